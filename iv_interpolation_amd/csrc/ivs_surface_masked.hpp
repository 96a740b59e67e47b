// Surfaces with MISSING quotes (NaN in sigma), 64 strikes x 16 maturities: the fast second pass.
//
// A missing quote changes the KNOT SET of its row, so every row has its own tridiagonal system -- the shared
// factorisation of the dense kernels does not apply -- and until now any NaN sent the surface to the correctness-first
// generic kernel (18 M surfaces/s: lane t runs row t's serial Thomas recurrence, 51 KB of LDS = 3 wavefronts per CU,
// one binary search per row and query, lane-private columns in LDS).  This kernel keeps the generic kernel's per-row
// solve but removes what made it slow:
//   * rows are compacted by wave ballot; besides the compacted quotes each row keeps a RANK table (valid knots at or
//     below strike k, one byte) so that the interval of a query in ANY row is one LDS byte away from the interval in
//     the full strike grid -- no per-row binary searches;
//   * the per-row system is eliminated from BOTH ends at once by two lanes ("burn at both ends"): the not-a-knot system
//     is symmetric under reversal of the strike axis (x -> -x, slopes change sign), so lane (row, 1) runs the same
//     recurrence on the mirrored row, the two meet in the middle, exchange their last (c', d') pair through one DPP
//     swap, close the 2 x 2 system and back-substitute their halves: half the dependent steps, no extra arithmetic,
//     the elimination's c' coefficients stay in registers (32 per lane);
//   * strike-pass values stay in registers; a column whose values are all there takes the dense register solve with the
//     batch-wide tables from the scalar cache, the few columns with a missing value (a row that lost its outermost
//     quotes does not reach the outermost output strikes) are solved in place by the generic per-column recurrence.
//     Surfaces with a row of fewer than 4 quotes, a column with too few values or more than 8 masked columns keep
//     their "redo" tag and fall through to the generic kernel (third launch, cheap when nothing is left).
// LDS: two planes [16][66] + two byte tables = 19.5 KB = 8 wavefronts per CU.
// The other dense methods ride on the same compaction: linear / slinear need no per-row table at all (the RANK byte gives
// the interval, the IDX byte the strike); pchip / akima compute a row's slopes with lane = compacted knot (secants of the
// neighbours through wave shifts, akima's row maximum through a wave reduction) -- no serial loop.
// Scope: uniform 64 x 16 batches, T / Tq shared, mK <= 64, linear / slinear / cubic / cubicspline / pchip / akima; runs
// in FILTER mode behind the dense / row-pass kernel (only surfaces tagged with the sentinel).
#pragma once
#include "ivs_surface_dense.hpp"

namespace ivs {

constexpr int MK_RS = 66;                          // row stride (doubles) of the compacted planes
// lerp methods keep no per-knot table next to the quotes (their strikes come through the IDX bytes): one plane, 11 KB
__host__ __device__ constexpr size_t masked_lds_bytes(bool lerp = false) { return (size_t)((lerp ? 1 : 2) * DT * MK_RS + DK) * 8 + 2 * DT * DK + DT * 4; }

// knots of one (possibly mirrored) compacted row: x(j), y(j) for j counted from the lane's own end
struct MaskedRow {
    const double* xs; const double* y; int n; bool mir;     // xs: the row's compacted strikes (S plane until overwritten)
    __device__ __forceinline__ int at(int j) const { return mir ? n - 1 - j : j; }
    __device__ __forceinline__ double x(int j) const { const double v = xs[at(j)]; return mir ? -v : v; }
    __device__ __forceinline__ double yv(int j) const { return y[at(j)]; }
};
struct MaskedT { const double* T; const uint8_t* idx; __device__ __forceinline__ double operator()(int i) const { return T[idx[i]]; } };
constexpr int MK_MAXCOL = 8;                      // masked output columns handled in place per surface (more: generic kernel)
// accessors of a compacted row for eval_cubic
struct MaskedX { const double* Ksh; const uint8_t* idx; __device__ __forceinline__ double operator()(int i) const { return Ksh[idx[i]]; } };

// maximum over the wavefront: DPP inside the rows of 16 (quad swaps, half mirror, mirror), one scalar read per row across
// them -- the six ds_bpermute round trips of a shuffle butterfly were a dependent LDS chain per ROW of an akima surface
__device__ __forceinline__ double wave_max_f64(double v) {
    v = __builtin_fmax(v, dpp_f64<DPP_QUAD_SWAP1>(v, v));
    v = __builtin_fmax(v, dpp_f64<DPP_QUAD_SWAP2>(v, v));
    v = __builtin_fmax(v, dpp_f64<0x141>(v, v));               // row_half_mirror
    v = __builtin_fmax(v, dpp_f64<0x140>(v, v));               // row_mirror: every lane holds its row's maximum
    return __builtin_fmax(__builtin_fmax(readlane_f64(v, 0), readlane_f64(v, 16)), __builtin_fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

template <int METHOD>
__global__ __launch_bounds__(64, (METHOD == IVS_LINEAR || METHOD == IVS_SLINEAR || d_is_step(METHOD)) ? 3 : 2) void surface_masked_kernel(SurfaceParams p) {
    constexpr bool NAK = METHOD == IVS_CUBIC || METHOD == IVS_CUBICSPLINE;
    constexpr bool LOCAL = d_is_local(METHOD);
    constexpr bool STEP = d_is_step(METHOD);                      // nearest / zero / from_derivatives: no per-knot table either
    constexpr bool LERP = METHOD == IVS_LINEAR || METHOD == IVS_SLINEAR || STEP;
    static_assert(NAK || LOCAL || LERP, "methods of the dense kernels only");
    constexpr int MINROW = NAK ? 4 : (METHOD == IVS_AKIMA ? 3 : 2);      // fewer quotes in a row: the generic kernel's business
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int mT = p.mT, mK = p.mK;
    double* YC = reinterpret_cast<double*>(smem);
    double* SS = YC + DT * MK_RS;                      // not carved for the lerp methods (never touched)
    double* Ksh = YC + (METHOD == IVS_LINEAR || METHOD == IVS_SLINEAR || d_is_step(METHOD) ? 1 : 2) * DT * MK_RS;
    uint8_t* IDX = reinterpret_cast<uint8_t*>(Ksh + DK);
    uint8_t* RANK = IDX + DT * DK;
    int* NROW = reinterpret_cast<int*>(RANK + DT * DK);
    const double nanv = __builtin_nan("");
    auto nostamp = [](int) {};

    TqTables tt;
    const double* TTp = nullptr;
    const double* Wp = nullptr;
    tq_from_shared(p.tqs, tt, TTp, Wp);

    const bool kq_shared = p.kq_stride == 0;
    const bool act = lane < mK;
    double xq = (kq_shared && act) ? p.Kq[lane] : nanv;
    const unsigned long long lt_mask = (1ull << lane) - 1ull, le_mask = lt_mask | (1ull << lane);

    const bool all = p.mode && *p.mode != 0;               // "missing quotes first": nothing was tagged, every surface is ours
    if (!all && p.redo && *p.redo == 0) return;            // nothing was tagged (wave-uniform)
    bool told = false;
    double* tag_at = nullptr;                              // the current surface's first output cell
    auto leave = [&]() {                                   // the surface keeps (or, in `all` mode, gets) its tag: generic kernel
        if (lane == 0) {
            if (all) reinterpret_cast<unsigned long long*>(tag_at)[0] = REDO_SENTINEL;
            if (p.redo && !told) { *reinterpret_cast<volatile int*>(p.redo + 1) = 1; told = true; }
        }
    };
    const int64_t n_outer = (p.B + 63) / 64;
    // blocks of 64 tags are claimed from a work queue (head 16 of the workspace; WorkQueue, ivs_surface_generic.hpp): the
    // tagged surfaces are spread unevenly over the blocks, and so is the speed of the workgroups
    WorkQueue wq;
    wq.init(p.queue + 16 * QUEUE_STRIDE, 1, n_outer, 1, lane);
    for (int64_t ob = p.queue ? wq.take() : (int64_t)blockIdx.x; ob >= 0 && ob < n_outer; ob = p.queue ? wq.take() : ob + gridDim.x) {
      const int64_t bi = ob * 64 + lane;
      const bool tagged = bi < p.B &&
          (all || reinterpret_cast<const unsigned long long*>(p.out + bi * (int64_t)mT * mK)[0] == REDO_SENTINEL);
      unsigned long long todo = __ballot(tagged);
      // quotes and strikes of the NEXT tagged surface of the block are requested while the current one is processed
      double vn[DT], kn = 0.0;
      auto request = [&](int64_t bb) {
          const double* sb = p.sigma + bb * (int64_t)(DT * DK);
#pragma unroll
          for (int t = 0; t < DT; ++t) vn[t] = sb[t * DK + lane];
          kn = p.K[bb * p.k_stride + lane];
      };
      if (todo) request(ob * 64 + __builtin_ctzll(todo));
      while (todo) {
        const int bit = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t b = ob * 64 + bit;
        double* outb = p.out + b * (int64_t)mT * mK;
        tag_at = outb;
        double v[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) v[t] = vn[t];
        const double kx = kn;
        if (!kq_shared) xq = act ? p.Kq[b * p.kq_stride + lane] : nanv;
        if (todo) request(ob * 64 + __builtin_ctzll(todo));
        __syncthreads();                                   // the previous surface's readers are done with LDS
        Ksh[lane] = kx;
        // ---- compact every row by ballot; RANK[t][k] = valid knots of row t at or below strike k
        bool give_up = tt.unsorted != 0;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const bool valid = !__builtin_isnan(v[t]);         // NaN = missing quote
            const unsigned long long m = __ballot(valid);
            give_up = give_up || __ballot(__builtin_isinf(v[t])) != 0ull;      // an infinity is a VALUE (it propagates): generic kernel
            const int rank = __popcll(m & lt_mask);
            // compacted quotes, their strike numbers, and -- in the S plane, until the elimination overwrites them knot by
            // knot -- the compacted strikes themselves (saves the index -> strike indirection inside the recurrence)
            if (valid) { YC[t * MK_RS + rank] = v[t]; IDX[t * DK + rank] = (uint8_t)lane; if (!LERP) SS[t * MK_RS + rank] = kx; }
            RANK[t * DK + lane] = (uint8_t)__popcll(m & le_mask);
            const int nt = __popcll(m);
            if (lane == 0) NROW[t] = nt;
            give_up = give_up || nt < MINROW;                  // too few knots (or an empty row): the generic kernel's business
        }
        if (give_up) { leave(); continue; }                    // wave-uniform; the sentinel stays, the generic pass redoes it
        __syncthreads();
        // ---- pchip / akima: a row's slopes with lane = compacted knot; the S plane's strikes are replaced by the slopes
        if (LOCAL) {
#pragma unroll 2
            for (int t = 0; t < DT; ++t) {
                const int n = NROW[t];
                double* xr = SS + t * MK_RS;
                const double* yr = YC + t * MK_RS;
                const int k = lane < n ? lane : n - 1, k1 = k + 1 < n ? k + 1 : n - 1;
                const double x0 = xr[k], x1 = xr[k1], y0 = yr[k], y1 = yr[k1];
                const double dxc = x1 - x0;                                       // dx_k, 0 from knot n-1 on
                const double mc = lane < n - 1 ? (y1 - y0) * refined_rcp(dxc) : 0.0;      // secant m_k
                double sk;
                if (METHOD == IVS_PCHIP) {
                    const double dxm = dpp0_f64<DPP_WAVE_SHR1>(dxc), dxp = dpp0_f64<DPP_WAVE_SHL1>(dxc), dxmm = dpp0_f64<DPP_WAVE_SHR1>(dxm);
                    const double mp = dpp0_f64<DPP_WAVE_SHR1>(mc), mn = dpp0_f64<DPP_WAVE_SHL1>(mc), mpp = dpp0_f64<DPP_WAVE_SHR1>(mp);
                    const bool first = lane == 0, last = lane == n - 1;
                    const double h0 = first ? dxc : dxm, h1 = first ? dxp : dxmm;      // one-sided three-point rule at the two ends
                    const double rs = refined_rcp(h0 + h1);
                    const double e = pchip_edge(first ? mc : mp, first ? mn : mpp, (2.0 * h0 + h1) * rs, h0 * rs);
                    const double v = pchip_knot(mp, mc, 2.0 * dxc + dxm, dxc + 2.0 * dxm);
                    sk = (first || last) ? e : v;
                    if (n == 2) sk = readlane_f64(mc, 0);
                } else {      // akima: secants extended linearly two steps beyond either end (scipy _cubic.py:520-528)
                    const double m0 = readlane_f64(mc, 0), m1 = readlane_f64(mc, 1);
                    const int nu = __builtin_amdgcn_readfirstlane(n);      // the row's knot count is wave-uniform: scalar lane reads, no ds_bpermute
                    const double l1 = readlane_f64(mc, nu - 2), l2 = readlane_f64(mc, nu - 3);
                    const double em1 = 2.0 * m0 - m1, em2 = 2.0 * em1 - m0, en1 = 2.0 * l1 - l2, en = 2.0 * en1 - l1;
                    const double E = lane <= n - 2 ? mc : en1;
                    const double es1 = dpp0_f64<DPP_WAVE_SHR1>(E), es2 = dpp0_f64<DPP_WAVE_SHR1>(es1), el1 = dpp0_f64<DPP_WAVE_SHL1>(E);
                    const double ma = lane >= 2 ? es2 : (lane == 1 ? em1 : em2);
                    const double mb = lane >= 1 ? es1 : em1;
                    const double md = lane <= n - 2 ? el1 : en;
                    const double fmax = wave_max_f64(lane < n ? akima_f12(ma, mb, E, md) : 0.0);
                    sk = akima_knot(ma, mb, E, md, 1e-9 * fmax);
                }
                __builtin_amdgcn_wave_barrier();               // every lane holds its strikes before the first slope lands
                if (lane < n) xr[lane] = sk;
            }
        }
        // ---- per-row not-a-knot solve, two lanes per row eliminating from the two ends (lanes 0..31)
        if (NAK && lane < 32) {
            const int t = lane >> 1;
            const bool mir = (lane & 1) != 0;
            const int n = NROW[t];
            const int m_top = n >> 1;
            const int cnt = mir ? n - m_top : m_top;           // rows this lane eliminates: 2 .. 32
            double* srow = SS + t * MK_RS;
            const MaskedRow R{srow, YC + t * MK_RS, n, mir};
            double cpv[32];
            double xm = R.x(0), xc = R.x(1), xp = R.x(2);
            double ym = R.yv(0), yc = R.yv(1), yp = R.yv(2);
            double dxm = xc - xm, dxc = xp - xc;
            double dlm = (yc - ym) * refined_rcp(dxm), dlc = (yp - yc) * refined_rcp(dxc);
            double cprev, dprev;
            {   // row 0 (not-a-knot): [dx1, x2 - x0], rhs = ((dx0 + 2d) dx1 d0 + dx0^2 d1) / d
                const double d = xp - xm;
                const double rhs = ((dxm + 2.0 * d) * dxc * dlm + dxm * dxm * dlc) * refined_rcp(d);
                const double rdx1 = refined_rcp(dxc);
                cprev = d * rdx1; dprev = rhs * rdx1;
                __builtin_amdgcn_wave_barrier();               // x(0..2) of BOTH lanes are in registers before anything is stored
                cpv[0] = cprev; srow[R.at(0)] = dprev;
                __builtin_amdgcn_wave_barrier();
            }
            int cmax = cnt;                                    // uniform bound of the unrolled loop
            cmax = max(cmax, __shfl_xor(cmax, 1)); cmax = max(cmax, __shfl_xor(cmax, 2)); cmax = max(cmax, __shfl_xor(cmax, 4));
            cmax = max(cmax, __shfl_xor(cmax, 8)); cmax = max(cmax, __shfl_xor(cmax, 16));
            cmax = __builtin_amdgcn_readfirstlane(cmax);
#pragma unroll
            for (int j = 1; j < 32; ++j) {
                if (j < cmax) {                               // wave-uniform: whole steps are skipped
                    // the knot two steps ahead is read BEFORE this step's result is stored: in a row with an odd knot
                    // count the partner lane overwrites exactly that slot of the S plane in this very step
                    double xn = 0.0, yn = 0.0;
                    if (j + 1 < cnt) { xn = R.x(j + 2); yn = R.yv(j + 2); }
                    __builtin_amdgcn_wave_barrier();
                    if (j < cnt) {
                        const double rhs = 3.0 * (dxc * dlm + dxm * dlc);
                        const double rw = refined_rcp(2.0 * (dxm + dxc) - dxc * cprev);
                        cprev = dxm * rw;
                        dprev = (rhs - dxc * dprev) * rw;
                        cpv[j] = cprev; srow[R.at(j)] = dprev;
                        if (j + 1 < cnt) {
                            xm = xc; xc = xp; xp = xn;
                            ym = yc; yc = yp; yp = yn;
                            dxm = dxc; dxc = xp - xc;
                            dlm = dlc; dlc = (yp - yc) * refined_rcp(dxc);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // meet in the middle: own relation  s_own = d_own - c_own * s_other', partner's the same with roles swapped
            const double c_oth = dpp_f64<DPP_QUAD_SWAP1>(cprev, cprev), d_oth = dpp_f64<DPP_QUAD_SWAP1>(dprev, dprev);
            double sn = (dprev + cprev * d_oth) * refined_rcp(1.0 - cprev * c_oth);
            srow[R.at(cnt - 1)] = mir ? -sn : sn;
#pragma unroll
            for (int j = 30; j >= 0; --j) {
                if (j < cnt - 1) {
                    sn = srow[R.at(j)] - cpv[j] * sn;
                    srow[R.at(j)] = mir ? -sn : sn;
                }
            }
        }
        __syncthreads();
        // ---- strike evaluation (q-lane): interval in the full grid once, per row one RANK byte away
        int jf = -1;
        if (Ksh[0] <= xq) {
            jf = 0;
#pragma unroll
            for (int m = 1; m < 8; ++m) jf += (Ksh[8 * m] <= xq) ? 8 : 0;
#pragma unroll
            for (int st = 4; st >= 1; st >>= 1) if (Ksh[jf + st] <= xq) jf += st;
        }
        double z[DT];
        bool all_ok = true;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            const int n = NROW[t];
            const int j = jf >= 0 ? (int)RANK[t * DK + jf] - 1 : -1;
            const MaskedX X{Ksh, IDX + t * DK};
            const CView Y{YC + t * MK_RS, 1}, S{SS + t * MK_RS, 1};
            if (METHOD == IVS_NEAREST) {
                // the row's interval j is known (RANK): one midpoint compare (ties to the left knot, as searchsorted side='left'
                // over the midpoints decides for strictly increasing strikes) instead of eval_nearest's binary search over the
                // midpoints -- 6 steps of two two-level LDS gathers per row (10 % of the quotes missing: 148 -> 242 M surfaces/s)
                double r = nanv;
                if (j >= 0 && xq <= X(n - 1)) {
                    const int jj = j > n - 2 ? n - 2 : j;
                    r = j > n - 2 ? Y(n - 1) : step_eval<IVS_NEAREST>(xq, X(jj), X(jj + 1), Y(jj), Y(jj + 1));
                }
                z[t] = r;
            } else if (STEP) z[t] = eval_method(METHOD, X, Y, S, n, j, xq);
            else if (LERP) z[t] = eval_linear(X, Y, n, j, xq, METHOD == IVS_LINEAR);      // (np.interp's division through div_shared_rcp: measured -5 % here -- the per-row range checks cost more than the 16 divisions)
            else z[t] = eval_cubic(X, Y, S, n, j, xq, d_extrap_right(METHOD));
            all_ok = all_ok && !__builtin_isnan(z[t]);
        }
        // ---- maturity direction.  A column whose strike-pass values are all there takes the dense register solve with the
        // batch-wide tables; a column with a missing value (a row whose outermost quotes are gone does not reach the
        // outermost output strikes) has its own maturity knot set: up to MK_MAXCOL of them are solved in place by the
        // generic kernel's per-column recurrence, in slots carved out of the (now dead) quote plane.
        const bool col_masked = act && !all_ok;
        const unsigned long long mm = __ballot(col_masked);
        if (mm != 0ull) {
            if (__popcll(mm) > MK_MAXCOL) { leave(); continue; }      // the generic kernel redoes the surface
            __syncthreads();                                   // every lane is done with the planes
            double* Tsh = Ksh;                                 // 16 maturities
            double* cz = YC; double* cs = YC + MK_MAXCOL * DT; double* ccp = YC + 2 * MK_MAXCOL * DT;
            uint8_t* cti = IDX;
            if (lane < DT) Tsh[lane] = p.T[lane];
            const int slot = __popcll(mm & lt_mask);
            int cn = 0;
            if (col_masked) {
#pragma unroll
                for (int t = 0; t < DT; ++t)
                    if (!__builtin_isnan(z[t])) { cz[slot * DT + cn] = z[t]; cti[slot * DT + cn] = (uint8_t)t; ++cn; }
            }
            if (__ballot(col_masked && cn > 0 && cn < method_min_knots(METHOD)) != 0ull) { leave(); continue; }    // too-few-knots status: generic kernel
            __syncthreads();
            if (col_masked) {
                const MaskedT cx{Tsh, cti + slot * DT};
                const CView cy{cz + slot * DT, 1};
                View csv{cs + slot * DT, 1}, cpw{ccp + slot * DT, 1};
                if (!LERP && cn >= 2) method_slopes(METHOD, cx, cy, csv, cpw, cn);
                const CView csr{cs + slot * DT, 1};
                int jc = -1;
                for (int tq = 0; tq < mT; ++tq) {
                    const double x = p.Tq[tq];
                    double r = nanv;
                    if (cn > 0) {
                        while (jc + 1 < cn && cx(jc + 1) <= x) ++jc;          // Tq ascending (checked: tt.unsorted)
                        r = eval_method(METHOD, cx, cy, csr, cn, jc, x);
                    }
                    outb[(int64_t)tq * mK + lane] = r;
                }
            }
        }
        if (LOCAL) {
#pragma unroll
            for (int r = 0; r < DT; ++r) asm volatile("" : "+v"(z[r]));      // see dense_maturity_pass
        }
        if (act && all_ok) dense_maturity_pass<METHOD, true, false, false, true, false>(z, tt, TTp, Wp, outb, 0, lane, true, mT, mK, nostamp);      // 8 wavefronts per CU by LDS: akima keeps its slopes in registers (-2 % with the rolling form)
        if (p.status && lane == 0) p.status[b] = IVS_ST_OK;
      }
    }
}

}  // namespace ivs
#include "ivs_surface_masked_pass.hpp"      // the not-a-knot methods' row-pass form (needs the definitions above)
namespace ivs {

#ifndef IVS_MASKED_PASS_LOCAL
#define IVS_MASKED_PASS_LOCAL 1      // pchip / akima with missing quotes on the two-pass layout too (0: the lane-per-knot kernel above)
#endif
#ifndef IVS_MASKED_PASS
#define IVS_MASKED_PASS 1      // 0: cubic / cubicspline on the two-lanes-per-row kernel above (A/B)
#endif

#ifndef IVS_DIAG_MINIMAL
// Second pass behind the dense / row-pass kernel for uniform 64 x 16 batches: returns true when launched.
inline bool launch_surface_masked(const SurfaceParams& p, const LaunchCtx& cx) {
    if (p.k_off || p.nK != DK || p.nT != DT || p.mK > 64 || p.mT > D_MAX_MT) return false;
    if (p.t_stride != 0 || p.tq_stride != 0 || !p.tqs) return false;
    const bool lerp = p.method == IVS_LINEAR || p.method == IVS_SLINEAR || d_is_step(p.method);
    const size_t lds = masked_lds_bytes(lerp);
    int64_t grid = (int64_t)cx.num_cu * (lerp ? 12 : 8);
    const int64_t work = (p.B + 63) / 64;
    if (grid > work) grid = work;
    if (p.method == IVS_QUADRATIC || (IVS_MASKED_PASS && (p.method == IVS_CUBIC || p.method == IVS_CUBICSPLINE)) ||
        (IVS_MASKED_PASS_LOCAL && d_is_local(p.method))) {
        int64_t g12 = (int64_t)cx.num_cu * 12;
        if (g12 > work) g12 = work;
        if (p.method == IVS_CUBIC) hipLaunchKernelGGL((surface_masked_pass_kernel<IVS_CUBIC>), dim3((unsigned)g12), dim3(64), masked_pass_lds_bytes(), cx.st, p);
        else if (p.method == IVS_QUADRATIC) hipLaunchKernelGGL((surface_masked_pass_kernel<IVS_QUADRATIC>), dim3((unsigned)g12), dim3(64), masked_pass_lds_bytes(), cx.st, p);
        else if (p.method == IVS_PCHIP) hipLaunchKernelGGL((surface_masked_pass_kernel<IVS_PCHIP>), dim3((unsigned)g12), dim3(64), masked_pass_lds_bytes(), cx.st, p);
        else if (p.method == IVS_AKIMA) hipLaunchKernelGGL((surface_masked_pass_kernel<IVS_AKIMA>), dim3((unsigned)g12), dim3(64), masked_pass_lds_bytes(), cx.st, p);
        else hipLaunchKernelGGL((surface_masked_pass_kernel<IVS_CUBICSPLINE>), dim3((unsigned)g12), dim3(64), masked_pass_lds_bytes(), cx.st, p);
        return true;
    }
    switch (p.method) {
#define IVS_MASKED_CASE(M) case M: hipLaunchKernelGGL((surface_masked_kernel<M>), dim3((unsigned)grid), dim3(64), lds, cx.st, p); break;
        IVS_MASKED_CASE(IVS_LINEAR) IVS_MASKED_CASE(IVS_SLINEAR) IVS_MASKED_CASE(IVS_CUBIC) IVS_MASKED_CASE(IVS_CUBICSPLINE)
        IVS_MASKED_CASE(IVS_PCHIP) IVS_MASKED_CASE(IVS_AKIMA) IVS_MASKED_CASE(IVS_NEAREST) IVS_MASKED_CASE(IVS_ZERO)
        IVS_MASKED_CASE(IVS_FROM_DERIVATIVES)
#undef IVS_MASKED_CASE
        default: return false;
    }
    return true;
}
#endif

}  // namespace ivs
