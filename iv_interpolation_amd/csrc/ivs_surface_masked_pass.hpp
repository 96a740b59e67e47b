// Surfaces with MISSING quotes, 64 strikes x 16 maturities, not-a-knot methods: the row-pass form of the masked kernel.
//
// ivs_surface_masked.hpp eliminates every row's own tridiagonal system with TWO lanes per row (from both ends), 32 of 64
// lanes busy on dependent chains of up to 32 steps, the elimination's c' coefficients in 64 VGPRs: 256 VGPRs, 19.5 KB of
// LDS, 8 wavefronts per CU, 42 % of the wave time parked on dependent LDS round trips (profiles/r02/final/pmc_nan10.json).
// Here the 16 rows are processed in two PASSES of 8 rows and a row's compacted system is solved by EIGHT lanes, lane =
// (row of the pass, segment of 8 compacted knots), as a segmented two-level elimination:
//   1. every lane normalises its (up to) 8 system rows (1 / b_i) and multiplies the 2 x 2 Moebius matrices of the pivot
//      recurrence c'_i = u_i / (1 - l_i c'_{i-1}) over its segment; a 3-step Kogge-Stone scan over the row's 8 lanes (DPP
//      row_shr inside the 8-lane group) gives every segment its incoming c';
//   2. the lane runs the recurrence over its segment (one reciprocal per knot), the forward right-hand-side sweep with a
//      zero carry and the running products of its multipliers; the carries cross the segments as a scan of affine maps;
//   3. the same backwards for the slopes.
// Chains of 8 steps instead of 32, all 64 lanes busy, 48 VGPRs of per-knot state instead of 64 + the mirror bookkeeping;
// planes of 8 rows: 10 KB of LDS -> 12 wavefronts per CU at 168 VGPRs.  Compaction (RANK / IDX byte tables), strike
// evaluation, the in-place solve of masked output columns and the maturity pass are those of ivs_surface_masked.hpp.
// Rounding differs from the serial Thomas order by ~1e-15 of the slope scale on the benchmark grids (the same
// conditioning remarks as for the scan-based dense kernels apply: DESIGN 3).
#pragma once
#include "ivs_surface_masked.hpp"

namespace ivs {

constexpr int MP_ROWS = 8;                                        // rows per pass
__host__ __device__ constexpr size_t masked_pass_lds_bytes() { return (size_t)(2 * MP_ROWS * MK_RS + DK) * 8 + 2 * MP_ROWS * DK + DT * 4; }

// Not-a-knot slopes of the 8 compacted rows staged in (YC, XS): XS holds the compacted strikes on entry and the slopes on
// exit.  All 64 lanes; the caller brackets it with barriers.  n = NROW[row] >= 4.
// QUAD: the collocation system of the quadratic B-spline (quad_row, ivs_device.hpp) instead -- rows (lo, di, up), right-hand
// side y_i, rows 0 and n - 1 the identity; the solution = the B-spline coefficients; n >= 3.
template <bool QUAD = false>
__device__ __forceinline__ void masked_solve8(const double* YC, double* XS, const int* NROW, int lane) {
    const int r = lane >> 3, seg = lane & 7, i0 = seg * 8;
    const int n = NROW[r];
    double* xr = XS + r * MK_RS;
    const double* yr = YC + r * MK_RS;
    auto cl = [&](int i) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); };
    // rolling window of the four intervals around knot i: (i-2,i-1) (i-1,i) (i,i+1) (i+1,i+2); clamped indices repeat
    // a knot (dx = 0 -> inf / NaN in entries that only masked-off rows or edge selects look at)
    double xa = xr[cl(i0 - 2)], xb = xr[cl(i0 - 1)], xc = xr[cl(i0)], xd = xr[cl(i0 + 1)], xe = xr[cl(i0 + 2)];
    double ya = yr[cl(i0 - 2)], yb = yr[cl(i0 - 1)], yc = yr[cl(i0)], yd = yr[cl(i0 + 1)], ye = yr[cl(i0 + 2)];
    double dx0 = xb - xa, dx1 = xc - xb, dx2 = xd - xc, dx3 = xe - xd;
    double dl0 = (yb - ya) * refined_rcp(dx0), dl1 = (yc - yb) * refined_rcp(dx1), dl2 = (yd - yc) * refined_rcp(dx2),
           dl3 = (ye - yd) * refined_rcp(dx3);
    double xl = xe, yl = ye;                                       // last knot of the window
    double u[8], l[8], rh[8];                                      // normalised rows: upper, lower, right-hand side
    double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;             // product of the segment's Moebius matrices
    // The two not-a-knot boundary rows are built ONCE, outside the loop over the segment (inside it they cost a reciprocal,
    // two right-hand sides and eight selects per knot for two knots of a row): row 0 from the initial window of segment 0,
    // row n - 1 from the row's last three knots; the loop takes them by one select per coefficient.
    double uF = 0.0, lF = 0.0, rF = 0.0, uL = 0.0, lL = 0.0, rL = 0.0;
    int kl = -1;                                                   // the lane's k of row n - 1 (or none)
    if (!QUAD) {
        {   // row 0: b = dx3, c = d, rhs = ((dx2 + 2 d) dx3 dl2 + dx2^2 dl3) / d with d = dx2 + dx3 (seg 0: xc = x_0)
            const double d = dx2 + dx3, rb = refined_rcp(dx3);
            uF = d * rb;
            rF = ((dx2 + 2.0 * d) * dx3 * dl2 + dx2 * dx2 * dl3) * refined_rcp(d) * rb;
        }
        {   // row n - 1: a = d, b = dx0, rhs = (dx1^2 dl0 + (2 d + dx1) dx0 dl1) / d on the knots n-3, n-2, n-1
            const double x3 = xr[n - 3], x2 = xr[n - 2], x1 = xr[n - 1], y3 = yr[n - 3], y2 = yr[n - 2], y1 = yr[n - 1];
            const double e0 = x2 - x3, e1 = x1 - x2, g0 = (y2 - y3) * refined_rcp(e0), g1 = (y1 - y2) * refined_rcp(e1);
            const double d = e0 + e1, rb = refined_rcp(e0);
            lL = d * rb;
            rL = (e1 * e1 * g0 + (2.0 * d + e1) * e0 * g1) * refined_rcp(d) * rb;
            kl = ((n - 1) >> 3) == seg ? ((n - 1) & 7) : -1;
        }
    }
    // QUAD: the data sites x(i-2 .. i+2) (clamped) and the values y(i .. i+2) roll along with the window, so the knots of the
    // collocation row (midpoints of neighbouring sites, the triple end knots) come from registers -- quad_row read them from
    // LDS again, eight loads with their clamps per site
    double qa = xa, qb = xb, qc = xc, qd = xd, qe = xe, qy0 = yc, qy1 = yd, qy2 = ye;
    const double x_first = xr[0], x_last = xr[n - 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = i0 + k;
        const bool in = i < n;
        double a = dx2, b = 2.0 * (dx1 + dx2), c = dx1, rr = 3.0 * (dx2 * dl1 + dx1 * dl2);
        if (QUAD) {
            const double tA = i <= 2 ? x_first : (i >= n ? x_last : (qb + qa) / 2.0);              // t_i .. t_{i+3} (quad_knot)
            const double tB = i + 1 <= 2 ? x_first : (i + 1 >= n ? x_last : (qc + qb) / 2.0);
            const double tC = i + 2 <= 2 ? x_first : (i + 2 >= n ? x_last : (qd + qc) / 2.0);
            const double tD = i + 3 >= n ? x_last : (qe + qd) / 2.0;
            double lo, di, up;
            quad_basis_t(tA, tB, tC, tD, qc, lo, di, up);
            const bool interior = i > 0 && i < n - 1;              // rows 0 and n - 1: c_i = y_i
            a = interior ? lo : 0.0; b = interior ? di : 1.0; c = interior ? up : 0.0;
            rr = qy0;
        }
        const double rb = refined_rcp(b);
        double uu = c * rb, ll = a * rb, hh = rr * rb;
        if (!QUAD) {
            if (k == 0) { const bool first = seg == 0; uu = first ? uF : uu; ll = first ? lF : ll; hh = first ? rF : hh; }
            const bool last = k == kl;
            uu = last ? uL : uu; ll = last ? lL : ll; hh = last ? rL : hh;
        }
        u[k] = in ? uu : 0.0; l[k] = in ? ll : 0.0; rh[k] = in ? hh : 0.0;
        const double n00 = u[k] * m10, n01 = u[k] * m11, n10 = m10 - l[k] * m00, n11 = m11 - l[k] * m01;      // M <- [[0,u],[-l,1]] M
        m00 = n00; m01 = n01; m10 = n10; m11 = n11;
        if (k < 7) {                                               // slide the window: the interval (i+2, i+3) enters
            const int j = cl(i + 3);
            const double xn = xr[j], yn = yr[j];
            dx0 = dx1; dx1 = dx2; dx2 = dx3; dx3 = xn - xl;
            dl0 = dl1; dl1 = dl2; dl2 = dl3; dl3 = (yn - yl) * refined_rcp(dx3);
            xl = xn; yl = yn;
            if (QUAD) { qa = qb; qb = qc; qc = qd; qd = qe; qe = xn; qy0 = qy1; qy1 = qy2; qy2 = yn; }
        }
    }
    // inclusive prefix products over the row's 8 segments: P_s = M_s P_{s - sh} (Kogge-Stone; a lane whose source would
    // lie in the neighbouring row's group keeps its value)
#define IVS_MP_MAT_STEP(SH)                                                                                   \
    {                                                                                                         \
        const double e = dpp_f64<DPP_ROW_SHR(SH)>(1.0, m00), f = dpp_f64<DPP_ROW_SHR(SH)>(0.0, m01);          \
        const double g = dpp_f64<DPP_ROW_SHR(SH)>(0.0, m10), h = dpp_f64<DPP_ROW_SHR(SH)>(1.0, m11);          \
        const bool has = seg >= SH;                                                                           \
        const double n00 = m00 * e + m01 * g, n01 = m00 * f + m01 * h, n10 = m10 * e + m11 * g, n11 = m10 * f + m11 * h; \
        m00 = has ? n00 : m00; m01 = has ? n01 : m01; m10 = has ? n10 : m10; m11 = has ? n11 : m11;           \
    }
    IVS_MP_MAT_STEP(1) IVS_MP_MAT_STEP(2) IVS_MP_MAT_STEP(4)
#undef IVS_MP_MAT_STEP
    // incoming c' of the segment: the previous segments' product applied to (p, q) = (0, 1)
    const double pin = dpp_f64<DPP_ROW_SHR(1)>(0.0, m01), qin = dpp_f64<DPP_ROW_SHR(1)>(1.0, m11);
    double cprev = seg == 0 ? 0.0 : pin * refined_rcp(qin);
    // ---- pivots, forward sweep with zero carry, running products (in place: u -> c', rh -> d', l -> prod(-al))
    double dprev = 0.0, prod = 1.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double rw = refined_rcp(1.0 - l[k] * cprev);
        cprev = u[k] * rw;
        const double al = l[k] * rw;
        dprev = rh[k] * rw - al * dprev;
        prod *= -al;
        u[k] = cprev; rh[k] = dprev; l[k] = prod;
    }
    {   // carries across the segments: d_out = E + P d_in, inclusive scan, then the exclusive value of the left neighbour
        double A = prod, B = dprev;
#define IVS_MP_AFF_STEP(SH)                                                                                   \
        { const double a_ = dpp_f64<DPP_ROW_SHR(SH)>(1.0, A), b_ = dpp_f64<DPP_ROW_SHR(SH)>(0.0, B);          \
          const bool has = seg >= SH; B = has ? __builtin_fma(A, b_, B) : B; A = has ? A * a_ : A; }
        IVS_MP_AFF_STEP(1) IVS_MP_AFF_STEP(2) IVS_MP_AFF_STEP(4)
#undef IVS_MP_AFF_STEP
        double din = dpp_f64<DPP_ROW_SHR(1)>(0.0, B);
        din = seg == 0 ? 0.0 : din;
#pragma unroll
        for (int k = 0; k < 8; ++k) rh[k] = __builtin_fma(l[k], din, rh[k]);
    }
    // ---- backward sweep with zero carry and running products (in place: rh -> slope, l -> prod(-c'))
    double nxt = 0.0;
    prod = 1.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        nxt = rh[k] - u[k] * nxt;
        prod *= -u[k];
        rh[k] = nxt; l[k] = prod;
    }
    {
        double A = prod, B = nxt;
#define IVS_MP_AFF_STEP(SH)                                                                                   \
        { const double a_ = dpp_f64<DPP_ROW_SHL(SH)>(1.0, A), b_ = dpp_f64<DPP_ROW_SHL(SH)>(0.0, B);          \
          const bool has = seg + SH <= 7; B = has ? __builtin_fma(A, b_, B) : B; A = has ? A * a_ : A; }
        IVS_MP_AFF_STEP(1) IVS_MP_AFF_STEP(2) IVS_MP_AFF_STEP(4)
#undef IVS_MP_AFF_STEP
        double sin_ = dpp_f64<DPP_ROW_SHL(1)>(0.0, B);
        sin_ = seg == 7 ? 0.0 : sin_;
#pragma unroll
        for (int k = 0; k < 8; ++k) rh[k] = __builtin_fma(l[k], sin_, rh[k]);
    }
    __builtin_amdgcn_wave_barrier();                               // every lane's strikes are in registers: the slopes may land
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k < n) xr[i0 + k] = rh[k];
}

// Local-slope methods (pchip, akima) on the same layout: lane = (row of the pass, segment of 8 compacted knots), the 8 slopes
// of the segment from the 11 secants around it (2 knots of either neighbour), no system to solve.  Formulas of the
// lane-per-knot form in ivs_surface_masked.hpp (pchip: weighted harmonic mean, one-sided three-point rule at knots 0 and
// n - 1, the secant for n == 2; akima: secants extended linearly two steps beyond either end, the row maximum of
// |m_{k+1} - m_k| + |m_{k-1} - m_{k-2}| over the row's 8 lanes by three DPP moves).  XS: strikes on entry, slopes on exit.
template <int METHOD>
__device__ __forceinline__ void masked_local8(const double* YC, double* XS, const int* NROW, int lane) {
    constexpr bool AK = METHOD == IVS_AKIMA;
    const int r = lane >> 3, seg = lane & 7, i0 = seg * 8;
    const int n = NROW[r];
    double* xr = XS + r * MK_RS;
    const double* yr = YC + r * MK_RS;
    auto cl = [&](int i) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); };
    // secants F[i] and widths DX[i] of the intervals i0 - 2 + i, i = 0..10 (intervals outside [0, n-2]: 0 here, fixed up below)
    double F[11], DX[11];
    {
        double xa = xr[cl(i0 - 2)], ya = yr[cl(i0 - 2)];
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int idx = i0 - 2 + i;
            const double xb = xr[cl(idx + 1)], yb = yr[cl(idx + 1)];
            const bool in = idx >= 0 && idx <= n - 2;
            const double dx = xb - xa;
            DX[i] = in ? dx : 0.0;
            F[i] = in ? (yb - ya) * refined_rcp(dx) : 0.0;
            xa = xb; ya = yb;
        }
    }
    double d[8];
    if (AK) {
        // linear extension: m_{-1} = 2 m_0 - m_1, m_{-2} = 2 m_{-1} - m_0; m_{n-1} = 2 m_{n-2} - m_{n-3}, m_n = 2 m_{n-1} - m_{n-2}
        const double x3 = xr[n - 3], x2 = xr[n - 2], x1 = xr[n - 1], y3 = yr[n - 3], y2 = yr[n - 2], y1 = yr[n - 1];
        const double mL1 = (y1 - y2) * refined_rcp(x1 - x2), mL2 = (y2 - y3) * refined_rcp(x2 - x3);      // m_{n-2}, m_{n-3}
        const double en1 = 2.0 * mL1 - mL2, en = 2.0 * en1 - mL1;
        const double m0 = (yr[1] - yr[0]) * refined_rcp(xr[1] - xr[0]), m1 = (yr[2] - yr[1]) * refined_rcp(xr[2] - xr[1]);
        const double em1 = 2.0 * m0 - m1, em2 = 2.0 * em1 - m0;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int idx = i0 - 2 + i;
            F[i] = idx == -1 ? em1 : (idx == -2 ? em2 : (idx == n - 1 ? en1 : (idx == n ? en : F[i])));
        }
        double fmax = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double f = akima_f12(F[k], F[k + 1], F[k + 2], F[k + 3]);
            fmax = (i0 + k < n) ? __builtin_fmax(fmax, f) : fmax;
        }
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP1>(fmax, fmax));
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP2>(fmax, fmax));
        fmax = __builtin_fmax(fmax, dpp_f64<0x141>(fmax, fmax));           // row_half_mirror: the other quad of the row's 8 lanes
        const double thr = 1e-9 * fmax;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = akima_knot(F[k], F[k + 1], F[k + 2], F[k + 3], thr);
    } else {
        // the one-sided rule at knot n - 1 from the row's last three knots, once (n >= 3; n == 2: the secant)
        double e_last = 0.0, m_only = 0.0;
        {
            const int a3 = n >= 3 ? n - 3 : 0, a2 = n >= 3 ? n - 2 : 0, a1 = n - 1;
            const double x3 = xr[a3], x2 = xr[a2], x1 = xr[a1], y3 = yr[a3], y2 = yr[a2], y1 = yr[a1];
            const double h0 = x1 - x2, h1 = x2 - x3;                                     // dx_{n-2}, dx_{n-3}
            const double mp = (y1 - y2) * refined_rcp(h0), mpp = (y2 - y3) * refined_rcp(h1);
            const double rs = refined_rcp(h0 + h1);
            e_last = pchip_edge(mp, mpp, (2.0 * h0 + h1) * rs, h0 * rs);
            m_only = (yr[n - 1] - yr[0]) * refined_rcp(xr[n - 1] - xr[0]);               // n == 2
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k;
            // knot i: secants m_{i-1} = F[k+1], m_i = F[k+2]; widths dx_{i-1} = DX[k+1], dx_i = DX[k+2]
            double v = pchip_knot(F[k + 1], F[k + 2], 2.0 * DX[k + 2] + DX[k + 1], DX[k + 2] + 2.0 * DX[k + 1]);
            if (k == 0) {       // knot 0 sits in segment 0 only: h0 = dx_0, h1 = dx_1, secants m_0, m_1
                const double h0 = DX[2], h1 = DX[3], rs = refined_rcp(h0 + h1);
                const double e = pchip_edge(F[2], F[3], (2.0 * h0 + h1) * rs, h0 * rs);
                v = seg == 0 ? e : v;
            }
            v = i == n - 1 ? e_last : v;
            d[k] = n == 2 ? m_only : v;
        }
    }
    __builtin_amdgcn_wave_barrier();                               // every lane's strikes are in registers: the slopes may land
#pragma unroll
    for (int k = 0; k < 8; ++k) if (i0 + k < n) xr[i0 + k] = d[k];
}

template <int METHOD>
__global__ __launch_bounds__(64, 3) void surface_masked_pass_kernel(SurfaceParams p) {
    static_assert(METHOD == IVS_CUBIC || METHOD == IVS_CUBICSPLINE || METHOD == IVS_QUADRATIC || d_is_local(METHOD), "tridiagonal and local-slope methods");
    constexpr bool QUAD = METHOD == IVS_QUADRATIC;
    constexpr bool LOCAL = d_is_local(METHOD);
    constexpr int MINROW = LOCAL ? (METHOD == IVS_AKIMA ? 3 : 2) : (QUAD ? 3 : 4);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int mT = p.mT, mK = p.mK;
    double* YC = reinterpret_cast<double*>(smem);
    double* XS = YC + MP_ROWS * MK_RS;
    double* Ksh = XS + MP_ROWS * MK_RS;
    uint8_t* IDX = reinterpret_cast<uint8_t*>(Ksh + DK);
    uint8_t* RANK = IDX + MP_ROWS * DK;
    int* NROW = reinterpret_cast<int*>(RANK + MP_ROWS * DK);
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
    WorkQueue wq;
    wq.init(p.queue + 16 * QUEUE_STRIDE, 1, n_outer, 1, lane);
    for (int64_t ob = p.queue ? wq.take() : (int64_t)blockIdx.x; ob >= 0 && ob < n_outer; ob = p.queue ? wq.take() : ob + gridDim.x) {
      const int64_t bi = ob * 64 + lane;
      const bool tagged = bi < p.B &&
          (all || reinterpret_cast<const unsigned long long*>(p.out + bi * (int64_t)mT * mK)[0] == REDO_SENTINEL);
      unsigned long long todo = __ballot(tagged);
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
        // ---- which quotes are there (NaN = missing); rows below four knots and infinities are the generic kernel's business
        bool give_up = tt.unsorted != 0;
        unsigned long long vm[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            vm[t] = __ballot(!__builtin_isnan(v[t]));
            give_up = give_up || __ballot(__builtin_isinf(v[t])) != 0ull || __popcll(vm[t]) < MINROW;
        }
        if (give_up) { leave(); continue; }                    // wave-uniform; the sentinel stays
        __syncthreads();                                       // the previous surface's readers are done with LDS
        Ksh[lane] = kx;
        double z[DT];
        bool all_ok = true;
        int jf = -1;
#pragma unroll
        for (int ps = 0; ps < DT / MP_ROWS; ++ps) {
            if (ps) __syncthreads();                           // the previous pass's evaluation is done with the planes
            // ---- compact the pass's rows by ballot; RANK[r][k] = valid knots of the row at or below strike k
#pragma unroll
            for (int r = 0; r < MP_ROWS; ++r) {
                const int t = ps * MP_ROWS + r;
                const unsigned long long m = vm[t];
                const int rank = __popcll(m & lt_mask);
                if ((m >> lane) & 1ull) { YC[r * MK_RS + rank] = v[t]; XS[r * MK_RS + rank] = kx; IDX[r * DK + rank] = (uint8_t)lane; }
                RANK[r * DK + lane] = (uint8_t)__popcll(m & le_mask);
                if (lane == 0) NROW[r] = __popcll(m);
            }
            __syncthreads();
            if (ps == 0 && Ksh[0] <= xq) {                     // interval of the lane's output strike in the FULL grid, once
                jf = 0;
#pragma unroll
                for (int m = 1; m < 8; ++m) jf += (Ksh[8 * m] <= xq) ? 8 : 0;
#pragma unroll
                for (int st = 4; st >= 1; st >>= 1) if (Ksh[jf + st] <= xq) jf += st;
            }
#ifndef IVS_MP_ABL
#define IVS_MP_ABL 0      // phase ablations (wrong results, timing only): 1 = no slope solve, 2 = no strike evaluation, 3 = no compaction tables
#endif
            if (IVS_MP_ABL == 1) {}
            else if (LOCAL) masked_local8<METHOD>(YC, XS, NROW, lane);
            else masked_solve8<QUAD>(YC, XS, NROW, lane);
            __syncthreads();
            // ---- strike evaluation of the pass's rows (q-lane): per row one RANK byte away from the full-grid interval
#pragma unroll
            for (int r = 0; r < MP_ROWS; ++r) {
                // the row's knot count and outermost strikes from its ballot mask (scalar: no LDS reads; the hull test of the
                // methods that do not extrapolate cost two dependent LDS reads per row through the IDX table -- 0.6 of 7.0 ms)
                const unsigned long long rm = vm[ps * MP_ROWS + r];
                const int n = __popcll(rm);
                const double x_last = readlane_f64(kx, 63 - __builtin_clzll(rm));
                const int j = jf >= 0 ? (int)RANK[r * DK + jf] - 1 : -1;
                const MaskedX X{Ksh, IDX + r * DK};
                const CView Y{YC + r * MK_RS, 1}, S{XS + r * MK_RS, 1};
                double zz;
                if (IVS_MP_ABL == 2) zz = Y(lane & 7) + xq;
                else if (QUAD) {
                    // eval_quadratic with the row's strikes read ONCE: the six sites around the interval (x(je-2 .. je+3), je =
                    // min(j, n-2)) hold every site the basis window [q, q+2], q in {je-1, je}, can ask for; the triple end knots
                    // come from the ballot mask.  (eval_quadratic read ~11 strikes per row through the two-level IDX table:
                    // the strike evaluation was 6.1 of the kernel's 11.4 ms.)  Same arithmetic: quad_knot / quad_basis_t.
                    zz = nanv;
                    if (j >= 0 && xq <= x_last) {
                        const double x_first = readlane_f64(kx, __builtin_ctzll(rm));
                        const int je = j > n - 2 ? n - 2 : j;
                        auto cl = [&](int i) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); };
                        double W[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) W[i] = X(cl(je - 2 + i));
                        int q = je - 1 + ((xq >= (W[3] + W[2]) / 2.0) ? 1 : 0);
                        q = j < n - 1 ? q : n - 3;
                        q = q < 0 ? 0 : (q > n - 3 ? n - 3 : q);
                        const bool hi = q == je;                                   // window x(q-1 .. q+3) = W[1..5], else W[0..4]
                        const double xs0 = hi ? W[1] : W[0], xs1 = hi ? W[2] : W[1], xs2 = hi ? W[3] : W[2], xs3 = hi ? W[4] : W[3],
                                     xs4 = hi ? W[5] : W[4];
                        const double tm1 = q + 1 <= 2 ? x_first : (xs1 + xs0) / 2.0;             // t_{q+1} .. t_{q+4}
                        const double t0 = q + 2 <= 2 ? x_first : (xs2 + xs1) / 2.0;
                        const double t1 = q + 3 >= n ? x_last : (xs3 + xs2) / 2.0;
                        const double t2 = q + 4 >= n ? x_last : (xs4 + xs3) / 2.0;
                        double h0, h1, h2;
                        quad_basis_t(tm1, t0, t1, t2, xq, h0, h1, h2);
                        zz = h0 * S(q) + h1 * S(q + 1) + h2 * S(q + 2);
                    }
                }
                else {
                    zz = nanv;
                    if (j >= 0 && (d_extrap_right(METHOD) || xq <= x_last)) {
                        const int jj = j > n - 2 ? n - 2 : j;
                        zz = hermite_eval(X(jj), X(jj + 1), Y(jj), Y(jj + 1), S(jj), S(jj + 1), xq);
                    }
                }
                z[ps * MP_ROWS + r] = zz;
                all_ok = all_ok && !__builtin_isnan(zz);
            }
        }
        // ---- maturity direction (as ivs_surface_masked.hpp): complete columns take the dense register solve, up to MK_MAXCOL
        // columns with a missing value are solved in place by the generic per-column recurrence
        const bool col_masked = act && !all_ok;
        const unsigned long long mm = __ballot(col_masked);
        if (mm != 0ull) {
            if (__popcll(mm) > MK_MAXCOL) { leave(); continue; }
            __syncthreads();                                   // every lane is done with the planes
            double* Tsh = Ksh;
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
            if (__ballot(col_masked && cn > 0 && cn < method_min_knots(METHOD)) != 0ull) { leave(); continue; }
            __syncthreads();
            if (col_masked) {
                const MaskedT cx{Tsh, cti + slot * DT};
                const CView cy{cz + slot * DT, 1};
                View csv{cs + slot * DT, 1}, cpw{ccp + slot * DT, 1};
                if (cn >= 2) method_slopes(METHOD, cx, cy, csv, cpw, cn);
                const CView csr{cs + slot * DT, 1};
                int jc = -1;
                for (int tq = 0; tq < mT; ++tq) {
                    const double x = p.Tq[tq];
                    double r = nanv;
                    if (cn > 0) {
                        while (jc + 1 < cn && cx(jc + 1) <= x) ++jc;
                        r = eval_method(METHOD, cx, cy, csr, cn, jc, x);
                    }
                    outb[(int64_t)tq * mK + lane] = r;
                }
            }
        }
        if (act && all_ok) dense_maturity_pass<METHOD, true, false, false, true>(z, tt, TTp, Wp, outb, 0, lane, true, mT, mK, nostamp);
        if (p.status && lane == 0) p.status[b] = IVS_ST_OK;
      }
    }
}

}  // namespace ivs
