// Row-pass kernels: the fast path of every dense method at 3 wavefronts per SIMD (not-a-knot solve described first).
//
// The one-wavefront-per-surface kernels of ivs_surface_dense*.hpp keep the whole surface (two planes of 16 rows) in
// LDS: 20 KB (64 strikes) or 40 KB (128 strikes), i.e. 8 wavefronts per CU -- and phase ablation (tools/ablate_api.hip,
// profiles/r02/ablation_onepass_*.txt) shows that at 8 wavefronts the arithmetic is simply ADDED to the streaming time: the
// skeleton (staging + stores) already runs at the device's streaming ceiling.  Here the strike direction is processed
// in PASSES of RP rows, the planes hold one pass:
//     64 strikes : 8 rows per pass, 2 passes, lane = (row, segment of 8 knots)     12.8 KB LDS, <= 168 VGPRs: 12 per CU
//    128 strikes : 4 rows per pass, 4 passes, lane = (row, 16 segments of 8 knots)
// The factorisation tables (one K-phase per surface) therefore get LDS of their own instead of aliasing the S plane.
// Per pass: the rs-lanes sweep their 8 knots forward, the carries cross the 8 (16) segments of a row as a Kogge-Stone
// scan of affine maps whose multipliers (products of segment products) are row independent and precomputed in the
// K-phase, then the backward sweep, the backward scan and the fix-up; slopes go to the S plane, the q-lanes gather the
// pass's rows into z[].  The maturity pass is dense_maturity_pass with the batch-wide tables in the scalar cache.
// Layouts: Y plane row stride KCAP + 2 doubles, rows contiguous (b128 reads of the rs-lanes conflict-free for 64
// strikes); S plane the same with the four 16-byte slots of a segment permuted by (segment >> 1) so that the b128 writes
// of eight neighbouring segments hit eight different bank groups; tables at segment stride 10 (b128 broadcast reads of
// 8 or 16 segments conflict-free), the scan multipliers in the two spare slots behind each table segment.
// 128 strikes: rows back to back (stride 128), both planes slot-swizzled over (segment >> 2).
// Other methods on the same structure: quadratic = the same solve on another tridiagonal system, B-spline coefficients in
// the S plane, four taps c[j-1 .. j+2]; linear / slinear / nearest / zero / from_derivatives carry no S plane and no tables
// (a pass = stage, two gathers per row, lerp_fast or step_eval); pchip / akima replace the K-phase and the sweeps by three
// per-knot tables and per-segment local slopes (pass_local_tables / pass_local_slopes).
// Work distribution: work queues (WorkQueue, ivs_surface_generic.hpp), not static striding.
// Scope: T and Tq shared by the batch, mK <= 64 (one block of output strikes: the pass structure would recompute the
// slopes per block), 4..16 maturities, 4..128 strikes; per-surface maturities for `linear` with mT <= 16 (TSH = false);
// 64 x 16 akima, per-surface maturities of the other methods and wider output grids stay on the one-pass kernels;
// `quadratic` runs the not-a-knot machinery on its collocation system (QUADM / quad_weights).
#pragma once
#include "ivs_surface_dense_var2.hpp"
#include "ivs_surface_masked.hpp"

#ifndef IVS_PASS_SL
#define IVS_PASS_SL 8      // knots per segment of the uniform 64 x 16 kernel (8: 12 workgroups / CU, 4: 16)
#endif

#ifndef IVS_PASS_CAP2
// workgroups per CU of the 128-strike kernel (LDS admits 9).  Under STATIC striding 9 lost 8.5 % against 8 (one SIMD of
// every CU carries three wavefronts, its workgroups ran at 2/3 of the others' pace and the launch waited for them); with
// the work queues the uneven wavefront counts balance out: 184.9 vs 175.9 M surfaces/s for the 65..128 class.
#define IVS_PASS_CAP2 12
#endif
#ifndef IVS_PASS_CHUNK
#define IVS_PASS_CHUNK 4      // surfaces per work-queue claim
#endif
#ifndef IVS_PASS_PFP4
#define IVS_PASS_PFP4 2
#endif
#ifndef IVS_PASS_WIDE
// A/B variant, REJECTED (round 3, profiles/r03/ab_cfg5_wide_staging_and_ablations.txt): run-time-shape kernels staging their
// quotes with 16 B per lane (a lane takes TWO neighbouring strikes of a row; 64 strikes: one load covers two rows, 128
// strikes: one row) instead of one row of 8-byte loads with the lanes beyond n idle.  Half the VMEM and LDS-store
// instructions per pass, bit-identical results -- and 2-5 % SLOWER in every size class (65..128: 186.7 -> 183.8 M surfaces/s,
// 8..64: 311.6 -> 298.3): these kernels are bound by their arithmetic phases (the ablations in the same file: K-phase +
// sweeps cost 25-37 % of the call, the no-math skeleton streams at 5.5-5.7 TB/s), not by load / store instruction issue,
// and the pair addressing costs 12-16 B of scratch at 168 VGPRs.
#define IVS_PASS_WIDE 0
#endif
namespace ivs {

template <int NKB, int SL = 8>
struct PassGeom {
    static constexpr int NSEG = 64 * NKB / SL;     // segments of SL (8 or 4) knots per row
    static constexpr int RP = 64 / NSEG;           // rows per pass
    static constexpr int NPASS = DT / RP;
    static constexpr int KCAP = 64 * NKB;
    // plane row stride (doubles).  64 strikes: two spare slots behind a row (read as y_{kb+SL} of its last segment; the odd
    // 16-byte-slot stride also staggers the rows of a b128 lane group over the banks).  128 strikes: rows back to back
    // (y_{kb+SL} of a row's last segment = the next row's / the S plane's first entry: finite, meets a zero coefficient)
    // and BOTH planes slot-swizzled over (segment >> 2), which makes every b128 lane group -- 16 distinct segments of
    // two rows -- conflict-free; with the former stride of 130 segments s and s+4 shared their banks (25 % of the LDS
    // cycles of config 5's upper class were conflicts).
    static constexpr int RS = NKB == 2 ? KCAP : KCAP + 2;
    static constexpr int PLANE = RP * RS;
    static constexpr int TS = SL + 2;              // table segment stride (SL entries + 2 spare)
    static constexpr int TN = NSEG * TS;           // one table
    static constexpr int NSCAN = NSEG == 8 ? 3 : 4; // carry scan steps (shift 1, 2, 4[, 8])
    static_assert(SL == 8 || (SL == 4 && NKB == 1), "segment length");
    static_assert(NSEG == 8 || NSEG == 16, "a row's segments live in one DPP row");
};
// 128 strikes, not-a-knot: the in-segment products PI / PSI are not tabulated but rebuilt as running products inside the
// fix-up loops (the same LDS reads -- AL / CP a second time instead of PI / PSI --, 16 more multiplications per pass), and
// the strikes alias the S plane (they are dead before the first sweep writes it): 16 896 -> 13 328 B = 9 -> 11 workgroups per CU.
template <int NKB> __host__ __device__ constexpr bool pass_runp() { return NKB == 2; }
constexpr int PASS_TQ_DOUBLES = DT * 4 + D_WLDS_MAX_MT * 4;      // per-surface maturity tables TT + W (TSH = false)
template <int NKB, bool VAR, int SL = 8, int KIND = 0>      // KIND 0: not-a-knot, 1: lerp, 2: local slopes
__host__ __device__ constexpr size_t pass_lds_bytes() {
    using G = PassGeom<NKB, SL>;
    // Y, S planes; AL CP PP QQ PI PSI tables (local slopes: R0 R1 R2); Ksh  (lerp methods: Y plane and Ksh only)
    if (KIND == 0 && pass_runp<NKB>()) return (size_t)(2 * G::PLANE + 4 * G::TN + 2) * 8;
    return KIND == 1 ? (size_t)(G::PLANE + G::KCAP) * 8 : (size_t)(2 * G::PLANE + (KIND == 2 ? 3 : 6) * G::TN + G::KCAP) * 8;
}

template <int SL> __device__ __forceinline__ int p_tix(int k) { return (k / SL) * (SL + 2) + (k % SL); }
// S plane, 8-knot segments: 16-byte slot c of segment s sits at c ^ (s >> 1) (conflict-free b128 writes); 4-knot segments
// need no swizzle (the odd row stride already interleaves the two rows of a 16-lane write group)
template <int SL, int NKB = 1> __device__ __forceinline__ int p_swz(int k) {
    return NKB == 2 ? k ^ (((k >> 5) & 3) << 1) : (SL == 8 ? k ^ (((k >> 4) & 3) << 1) : k);
}
// Y plane: swizzled like the S plane for 128 strikes, plain otherwise
template <int NKB> __device__ __forceinline__ int y_swz(int k) { return NKB == 2 ? k ^ (((k >> 5) & 3) << 1) : k; }

// inclusive prefix / suffix products within aligned groups of SL lanes (several groups per DPP row: a shifted value that
// comes from the neighbouring group is replaced by 1.0)
template <int SL>
__device__ __forceinline__ double seg_prefix_prod(double v, int lane) {
    const int i = lane & (SL - 1);
    double t = dpp_f64<DPP_ROW_SHR(1)>(1.0, v); v *= i >= 1 ? t : 1.0;
    t = dpp_f64<DPP_ROW_SHR(2)>(1.0, v); v *= i >= 2 ? t : 1.0;
    if (SL > 4) { t = dpp_f64<DPP_ROW_SHR(4)>(1.0, v); v *= i >= 4 ? t : 1.0; }
    return v;
}
template <int SL>
__device__ __forceinline__ double seg_suffix_prod(double v, int lane) {
    const int i = lane & (SL - 1);
    double t = dpp_f64<DPP_ROW_SHL(1)>(1.0, v); v *= i <= SL - 2 ? t : 1.0;
    t = dpp_f64<DPP_ROW_SHL(2)>(1.0, v); v *= i <= SL - 3 ? t : 1.0;
    if (SL > 4) { t = dpp_f64<DPP_ROW_SHL(4)>(1.0, v); v *= i <= 3 ? t : 1.0; }
    return v;
}

// K-phase: factorisation tables of the not-a-knot system on n knots (n = KCAP when !VAR) at p_tix(k), the segment
// products P_j = prod(-AL) / Q_j = prod(-CP) of every 8-knot segment, and from them the multipliers of the carry scans
// (see pass_sweeps).  Ends with the tables visible to every lane.
template <int NKB, bool VAR, int SL = 8, bool QUADM = false>      // QUADM: the collocation system of the quadratic B-spline instead
__device__ __forceinline__ void pass_factor_tables(const double* X, int n, int lane, double* TB, double* SCR = nullptr) {
    using G = PassGeom<NKB, SL>;
    constexpr int TS = G::TS;
    constexpr int TN = G::TN, NSEG = G::NSEG;
    constexpr bool RUNP = pass_runp<NKB>();         // no PI / PSI tables: segment products go through SCR (free LDS), pm_last behind QQ
    double* AL = TB; double* CP = TB + TN; double* PP = TB + 2 * TN; double* QQ = TB + 3 * TN;
    double* PI = TB + 4 * TN; double* PSI = TB + 5 * TN;
    double c00 = 1.0, c01 = 0.0, c10 = 0.0, c11 = 1.0;          // product of all matrices of the previous blocks
    double carry_crb = 0.0, carry_rdx = 0.0, carry_rdx_prev = 0.0;
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk) {
        const int ir = blk * 64 + lane;
        const bool in = ir < n;
        const int i = in ? ir : n - 1;
        const double x0 = X[i];
        const double xp = X[i + 1 < n ? i + 1 : n - 1];
        const double xpp = X[i + 2 < n ? i + 2 : n - 1];
        const double xm = X[i > 0 ? i - 1 : 0];
        const double xmm = X[i > 1 ? i - 2 : 0];
        const double dxc = xp - x0, dxm = x0 - xm, dxp = xpp - xp, dxmm = xm - xmm;
        const double rdxc = refined_rcp(dxc);
        const bool first = ir == 0, last = ir == n - 1;
        double a, b, c;
        if (QUADM) {
            const CView xv{X, 1};
            quad_row(xv, n, i, a, b, c);
        } else
        if (first) { a = 0.0; b = dxp; c = dxc + dxp; }
        else if (last) { a = dxmm + dxm; b = dxmm; c = 0.0; }
        else { a = dxc; b = 2.0 * (dxm + dxc); c = dxm; }
        if (!in) { a = 0.0; b = 1.0; c = 0.0; }
        const double rb = refined_rcp(b);
        const double crb = c * rb;
        double crb_prev = dpp0_f64<DPP_WAVE_SHR1>(crb);
        if (blk > 0 && lane == 0) crb_prev = carry_crb;
        const bool ident = first || !in;
        const double g = ident ? 0.0 : a * rb * crb_prev;
        double p00 = 1.0, p01 = ident ? 0.0 : -g, p10 = ident ? 0.0 : 1.0, p11 = ident ? 1.0 : 0.0;
        scan_mat2<64>(p00, p01, p10, p11, lane);
        if (blk > 0) {                                           // append the previous blocks' product on the right
            const double n00 = p00 * c00 + p01 * c10, n01 = p00 * c01 + p01 * c11;
            const double n10 = p10 * c00 + p11 * c10, n11 = p10 * c01 + p11 * c11;
            p00 = n00; p01 = n01; p10 = n10; p11 = n11;
        }
        const double num = p00 + p01, den = p10 + p11;
        const double rw = first ? rb : den * rb * refined_rcp(num);
        const double al = a * rw, cp = c * rw;
        double rdx_prev = dpp0_f64<DPP_WAVE_SHR1>(rdxc);
        if (blk > 0 && lane == 0) rdx_prev = carry_rdx;
        const double rdx_next = dpp0_f64<DPP_WAVE_SHL1>(rdxc);          // only row 0 uses it (never crosses a block)
        double rdxmm = dpp0_f64<DPP_WAVE_SHR1>(rdx_prev);
        if (blk > 0 && lane == 0) rdxmm = carry_rdx_prev;
        const double d = first ? dxc + dxp : dxmm + dxm;
        const double rd = refined_rcp(d);
        double pm = 0.0, pp, qq;
        if (QUADM) {
            pp = rw; qq = 0.0;                                   // right-hand side = y_i / pivot
        } else
        if (first) {
            pp = (dxc + 2.0 * d) * dxp * rdxc * rd * rw;         // * dy_0
            qq = dxc * dxc * rdx_next * rd * rw;                 // * dy_1
        } else if (last) {
            pm = dxm * dxm * rdxmm * rd * rw;                    // * dy_{n-3}
            pp = (2.0 * d + dxm) * dxmm * rdx_prev * rd * rw;    // * dy_{n-2}
            qq = 0.0;
            if (!VAR) { qq = pp; pp = pm; }                      // fixed n: the sweep feeds (dy_{n-3}, dy_{n-2}) to the last row
        } else {
            pp = 3.0 * dxc * rdx_prev * rw;                      // * dy_{i-1}
            qq = 3.0 * dxm * rdxc * rw;                          // * dy_i
        }
        const double pi = seg_prefix_prod<SL>(in ? -al : 1.0, lane);
        const double psi = seg_suffix_prod<SL>(in ? -cp : 1.0, lane);
        {   // beyond n the NEUTRAL row (AL = -1, everything else 0): the forward sweep holds its value, the last system
            // row's CP = 0 cuts the backward recurrence off from whatever lies to its right (see factor_tables_var)
            const int kl = p_tix<SL>(ir);
            AL[kl] = in ? al : -1.0; CP[kl] = in ? cp : 0.0; PP[kl] = in ? pp : 0.0; QQ[kl] = in ? qq : 0.0;
            if (RUNP) {
                if (VAR && last) TB[4 * TN] = pm;
                if ((lane & (SL - 1)) == SL - 1) SCR[ir / SL] = pi;
                if ((lane & (SL - 1)) == 0) SCR[NSEG + ir / SL] = in ? psi : 0.0;
            } else {
                if (VAR && last) PI[SL + 1] = pm;      // the third tap exists in the last system row only: one scalar, not a table
                PI[kl] = pi; PSI[kl] = psi;
                if ((lane & (SL - 1)) == SL - 1) PI[(ir / SL) * TS + SL] = pi;               // P_j: product of (-AL) over segment j (spare slot)
                if ((lane & (SL - 1)) == 0) PSI[(ir / SL) * TS + SL] = in ? psi : 0.0;       // Q_j: product of (-CP)
            }
        }
        if (blk + 1 < NKB) {
            c00 = readlane_f64(p00, 63); c01 = readlane_f64(p01, 63); c10 = readlane_f64(p10, 63); c11 = readlane_f64(p11, 63);
            carry_crb = readlane_f64(crb, 63); carry_rdx = readlane_f64(rdxc, 63); carry_rdx_prev = readlane_f64(rdx_prev, 63);
        }
    }
    __syncthreads();
    {   // scan multipliers: forward step s (1, 2, 4, 8) multiplies the value 2^s segments to the left by
        // P_j P_{j-1} .. P_{j-s+1}; 0 where no such segment exists (it also silences the DPP sources of the neighbouring
        // row when two rows share a DPP row).  Backward: Q_j .. Q_{j+s-1} and the segment to the right.
        const int j = lane < NSEG ? lane : NSEG - 1;
        double pw = lane < NSEG ? (RUNP ? SCR[j] : PI[j * TS + SL]) : 1.0, qw = lane < NSEG ? (RUNP ? SCR[NSEG + j] : PSI[j * TS + SL]) : 1.0;
        double fm[4], bm[4];
#pragma unroll
        for (int s = 0; s < G::NSCAN; ++s) {
            fm[s] = j >= (1 << s) ? pw : 0.0;
            bm[s] = j + (1 << s) <= NSEG - 1 ? qw : 0.0;
            if (s == 0) { pw *= dpp_f64<DPP_ROW_SHR(1)>(1.0, pw); qw *= dpp_f64<DPP_ROW_SHL(1)>(1.0, qw); }
            if (s == 1) { pw *= dpp_f64<DPP_ROW_SHR(2)>(1.0, pw); qw *= dpp_f64<DPP_ROW_SHL(2)>(1.0, qw); }
            if (s == 2) { pw *= dpp_f64<DPP_ROW_SHR(4)>(1.0, pw); qw *= dpp_f64<DPP_ROW_SHL(4)>(1.0, qw); }
        }
        if (lane < NSEG) {      // the two spare slots of segment j in the AL / CP / PP / QQ tables
            AL[j * TS + SL] = fm[0]; AL[j * TS + SL + 1] = fm[1]; CP[j * TS + SL] = fm[2];
            PP[j * TS + SL] = bm[0]; PP[j * TS + SL + 1] = bm[1]; QQ[j * TS + SL] = bm[2];
            if (NSEG > 8) { CP[j * TS + SL + 1] = fm[3]; QQ[j * TS + SL + 1] = bm[3]; }
        }
    }
    __syncthreads();
}

// One pass: slopes of the RP rows staged in Yp -> Sp.  All 64 lanes; no barrier inside (the caller brackets it).
template <int NKB, bool VAR, int SL = 8, bool QUADM = false>
__device__ __forceinline__ void pass_sweeps(const double* Yp, double* Sp, const double* TB, int lane, int n) {
    using G = PassGeom<NKB, SL>;
    constexpr int TN = G::TN, NSEG = G::NSEG, RS = G::RS, TS = G::TS;
    const double* AL = TB; const double* CP = TB + TN; const double* PP = TB + 2 * TN; const double* QQ = TB + 3 * TN;
    constexpr bool RUNP = pass_runp<NKB>();
    const double* PI = TB + 4 * TN; const double* PSI = TB + 5 * TN;
    const double pm_last = VAR ? (RUNP ? TB[4 * TN] : PI[SL + 1]) : 0.0;  // run-time n: third tap of the last system row (wave-uniform)
    const int tl = lane / NSEG, seg = lane % NSEG;
    const int kb = seg * SL, tb = seg * TS;
    const bool s_first = seg == 0, s_last = seg == NSEG - 1;
    const int mlast = n - 1 - kb;                   // position of the last system row in this lane's segment (VAR)
    const double* yr = Yp + tl * RS + kb;
    const int fy = NKB == 2 ? (seg >> 2) & 3 : 0;   // slot swizzle of this segment in the Y plane
    double y[SL + 3];                                   // y[j] = y_{kb + j - 2}
#pragma unroll
    for (int c = 0; c < SL / 2; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(yr + 2 * (c ^ fy));
        y[2 + 2 * c] = v.x; y[3 + 2 * c] = v.y;
    }
    {   // the two knots to the left (segment 0: re-reads its own first pair, selected away / multiplied by 0 below)
        const int fl = NKB == 2 ? ((seg - 1) >> 2) & 3 : 0;
        const double2 v = *reinterpret_cast<const double2*>(s_first ? yr : yr - SL + 2 * ((SL / 2 - 1) ^ fl));
        y[0] = v.x; y[1] = v.y;
    }
    {   // first knot of the next segment (last segment: the spare slot behind the row / the next row's first entry)
        const int fn = NKB == 2 ? ((seg + 1) >> 2) & 3 : 0;
        y[SL + 2] = yr[SL + 2 * fn];
    }
    auto tab2 = [&](const double* T, int m) { return *reinterpret_cast<const double2*>(T + tb + m); };   // entries m, m+1
    double d[SL];
    double prev = 0.0;
#pragma unroll
    for (int mm = 0; mm < SL; mm += 2) {
        const double2 tpp = tab2(PP, mm), tqq = tab2(QQ, mm), tal = tab2(AL, mm);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = mm + u;
            double dM = y[m + 1] - y[m], dA = y[m + 2] - y[m + 1], dB = y[m + 3] - y[m + 2];   // dy_{i-2}, dy_{i-1}, dy_i
            if (m == 0) { const double e = y[4] - y[3]; dA = s_first ? dB : dA; dB = s_first ? e : dB; }      // row 0: (dy_0, dy_1)
            if (!VAR && m == SL - 1) { dB = s_last ? dA : dB; dA = s_last ? dM : dA; }                             // row n-1: (dy_{n-3}, dy_{n-2})
            double r = (u ? tpp.y : tpp.x) * dA + (u ? tqq.y : tqq.x) * dB;
            if (QUADM) r = (u ? tpp.y : tpp.x) * y[m + 2];
            else if (VAR) r = (m == mlast) ? __builtin_fma(pm_last, dM, r) : r;
            prev = r - (u ? tal.y : tal.x) * prev;
            d[m] = prev;
        }
        if (SL == 8 && mm == 2) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- forward carries: tot_j = E_j + P_j tot_{j-1} over the segments of the row, Kogge-Stone on the E part
    const double2 fm01 = *reinterpret_cast<const double2*>(AL + tb + SL);
    const double2 fm23 = *reinterpret_cast<const double2*>(CP + tb + SL);
    double e = prev;
    e = __builtin_fma(fm01.x, dpp0_f64<DPP_ROW_SHR(1)>(e), e);
    e = __builtin_fma(fm01.y, dpp0_f64<DPP_ROW_SHR(2)>(e), e);
    e = __builtin_fma(fm23.x, dpp0_f64<DPP_ROW_SHR(4)>(e), e);
    if (NSEG > 8) e = __builtin_fma(fm23.y, dpp0_f64<DPP_ROW_SHR(8)>(e), e);
    double din = dpp0_f64<DPP_ROW_SHR(1)>(e);
    din = s_first ? 0.0 : din;
    // ---- local backward sweep with the forward fix-up folded in
    double nxt = 0.0;
    if (RUNP) {      // PI rebuilt as a running product of (-AL), front to back, before the backward sweep
        double pi = 1.0;
#pragma unroll
        for (int mm = 0; mm < SL; mm += 2) {
            const double2 tal = tab2(AL, mm);
            pi *= -tal.x; d[mm] = __builtin_fma(pi, din, d[mm]);
            pi *= -tal.y; d[mm + 1] = __builtin_fma(pi, din, d[mm + 1]);
        }
    }
#pragma unroll
    for (int mm = SL - 2; mm >= 0; mm -= 2) {
        double2 tpi = double2{0.0, 0.0};
        if (!RUNP) tpi = tab2(PI, mm);
        const double2 tcp = tab2(CP, mm);
#pragma unroll
        for (int u = 1; u >= 0; --u) {
            const int m = mm + u;
            const double dp = RUNP ? d[m] : d[m] + (u ? tpi.y : tpi.x) * din;
            nxt = dp - (u ? tcp.y : tcp.x) * nxt;
            d[m] = nxt;
        }
        if (SL == 8 && mm == 4) __builtin_amdgcn_sched_barrier(0);
    }
    // ---- backward carries: first_j = F_j + Q_j first_{j+1}
    const double2 bm01 = *reinterpret_cast<const double2*>(PP + tb + SL);
    const double2 bm23 = *reinterpret_cast<const double2*>(QQ + tb + SL);
    double f = nxt;
    f = __builtin_fma(bm01.x, dpp0_f64<DPP_ROW_SHL(1)>(f), f);
    f = __builtin_fma(bm01.y, dpp0_f64<DPP_ROW_SHL(2)>(f), f);
    f = __builtin_fma(bm23.x, dpp0_f64<DPP_ROW_SHL(4)>(f), f);
    if (NSEG > 8) f = __builtin_fma(bm23.y, dpp0_f64<DPP_ROW_SHL(8)>(f), f);
    double sin_ = dpp0_f64<DPP_ROW_SHL(1)>(f);
    sin_ = s_last ? 0.0 : sin_;
    double* srow = Sp + tl * RS + kb;
    const int sx = NKB == 2 ? (seg >> 2) & 3 : (SL == 8 ? (seg >> 1) & 3 : 0);
    if (RUNP) {      // PSI rebuilt as a running product of (-CP), back to front
        double psi = 1.0;
#pragma unroll
        for (int c = SL / 2 - 1; c >= 0; --c) {
            const double2 tcp = tab2(CP, 2 * c);
            double2 v;
            psi *= -tcp.y; v.y = __builtin_fma(psi, sin_, d[2 * c + 1]);
            psi *= -tcp.x; v.x = __builtin_fma(psi, sin_, d[2 * c]);
            *reinterpret_cast<double2*>(srow + 2 * (c ^ sx)) = v;
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < SL / 2; ++c) {
        const double2 tps = tab2(PSI, 2 * c);
        double2 v;
        v.x = d[2 * c] + tps.x * sin_; v.y = d[2 * c + 1] + tps.y * sin_;
        *reinterpret_cast<double2*>(srow + 2 * (c ^ sx)) = v;
    }
}

// ---- local-slope methods (pchip, akima) on the row-pass structure: no system to solve, three per-knot tables
// R0 = 1/dx_k (0 from interval n-1 on), R1 / R2 = pchip's weights (the one-sided end coefficients at knots 0 and n-1),
// rule set of dense_strike_slopes_local[_var] (ivs_surface_dense[_var].hpp) re-cut for 8-knot segments.
template <int NKB, bool VAR>
__device__ __forceinline__ void pass_local_tables(const double* X, int n, int lane, double* TB) {
    using G = PassGeom<NKB, 8>;
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk) {
        const int k = blk * 64 + lane;
        double r0, r1, r2;
        local_tables_rt(X, n, k, r0, r1, r2);
        const int kl = p_tix<8>(k);
        TB[kl] = r0; TB[G::TN + kl] = r1; TB[2 * G::TN + kl] = r2;
    }
    __syncthreads();
}
// One pass: local slopes of the RP rows staged in Yp -> Sp.  Lane = (row of the pass, 8-knot segment).  No barrier inside.
template <int METHOD, int NKB, bool VAR>
__device__ __forceinline__ void pass_local_slopes(const double* Yp, double* Sp, const double* TB, int lane, int n) {
    using G = PassGeom<NKB, 8>;
    constexpr bool AK = METHOD == IVS_AKIMA;
    constexpr int TN = G::TN, NSEG = G::NSEG, RS = G::RS, TS = G::TS;
    const double* R0 = TB; const double* R1 = TB + TN; const double* R2 = TB + 2 * TN;
    const int tl = lane / NSEG, seg = lane % NSEG;
    const int kb = seg * 8, tb = seg * TS;
    const bool s_first = seg == 0;
    const double* yrow = Yp + tl * RS;
    const double* yr = yrow + kb;
    const int fy = NKB == 2 ? (seg >> 2) & 3 : 0;
    double y[12];                                       // y[i] = y_{kb - 2 + i}
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(yr + 2 * (c ^ fy));
        y[2 + 2 * c] = v.x; y[3 + 2 * c] = v.y;
    }
    {   // two knots to the left (segment 0: its own first pair, never used) and two to the right (last segment: the spare
        // slots behind the row / the next row's first pair: finite, meet secant factors that are replaced or multiplied by 0)
        const int fl = NKB == 2 ? ((seg - 1) >> 2) & 3 : 0, fn = NKB == 2 ? ((seg + 1) >> 2) & 3 : 0;
        const double2 l = *reinterpret_cast<const double2*>(s_first ? yr : yr - 8 + 2 * (3 ^ fl));
        const double2 r = *reinterpret_cast<const double2*>(yr + 8 + 2 * fn);
        y[0] = l.x; y[1] = l.y; y[10] = r.x; y[11] = r.y;
    }
    double F[11];                                       // F[i] = secant of interval kb - 2 + i
    {
        const double2 rl = *reinterpret_cast<const double2*>(R0 + (s_first ? tb : tb - TS + 6));       // intervals kb-2, kb-1
        F[0] = (y[1] - y[0]) * rl.x; F[1] = (y[2] - y[1]) * rl.y;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double2 rr = *reinterpret_cast<const double2*>(R0 + tb + 2 * c);
            F[2 + 2 * c] = (y[3 + 2 * c] - y[2 + 2 * c]) * rr.x; F[3 + 2 * c] = (y[4 + 2 * c] - y[3 + 2 * c]) * rr.y;
        }
        F[10] = (y[11] - y[10]) * R0[tb + TS];                                                          // interval kb+8 (akima only)
    }
    double thr = 0.0;
    if (AK) {   // secants left of knot 0 and from interval n-1 on: linear extension F(i) = 2 F(i-1) - F(i-2)
        const double l1 = 2.0 * F[2] - F[3], l0 = 2.0 * l1 - F[2];
        F[1] = s_first ? l1 : F[1]; F[0] = s_first ? l0 : F[0];
#pragma unroll
        for (int i = 2; i < 11; ++i) {
            const int idx = kb - 2 + i;
            F[i] = (idx == n - 1 || idx == n) ? 2.0 * F[i - 1] - F[i - 2] : F[i];
        }
        double fmax = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const double f = akima_f12(F[m], F[m + 1], F[m + 2], F[m + 3]);
            fmax = (kb + m < n) ? __builtin_fmax(fmax, f) : fmax;
        }
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP1>(fmax, fmax));      // the row's 8 (16) segments: aligned lane groups
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP2>(fmax, fmax));
        // the other quad of the 8-lane group / the other half of the 16-lane row: every lane of a quad (half) holds the same
        // value by now, so the mirrored lane serves -- DPP moves instead of two ds_bpermute round trips per pass
        fmax = __builtin_fmax(fmax, dpp_f64<0x141>(fmax, fmax));               // row_half_mirror: lane i <- lane 7 - i (of its 8)
        if (NSEG > 8) fmax = __builtin_fmax(fmax, dpp_f64<0x140>(fmax, fmax)); // row_mirror: lane i <- lane 15 - i (of its 16)
        thr = 1e-9 * fmax;
    }
    // pchip: one-sided rule at the last knot n-1, wherever it falls: computed once from LDS, selected in below
    const int mlast = n - 1 - kb;
    double e_last = 0.0;
    if (!AK) {
        const int k1 = y_swz<NKB>(n - 1), k2 = y_swz<NKB>(n - 2), k3 = y_swz<NKB>(n - 3);
        const int t1 = p_tix<8>(n - 1), t2 = p_tix<8>(n - 2), t3 = p_tix<8>(n - 3);
        const double m2 = (yrow[k1] - yrow[k2]) * R0[t2], m3 = (yrow[k2] - yrow[k3]) * R0[t3];
        e_last = pchip_edge(m2, m3, R1[t1], R2[t1]);
    }
    double d[8];
#pragma unroll
    for (int mm = 0; mm < 8; mm += 2) {
        double2 w1 = double2{0.0, 0.0}, w2 = double2{0.0, 0.0};
        if (!AK) { w1 = *reinterpret_cast<const double2*>(R1 + tb + mm); w2 = *reinterpret_cast<const double2*>(R2 + tb + mm); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = mm + u;
            double v;
            if (AK) v = akima_knot(F[m], F[m + 1], F[m + 2], F[m + 3], thr);
            else {
                v = pchip_knot(F[m + 1], F[m + 2], u ? w1.y : w1.x, u ? w2.y : w2.x);
                if (m == 0) { const double e = pchip_edge(F[2], F[3], w1.x, w2.x); v = s_first ? e : v; }
                v = (m == mlast) ? e_last : v;
            }
            d[m] = v;
        }
        if (mm == 2) __builtin_amdgcn_sched_barrier(0);
    }
    double* srow = Sp + tl * RS + kb;
    const int sx = NKB == 2 ? (seg >> 2) & 3 : (seg >> 1) & 3;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double2 v; v.x = d[2 * c]; v.y = d[2 * c + 1];
        *reinterpret_cast<double2*>(srow + 2 * (c ^ sx)) = v;
    }
}

// VAR = false: uniform batch of 64 x 16 surfaces (BASELINE configs 2/3), surface -> workgroup mapping as surface_dense_kernel.
// VAR = true : work list of a size class (n <= 64 * NKB strikes per surface, run-time maturity count), or a uniform
//              batch with nK != 64.
#ifdef IVS_PASS_ENDSTAMP      // diagnostic build (tools/pass_api.hip): start / end time of every workgroup, 100 MHz wall clock
__device__ unsigned long long* d_pass_ends = nullptr;
#endif

#ifndef IVS_PCHIP3
#define IVS_PCHIP3 3      // bit 0: pchip NT16 <= 64 strikes at 3 wavefronts per SIMD (161 VGPRs, no scratch); bit 1: 65..128 strikes too (48 B scratch)
#endif
#ifndef IVS_AKIMA3
#define IVS_AKIMA3 3      // the same for akima (164 / 166 VGPRs with the rolling maturity slopes, IVS_AKIMA_ROLL; 240 before)
#endif
__host__ __device__ constexpr bool pass_local_three(int method, int nkb, bool nt16) {
    return nt16 && ((method == IVS_PCHIP && ((IVS_PCHIP3 >> (nkb - 1)) & 1)) || (method == IVS_AKIMA && ((IVS_AKIMA3 >> (nkb - 1)) & 1)));
}
// NT16 (VAR only): the batch has the full 16 maturities (BASELINE config 5): the run-time maturity count -- masked rows, the
// third tap of the last system row, the select chain for the hold row -- compiles away (~60 VALU instructions per surface).
template <int METHOD, int NKB, bool VAR, int SL = 8, bool TSH = true, bool NT16 = false>
// Wavefronts per SIMD: 3 (168 VGPRs); the run-time-shape instantiations of pchip / akima need ~200 (the per-lane maturity
// solve of the local rules) and run at 2 without scratch -- at 3 they spilled 14-36 registers and lost 3-30 %.
__global__ __launch_bounds__(64, SL == 4 ? 4 : ((d_is_local(METHOD) && VAR && !pass_local_three(METHOD, NKB, NT16)) ? 2 : 3))
void surface_pass_kernel(SurfaceParams p, VarList list) {
#ifdef IVS_PASS_ENDSTAMP
    if (threadIdx.x == 0 && d_pass_ends) d_pass_ends[blockIdx.x * 2] = wall_clock64();
#endif
    if (!VAR && p.mode && *((const int IVS_CONST*)p.mode)) return;      // "missing quotes first": the compaction kernel takes the batch
    using G = PassGeom<NKB, SL>;
    constexpr int RP = G::RP, NPASS = G::NPASS, KCAP = G::KCAP, RS = G::RS, TN = G::TN;
    // passes in flight per lane (8-knot segments, 64 strikes: a whole surface; the local-slope methods need the registers: one pass)
    constexpr int PFP = (NKB > 1 || d_is_local(METHOD)) ? 1 : (SL == 4 ? IVS_PASS_PFP4 : NPASS);
    constexpr int PPL = RP * NKB;                    // doubles per lane and pass
    static_assert(NPASS % PFP == 0, "prefetch slots rotate with the passes");
    constexpr bool STEP = d_is_step(METHOD);                                   // nearest / zero / from_derivatives
    constexpr bool LERP = METHOD == IVS_LINEAR || METHOD == IVS_SLINEAR || STEP;     // no slopes: only the Y plane and the strikes live in LDS
    constexpr bool LOCAL = d_is_local(METHOD);                                  // pchip / akima: three tables instead of six, no sweeps
    constexpr bool QUADK = d_is_quad(METHOD);                                   // quadratic B-spline: the not-a-knot machinery on another system
    static_assert(LERP || LOCAL || QUADK || METHOD == IVS_CUBIC || METHOD == IVS_CUBICSPLINE, "methods of the dense kernels");
    static_assert(!LOCAL || SL == 8, "local slopes are cut for 8-knot segments");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int mT = p.mT, mK = p.mK;
    double* Yp = reinterpret_cast<double*>(smem);
    double* Sp = Yp + G::PLANE;
    double* TB = Sp + G::PLANE;
    constexpr bool RUNP = !LERP && !LOCAL && pass_runp<NKB>();
    // strikes: behind the tables; 128-strike not-a-knot kernels park them in the S plane (dead before the first sweep writes it)
    double* Ksh = LERP ? Yp + G::PLANE : (RUNP ? Sp : TB + (LOCAL ? 3 : 6) * TN);
    const double nanv = __builtin_nan(""), inf = __builtin_inf();
    auto nostamp = [](int) {};

    // TSH = false: maturities / query maturities differ per surface (time to expiry shrinks from snapshot to snapshot): the
    // T-phase runs per surface into LDS tables of its own (TT, W: 1 KB behind everything else), scratch = the planes, which
    // are free between two surfaces; mT <= 16 (weights in LDS).  Shipped for `linear` only: the not-a-knot instantiations hold
    // the solve tables in VGPRs (no scalar cache without batch-wide tables), spilled 70-80 registers and ran at 153 instead of
    // the one-pass kernel's 242 M surfaces/s; at 2 wavefronts per SIMD (no scratch) they reach 227 M -- still behind it.
    static_assert(TSH || METHOD == IVS_LINEAR, "per-surface maturities: linear only (the not-a-knot instantiations spilled 70-80 registers)");
    double* TTl = (LERP ? Ksh + KCAP : (RUNP ? TB + 4 * TN + 2 : Ksh + KCAP));
    double* Wl = TTl + DT * 4;
    TqTables tt;
    const double* TTp = nullptr;
    const double* Wp = nullptr;
    if (TSH) tq_from_shared(p.tqs, tt, TTp, Wp);
    constexpr bool NTR = VAR && !NT16;               // run-time maturity count
    const int nT = NTR ? p.nT : DT;

    // spare slots that are read but never staged must hold finite numbers (they meet zero coefficients)
    for (int i = lane; i < (LERP ? 1 : 2) * G::PLANE; i += 64) Yp[i] = 0.0;

    const bool kq_shared = p.kq_stride == 0;
    const bool act = lane < mK;
    double xq = (kq_shared && act) ? p.Kq[lane] : nanv;

    // ---- work distribution: work queues (WorkQueue, ivs_surface_generic.hpp)
    bool told = false;
    WorkQueue wq;
    if (!VAR) wq.init(p.queue, p.map_groups, p.B, IVS_PASS_CHUNK, lane);
    else wq.init(p.queue + list.qslot * 8 * QUEUE_STRIDE, 8, list.items ? (int64_t)*list.count : p.B, IVS_PASS_CHUNK, lane);
    auto at = [&](int64_t i, int& n, int64_t& koff) -> int64_t {
        if (VAR && list.items) { const VarItem v = list.items[i]; n = v.n; koff = v.koff; return v.b; }
        n = p.nK; koff = i * p.k_stride; return i;
    };

    // ---- prefetch registers: pass ps of a surface lands in slot ps % PFP, requested PFP passes ahead of its staging
    double pre[PFP * PPL], pre_k[NKB];
    auto issue_pass = [&](int64_t b, int64_t ko, int nn, int ps) {
        const int slot = ps % PFP;
        if (!VAR) {      // chunk c of the surface = 16 B at c*1024 + lane*16: rows 2c + (lane >> 5), strikes 2(lane & 31), +1
            const double2* s2 = reinterpret_cast<const double2*>(p.sigma + b * (int64_t)(DT * DK));
#pragma unroll
            for (int c = 0; c < RP / 2; ++c) {
                const double2 v = s2[(ps * (RP / 2) + c) * 64 + lane];
                pre[slot * RP + 2 * c] = v.x; pre[slot * RP + 2 * c + 1] = v.y;
            }
        } else {
            const double* sb = p.k_off ? p.sigma + (int64_t)nT * ko : p.sigma + b * (int64_t)nT * p.nK;
#if IVS_PASS_WIDE
            // lane = (row of a row pair, strike pair): rows are nn doubles apart, so a pair is 8-byte aligned only (global
            // loads take that); the last strike of an odd row is fetched alone -- its pair would reach into the next row,
            // and behind the last row of the batch past the end of the caller's array
            typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));
            constexpr int LPR = 32 * NKB;                        // lanes per row
#pragma unroll
            for (int c = 0; c < PPL / 2; ++c) {
                const int t = ps * RP + (NKB == 1 ? 2 * c + (lane >> 5) : c), k = 2 * (lane & (LPR - 1));
                const double* src = sb + (int64_t)t * nn + k;
                double2 v = double2{0.0, 0.0};                   // rows beyond nT, strikes beyond n: zeros
                if (t < nT && k + 1 < nn) { const v2d_a8 w = *reinterpret_cast<const v2d_a8*>(src); v.x = w.x; v.y = w.y; }
                else if (t < nT && k < nn) v.x = *src;
                pre[slot * PPL + 2 * c] = v.x; pre[slot * PPL + 2 * c + 1] = v.y;
            }
#else
#pragma unroll
            for (int r = 0; r < RP; ++r)
#pragma unroll
                for (int blk = 0; blk < NKB; ++blk) {
                    const int t = ps * RP + r, k = blk * 64 + lane;
                    pre[(slot * RP + r) * NKB + blk] = (t < nT && k < nn) ? sb[(int64_t)t * nn + k] : 0.0;      // rows beyond nT: zeros
                }
#endif
        }
        if (ps == 0) {
#pragma unroll
            for (int blk = 0; blk < NKB; ++blk) {
                const int k = blk * 64 + lane;
                pre_k[blk] = VAR ? (k < nn ? p.K[ko + k] : inf) : p.K[b * p.k_stride + k];
            }
        }
    };

    int n = KCAP, n_next = KCAP;
    int64_t koff = 0, koff_next = 0, b = 0, b_next = 0;
    int64_t it = wq.take();
    int64_t it_next = it >= 0 ? wq.take() : -1;
    if (it >= 0) {
        b = at(it, n, koff);
#pragma unroll
        for (int ps = 0; ps < PFP; ++ps) issue_pass(b, koff, n, ps);
    }

    // Strike-dependent state survives from surface to surface: consecutive snapshots of one option chain (and every batch with
    // a shared strike grid) repeat their strikes, and then the K-phase -- a quarter of the kernel's arithmetic -- is skipped.
    int j = 0, jj = 0;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    bool l_left = false, l_right = false, l_hold = false, l_slow = false;
    double kprev[NKB];
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk) kprev[blk] = nanv;
    int n_prev = -1;
    bool tables_ok = false;

    while (it >= 0) {
        double* outb = p.out + b * (int64_t)mT * mK;
        const bool more = it_next >= 0;
        if (more) b_next = at(it_next, n_next, koff_next);
        const WorkQueue::Pending pend = wq.begin(more);       // the surface after next may open a new chunk: claim it now
        if (!kq_shared) xq = act ? p.Kq[b * p.kq_stride + lane] : nanv;      // issued ahead of the next prefetch (vmcnt is in order)
        if (!TSH) {
            __syncthreads();                                   // the previous surface is done with the planes and with TT / W
            dense_t_phase<METHOD, true, NTR>(p.T + b * p.t_stride, p.Tq + b * p.tq_stride, mT, lane, LERP ? Yp : Yp + 600, TTl, Wl, tt, nT, Yp);
            // the scratch may leave non-finite numbers in the rows' spare slots (they meet zero coefficients): clear them
            if (NKB == 1 && !LERP && lane < 2 * RP) { Yp[(lane >> 1) * RS + KCAP + (lane & 1)] = 0.0; Sp[(lane >> 1) * RS + KCAP + (lane & 1)] = 0.0; }
        }
        bool ok = !tt.unsorted;
        double z[DT];
        bool same_k = false;                               // this surface's strikes are the previous surface's: tables (and, with a
                                                           // shared query grid, the interval search and weights) are kept
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int slot = ps % PFP;
            __syncthreads();                                   // the previous pass's gathers are done with the planes
            // ---- stage pass ps; any non-finite quote (NaN = missing, or an infinity) sends the surface to the generic kernel
            double acc = 0.0;
            if (!VAR) {
#pragma unroll
                for (int c = 0; c < RP / 2; ++c) {
                    const int tl = 2 * c + (lane >> 5), k = 2 * (lane & 31);
                    double2 v; v.x = pre[slot * RP + 2 * c]; v.y = pre[slot * RP + 2 * c + 1];
                    *reinterpret_cast<double2*>(&Yp[tl * RS + k]) = v;
                }
#pragma unroll
                for (int c = 0; c < RP; ++c) acc = __builtin_fma(pre[slot * RP + c], 0.0, acc);
            } else {
#if IVS_PASS_WIDE
#pragma unroll
                for (int c = 0; c < PPL / 2; ++c) {
                    const int tl = NKB == 1 ? 2 * c + (lane >> 5) : c, k = 2 * (lane & (32 * NKB - 1));
                    double2 v; v.x = pre[slot * PPL + 2 * c]; v.y = pre[slot * PPL + 2 * c + 1];
                    *reinterpret_cast<double2*>(&Yp[tl * RS + y_swz<NKB>(k)]) = v;       // the slot swizzle keeps a pair together
                    acc = __builtin_fma(v.x, 0.0, acc); acc = __builtin_fma(v.y, 0.0, acc);
                }
#else
#pragma unroll
                for (int r = 0; r < RP; ++r)
#pragma unroll
                    for (int blk = 0; blk < NKB; ++blk) {
                        const double v = pre[(slot * RP + r) * NKB + blk];
                        Yp[r * RS + y_swz<NKB>(blk * 64 + lane)] = v;
                        acc = __builtin_fma(v, 0.0, acc);
                    }
#endif
            }
            if (ps == 0) {
                bool same = tables_ok && n == n_prev;
#pragma unroll
                for (int blk = 0; blk < NKB; ++blk) {
                    Ksh[blk * 64 + lane] = pre_k[blk];
                    same = same && __ballot(pre_k[blk] != kprev[blk]) == 0ull;      // NaN strikes never compare equal
                    kprev[blk] = pre_k[blk];
                }
                n_prev = n;
                same_k = same;
            }
            ok = ok && __ballot(acc != 0.0) == 0ull;
            if (ps == 0) tables_ok = ok;                       // a surface that is redone elsewhere leaves no tables behind
            {                                                  // the slot's registers are free: request the pass PFP ahead
                const int nx = ps + PFP;
                if (nx < NPASS) issue_pass(b, koff, n, nx);
                else if (more) issue_pass(b_next, koff_next, n_next, nx - NPASS);
            }
            __syncthreads();
            if (ok) {
                if (ps == 0 && !same_k) {
                    if (LOCAL) pass_local_tables<NKB, VAR>(Ksh, n, lane, TB);
                    else if (!LERP && ABL != 1 && ABL != 6) pass_factor_tables<NKB, VAR, SL, QUADK>(Ksh, n, lane, TB, Sp + KCAP);
                }
                if (ps == 0 && !(same_k && kq_shared)) {
                    // ---- strike search + weights of this lane's output strike (once per strike grid / query grid)
                    j = 0;
#pragma unroll
                    for (int m = 1; m < 8 * NKB; ++m) j += (Ksh[8 * m] <= xq) ? 8 : 0;
#pragma unroll
                    for (int st = 4; st >= 1; st >>= 1) if (Ksh[j + st] <= xq) j += st;
                    const double xl = Ksh[n - 1];
                    const bool left = !(Ksh[0] <= xq);
                    jj = j > n - 2 ? n - 2 : j;
                    const double x0 = Ksh[jj], x1 = Ksh[jj + 1];
                    if (LERP) {      // np.interp's interval data; w0..w3 = x0, x1, dx, 1/dx (shared by the 16 rows)
                        l_left = left; l_right = j >= n - 1;
                        l_hold = l_right && (METHOD == IVS_LINEAR || xq == xl);
                        w0 = x0; w1 = x1; w2 = x1 - x0; w3 = refined_rcp(w2);
                        l_slow = !STEP && !div_safe(w2);
                    } else if (QUADK) {      // weights of the coefficients c[jj-1 .. jj+2]
                        const CView kx{Ksh, 1};
                        quad_weights(kx, n, j, xq, w0, w1, w2, w3);
                        if (left || !(xq <= xl)) w0 = nanv;
                    } else {
                        const bool okq = !left && ((xq <= xl) || d_extrap_right(METHOD));
                        const double u = xq - x0, t = u * refined_rcp(x1 - x0), omt = 1.0 - t;
                        w0 = okq ? (1.0 + 2.0 * t) * omt * omt : nanv;
                        w1 = t * t * (3.0 - 2.0 * t);
                        w2 = u * omt * omt;
                        w3 = u * t * (t - 1.0);
                    }
                }
                const int o0 = y_swz<NKB>(jj), o1 = y_swz<NKB>(jj + 1), q0 = p_swz<SL, NKB>(jj), q1 = p_swz<SL, NKB>(jj + 1);
                if (LERP) {      // two gathers per row, RP in flight; the same arithmetic (and bits) as the one-pass kernel
                    double g0[RP], g1[RP];
#pragma unroll
                    for (int r = 0; r < RP; ++r) { g0[r] = Yp[r * RS + o0]; g1[r] = Yp[r * RS + o1]; }
                    bool slow = l_slow;
#pragma unroll
                    for (int r = 0; r < RP; ++r) {
                        double v;
                        if (STEP) {      // NaN outside the hull; at the last knot the rule itself yields the last quote
                            v = step_eval<METHOD>(xq, w0, w1, g0[r], g1[r]);
                            if (l_right && !l_hold) v = nanv;
                        } else {
                            v = lerp_fast(xq, w0, g0[r], g1[r], w2, w3, slow);
                            if (l_right) v = l_hold ? g1[r] : nanv;      // jj = n - 2 -> g1 is the last quote
                        }
                        if (l_left) v = nanv;
                        z[ps * RP + r] = v;
                    }
                    if (STEP) continue;
                    if (__builtin_expect(__ballot(slow && !l_right && !l_left) != 0ull, 0)) {      // rare: full IEEE path
                        if (slow && !l_right && !l_left) {
#pragma unroll
                            for (int r = 0; r < RP; ++r) z[ps * RP + r] = lerp_np(xq, w0, Yp[r * RS + o0], w1, Yp[r * RS + o1]);
                        }
                    }
                    continue;
                }
                if (LOCAL) pass_local_slopes<METHOD, NKB, VAR>(Yp, Sp, TB, lane, n);
                else if (ABL != 1 && ABL != 6) pass_sweeps<NKB, VAR, SL, QUADK>(Yp, Sp, TB, lane, n);
                __syncthreads();
                // ---- strike evaluation of the pass's rows (q-lane), gathers pipelined LA rows ahead
                if (ABL == 2 || ABL == 6) {
#pragma unroll
                    for (int r = 0; r < RP; ++r) z[ps * RP + r] = Yp[r * RS + lane] + w0;
                    continue;
                }
                constexpr int LA = RP < 4 ? RP - 1 : 3;
                double g0[4], g1[4], g2[4], g3[4];
                // operands of the four weights: (y_j, y_j+1, s_j, s_j+1), or for the quadratic spline the coefficients c[jj-1 .. jj+2]
                const double* PA = QUADK ? Sp : Yp;
                const int a0i = QUADK ? p_swz<SL, NKB>(jj > 0 ? jj - 1 : 0) : o0, a1i = QUADK ? q0 : o1;
                const int b0i = QUADK ? q1 : q0, b1i = QUADK ? p_swz<SL, NKB>(jj + 2 < KCAP ? jj + 2 : KCAP - 1) : q1;
#pragma unroll
                for (int r = 0; r < LA; ++r) {
                    g0[r] = PA[r * RS + a0i]; g1[r] = PA[r * RS + a1i]; g2[r] = Sp[r * RS + b0i]; g3[r] = Sp[r * RS + b1i];
                }
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    if (r + LA < RP) {
                        const int nn = r + LA;
                        g0[nn & 3] = PA[nn * RS + a0i]; g1[nn & 3] = PA[nn * RS + a1i];
                        g2[nn & 3] = Sp[nn * RS + b0i]; g3[nn & 3] = Sp[nn * RS + b1i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    z[ps * RP + r] = w0 * g0[r & 3] + w1 * g1[r & 3] + w2 * g2[r & 3] + w3 * g3[r & 3];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (ok) {
            if (LOCAL) {
#pragma unroll
                for (int r = 0; r < DT; ++r) asm volatile("" : "+v"(z[r]));      // see dense_maturity_pass
            }
            if (act) dense_maturity_pass<METHOD, true, false, NTR, TSH>(z, tt, TSH ? TTp : TTl, TSH ? Wp : Wl, outb, 0, lane, true, mT, mK,
                                                                        nostamp, 0, 0, nT);
            if (p.status && lane == 0) p.status[b] = IVS_ST_OK;
        } else if (lane == 0) {
            reinterpret_cast<unsigned long long*>(outb)[0] = D_SENTINEL;     // redone by the compaction / generic kernel (later launches)
            count_redo(p, told);
        }
        const int64_t it_next2 = wq.finish(pend, more);
        it = it_next; it_next = it_next2; b = b_next; n = n_next; koff = koff_next;
    }
#ifdef IVS_PASS_ENDSTAMP
    if (threadIdx.x == 0 && d_pass_ends) d_pass_ends[blockIdx.x * 2 + 1] = wall_clock64();
#endif
}

// one launch; run-time-shape batches with the full 16 maturities take the NT16 instantiation
template <int METHOD, int NKB, bool VAR, int SL>
inline void launch_pass_nt(bool nt16, int64_t grid, size_t lds, hipStream_t st, const SurfaceParams& p, const VarList& list) {
    if constexpr (VAR) {
        if (nt16) { hipLaunchKernelGGL((surface_pass_kernel<METHOD, NKB, true, SL, true, true>), dim3((unsigned)grid), dim3(64), lds, st, p, list); return; }
    }
    hipLaunchKernelGGL((surface_pass_kernel<METHOD, NKB, VAR, SL, true, false>), dim3((unsigned)grid), dim3(64), lds, st, p, list);
}
#ifndef IVS_AKIMA_FIXED
#define IVS_AKIMA_FIXED 1      // 64 x 16 akima on the row-pass kernel (0: the one-pass kernel, rounds 1-2)
#endif
template <int NKB, bool VAR>
inline void launch_pass_akima(bool nt16, int64_t grid, size_t lds, hipStream_t st, const SurfaceParams& p, const VarList& list) {
    if constexpr (VAR || (IVS_AKIMA_FIXED && NKB == 1)) launch_pass_nt<IVS_AKIMA, NKB, VAR, 8>(nt16, grid, lds, st, p, list);
}

// Dispatch of the row-pass kernels.  Returns 1 if dispatched (pass kernel(s) + filtered generic redo pass), 0 if the
// call is outside their scope (see the head of this file), -1 on a launch error.
inline int launch_surface_pass(const SurfaceParams& p_in, const LaunchCtx& cx, const char** name) {
    SurfaceParams p = p_in;
    hipStream_t st = cx.st;
    const bool lerp = p.method == IVS_LINEAR || p.method == IVS_SLINEAR || d_is_step(p.method);
    const bool local = d_is_local(p.method);
    if (!(p.method == IVS_CUBIC || p.method == IVS_CUBICSPLINE || p.method == IVS_QUADRATIC || lerp || local)) return 0;
    const bool tsh = p.t_stride == 0 && p.tq_stride == 0;
    if (!tsh && !(p.method == IVS_LINEAR && p.mT <= D_WLDS_MAX_MT)) return 0;
    if (p.mK > 64 || p.mT > D_MAX_MT) return 0;
    if (p.nT < 4 || p.nT > DT || p.nK < 4 || p.nK > 128) return 0;
    if (!p.k_off && p.k_stride != 0 && p.k_stride < p.nK) return 0;
    if (p.k_off && p.B > 0x7fffffffLL) return 0;
    const bool fixed64 = !p.k_off && p.nK == DK && p.nT == DT && !(reinterpret_cast<uintptr_t>(p.sigma) & 15);
#ifndef IVS_PASS_NT16
#define IVS_PASS_NT16 1      // 0: A/B builds without the NT16 instantiations
#endif
    const bool nt16 = IVS_PASS_NT16 && p.nT == DT;       // run-time-shape kernels: maturity count fixed at compile time (NT16)
    if (!IVS_AKIMA_FIXED && fixed64 && p.method == IVS_AKIMA) return 0;      // rounds 1-2: 64 x 16 akima stayed on the one-pass kernel (at 168 VGPRs the row-pass form spilled 50 registers; see IVS_AKIMA_ROLL)
    TqShared* tq = reinterpret_cast<TqShared*>(cx.ws);
    if (tsh) {
        if (fixed64 || nt16) launch_tq_tables<false>(p, tq, st); else launch_tq_tables<true>(p, tq, st);      // 16 maturities: the fixed-count tables
        p.tqs = tq;
        p.redo = tq->redo;
        p.mode = fixed64 ? &tq->mode : nullptr;
    } else {      // per-surface maturities: the T-phase runs inside the kernel; only the queue heads need zeroing
        if (hipMemsetAsync(tq->queue, 0, sizeof(TqShared::queue), st) != hipSuccess) return -1;
        p.tqs = nullptr;
        p.redo = nullptr;
    }
    p.queue = tq->queue;
    auto grid_for = [&](size_t lds, int64_t work, int wg_cap = 12) {
        int per_cu = (int)((160 * 1024) / (((lds + 1279) / 1280) * 1280));     // LDS is granted in 1280-byte granules
        per_cu = per_cu > wg_cap ? wg_cap : (per_cu < 1 ? 1 : per_cu);         // 3 wavefronts per SIMD (168 VGPRs); 4 with 4-knot segments (128)
#ifdef IVS_PASS_PER_CU
        per_cu = IVS_PASS_PER_CU;                                              // diagnostic builds (tools/pass_api.hip)
#endif
        const int64_t g = (int64_t)cx.num_cu * per_cu;
        return g > work ? work : g;
    };
    const VarList none{nullptr, nullptr};
    static const char* const names[2][10] = {
        {"surface_pass_kernel<cubic>", "surface_pass_kernel<cubicspline>", "surface_pass_kernel<linear>", "surface_pass_kernel<slinear>",
         "surface_pass_kernel<pchip>", "surface_pass_kernel<akima>", "surface_pass_kernel<nearest>", "surface_pass_kernel<zero>",
         "surface_pass_kernel<from_derivatives>", "surface_pass_kernel<quadratic>"},
        {"surface_pass_var_kernel<cubic>", "surface_pass_var_kernel<cubicspline>", "surface_pass_var_kernel<linear>",
         "surface_pass_var_kernel<slinear>", "surface_pass_var_kernel<pchip>", "surface_pass_var_kernel<akima>",
         "surface_pass_var_kernel<nearest>", "surface_pass_var_kernel<zero>", "surface_pass_var_kernel<from_derivatives>",
         "surface_pass_var_kernel<quadratic>"}};
    int mi = 0;
    switch (p.method) {
        case IVS_CUBIC: mi = 0; break; case IVS_CUBICSPLINE: mi = 1; break; case IVS_LINEAR: mi = 2; break; case IVS_SLINEAR: mi = 3; break;
        case IVS_PCHIP: mi = 4; break; case IVS_AKIMA: mi = 5; break; case IVS_NEAREST: mi = 6; break; case IVS_ZERO: mi = 7; break;
        case IVS_QUADRATIC: mi = 9; break;
        default: mi = 8; break;
    }
    // one launch of surface_pass_kernel<method, NKB, VAR, SL> over `list`; the lerp methods carry no S plane and no tables
#ifdef IVS_DIAG_MINIMAL
#define IVS_PASS_LAUNCH(NKB_, VAR_, SL_, CAP, LIST)                                                                                      \
    {                                                                                                                                    \
        const size_t lds = pass_lds_bytes<NKB_, VAR_, SL_, 0>();                                                                         \
        const int64_t grid = grid_for(lds, p.B, CAP);                                                                                    \
        if (!(VAR_)) p.map_groups = dense_map_groups(grid, p.B, cx.map_groups);                                                          \
        hipLaunchKernelGGL((surface_pass_kernel<IVS_CUBIC, NKB_, VAR_, SL_>), dim3((unsigned)grid), dim3(64), lds, st, p, LIST);         \
    }
#else
#define IVS_PASS_LAUNCH(NKB_, VAR_, SL_, CAP, LIST)                                                                                      \
    {                                                                                                                                    \
        size_t lds = lerp ? pass_lds_bytes<NKB_, VAR_, SL_, 1>() : (local ? pass_lds_bytes<NKB_, VAR_, SL_, 2>() : pass_lds_bytes<NKB_, VAR_, SL_, 0>());  \
        if (!tsh) lds += PASS_TQ_DOUBLES * 8;                                                                                            \
        const int64_t grid = grid_for(lds, p.B, (local && (VAR_) && !pass_local_three(p.method, NKB_, nt16)) ? 8 : (CAP));               \
        if (!(VAR_)) p.map_groups = dense_map_groups(grid, p.B, cx.map_groups);                                                          \
        if (!tsh) {                                                                                                                      \
            hipLaunchKernelGGL((surface_pass_kernel<IVS_LINEAR, NKB_, VAR_, 8, false>), dim3((unsigned)grid), dim3(64), lds, st, p, LIST);                              \
        } else                                                                                                                           \
        switch (p.method) {                                                                                                              \
            case IVS_CUBIC: launch_pass_nt<IVS_CUBIC, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break;             \
            case IVS_CUBICSPLINE: launch_pass_nt<IVS_CUBICSPLINE, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break; \
            case IVS_LINEAR: launch_pass_nt<IVS_LINEAR, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break;           \
            case IVS_NEAREST: launch_pass_nt<IVS_NEAREST, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break;         \
            case IVS_ZERO: launch_pass_nt<IVS_ZERO, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break;               \
            case IVS_FROM_DERIVATIVES: launch_pass_nt<IVS_FROM_DERIVATIVES, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break; \
            case IVS_QUADRATIC: launch_pass_nt<IVS_QUADRATIC, NKB_, VAR_, 8>(nt16, grid, lds, st, p, LIST); break;       \
            case IVS_PCHIP: launch_pass_nt<IVS_PCHIP, NKB_, VAR_, 8>(nt16, grid, lds, st, p, LIST); break;               \
            case IVS_AKIMA: launch_pass_akima<NKB_, VAR_>(nt16, grid, lds, st, p, LIST); break;                                                \
            default: launch_pass_nt<IVS_SLINEAR, NKB_, VAR_, SL_>(nt16, grid, lds, st, p, LIST); break;                  \
        }                                                                                                                                \
    }
#endif
    if (fixed64) {
        constexpr int FSL = IVS_PASS_SL;
        IVS_PASS_LAUNCH(1, false, FSL, (FSL == 4 ? 16 : 12), none)
        *name = names[0][mi];
    } else {
        VarItem* lists = nullptr;
        int32_t* counts = nullptr;
        if (p.k_off) {      // classify once into one work list per size class (workspace: counters, then B items per class)
            counts = reinterpret_cast<int32_t*>(cx.ws + WS_TQ_BYTES);
            lists = reinterpret_cast<VarItem*>(cx.ws + WS_TQ_BYTES + WS_COUNTS_BYTES);
            if (hipMemsetAsync(counts, 0, WS_COUNTS_BYTES, st) != hipSuccess) return -1;
            int64_t cb = (p.B + 1023) / 1024;
            const int64_t cap = (int64_t)cx.num_cu * 8;
            if (cb > cap) cb = cap;
            hipLaunchKernelGGL(var_classify_kernel, dim3((unsigned)cb), dim3(256), 0, st, p, lists, lists + p.B, counts);
        }
        const VarList wl1{lists, counts, 0}, wl2{lists ? lists + p.B : nullptr, counts ? counts + 1 : nullptr, 1};
        const bool need1 = p.k_off ? true : p.nK <= 64, need2 = p.nK > 64;
        if (need1) IVS_PASS_LAUNCH(1, true, 8, 12, wl1)
        if (need2) IVS_PASS_LAUNCH(2, true, 8, IVS_PASS_CAP2, wl2)
        *name = names[1][mi];
    }
#undef IVS_PASS_LAUNCH
    if (hipGetLastError() != hipSuccess) return -1;
#ifndef IVS_DIAG_MINIMAL
    if (fixed64 && launch_surface_masked(p, cx) && p.redo) ++p.redo;   // tagged surfaces (missing quotes): the masked fast pass first ...
#endif
    launch_surface_generic<true>(p, cx);     // ... then the generic kernel for whatever is still tagged (cheap when nothing is)
    return 1;
}

}  // namespace ivs
