// Dense fast path for VARIABLE strike counts: 4 <= nK <= 64 per surface on one wavefront (uniform batches with
// nK != 64 and ragged CSR batches, BASELINE config 5), 4..16 maturities (run time, uniform over the batch), T/Tq shared
// or per surface, no missing quotes.  Surfaces with 65..128 strikes use the two-wavefront kernel of
// ivs_surface_dense_var2.hpp, which reuses the pieces below.
//
// Same three-layout scheme as ivs_surface_dense.hpp, with the strike count n a run-time value:
//   * knots beyond n are neutral elements: +inf in the strike array (searches stop), identity matrices in the
//     pivot scan, 1.0 in the segment products, masked steps in the sweeps;
//   * the last row of the not-a-knot system sits at a run-time position, so the sweeps use a three-tap right-hand
//     side  r_i = PM_i*dy_{i-2} + PP_i*dy_{i-1} + QQ_i*dy_i  (PM is zero except on the last row);
//   * pchip / akima: per-segment secant sequences with run-time fix-ups (segment_secants / segment_slopes);
//   * a ragged batch is served by one launch per size class, each launch skipping the surfaces of the other class
//     (wave-uniform test on k_off).
// The templates keep an NKB parameter (strike blocks of 64 per k-lane): NKB = 2 was the one-wavefront kernel for
// 65..128 strikes (71 M surfaces/s), superseded by the two-wavefront kernel (98 M) and no longer instantiated.
// Surfaces outside [4,128], with a NaN quote, or batches with unsorted Tq are tagged for the generic redo pass.
#pragma once
#include "ivs_surface_dense.hpp"

namespace ivs {

constexpr int DPP_QUAD_BCAST0 = 0x00, DPP_QUAD_BCAST3 = 0xFF;   // quad_perm:[0,0,0,0] / [3,3,3,3]

template <int NKB>
__host__ __device__ inline size_t dense_var_lds_bytes(int mT) {
    // Y, S planes [16][NKB*72]; Ksh [NKB*64]; RDX [NKB*64]; TT [16][4]; W [16][4] when mT <= 16
    return (size_t)(2 * DT * NKB * 72 + NKB * 64 + NKB * 64 + 64 + (mT <= D_WLDS_MAX_MT ? 4 * D_WLDS_MAX_MT : 0)) * 8;
}

// Factor tables for n knots (run time) spread over NKB blocks of 64 lanes.  Tables are written at d_sl(i), i < n.
template <int NKB>
__device__ __forceinline__ void factor_tables_var(const double* X, int n, int lane, double* AL, double* CP, double* PP,
                                                  double* QQ, double* PM, double* PI, double* PSI, double* RDX) {
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
        if (first) {
            pp = (dxc + 2.0 * d) * dxp * rdxc * rd * rw;         // * dy_0
            qq = dxc * dxc * rdx_next * rd * rw;                 // * dy_1
        } else if (last) {
            pm = dxm * dxm * rdxmm * rd * rw;                    // * dy_{n-3}
            pp = (2.0 * d + dxm) * dxmm * rdx_prev * rd * rw;    // * dy_{n-2}
            qq = 0.0;
        } else {
            pp = 3.0 * dxc * rdx_prev * rw;                      // * dy_{i-1}
            qq = 3.0 * dxm * rdxc * rw;                          // * dy_i
        }
        const double pi = seg16_prefix_prod(in ? -al : 1.0, lane);
        const double psi = seg16_suffix_prod(in ? -cp : 1.0, lane);
        {   // every slot is written: beyond n the NEUTRAL row (AL = -1, everything else 0) makes the forward sweep hold its
            // value and the backward sweep harmless, so the sweeps need no per-step masks (the last system row has
            // CP = 0, which cuts the backward recurrence off from whatever lies to its right)
            const int kl = d_sl(ir);
            AL[kl] = in ? al : -1.0; CP[kl] = in ? cp : 0.0; PP[kl] = in ? pp : 0.0; QQ[kl] = in ? qq : 0.0;
            PM[kl] = in ? pm : 0.0; PI[kl] = pi; PSI[kl] = psi;
            RDX[ir] = rdxc;
        }
        if (blk + 1 < NKB) {
            c00 = readlane_f64(p00, 63); c01 = readlane_f64(p01, 63); c10 = readlane_f64(p10, 63); c11 = readlane_f64(p11, 63);
            carry_crb = readlane_f64(crb, 63); carry_rdx = readlane_f64(rdxc, 63); carry_rdx_prev = readlane_f64(rdx_prev, 63);
        }
    }
}

// Slopes of all 16 rows for run-time n.  Y/S planes have row stride RS = NKB*72; tables alias the S plane.
template <int NKB>
__device__ __forceinline__ void dense_strike_slopes_var(const double* Y, double* S, const double* Ksh, double* RDX, int n,
                                                        int lane) {
    constexpr int RS = NKB * 72;
    double* AL = S;
    double* CP = S + RS;
    double* PP = S + 2 * RS;
    double* QQ = S + 3 * RS;
    double* PM = S + 4 * RS;
    double* PI = S + 5 * RS;
    double* PSI = S + 6 * RS;
    factor_tables_var<NKB>(Ksh, n, lane, AL, CP, PP, QQ, PM, PI, PSI, RDX);
    __syncthreads();
    const int rs_t = lane >> 2, rs_seg = lane & 3;
    double d[NKB][16];
    int len[NKB];
    double endv[NKB], pie[NKB], din[NKB];
    // ---- local forward sweeps
#pragma unroll
    for (int u = 0; u < NKB; ++u) {
        const int sg = rs_seg + 4 * u, kb = 16 * sg, kp = 18 * sg;
        int ln = n - kb; ln = ln < 0 ? 0 : (ln > 16 ? 16 : ln);
        len[u] = ln;
        const double* yrow = Y + rs_t * RS;
        double y[19];                                          // y[j] = y_{kb + j - 2}
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const double2 v = *reinterpret_cast<const double2*>(yrow + kp + 2 * c);
            y[2 + 2 * c] = v.x; y[3 + 2 * c] = v.y;
        }
        y[0] = kb >= 2 ? yrow[d_sl(kb - 2)] : 0.0;
        y[1] = kb >= 1 ? yrow[d_sl(kb - 1)] : 0.0;
        y[18] = kb + 16 < n ? yrow[d_sl(kb + 16)] : 0.0;
        double prev = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            double dM = y[m + 1] - y[m], dA = y[m + 2] - y[m + 1], dB = y[m + 3] - y[m + 2];   // dy_{i-2}, dy_{i-1}, dy_i
            if (u == 0 && m == 0) {                            // row 0 of the system uses (dy_0, dy_1)
                const double e = y[4] - y[3];
                if (rs_seg == 0) { dA = dB; dB = e; }
            }
            const double r = PM[kp + m] * dM + PP[kp + m] * dA + QQ[kp + m] * dB;
            prev = r - AL[kp + m] * prev;                      // neutral rows beyond n: r = 0, AL = -1 -> holds
            d[u][m] = prev;
            if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        endv[u] = prev;
        pie[u] = PI[kp + 15];                                  // neutral factors are 1: = the product up to the last knot
    }
    // ---- forward carries across the 4*NKB logical segments (0,1,2,3 of u = 0, then of u = 1, ...)
    double carry = 0.0;
#pragma unroll
    for (int u = 0; u < NKB; ++u) {
        double di = rs_seg == 0 ? carry : 0.0;
        double tot = endv[u] + pie[u] * di;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const double v = dpp0_f64<DPP_ROW_SHR(1)>(tot);
            if (rs_seg == j) { di = v; tot = endv[u] + pie[u] * di; }
        }
        din[u] = di;
        carry = dpp0_f64<DPP_QUAD_BCAST3>(tot);            // value at the end of logical segment 3 + 4u
    }
    // ---- local backward sweeps (with the forward fix-up folded in)
    double firstv[NKB], psb[NKB], sin_[NKB];
#pragma unroll
    for (int u = NKB - 1; u >= 0; --u) {
        const int kp = 18 * (rs_seg + 4 * u);
        double nxt = 0.0;
#pragma unroll
        for (int m = 15; m >= 0; --m) {
            const double dp = d[u][m] + PI[kp + m] * din[u];
            nxt = dp - CP[kp + m] * nxt;                       // slots beyond n: finite garbage, cut off by CP[n-1] = 0
            d[u][m] = nxt;
            if ((m & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        firstv[u] = d[u][0];
        psb[u] = PSI[kp];
    }
    carry = 0.0;
#pragma unroll
    for (int u = NKB - 1; u >= 0; --u) {
        double si = rs_seg == 3 ? carry : 0.0;
        double tot = firstv[u] + psb[u] * si;
#pragma unroll
        for (int j = 2; j >= 0; --j) {
            const double v = dpp0_f64<DPP_ROW_SHL(1)>(tot);
            if (rs_seg == j) { si = v; tot = firstv[u] + psb[u] * si; }
        }
        sin_[u] = si;
        carry = dpp0_f64<DPP_QUAD_BCAST0>(tot);            // slope at the first knot of logical segment 4u
    }
#pragma unroll
    for (int u = 0; u < NKB; ++u) {
        const int kp = 18 * (rs_seg + 4 * u);
#pragma unroll
        for (int m = 0; m < 16; ++m) d[u][m] = d[u][m] + PSI[kp + m] * sin_[u];
    }
    __syncthreads();                                           // table reads done: the S plane may be overwritten
#pragma unroll
    for (int u = 0; u < NKB; ++u) {
        // all 16 slots of the segment are written (b128, conflict-free); those beyond n are never read
        double* srow = S + rs_t * RS + 18 * (rs_seg + 4 * u);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double2 v; v.x = d[u][2 * c]; v.y = d[u][2 * c + 1];
            *reinterpret_cast<double2*>(srow + 2 * c) = v;
        }
    }
}

// ---- local-slope methods (pchip, akima) with a run-time knot count (scheme: dense_strike_slopes_local)
__device__ __forceinline__ void local_tables_rt(const double* X, int n, int i, double& r0, double& r1, double& r2) {
    const bool in = i < n;
    const int ii = in ? i : n - 1;
    const double x0 = X[ii];
    const double xp = X[ii + 1 < n ? ii + 1 : n - 1], xm = X[ii > 0 ? ii - 1 : 0];
    const double dxc = xp - x0, dxm = x0 - xm;
    const bool first = ii == 0, last = ii == n - 1;
    const double h0 = first ? dxc : dxm;
    const double h1 = first ? X[2] - X[1] : X[n - 2] - X[n - 3];
    const double rs = refined_rcp(h0 + h1);
    r0 = (in && !last) ? refined_rcp(dxc) : 0.0;
    r1 = (first || last) ? (2.0 * h0 + h1) * rs : 2.0 * dxc + dxm;
    r2 = (first || last) ? h0 * rs : dxc + 2.0 * dxm;
    if (!in) { r1 = 0.0; r2 = 0.0; }
}

// Final secant sequence F[i] = m_{kb-2+i}, i = 0..18, of one 16-knot segment of a row (yr: the row in the padded LDS
// layout, kp = padded index of knot kb, R0 = 1/dx table, zero from interval n-1 on).  Akima: secants left of knot 0 and
// beyond knot n-1 by linear extension of the secant sequence, F(i) = 2 F(i-1) - F(i-2).
template <bool AK>
__device__ __forceinline__ void segment_secants(const double* yr, const double* R0, int kb, int kp, int n, double (&F)[19]) {
    const bool s_first = kb == 0;
    double y[20];                                   // y[m+2] = y_{kb+m}, m = -2..17
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(yr + kp + 2 * c);
        y[2 + 2 * c] = v.x; y[3 + 2 * c] = v.y;
    }
    y[1] = yr[s_first ? 0 : kp - 3];
    y[18] = yr[kp + 18];
    y[0] = yr[s_first ? 0 : kp - 4];               // pchip too: the one-sided rule at knot n-1 = kb needs m_{n-3}
    y[19] = AK ? yr[kp + 19] : 0.0;
#pragma unroll
    for (int i = 0; i < 19; ++i) {
        const int ti = i == 0 ? kp - 4 : (i == 1 ? kp - 3 : (i == 18 ? kp + 18 : kp + i - 2));
        F[i] = (y[i + 1] - y[i]) * R0[(s_first && i < 2) ? 0 : ti];
    }
    if (AK) {
        const double l1 = 2.0 * F[2] - F[3], l0 = 2.0 * l1 - F[2];
        F[1] = s_first ? l1 : F[1]; F[0] = s_first ? l0 : F[0];
#pragma unroll
        for (int i = 2; i < 19; ++i) {
            const int idx = kb - 2 + i;
            F[i] = (idx == n - 1 || idx == n) ? 2.0 * F[i - 1] - F[i - 2] : F[i];
        }
    }
}
// largest |second difference| over the segment's knots below n (akima's switch to the arithmetic mean)
__device__ __forceinline__ double segment_akima_fmax(const double (&F)[19], int kb, int n) {
    double fmax = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const double f = akima_f12(F[m], F[m + 1], F[m + 2], F[m + 3]);
        fmax = (kb + m < n) ? __builtin_fmax(fmax, f) : fmax;
    }
    return fmax;
}
template <int METHOD>
__device__ __forceinline__ void segment_slopes(const double (&F)[19], const double* R1, const double* R2, int kb, int kp,
                                               int n, double thr, double (&d)[16]) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        if (METHOD == IVS_AKIMA) d[m] = akima_knot(F[m], F[m + 1], F[m + 2], F[m + 3], thr);
        else {
            const double w1 = R1[kp + m], w2 = R2[kp + m];
            double v = pchip_knot(F[m + 1], F[m + 2], w1, w2);
            if (m == 0) { const double e = pchip_edge(F[2], F[3], w1, w2); v = kb == 0 ? e : v; }
            d[m] = v;                                                   // knot n-1 is patched by pchip_last_slope
        }
        if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// pchip's one-sided rule at the last knot n-1 (run-time position): computed once per row from LDS instead of being
// evaluated and selected away at every knot; the lane that owns knot n-1 overwrites its slope after the row is stored.
__device__ __forceinline__ double pchip_last_slope(const double* yr, const double* R0, const double* R1, const double* R2, int n) {
    const int k1 = d_sl(n - 1), k2 = d_sl(n - 2), k3 = d_sl(n - 3);
    const double m2 = (yr[k1] - yr[k2]) * R0[k2], m3 = (yr[k2] - yr[k3]) * R0[k3];     // m_{n-2}, m_{n-3}
    return pchip_edge(m2, m3, R1[k1], R2[k1]);
}

// Strike-direction slopes, local methods, n <= 64 (one wavefront per surface).  Tables alias the S plane.
template <int METHOD>
__device__ __forceinline__ void dense_strike_slopes_local_var(const double* Y, double* S, const double* Ksh, double* RDX,
                                                              int n, int lane) {
    constexpr bool AK = METHOD == IVS_AKIMA;
    constexpr int RS = 72;
    double* R0 = S;
    double* R1 = S + 80;
    double* R2 = S + 160;
    {
        double r0, r1, r2;
        local_tables_rt(Ksh, n, lane, r0, r1, r2);
        const int kl = d_sl(lane);
        R0[kl] = r0; R1[kl] = r1; R2[kl] = r2; RDX[lane] = r0;
    }
    __syncthreads();
    const int rs_t = lane >> 2, rs_seg = lane & 3;
    const int kb = 16 * rs_seg, kp = 18 * rs_seg;
    double F[19], d[16];
    segment_secants<AK>(Y + rs_t * RS, R0, kb, kp, n, F);
    double thr = 0.0;
    if (AK) {
        double fmax = segment_akima_fmax(F, kb, n);
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP1>(fmax, fmax));      // the row's 4 segments = one quad
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP2>(fmax, fmax));
        thr = 1e-9 * fmax;
    }
    segment_slopes<METHOD>(F, R1, R2, kb, kp, n, thr, d);
    const double e_last = AK ? 0.0 : pchip_last_slope(Y + rs_t * RS, R0, R1, R2, n);
    __syncthreads();                               // table reads done: the S plane may be overwritten
    double* srow = S + rs_t * RS + kp;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double2 v; v.x = d[2 * c]; v.y = d[2 * c + 1];
        *reinterpret_cast<double2*>(srow + 2 * c) = v;
    }
    if (!AK && rs_seg == ((n - 1) >> 4)) S[rs_t * RS + d_sl(n - 1)] = e_last;     // same lane, after its row store
}

// Work list of a launch.  Ragged batches are classified once per call (var_classify_kernel) into one list per size
// class, so that no kernel walks over -- and reads the offsets of -- the other class's surfaces (that cost 9-18 % of
// the launch on config 5).  items == nullptr: the launch serves surfaces 0..B-1 of a uniform batch.
struct VarItem { int32_t b; int32_t n; int64_t koff; };
struct VarList { const VarItem* items; const int32_t* count; int qslot = 0; };      // qslot: work-queue head of this launch

__global__ __launch_bounds__(256) void var_classify_kernel(SurfaceParams p, VarItem* l1, VarItem* l2, int32_t* counts) {
    // 1024 consecutive surfaces per block and step, 4 per thread; block-level exclusive scan of the per-thread class
    // counts, ONE global atomic per class, block and step (a per-wavefront atomic cost 0.36 ms per million surfaces)
    __shared__ int wsum[2][4];
    __shared__ int base_s[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t b0 = (int64_t)blockIdx.x * 1024; b0 < p.B; b0 += (int64_t)gridDim.x * 1024) {      // block-uniform trip count
        int64_t ko[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { const int64_t b = b0 + 4 * tid + k; ko[k] = b <= p.B ? p.k_off[b] : 0; }
        int cls[4]; int c[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t b = b0 + 4 * tid + k;
            const int64_t nn = ko[k + 1] - ko[k];
            cls[k] = b >= p.B ? -1 : ((nn >= 4 && nn <= 64) ? 0 : ((nn >= 65 && nn <= 128) ? 1 : 2));
            if (cls[k] >= 0 && (nn < 0 || nn > p.nK || ko[k] < 0 || (p.k_total > 0 && ko[k + 1] > p.k_total))) {
                // offsets no kernel can serve (the LDS carve is sized by nK) or that leave the caller's K / sigma arrays
                cls[k] = -1;
                if (p.status) p.status[b] = IVS_ST_BAD_SHAPE;
            }
            if (cls[k] == 2) {                                  // not served by a dense kernel: generic redo pass
                reinterpret_cast<unsigned long long*>(p.out + b * (int64_t)p.mT * p.mK)[0] = D_SENTINEL;
                count_redo(p);
            }
            if (cls[k] == 0) ++c[0];
            if (cls[k] == 1) ++c[1];
        }
        int pre[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int incl = c[j];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
            if (lane == 63) wsum[j][wave] = incl;
            pre[j] = incl - c[j];
        }
        __syncthreads();
        if (tid < 2) {
            const int tot = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
            base_s[tid] = tot ? atomicAdd(&counts[tid], tot) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int w0 = 0;
            for (int w = 0; w < wave; ++w) w0 += wsum[j][w];
            pre[j] += base_s[j] + w0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (cls[k] == 0) l1[pre[0]++] = VarItem{(int32_t)(b0 + 4 * tid + k), (int32_t)(ko[k + 1] - ko[k]), ko[k]};
            if (cls[k] == 1) l2[pre[1]++] = VarItem{(int32_t)(b0 + 4 * tid + k), (int32_t)(ko[k + 1] - ko[k]), ko[k]};
        }
        __syncthreads();                                         // wsum / base_s are reused by the next step
    }
}

// TSHARED: T and Tq shared by the batch (T-phase once per workgroup); otherwise per surface (t_stride / tq_stride),
// the T-phase then runs inside the loop with its scratch in the S plane, which is free until the slopes are written.
template <int METHOD, int NKB, bool WLDS, bool TSHARED = true>
__global__ __launch_bounds__(64, 2) void surface_dense_var_kernel(SurfaceParams p, VarList list) {
    constexpr bool CUB = d_is_hermite(METHOD);
    constexpr int RS = NKB * 72;
    constexpr int KCAP = NKB * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int mT = p.mT, mK = p.mK;
    double* Y = reinterpret_cast<double*>(smem);
    double* S = Y + DT * RS;
    double* Ksh = S + DT * RS;             // KCAP entries; strikes beyond n are +inf (searches never index past KCAP-1)
    double* RDX = Ksh + KCAP;
    double* TT = RDX + KCAP;
    double* W = TT + 64;
    const double nanv = __builtin_nan(""), inf = __builtin_inf();
    auto nostamp = [](int) {};

    TqTables tt;
    const int nT = p.nT;                                                   // 4..16, uniform over the batch
    const double* TTp = TT;                // maturity tables: LDS (per-surface T) or the published TqShared (scalar cache)
    const double* Wp = W;
    if (TSHARED) tq_from_shared(p.tqs, tt, TTp, Wp);

    constexpr int XQ_REG = 4;
    double xq_reg[XQ_REG];
    const bool kq_shared = p.kq_stride == 0;
    auto load_xq = [&](const double* Kqb) {
#pragma unroll
        for (int i = 0; i < XQ_REG; ++i) xq_reg[i] = (i * 64 + lane < mK) ? Kqb[i * 64 + lane] : nanv;
    };
    if (kq_shared) load_xq(p.Kq);

    // surface at position `it` of this launch's work list (uniform batches: position = surface number)
    const int64_t limit = list.items ? (int64_t)*list.count : p.B;
    auto at = [&](int64_t it, int& n, int64_t& koff) -> int64_t {
        if (list.items) { const VarItem v = list.items[it]; n = v.n; koff = v.koff; return v.b; }
        n = p.nK; koff = it * p.k_stride; return it;
    };
    bool told = false;
    auto tag = [&](int64_t b) {
        if (lane == 0) { reinterpret_cast<unsigned long long*>(p.out + b * (int64_t)mT * mK)[0] = D_SENTINEL; count_redo(p, told); }
    };

    double pre[DT * NKB], pre_k[NKB];
    int n = 0, n_next = 0;
    int64_t koff = 0, koff_next = 0;
    auto sigma_of = [&](int64_t b, int64_t ko) -> const double* {
        return p.k_off ? p.sigma + (int64_t)nT * ko : p.sigma + b * (int64_t)nT * p.nK;
    };
    auto issue_loads = [&](int64_t b, int64_t ko, int nn) {
        const double* sb = sigma_of(b, ko);
        const double* Kb = p.K + ko;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int blk = 0; blk < NKB; ++blk) {
                const int k = blk * 64 + lane;
                pre[t * NKB + blk] = (t < nT && k < nn) ? sb[(int64_t)t * nn + k] : 0.0;      // rows beyond nT: zeros
            }
#pragma unroll
        for (int blk = 0; blk < NKB; ++blk) { const int k = blk * 64 + lane; pre_k[blk] = k < nn ? Kb[k] : inf; }
    };

    WorkQueue wq;                                          // work queues instead of static striding (ivs_surface_generic.hpp)
    wq.init(p.queue + list.qslot * 8 * QUEUE_STRIDE, 8, limit, WQ_CHUNK, lane);
    int64_t it = wq.take(), it_next = it >= 0 ? wq.take() : -1, b = 0;
    if (it >= 0) { b = at(it, n, koff); issue_loads(b, koff, n); }

    while (it >= 0) {
        const bool more = it_next >= 0;
        const WorkQueue::Pending pend = wq.begin(more);
        __syncthreads();
        unsigned long long bad = 0ull;                     // wave-level NaN mask in scalar registers
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int blk = 0; blk < NKB; ++blk) {
                const double v = pre[t * NKB + blk];
                Y[t * RS + d_sl(blk * 64 + lane)] = v;
                bad |= __ballot(v != v);
            }
#pragma unroll
        for (int blk = 0; blk < NKB; ++blk) Ksh[blk * 64 + lane] = pre_k[blk];
        double* outb = p.out + b * (int64_t)mT * mK;
        const double* Kqb = p.Kq + b * p.kq_stride;
        int64_t b_next = 0;
        if (more) b_next = at(it_next, n_next, koff_next);
        if (!TSHARED) dense_t_phase<METHOD, WLDS, true>(p.T + b * p.t_stride, p.Tq + b * p.tq_stride, mT, lane, S + 600, TT, W,
                                                        tt, nT, S);
        const bool redo = bad != 0ull || tt.unsorted;
        if (redo) {
            tag(b);
        } else {
            __syncthreads();
            if (d_is_nak(METHOD)) {
                dense_strike_slopes_var<NKB>(Y, S, Ksh, RDX, n, lane);
                __syncthreads();
            } else if (d_is_local(METHOD)) {
                dense_strike_slopes_local_var<METHOD>(Y, S, Ksh, RDX, n, lane);
                __syncthreads();
            }
            if (!kq_shared) load_xq(Kqb);
        }
        if (more) issue_loads(b_next, koff_next, n_next);   // next surface flies during evaluation + maturity pass
        if (!redo) {
#pragma unroll 1
            for (int q0 = 0, qb = 0; q0 < mK; q0 += 64, ++qb) {
                const int q = q0 + lane;
                const bool act = q < mK;
                double xq;
                if (qb < XQ_REG) xq = qb == 0 ? xq_reg[0] : (qb == 1 ? xq_reg[1] : (qb == 2 ? xq_reg[2] : xq_reg[3]));
                else xq = act ? Kqb[q] : nanv;
                int j = 0;
#pragma unroll
                for (int m = 1; m < 8 * NKB; ++m) j += (Ksh[8 * m] <= xq) ? 8 : 0;
#pragma unroll
                for (int st = 4; st >= 1; st >>= 1) if (Ksh[j + st] <= xq) j += st;
                const double xl = Ksh[n - 1];
                const bool left = !(Ksh[0] <= xq);
                const int jj = j > n - 2 ? n - 2 : j;
                const double x0 = Ksh[jj], x1 = Ksh[jj + 1];
                const int o0 = d_sl(jj), o1 = d_sl(jj + 1);
                double z[DT];
                if (CUB) {
                    const bool ok = !left && ((xq <= xl) || d_extrap_right(METHOD));
                    const double u = xq - x0, t = u * RDX[jj], omt = 1.0 - t;
                    const double w0 = ok ? (1.0 + 2.0 * t) * omt * omt : nanv;
                    const double w1 = t * t * (3.0 - 2.0 * t);
                    const double w2 = u * omt * omt;
                    const double w3 = u * t * (t - 1.0);
                    constexpr int LA = 3;
                    double g0[4], g1[4], g2[4], g3[4];
#pragma unroll
                    for (int r = 0; r < LA; ++r) {
                        g0[r] = Y[r * RS + o0]; g1[r] = Y[r * RS + o1]; g2[r] = S[r * RS + o0]; g3[r] = S[r * RS + o1];
                    }
#pragma unroll
                    for (int r = 0; r < DT; ++r) {
                        if (r + LA < DT) {
                            const int nn = r + LA;
                            g0[nn & 3] = Y[nn * RS + o0]; g1[nn & 3] = Y[nn * RS + o1];
                            g2[nn & 3] = S[nn * RS + o0]; g3[nn & 3] = S[nn * RS + o1];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        z[r] = w0 * g0[r & 3] + w1 * g1[r & 3] + w2 * g2[r & 3] + w3 * g3[r & 3];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    const bool right = j >= n - 1;
                    const bool hold = right && (METHOD == IVS_LINEAR || xq == xl);
                    const double dx = x1 - x0, rdx = refined_rcp(dx);
                    bool slow = !div_safe(dx);
                    constexpr int LA = 6;
                    double g0[8], g1[8];
#pragma unroll
                    for (int r = 0; r < LA; ++r) { g0[r] = Y[r * RS + o0]; g1[r] = Y[r * RS + o1]; }
#pragma unroll
                    for (int r = 0; r < DT; ++r) {
                        if (r + LA < DT) { const int nn = r + LA; g0[nn & 7] = Y[nn * RS + o0]; g1[nn & 7] = Y[nn * RS + o1]; }
                        __builtin_amdgcn_sched_barrier(0);
                        const double y0 = g0[r & 7], y1 = g1[r & 7];
                        double v = lerp_fast(xq, x0, y0, y1, dx, rdx, slow);
                        if (right) v = hold ? y1 : nanv;
                        if (left) v = nanv;
                        z[r] = v;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (__builtin_expect(__ballot(slow && !right && !left) != 0ull, 0)) {
                        if (slow && !right && !left) {
#pragma unroll
                            for (int r = 0; r < DT; ++r) z[r] = lerp_np(xq, x0, Y[r * RS + o0], x1, Y[r * RS + o1]);
                        }
                    }
                }
                if (d_is_local(METHOD)) {
#pragma unroll
                    for (int r = 0; r < DT; ++r) asm volatile("" : "+v"(z[r]));      // see dense_maturity_pass
                }
                if (act) dense_maturity_pass<METHOD, WLDS, false, true, TSHARED>(z, tt, TTp, Wp, outb, q0, lane, true, mT, mK, nostamp, 0, 0, nT);
            }
            if (p.status && lane == 0) p.status[b] = IVS_ST_OK;
        }
        const int64_t it_next2 = wq.finish(pend, more);
        it = it_next; it_next = it_next2; b = b_next; n = n_next; koff = koff_next;
    }
}

}  // namespace ivs
