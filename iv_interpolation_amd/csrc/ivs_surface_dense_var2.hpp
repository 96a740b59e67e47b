// Dense fast path for 65..128 strikes per surface with TWO wavefronts per surface.
//
// The single-wavefront kernel of ivs_surface_dense_var.hpp (NKB = 2) needs 40 KB of LDS per surface, which leaves
// 4 wavefronts per CU and every phase twice as long.  Here a 128-thread workgroup shares the same 40 KB: the LDS
// budget still allows 4 workgroups per CU, but that is now 8 wavefronts, each doing half of every phase:
//   K-phase   wave w factorises strike block w (64 knots); wave 1 appends wave 0's Moebius product (LDS exchange)
//   sweeps    wave w sweeps the logical segments 4w..4w+3 of all 16 rows; the forward carry crosses from wave 0
//             to wave 1 and the backward carry from wave 1 to wave 0 through 16-entry LDS mailboxes
//   strike evaluation + maturity pass
//             mK <= 64: both waves search/weight the same 64 queries, each gathers 8 of the 16 rows, the halves
//                       are exchanged through LDS and each wave writes half of the output rows;
//             mK  > 64: the 64-query blocks alternate between the waves, no exchange.
// Every barrier below is reached by both waves on every path (the redo decision is workgroup-uniform).
#pragma once
#include <type_traits>

#include "ivs_surface_dense_var.hpp"

namespace ivs {

constexpr int V2_RS = 144;                     // row stride of the Y / S planes (8 segments x 18)
// 144 doubles = 288 words = 32 (mod 64 banks): rows t and t+2 would share banks in the rs-lane accesses (16 lanes =
// 4 rows x 4 segments per pass).  Every second pair of rows is pushed 8 doubles further: the four rows of a pass sit
// at bank offsets {0, 32, 16, 48} + const.
__device__ __forceinline__ int v2_row(int t) { return t * V2_RS + 8 * (t >> 1); }
constexpr int V2_PLANE = DT * V2_RS + 64;
constexpr int V2_XCH = 64;                     // doubles in the exchange area
// exchange slots
constexpr int X_MAT = 0;                       // [0..3] wave 0's matrix product
constexpr int X_FWD = 8;                       // [8..23] forward carry of row t
constexpr int X_BWD = 24;                      // [24..39] backward carry of row t
constexpr int X_BAD = 40;                      // [40..41] NaN flag of wave w

__host__ __device__ inline size_t dense_var2_lds_bytes(int mT) {
    return (size_t)(2 * V2_PLANE + 128 + 64 + V2_XCH + (mT <= D_WLDS_MAX_MT ? 4 * D_WLDS_MAX_MT : 0)) * 8;
}

// K-phase of wave w (strike block w).  n in 65..128, so knot 63 is always an interior row and wave 1 can rebuild
// the three scalars it needs from block 0 (c/b, 1/dx_63, 1/dx_62) from the staged strikes itself; only the scan
// total of block 0 has to be handed over.
__device__ __forceinline__ void factor_tables_var2(const double* X, int n, int lane, int w, double* AL, double* CP,
                                                   double* PP, double* QQ, double* PM, double* PI, double* PSI,
                                                   double* XCH) {
    const int ir = w * 64 + lane;
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
    // knot 63 (interior): c = dx_62, b = 2 (dx_62 + dx_63)
    const double k62 = X[62], k63 = X[63], k64 = X[64];
    const double e_crb = (k63 - k62) * refined_rcp(2.0 * ((k63 - k62) + (k64 - k63)));    // same expression as block 0
    const double e_rdx63 = refined_rcp(k64 - k63), e_rdx62 = refined_rcp(k63 - k62);
    double crb_prev = dpp0_f64<DPP_WAVE_SHR1>(crb);
    if (w > 0 && lane == 0) crb_prev = e_crb;
    const bool ident = first || !in;
    const double g = ident ? 0.0 : a * rb * crb_prev;
    double p00 = 1.0, p01 = ident ? 0.0 : -g, p10 = ident ? 0.0 : 1.0, p11 = ident ? 1.0 : 0.0;
    scan_mat2<64>(p00, p01, p10, p11, lane);
    if (w == 0 && lane == 63) { XCH[X_MAT + 0] = p00; XCH[X_MAT + 1] = p01; XCH[X_MAT + 2] = p10; XCH[X_MAT + 3] = p11; }
    __syncthreads();
    if (w > 0) {                                               // append block 0's product on the right
        const double c00 = XCH[X_MAT + 0], c01 = XCH[X_MAT + 1], c10 = XCH[X_MAT + 2], c11 = XCH[X_MAT + 3];
        const double n00 = p00 * c00 + p01 * c10, n01 = p00 * c01 + p01 * c11;
        const double n10 = p10 * c00 + p11 * c10, n11 = p10 * c01 + p11 * c11;
        p00 = n00; p01 = n01; p10 = n10; p11 = n11;
    }
    const double num = p00 + p01, den = p10 + p11;
    const double rw = first ? rb : den * rb * refined_rcp(num);
    const double al = a * rw, cp = c * rw;
    double rdx_prev = dpp0_f64<DPP_WAVE_SHR1>(rdxc);
    if (w > 0 && lane == 0) rdx_prev = e_rdx63;
    const double rdx_next = dpp0_f64<DPP_WAVE_SHL1>(rdxc);          // only row 0 uses it
    double rdxmm = dpp0_f64<DPP_WAVE_SHR1>(rdx_prev);
    if (w > 0 && lane == 0) rdxmm = e_rdx62;
    const double d = first ? dxc + dxp : dxmm + dxm;
    const double rd = refined_rcp(d);
    double pm = 0.0, pp, qq;
    if (first) {
        pp = (dxc + 2.0 * d) * dxp * rdxc * rd * rw;
        qq = dxc * dxc * rdx_next * rd * rw;
    } else if (last) {
        pm = dxm * dxm * rdxmm * rd * rw;
        pp = (2.0 * d + dxm) * dxmm * rdx_prev * rd * rw;
        qq = 0.0;
    } else {
        pp = 3.0 * dxc * rdx_prev * rw;
        qq = 3.0 * dxm * rdxc * rw;
    }
    const double pi = seg16_prefix_prod(in ? -al : 1.0, lane);
    const double psi = seg16_suffix_prod(in ? -cp : 1.0, lane);
    {   // neutral rows beyond n (see factor_tables_var): the sweeps run unmasked
        const int kl = d_sl(ir);
        AL[kl] = in ? al : -1.0; CP[kl] = in ? cp : 0.0; PP[kl] = in ? pp : 0.0; QQ[kl] = in ? qq : 0.0;
        PM[kl] = in ? pm : 0.0; PI[kl] = pi; PSI[kl] = psi;
    }
}

// Slopes of all 16 rows, run-time n in 65..128; wave w owns the logical segments 4w .. 4w+3.
__device__ __forceinline__ void dense_strike_slopes_var2(const double* Y, double* S, const double* Ksh, double* XCH,
                                                         int n, int lane, int w) {
    constexpr int RS = V2_RS;
    double* AL = S;
    double* CP = S + RS;
    double* PP = S + 2 * RS;
    double* QQ = S + 3 * RS;
    double* PM = S + 4 * RS;
    double* PI = S + 5 * RS;
    double* PSI = S + 6 * RS;
    factor_tables_var2(Ksh, n, lane, w, AL, CP, PP, QQ, PM, PI, PSI, XCH);
    __syncthreads();
    const int rs_t = lane >> 2, rs_seg = lane & 3;
    const int sg = rs_seg + 4 * w, kb = 16 * sg, kp = 18 * sg;
    int ln = n - kb; ln = ln < 0 ? 0 : (ln > 16 ? 16 : ln);
    const double* yrow = Y + v2_row(rs_t);
    double d[16];
    double endv, pie;
    {   // ---- local forward sweep
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
            if (m == 0) {                                      // row 0 of the system uses (dy_0, dy_1)
                const double e = y[4] - y[3];
                const bool r0 = sg == 0;
                dA = r0 ? dB : dA; dB = r0 ? e : dB;
            }
            const double r = PM[kp + m] * dM + PP[kp + m] * dA + QQ[kp + m] * dB;
            prev = r - AL[kp + m] * prev;
            d[m] = prev;
            if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        endv = prev;
        pie = PI[kp + 15];
    }
    // ---- forward carries: wave 0 chains its four segments and posts the row's value at the end of segment 3;
    // wave 1 picks it up after the barrier
    auto fwd_chain = [&](double carry_in, double& di, double& tot) {
        di = rs_seg == 0 ? carry_in : 0.0;
        tot = endv + pie * di;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const double v = dpp0_f64<DPP_ROW_SHR(1)>(tot);
            if (rs_seg == j) { di = v; tot = endv + pie * di; }
        }
    };
    double din = 0.0, tot = 0.0;
    if (w == 0) {
        fwd_chain(0.0, din, tot);
        if (rs_seg == 3) XCH[X_FWD + rs_t] = tot;
    }
    __syncthreads();
    if (w == 1) fwd_chain(XCH[X_FWD + rs_t], din, tot);
    // ---- local backward sweep (forward fix-up folded in)
    {
        double nxt = 0.0;
#pragma unroll
        for (int m = 15; m >= 0; --m) {
            const double dp = d[m] + PI[kp + m] * din;
            nxt = dp - CP[kp + m] * nxt;
            d[m] = nxt;
            if ((m & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const double firstv = d[0];
    const double psb = PSI[kp];
    auto bwd_chain = [&](double carry_in, double& si, double& t2) {
        si = rs_seg == 3 ? carry_in : 0.0;
        t2 = firstv + psb * si;
#pragma unroll
        for (int j = 2; j >= 0; --j) {
            const double v = dpp0_f64<DPP_ROW_SHL(1)>(t2);
            if (rs_seg == j) { si = v; t2 = firstv + psb * si; }
        }
    };
    double sin_ = 0.0, t2 = 0.0;
    if (w == 1) {
        bwd_chain(0.0, sin_, t2);
        if (rs_seg == 0) XCH[X_BWD + rs_t] = t2;               // slope at the first knot of logical segment 4
    }
    __syncthreads();
    if (w == 0) bwd_chain(XCH[X_BWD + rs_t], sin_, t2);
#pragma unroll
    for (int m = 0; m < 16; ++m) d[m] = d[m] + PSI[kp + m] * sin_;
    __syncthreads();                                           // table reads done: the S plane may be overwritten
    // all 16 slots of the segment are written (b128, conflict-free); those beyond n are never read
    double* srow = S + v2_row(rs_t) + kp;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double2 v; v.x = d[2 * c]; v.y = d[2 * c + 1];
        *reinterpret_cast<double2*>(srow + 2 * c) = v;
    }
}

// Local methods (pchip, akima), 65..128 strikes: wave w owns the logical segments 4w .. 4w+3.  No carries cross the
// waves (neighbouring quotes are read from LDS); akima's row maximum over all 8 segments = the two waves' maxima,
// exchanged through the LDS mailboxes (one barrier; scanning the partner wave's segment instead cost ~250 instructions).
template <int METHOD>
__device__ __forceinline__ void dense_strike_slopes_local_var2(const double* Y, double* S, const double* Ksh, double* XCH,
                                                               int n, int lane, int w) {
    constexpr bool AK = METHOD == IVS_AKIMA;
    double* R0 = S;
    double* R1 = S + 152;
    double* R2 = S + 304;
    {
        double r0, r1, r2;
        local_tables_rt(Ksh, n, w * 64 + lane, r0, r1, r2);
        const int kl = d_sl(w * 64 + lane);
        R0[kl] = r0; R1[kl] = r1; R2[kl] = r2;
    }
    __syncthreads();
    const int rs_t = lane >> 2, rs_seg = lane & 3;
    const int sg = rs_seg + 4 * w, kb = 16 * sg, kp = 18 * sg;
    const double* yr = Y + v2_row(rs_t);
    double F[19], d[16];
    double thr = 0.0;
    if (AK) {
        segment_secants<true>(yr, R0, kb, kp, n, F);
        double fmax = segment_akima_fmax(F, kb, n);
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP1>(fmax, fmax));
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP2>(fmax, fmax));
        if (rs_seg == 0) XCH[X_FWD + 16 * w + rs_t] = fmax;                 // X_FWD / X_BWD: 32 contiguous slots
        __syncthreads();
        fmax = __builtin_fmax(fmax, XCH[X_FWD + 16 * (1 - w) + rs_t]);
        thr = 1e-9 * fmax;
    } else {
        segment_secants<false>(yr, R0, kb, kp, n, F);
    }
    segment_slopes<METHOD>(F, R1, R2, kb, kp, n, thr, d);
    const double e_last = AK ? 0.0 : pchip_last_slope(yr, R0, R1, R2, n);
    __syncthreads();                               // table reads done: the S plane may be overwritten
    double* srow = S + v2_row(rs_t) + kp;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double2 v; v.x = d[2 * c]; v.y = d[2 * c + 1];
        *reinterpret_cast<double2*>(srow + 2 * c) = v;
    }
    if (!AK && sg == ((n - 1) >> 4)) S[v2_row(rs_t) + d_sl(n - 1)] = e_last;      // same lane, after its row store
}

template <int METHOD, bool WLDS, bool TSHARED = true>
__global__ __launch_bounds__(128, 2) void surface_dense_var2_kernel(SurfaceParams p, VarList list) {
    constexpr bool CUB = d_is_hermite(METHOD);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // wave index, scalar
    const int mT = p.mT, mK = p.mK;
    double* Y = reinterpret_cast<double*>(smem);
    double* S = Y + V2_PLANE;
    double* Ksh = S + V2_PLANE;            // 128 entries; strikes beyond n are +inf
    double* TT = Ksh + 128;
    double* XCH = TT + 64;
    double* W = XCH + V2_XCH;
    const double nanv = __builtin_nan(""), inf = __builtin_inf();
    auto nostamp = [](int) {};

    TqTables tt;
    const int nT = p.nT;                                                   // 4..16, uniform over the batch
    const double* TTp = TT;
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

    const int64_t limit = list.items ? (int64_t)*list.count : p.B;
    auto at = [&](int64_t it, int& n, int64_t& koff) -> int64_t {
        if (list.items) { const VarItem v = list.items[it]; n = v.n; koff = v.koff; return v.b; }
        n = p.nK; koff = it * p.k_stride; return it;
    };
    bool told = false;
    auto tag = [&](int64_t b) {
        if (threadIdx.x == 0) { reinterpret_cast<unsigned long long*>(p.out + b * (int64_t)mT * mK)[0] = D_SENTINEL; count_redo(p, told); }
    };

    double pre[DT], pre_k;
    int n = 0, n_next = 0;
    int64_t koff = 0, koff_next = 0;
    auto issue_loads = [&](int64_t b, int64_t ko, int nn) {
        const double* sb = p.k_off ? p.sigma + (int64_t)nT * ko : p.sigma + b * (int64_t)nT * p.nK;
        const int k = w * 64 + lane;
#pragma unroll
        for (int t = 0; t < DT; ++t) pre[t] = (t < nT && k < nn) ? sb[(int64_t)t * nn + k] : 0.0;
        pre_k = k < nn ? p.K[ko + k] : inf;
    };

    int64_t it = blockIdx.x, b = 0;
    if (it < limit) { b = at(it, n, koff); issue_loads(b, koff, n); }

    while (it < limit) {
        __syncthreads();                                   // previous surface fully consumed by both waves
        unsigned long long bad = 0ull;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
            Y[v2_row(t) + d_sl(w * 64 + lane)] = pre[t];
            bad |= __ballot(pre[t] != pre[t]);
        }
        Ksh[w * 64 + lane] = pre_k;
        if (lane == 0) XCH[X_BAD + w] = bad != 0ull ? 1.0 : 0.0;
        double* outb = p.out + b * (int64_t)mT * mK;
        const double* Kqb = p.Kq + b * p.kq_stride;
        const int64_t it_next = it + gridDim.x;
        int64_t b_next = 0;
        if (it_next < limit) b_next = at(it_next, n_next, koff_next);
        if (!TSHARED) dense_t_phase<METHOD, WLDS, true>(p.T + b * p.t_stride, p.Tq + b * p.tq_stride, mT, lane, S + 600, TT, W,
                                                        tt, nT, S);      // per-surface maturities; scratch = the free S plane
        __syncthreads();                                   // staging and both flags visible
        const bool redo = XCH[X_BAD] != 0.0 || XCH[X_BAD + 1] != 0.0 || tt.unsorted;       // workgroup-uniform
        if (redo) {
            tag(b);
        } else {
            if (d_is_nak(METHOD)) {
                dense_strike_slopes_var2(Y, S, Ksh, XCH, n, lane, w);
                __syncthreads();
            } else if (d_is_local(METHOD)) {
                dense_strike_slopes_local_var2<METHOD>(Y, S, Ksh, XCH, n, lane, w);
                __syncthreads();
            }
            if (!kq_shared) load_xq(Kqb);
        }
        if (it_next < limit) issue_loads(b_next, koff_next, n_next);
        if (!redo) {
            const bool split_rows = mK <= 64;              // uniform: one query block -> the waves share it by rows
#pragma unroll 1
            for (int q0 = 0, qb = 0; q0 < mK; q0 += 64, ++qb) {
                if (!split_rows && (qb & 1) != w) continue;
                const int q = q0 + lane;
                const bool act = q < mK;
                double xq;
                if (qb < XQ_REG) xq = qb == 0 ? xq_reg[0] : (qb == 1 ? xq_reg[1] : (qb == 2 ? xq_reg[2] : xq_reg[3]));
                else xq = act ? Kqb[q] : nanv;
                int j = 0;
#pragma unroll
                for (int m = 1; m < 16; ++m) j += (Ksh[8 * m] <= xq) ? 8 : 0;
#pragma unroll
                for (int st = 4; st >= 1; st >>= 1) if (Ksh[j + st] <= xq) j += st;
                const double xl = Ksh[n - 1];
                const bool left = !(Ksh[0] <= xq);
                const int jj = j > n - 2 ? n - 2 : j;
                const double x0 = Ksh[jj], x1 = Ksh[jj + 1];
                const int o0 = d_sl(jj), o1 = d_sl(jj + 1);
                // rows this wave evaluates in the strike direction
                const int r_lo = split_rows ? 8 * w : 0;
                double z[DT];
                auto strike_rows = [&](auto NR_TAG, int rbase) {
                    constexpr int NR = decltype(NR_TAG)::value;
                    const double* Yb = Y;
                    const double* Sb = S;
                    if (CUB) {
                        const bool ok = !left && ((xq <= xl) || d_extrap_right(METHOD));
                        const double u = xq - x0, t = u * refined_rcp(x1 - x0), omt = 1.0 - t;   // no 1/dx table: LDS budget
                        const double w0 = ok ? (1.0 + 2.0 * t) * omt * omt : nanv;
                        const double w1 = t * t * (3.0 - 2.0 * t);
                        const double w2 = u * omt * omt;
                        const double w3 = u * t * (t - 1.0);
                        constexpr int LA = 3;
                        double g0[4], g1[4], g2[4], g3[4];
#pragma unroll
                        for (int r = 0; r < LA; ++r) {
                            g0[r] = Yb[v2_row(rbase + r) + o0]; g1[r] = Yb[v2_row(rbase + r) + o1]; g2[r] = Sb[v2_row(rbase + r) + o0]; g3[r] = Sb[v2_row(rbase + r) + o1];
                        }
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            if (r + LA < NR) {
                                const int nn = r + LA;
                                g0[nn & 3] = Yb[v2_row(rbase + nn) + o0]; g1[nn & 3] = Yb[v2_row(rbase + nn) + o1];
                                g2[nn & 3] = Sb[v2_row(rbase + nn) + o0]; g3[nn & 3] = Sb[v2_row(rbase + nn) + o1];
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
                        for (int r = 0; r < LA; ++r) { g0[r] = Yb[v2_row(rbase + r) + o0]; g1[r] = Yb[v2_row(rbase + r) + o1]; }
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            if (r + LA < NR) { const int nn = r + LA; g0[nn & 7] = Yb[v2_row(rbase + nn) + o0]; g1[nn & 7] = Yb[v2_row(rbase + nn) + o1]; }
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
                                for (int r = 0; r < NR; ++r) z[r] = lerp_np(xq, x0, Yb[v2_row(rbase + r) + o0], x1, Yb[v2_row(rbase + r) + o1]);
                            }
                        }
                    }
                };
                if (split_rows) {
                    strike_rows(std::integral_constant<int, 8>{}, r_lo);
                    __syncthreads();                       // every read of the Y / S planes is done
                    double* ZX = Y;                        // [16][64] exchange of the strike-pass values
#pragma unroll
                    for (int r = 0; r < 8; ++r) ZX[(r_lo + r) * 64 + lane] = z[r];
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < DT; ++r) z[r] = ZX[r * 64 + lane];
                    if (d_is_local(METHOD)) {
#pragma unroll
                        for (int r = 0; r < DT; ++r) asm volatile("" : "+v"(z[r]));      // see dense_maturity_pass
                    }
                    const int half = (mT + 1) >> 1;
                    if (act) dense_maturity_pass<METHOD, WLDS, true, true, TSHARED>(z, tt, TTp, Wp, outb, q0, lane, true, mT, mK, nostamp,
                                                                                    w == 0 ? 0 : half, w == 0 ? half : mT, nT);
                } else {
                    strike_rows(std::integral_constant<int, DT>{}, 0);
                    if (d_is_local(METHOD)) {
#pragma unroll
                        for (int r = 0; r < DT; ++r) asm volatile("" : "+v"(z[r]));
                    }
                    if (act) dense_maturity_pass<METHOD, WLDS, false, true, TSHARED>(z, tt, TTp, Wp, outb, q0, lane, true, mT, mK, nostamp, 0, 0, nT);
                }
            }
            if (p.status && threadIdx.x == 0) p.status[b] = IVS_ST_OK;
        }
        it = it_next; b = b_next; n = n_next; koff = koff_next;
    }
}

#ifndef IVS_DIAG_MINIMAL
// Dispatch for variable strike counts: class 4..64 on the one-wavefront kernel, class 65..128 on the two-wavefront
// kernel, then the filtered generic redo pass.  Returns 1 if dispatched, 0 if the batch is not covered.
// Scratch (ragged work lists + counters, batch-wide maturity tables) comes from the caller's workspace: no allocation.
inline int launch_surface_dense_var(const SurfaceParams& p_in, const LaunchCtx& cx, const char** name) {
    SurfaceParams p = p_in;
    hipStream_t st = cx.st;
    if (p.nT < 4 || p.nT > DT || p.mT > D_MAX_MT) return 0;
    const bool tsh = p.t_stride == 0 && p.tq_stride == 0;
    if (p.nK < 4 || p.nK > 128) return 0;
    if (!p.k_off && p.k_stride != 0 && p.k_stride < p.nK) return 0;
    if (generic_lds_bytes(p.nK, p.nT) > 160 * 1024) return 0;
    if (!(d_is_hermite(p.method) || p.method == IVS_LINEAR || p.method == IVS_SLINEAR)) return 0;
    const bool wl = p.mT <= D_WLDS_MAX_MT;
    const bool need1 = p.k_off ? true : p.nK <= 64;
    const bool need2 = p.nK > 64;
    if (tsh) {
        TqShared* tq = reinterpret_cast<TqShared*>(cx.ws);
        launch_tq_tables<true>(p, tq, st);
        p.tqs = tq;
        p.redo = tq->redo;
    } else {                                     // per-surface maturities: no table kernel runs, the queue heads are zeroed here
        if (hipMemsetAsync(reinterpret_cast<TqShared*>(cx.ws)->queue, 0, sizeof(TqShared::queue), st) != hipSuccess) return -1;
    }
    p.queue = reinterpret_cast<TqShared*>(cx.ws)->queue;
    // ragged batch: classify once into one work list per size class (workspace: counters, then V_NCLASS x B items)
    VarItem* lists = nullptr;
    int32_t* counts = nullptr;
    if (p.k_off) {
        if (p.B > 0x7fffffffLL) return 0;
        counts = reinterpret_cast<int32_t*>(cx.ws + WS_TQ_BYTES);
        lists = reinterpret_cast<VarItem*>(cx.ws + WS_TQ_BYTES + WS_COUNTS_BYTES);
        if (hipMemsetAsync(counts, 0, WS_COUNTS_BYTES, st) != hipSuccess) return -1;
        int64_t cb = (p.B + 1023) / 1024;
        const int64_t cap = (int64_t)cx.num_cu * 8;
        if (cb > cap) cb = cap;
        hipLaunchKernelGGL(var_classify_kernel, dim3((unsigned)cb), dim3(256), 0, st, p, lists, lists + p.B, counts);
    }
    const VarList wl1{lists, counts, 0}, wl2{lists ? lists + p.B : nullptr, counts ? counts + 1 : nullptr, 1};
    auto grid_for = [&](size_t lds) {
        int per_cu = (int)((160 * 1024) / lds);
        per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
        int64_t g = (int64_t)cx.num_cu * per_cu;
        return g > p.B ? p.B : g;
    };
#define IVS_VAR_LAUNCH1(M)                                                                                           \
    {                                                                                                                \
        const size_t lds = dense_var_lds_bytes<1>(p.mT);                                                             \
        const int64_t grid = grid_for(lds);                                                                          \
        if (wl && tsh) hipLaunchKernelGGL((surface_dense_var_kernel<M, 1, true, true>), dim3((unsigned)grid), dim3(64), lds, st, p, wl1);    \
        else if (tsh) hipLaunchKernelGGL((surface_dense_var_kernel<M, 1, false, true>), dim3((unsigned)grid), dim3(64), lds, st, p, wl1);    \
        else if (wl) hipLaunchKernelGGL((surface_dense_var_kernel<M, 1, true, false>), dim3((unsigned)grid), dim3(64), lds, st, p, wl1);     \
        else hipLaunchKernelGGL((surface_dense_var_kernel<M, 1, false, false>), dim3((unsigned)grid), dim3(64), lds, st, p, wl1);            \
    }
#define IVS_VAR_LAUNCH2(M)                                                                                           \
    {                                                                                                                \
        const size_t lds = dense_var2_lds_bytes(p.mT);          /* 40 KB: below the 64 KiB default limit */          \
        const int64_t grid = grid_for(lds);                                                                          \
        if (wl && tsh) hipLaunchKernelGGL((surface_dense_var2_kernel<M, true, true>), dim3((unsigned)grid), dim3(128), lds, st, p, wl2);     \
        else if (tsh) hipLaunchKernelGGL((surface_dense_var2_kernel<M, false, true>), dim3((unsigned)grid), dim3(128), lds, st, p, wl2);     \
        else if (wl) hipLaunchKernelGGL((surface_dense_var2_kernel<M, true, false>), dim3((unsigned)grid), dim3(128), lds, st, p, wl2);      \
        else hipLaunchKernelGGL((surface_dense_var2_kernel<M, false, false>), dim3((unsigned)grid), dim3(128), lds, st, p, wl2);             \
    }
#define IVS_VAR_CASE(M, NAME)                                                  \
    case M:                                                                    \
        if (need1) IVS_VAR_LAUNCH1(M)                                          \
        if (need2) IVS_VAR_LAUNCH2(M)                                          \
        *name = NAME;                                                          \
        break;
    bool known = true;
    switch (p.method) {
        IVS_VAR_CASE(IVS_LINEAR, "surface_dense_var_kernel<linear>")
        IVS_VAR_CASE(IVS_CUBIC, "surface_dense_var_kernel<cubic>")
        IVS_VAR_CASE(IVS_CUBICSPLINE, "surface_dense_var_kernel<cubicspline>")
        IVS_VAR_CASE(IVS_SLINEAR, "surface_dense_var_kernel<slinear>")
        IVS_VAR_CASE(IVS_PCHIP, "surface_dense_var_kernel<pchip>")
        IVS_VAR_CASE(IVS_AKIMA, "surface_dense_var_kernel<akima>")
        default: known = false;
    }
#undef IVS_VAR_CASE
#undef IVS_VAR_LAUNCH1
#undef IVS_VAR_LAUNCH2
    if (!known) return 0;
    const bool launched = hipGetLastError() == hipSuccess;
    if (launched) launch_surface_generic<true>(p, cx);
    return launched ? 1 : -1;
}

#endif  // IVS_DIAG_MINIMAL

}  // namespace ivs
