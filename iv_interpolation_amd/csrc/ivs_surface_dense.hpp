// Dense fast path: 64 strikes x 16 maturities per surface, no missing quotes.
//
// One wavefront (64-lane workgroup) per surface, persistent grid-stride loop, next surface's
// quotes prefetched into registers while the current one is computed.  Three lane layouts:
//   k-lane   lane = strike k           coalesced loads; tridiagonal FACTORISATION of the not-a-knot
//                                      system by a Kogge-Stone scan of 2x2 Moebius matrices (pivots
//                                      w_i = b_i - a_i c_{i-1} / w_{i-1} without a serial chain)
//   rs-lane  lane = (row t, segment)   the 16 right-hand sides (one per maturity row) are swept
//                                      with 4 lanes per row, 16 knots per lane, two-level (local
//                                      sweep + 3-step cross-segment carry), all 64 lanes busy
//   q-lane   lane = output strike q    gather (y_j, y_j+1, s_j, s_j+1) from LDS, Hermite weights
//                                      once per lane; then the maturity-direction system is
//                                      solved per lane entirely in registers and evaluated at Tq
// Quotes are staged in LDS in a padded layout (row stride 72, segment stride 18 doubles) that makes
// the b128 reads of the rs-lanes and the b128 writes of the k-lanes bank-conflict free.
// Surfaces that contain a NaN quote (or batches whose Tq is not ascending) are tagged with a
// sentinel in out[b][0] and redone by the generic kernel in a second, filtered launch.
#pragma once
#include "ivs_surface_generic.hpp"

namespace ivs {

constexpr int DK = 64, DT = 16;
constexpr int D_RS = 72;                         // row stride in doubles (4 x (16 + 2))
constexpr unsigned long long D_SENTINEL = REDO_SENTINEL;           // NaN payload = "redo generically"

__device__ __forceinline__ int d_sl(int k) { return (k >> 4) * 18 + (k & 15); }

__host__ __device__ inline size_t dense_lds_bytes(int mT) {
    // Y, S planes; Ksh, RDX, AL, CP, PP, QQ, PI, PSI (64 each); Tsh + T tables [16][4]; W [mT][4]; CNT
    return (size_t)(2 * DT * D_RS + 8 * 64 + 16 + 64 + 4 * (size_t)mT) * 8 + 32 * 4;
}

// codes of a query row in the maturity direction
constexpr int TQ_LEFT = -1, TQ_HOLD = 15, TQ_NAN = 16;

// Factorisation tables of the not-a-knot slope system on N knots X[0..N) (N = 64 or 16), computed by
// lanes 0..N-1.  Conventions (oracle nak_slopes()):  forward  dp_i = PP_i*dyA + QQ_i*dyB - AL_i*dp_{i-1},
// backward s_i = dp_i - CP_i*s_{i+1};  (dyA, dyB) = (dy_{i-1}, dy_i) for interior rows,
// (dy_0, dy_1) for row 0 and (dy_{N-3}, dy_{N-2}) for row N-1.
template <int N>
__device__ __forceinline__ void factor_tables(const double* X, int lane, double& al, double& cp, double& pp,
                                              double& qq, double& rdx_out) {
    const int i = lane < N ? lane : N - 1;
    const double x0 = X[i];
    const double xp = X[i + 1 < N ? i + 1 : N - 1];
    const double xpp = X[i + 2 < N ? i + 2 : N - 1];
    const double xm = X[i > 0 ? i - 1 : 0];
    const double xmm = X[i > 1 ? i - 2 : 0];
    const double dxc = xp - x0;          // dx[i]
    const double dxm = x0 - xm;          // dx[i-1]
    const double dxp = xpp - xp;         // dx[i+1]
    const double dxmm = xm - xmm;        // dx[i-2]
    const double rdxc = 1.0 / dxc;
    double a, b, c;
    if (i == 0) { a = 0.0; b = dxp; c = dxc + dxp; }
    else if (i == N - 1) { a = dxmm + dxm; b = dxmm; c = 0.0; }
    else { a = dxc; b = 2.0 * (dxm + dxc); c = dxm; }
    // g_i = a_i c_{i-1} / (b_i b_{i-1})
    const double c_prev = __shfl_up(c, 1);
    const double b_prev = __shfl_up(b, 1);
    const double g = i == 0 ? 0.0 : (a * c_prev) / (b * b_prev);
    // omega_i = 1 - g_i / omega_{i-1}: prefix product of M_i = [[1,-g_i],[1,0]] (identity on lane 0)
    double p00 = 1.0, p01 = i == 0 ? 0.0 : -g, p10 = i == 0 ? 0.0 : 1.0, p11 = i == 0 ? 1.0 : 0.0;
#pragma unroll
    for (int s = 1; s < N; s <<= 1) {
        const double e = __shfl_up(p00, s), f = __shfl_up(p01, s), gg = __shfl_up(p10, s), h = __shfl_up(p11, s);
        if (i >= s) {
            const double n00 = p00 * e + p01 * gg, n01 = p00 * f + p01 * h;
            const double n10 = p10 * e + p11 * gg, n11 = p10 * f + p11 * h;
            p00 = n00; p01 = n01; p10 = n10; p11 = n11;
        }
    }
    const double num = p00 + p01, den = p10 + p11;          // omega_i = num / den (applied to omega_0 = 1)
    const double rw = i == 0 ? 1.0 / b : den / (b * num);   // 1 / w_i
    al = a * rw;
    cp = c * rw;
    const double rdx_prev = __shfl_up(rdxc, 1);             // 1/dx[i-1]
    const double rdx_next = __shfl_down(rdxc, 1);           // 1/dx[i+1]
    if (i == 0) {
        const double d = dxc + dxp;
        pp = (dxc + 2.0 * d) * dxp * rdxc / d * rw;          // * dy_0
        qq = dxc * dxc * rdx_next / d * rw;                  // * dy_1
    } else if (i == N - 1) {
        const double d = dxmm + dxm;
        pp = dxm * dxm / (dxmm * d) * rw;                    // * dy_{N-3}
        qq = (2.0 * d + dxm) * dxmm / (dxm * d) * rw;        // * dy_{N-2}
    } else {
        pp = 3.0 * dxc * rdx_prev * rw;                      // * dy_{i-1}
        qq = 3.0 * dxm * rdxc * rw;                          // * dy_i
    }
    rdx_out = rdxc;
}

// segmented inclusive prefix product within aligned groups of 16 lanes (towards higher lanes)
__device__ __forceinline__ double seg16_prefix_prod(double v, int lane) {
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
        const double o = __shfl_up(v, s);
        if ((lane & 15) >= s) v *= o;
    }
    return v;
}
__device__ __forceinline__ double seg16_suffix_prod(double v, int lane) {
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
        const double o = __shfl_down(v, s);
        if ((lane & 15) + s < 16) v *= o;
    }
    return v;
}

template <int METHOD>
__global__ __launch_bounds__(64) void surface_dense_kernel(SurfaceParams p) {
    constexpr bool CUB = METHOD == IVS_CUBIC || METHOD == IVS_CUBICSPLINE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int mT = p.mT, mK = p.mK;
    double* Y = reinterpret_cast<double*>(smem);
    double* S = Y + DT * D_RS;
    double* Ksh = S + DT * D_RS;
    double* RDX = Ksh + 64;
    double* AL = RDX + 64;
    double* CP = AL + 64;
    double* PP = CP + 64;
    double* QQ = PP + 64;
    double* PI = QQ + 64;
    double* PSI = PI + 64;
    double* Tsh = PSI + 64;            // 16
    double* TT = Tsh + 16;             // [16][4] = {PP, QQ, AL, CP} of the maturity system
    double* W = TT + 64;               // [mT][4]
    int* CNT = reinterpret_cast<int*>(W + 4 * (size_t)mT);   // [0]=left, [1..15]=interval 0..14, [16]=hold, [17]=nan, [18]=unsorted

    const bool t_shared = p.t_stride == 0 && p.tq_stride == 0;
    const int rs_t = lane >> 2, rs_seg = lane & 3;

    // ---- T-phase: maturity-direction tables + per-query-row weights
    auto t_phase = [&](const double* Tb, const double* Tqb) {
        if (lane < DT) Tsh[lane] = Tb[lane];
        __syncthreads();
        if (CUB) {
            double al, cp, pp, qq, rdx;
            factor_tables<DT>(Tsh, lane, al, cp, pp, qq, rdx);
            if (lane < DT) { TT[lane * 4 + 0] = pp; TT[lane * 4 + 1] = qq; TT[lane * 4 + 2] = al; TT[lane * 4 + 3] = cp; }
        }
        int c_left = 0, c_hold = 0, c_nan = 0, unsorted = 0;
        int c_iv[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) c_iv[j] = 0;
        double carry = -__builtin_inf();
        for (int q0 = 0; q0 < mT; q0 += 64) {
            const int tq = q0 + lane;
            const bool act = tq < mT;
            const double x = act ? Tqb[tq] : __builtin_inf();
            // largest j with Tsh[j] <= x, or -1
            int j = -1;
            if (Tsh[0] <= x) {
                j = 0;
#pragma unroll
                for (int st = 8; st >= 1; st >>= 1) if (Tsh[j + st] <= x) j += st;
            }
            int code;
            const double tl = Tsh[DT - 1];
            if (j < 0) code = TQ_LEFT;
            else if (j >= DT - 1) {
                if (METHOD == IVS_LINEAR) code = TQ_HOLD;
                else if (METHOD == IVS_SLINEAR) code = (x == tl) ? TQ_HOLD : TQ_NAN;
                else if (METHOD == IVS_CUBIC) code = (x == tl) ? DT - 2 : TQ_NAN;
                else code = DT - 2;
            } else code = j;
            if (!(x == x)) code = TQ_LEFT;                    // NaN query -> NaN row
            if (act) {
                const int jj = code >= 0 && code <= DT - 2 ? code : 0;
                const double x0 = Tsh[jj], x1 = Tsh[jj + 1];
                double w0, w1, w2, w3;
                if (CUB) {
                    const double h = x1 - x0, u = x - x0, t = u / h, omt = 1.0 - t;
                    w0 = (1.0 + 2.0 * t) * omt * omt;         // h00
                    w1 = t * t * (3.0 - 2.0 * t);             // h01
                    w2 = u * omt * omt;                       // h * h10
                    w3 = u * t * (t - 1.0);                   // h * h11
                } else { w0 = x; w1 = x0; w2 = x1; w3 = 0.0; }
                W[tq * 4 + 0] = w0; W[tq * 4 + 1] = w1; W[tq * 4 + 2] = w2; W[tq * 4 + 3] = w3;
            }
            c_left += __popcll(__ballot(act && code == TQ_LEFT));
            c_hold += __popcll(__ballot(act && code == TQ_HOLD));
            c_nan += __popcll(__ballot(act && code == TQ_NAN));
#pragma unroll
            for (int jv = 0; jv < 15; ++jv) c_iv[jv] += __popcll(__ballot(act && code == jv));
            // ascending check (codes must come out grouped: left, 0..14, hold/nan)
            double xprev = __shfl_up(x, 1);
            if (lane == 0) xprev = carry;
            unsorted |= __ballot(act && ((x < xprev) || !(x == x))) != 0ull;
            carry = __shfl(x, 63);
        }
        if (lane == 0) {
            CNT[0] = c_left; CNT[16] = c_hold; CNT[17] = c_nan; CNT[18] = unsorted;
#pragma unroll
            for (int jv = 0; jv < 15; ++jv) CNT[1 + jv] = c_iv[jv];
        }
        __syncthreads();
    };

    if (t_shared) t_phase(p.T, p.Tq);

    // ---- prefetch registers (k-lane layout: chunk i of the surface = 16 B at i*1024 + lane*16)
    double2 pre[8];
    double pre_k;
    auto prefetch = [&](int64_t b) {
        const double* sb = p.sigma + b * (int64_t)(DT * DK);
        const double2* s2 = reinterpret_cast<const double2*>(sb);
#pragma unroll
        for (int i = 0; i < 8; ++i) pre[i] = s2[i * 64 + lane];
        pre_k = p.K[b * p.k_stride + lane];
    };
    int64_t b = blockIdx.x;
    if (b < p.B) prefetch(b);

    for (; b < p.B; b += gridDim.x) {
        __syncthreads();                                   // everyone is done reading the previous surface's LDS
        // ---- stage quotes: chunk i, lane -> row t = 2i + (lane>>5), k = 2*(lane&31)
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = 2 * i + (lane >> 5), k = 2 * (lane & 31);
            *reinterpret_cast<double2*>(&Y[t * D_RS + d_sl(k)]) = pre[i];
            bad |= !(pre[i].x == pre[i].x) || !(pre[i].y == pre[i].y);
        }
        Ksh[lane] = pre_k;
        const int64_t bn = b + gridDim.x;
        if (bn < p.B) prefetch(bn);                        // next surface's loads fly during this one's math
        double* outb = p.out + b * (int64_t)mT * mK;
        if (!t_shared) t_phase(p.T + b * p.t_stride, p.Tq + b * p.tq_stride);   // contains a barrier
        const bool redo = __ballot(bad) != 0ull || CNT[18] != 0;
        if (redo) {                                        // wave-uniform
            if (lane == 0) reinterpret_cast<unsigned long long*>(outb)[0] = D_SENTINEL;
            continue;
        }
        __syncthreads();
        if (CUB) {
            // ---- K-phase (k-lane)
            double al, cp, pp, qq, rdx;
            factor_tables<DK>(Ksh, lane, al, cp, pp, qq, rdx);
            AL[lane] = al; CP[lane] = cp; PP[lane] = pp; QQ[lane] = qq; RDX[lane] = rdx;
            PI[lane] = seg16_prefix_prod(-al, lane);       // prod_{seg start..i} (-AL)
            PSI[lane] = seg16_suffix_prod(-cp, lane);      // prod_{i..seg end} (-CP)
            __syncthreads();
            // ---- strike sweeps (rs-lane): row rs_t, knots rs_seg*16 .. +15
            const double* yrow = Y + rs_t * D_RS + rs_seg * 18;
            const int kb = rs_seg * 16;
            double y[18];                                  // y[m+1] = y_{kb+m}, m = -1..16
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double2 v = *reinterpret_cast<const double2*>(yrow + 2 * c);
                y[1 + 2 * c] = v.x; y[2 + 2 * c] = v.y;
            }
            y[0] = rs_seg > 0 ? Y[rs_t * D_RS + d_sl(kb - 1)] : 0.0;
            y[17] = rs_seg < 3 ? Y[rs_t * D_RS + d_sl(kb + 16)] : 0.0;
            double dy[17];                                 // dy[m+1] = y_{kb+m+1} - y_{kb+m}, m = -1..15
#pragma unroll
            for (int m = 0; m < 17; ++m) dy[m] = y[m + 1] - y[m];
            double d[16];
            double prev = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                double dA = dy[m], dB = dy[m + 1];         // (dy_{i-1}, dy_i)
                if (m == 0) { dA = rs_seg == 0 ? dy[1] : dy[0]; dB = rs_seg == 0 ? dy[2] : dy[1]; }
                if (m == 15) { dA = rs_seg == 3 ? dy[14] : dy[15]; dB = rs_seg == 3 ? dy[15] : dy[16]; }
                const double r = PP[kb + m] * dA + QQ[kb + m] * dB;
                prev = r - AL[kb + m] * prev;
                d[m] = prev;
            }
            // carry across the 4 segments of the row
            const double pie = PI[kb + 15];
            double din = 0.0, tot = d[15];
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                const double v = __shfl_up(tot, 1);
                if (rs_seg == j) { din = v; tot = d[15] + pie * din; }
            }
            double nxt = 0.0;
#pragma unroll
            for (int m = 15; m >= 0; --m) {
                const double dp = d[m] + PI[kb + m] * din;
                nxt = dp - CP[kb + m] * nxt;
                d[m] = nxt;                                // local backward solution
            }
            const double psb = PSI[kb];
            double sin_ = 0.0;
            tot = d[0];
#pragma unroll
            for (int j = 2; j >= 0; --j) {
                const double v = __shfl_down(tot, 1);
                if (rs_seg == j) { sin_ = v; tot = d[0] + psb * sin_; }
            }
            double* srow = S + rs_t * D_RS + rs_seg * 18;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                double2 v;
                v.x = d[2 * c] + PSI[kb + 2 * c] * sin_;
                v.y = d[2 * c + 1] + PSI[kb + 2 * c + 1] * sin_;
                *reinterpret_cast<double2*>(srow + 2 * c) = v;
            }
            __syncthreads();
        }
        // counts of query rows per maturity interval -> scalar registers
        const int n_left = __builtin_amdgcn_readfirstlane(CNT[0]);
        const int n_hold = __builtin_amdgcn_readfirstlane(CNT[16]);
        const int n_nan = __builtin_amdgcn_readfirstlane(CNT[17]);

        const double* Kqb = p.Kq + b * p.kq_stride;
        for (int q0 = 0; q0 < mK; q0 += 64) {
            const int q = q0 + lane;
            const bool act = q < mK;
            const double xq = act ? Kqb[q] : __builtin_nan("");
            // ---- strike evaluation (q-lane)
            int j = -1;
            if (Ksh[0] <= xq) {
                j = 0;
#pragma unroll
                for (int st = 32; st >= 1; st >>= 1) if (Ksh[j + st] <= xq) j += st;
            }
            const double xl = Ksh[DK - 1];
            const int jj = j < 0 ? 0 : (j > DK - 2 ? DK - 2 : j);
            const double x0 = Ksh[jj], x1 = Ksh[jj + 1];
            const int o0 = d_sl(jj), o1 = d_sl(jj + 1);
            double z[DT];
            if (CUB) {
                const bool ok = j >= 0 && ((xq <= xl) || METHOD == IVS_CUBICSPLINE);
                const double u = xq - x0, t = u * RDX[jj], omt = 1.0 - t;
                const double nanv = __builtin_nan("");
                const double w0 = ok ? (1.0 + 2.0 * t) * omt * omt : nanv;
                const double w1 = t * t * (3.0 - 2.0 * t);
                const double w2 = u * omt * omt;
                const double w3 = u * t * (t - 1.0);
#pragma unroll
                for (int tt = 0; tt < DT; ++tt)
                    z[tt] = w0 * Y[tt * D_RS + o0] + w1 * Y[tt * D_RS + o1] + w2 * S[tt * D_RS + o0] + w3 * S[tt * D_RS + o1];
            } else {
                const bool left = j < 0;
                const bool right = j >= DK - 1;
                const bool hold = right && (METHOD == IVS_LINEAR || xq == xl);
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    const double y0 = Y[tt * D_RS + o0], y1 = Y[tt * D_RS + o1];
                    double r = lerp_np(xq, x0, y0, x1, y1);
                    if (right) r = hold ? y1 : __builtin_nan("");       // jj = 62 -> y1 is the last quote
                    if (left) r = __builtin_nan("");
                    z[tt] = r;
                }
            }
            // ---- maturity direction (q-lane, registers)
            double* orow = outb + q;
            int tq = 0;
            const double nanv = __builtin_nan("");
            for (int c = 0; c < n_left; ++c, ++tq) if (act) orow[(int64_t)tq * mK] = nanv;
            if (CUB) {
                double dz[15];
#pragma unroll
                for (int i = 0; i < 15; ++i) dz[i] = z[i + 1] - z[i];
                double s[DT];
                double prev = 0.0;
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const double2 pq = *reinterpret_cast<const double2*>(TT + i * 4);
                    const double2 ac = *reinterpret_cast<const double2*>(TT + i * 4 + 2);
                    const double dA = i == 0 ? dz[0] : (i == DT - 1 ? dz[DT - 3] : dz[i - 1]);
                    const double dB = i == 0 ? dz[1] : (i == DT - 1 ? dz[DT - 2] : dz[i]);
                    prev = (pq.x * dA + pq.y * dB) - ac.x * prev;
                    s[i] = prev;
                }
#pragma unroll
                for (int i = DT - 2; i >= 0; --i) s[i] = s[i] - TT[i * 4 + 3] * s[i + 1];
#pragma unroll
                for (int jv = 0; jv < DT - 1; ++jv) {
                    const int n = __builtin_amdgcn_readfirstlane(CNT[1 + jv]);
                    for (int c = 0; c < n; ++c, ++tq) {
                        const double2 wa = *reinterpret_cast<const double2*>(W + tq * 4);
                        const double2 wb = *reinterpret_cast<const double2*>(W + tq * 4 + 2);
                        const double r = wa.x * z[jv] + wa.y * z[jv + 1] + wb.x * s[jv] + wb.y * s[jv + 1];
                        if (act) orow[(int64_t)tq * mK] = r;
                    }
                }
            } else {
#pragma unroll
                for (int jv = 0; jv < DT - 1; ++jv) {
                    const int n = __builtin_amdgcn_readfirstlane(CNT[1 + jv]);
                    for (int c = 0; c < n; ++c, ++tq) {
                        const double2 wa = *reinterpret_cast<const double2*>(W + tq * 4);   // {x, T_j}
                        const double x1t = W[tq * 4 + 2];                                     // T_{j+1}
                        const double r = lerp_np(wa.x, wa.y, z[jv], x1t, z[jv + 1]);
                        if (act) orow[(int64_t)tq * mK] = r;
                    }
                }
            }
            for (int c = 0; c < n_hold; ++c, ++tq) if (act) orow[(int64_t)tq * mK] = z[DT - 1];
            for (int c = 0; c < n_nan; ++c, ++tq) if (act) orow[(int64_t)tq * mK] = nanv;
        }
        if (p.status && lane == 0) p.status[b] = IVS_ST_OK;
    }
}

inline void set_max_lds(const void* fn) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// Launch the generic kernel (FILTER selects the "only tagged surfaces" variant).  Returns false if the
// shape does not fit in LDS.
template <bool FILTER>
inline bool launch_surface_generic(const SurfaceParams& p, int num_cu, hipStream_t st) {
    const size_t lds = generic_lds_bytes(p.nK, p.nT);
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) { set_max_lds(reinterpret_cast<const void*>(surface_generic_kernel<FILTER>)); attr = true; }
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
    int64_t grid = (int64_t)num_cu * per_cu * 2;
    if (grid > p.B) grid = p.B;
    hipLaunchKernelGGL(surface_generic_kernel<FILTER>, dim3((unsigned)grid), dim3(64), lds, st, p);
    return true;
}

// Dense dispatch.  Returns 1 if dispatched (dense kernel + filtered generic redo pass), 0 if the
// shape is not covered by a dense kernel.
inline int launch_surface_dense(const SurfaceParams& p, int num_cu, hipStream_t st, const char** name) {
    if (p.k_off || p.nK != DK || p.nT != DT) return 0;
    if (p.k_stride != 0 && p.k_stride < DK) return 0;
    if (reinterpret_cast<uintptr_t>(p.sigma) & 15) return 0;
    const size_t lds = dense_lds_bytes(p.mT);
    if (lds > 64 * 1024) return 0;
    if (generic_lds_bytes(p.nK, p.nT) > 160 * 1024) return 0;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 8 ? 8 : per_cu;
    int64_t grid = (int64_t)num_cu * per_cu;
    if (grid > p.B) grid = p.B;
#define IVS_DENSE_CASE(M, NAME)                                                                            \
    case M: {                                                                                              \
        static bool attr = false;                                                                          \
        if (!attr) { set_max_lds(reinterpret_cast<const void*>(surface_dense_kernel<M>)); attr = true; }   \
        hipLaunchKernelGGL(surface_dense_kernel<M>, dim3((unsigned)grid), dim3(64), lds, st, p);           \
        *name = NAME;                                                                                      \
        break;                                                                                             \
    }
    switch (p.method) {
        IVS_DENSE_CASE(IVS_LINEAR, "surface_dense_kernel<linear>")
        IVS_DENSE_CASE(IVS_CUBIC, "surface_dense_kernel<cubic>")
        IVS_DENSE_CASE(IVS_CUBICSPLINE, "surface_dense_kernel<cubicspline>")
        IVS_DENSE_CASE(IVS_SLINEAR, "surface_dense_kernel<slinear>")
        default: return 0;
    }
#undef IVS_DENSE_CASE
    if (hipGetLastError() != hipSuccess) return -1;
    launch_surface_generic<true>(p, num_cu, st);     // redo pass for tagged surfaces (cheap when none are)
    return 1;
}

}  // namespace ivs
