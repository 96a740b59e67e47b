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
#include <atomic>
#include <cstddef>
#include <cstdlib>

#include "ivs_surface_generic.hpp"

namespace ivs {

// IVS_ABLATE (tools/ablate_api.hip only, never the product build): what a phase costs in THROUGHPUT, measured by leaving
// it out -- results are wrong by construction.  1 no strike slopes, 2 strike evaluation = one LDS read per row, 3 no
// maturity solve, 4 one output row in 8 stored, 5 no prefetch after the first surface, 6 = 1+2+3 (staging + stores only).
#ifndef IVS_ABLATE
#define IVS_ABLATE 0
#endif
constexpr int ABL = IVS_ABLATE;

constexpr int DK = 64, DT = 16;
constexpr int D_RS = 72;                         // row stride in doubles (4 x (16 + 2))
constexpr int D_MAX_MT = 64;                     // query maturities per surface handled by the dense kernel
constexpr unsigned long long D_SENTINEL = REDO_SENTINEL;           // NaN payload = "redo generically"

__device__ __forceinline__ int d_sl(int k) { return (k >> 4) * 18 + (k & 15); }

constexpr int D_WLDS_MAX_MT = 16;                // query-row weights live in LDS up to this many rows
__host__ __device__ inline size_t dense_lds_bytes(int mT) {
    // Y, S planes (the strike tables AL..PSI and the T-phase scratch alias them); Ksh, RDX; TT [16][4];
    // W [16][4] when mT <= 16 (exactly 20480 B = 8 workgroups per CU), else weights stay in registers
    return (size_t)(2 * DT * D_RS + 64 + 64 + 64 + (mT <= D_WLDS_MAX_MT ? 4 * D_WLDS_MAX_MT : 0)) * 8;
}

// method families of the dense kernels
__host__ __device__ constexpr bool d_is_nak(int m) { return m == IVS_CUBIC || m == IVS_CUBICSPLINE; }      // tridiagonal slopes
__host__ __device__ constexpr bool d_is_local(int m) { return m == IVS_PCHIP || m == IVS_AKIMA; }          // 3-/5-point slopes
__host__ __device__ constexpr bool d_is_hermite(int m) { return d_is_nak(m) || d_is_local(m); }
__host__ __device__ constexpr bool d_extrap_right(int m) { return m == IVS_CUBICSPLINE || m == IVS_PCHIP; }
// 'quadratic' (row-pass kernels only): B-spline coefficients from a tridiagonal collocation system, evaluation = 4 weights
// over the coefficients c[j-1 .. j+2] of the data interval j (one of the outer weights is 0)
__host__ __device__ constexpr bool d_is_quad(int m) { return m == IVS_QUADRATIC; }
// two-knot rules without slopes besides np.interp: nearest / zero / from_derivatives (row-pass kernels only)
__host__ __device__ constexpr bool d_is_step(int m) { return m == IVS_NEAREST || m == IVS_ZERO || m == IVS_FROM_DERIVATIVES; }

// codes of a query row in the maturity direction
constexpr int TQ_LEFT = -1, TQ_HOLD = 15, TQ_NAN = 16;

// ---- cross-lane helpers (DPP moves stay in the VALU; ds_bpermute shuffles cost an LDS round trip)
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// old = 0.0 with full row / bank masks: bound_ctrl supplies the zero for lanes without a source, which saves the two
// v_mov that would otherwise seed the destination with `old` before every pair of DPP moves
template <int CTRL>
__device__ __forceinline__ double dpp0_f64(double src) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHL(int n) { return 0x100 | n; }   // lane i <- lane i+n (within a row of 16)
constexpr int DPP_ROW_SHR(int n) { return 0x110 | n; }   // lane i <- lane i-n (within a row of 16)
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// exponent of |x| inside [2^-500, 2^500]: the range where the division expansion applies no scaling
__device__ __forceinline__ bool div_safe(double x) {
    const double ax = __builtin_fabs(x);
    return ax >= 0x1p-500 && ax <= 0x1p500;
}
// Correctly rounded a/b from y = refined_rcp(b): q = a*y, r = a - q*b, q' = q + r*y.  This IS the tail of
// the hardware IEEE expansion (v_div_scale is the identity in the safe range), so results are bit-identical
// to `a / b` there; outside it (or for non-finite operands) fall back to the full division.
__device__ __forceinline__ double div_shared_rcp(double a, double b, double y, bool b_safe) {
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    double res = __builtin_fma(r, y, q);
    if (__builtin_expect(!(b_safe && (a == 0.0 || div_safe(a))), 0)) res = a / b;
    return res;
}
// np.interp's interior formula with the shared-reciprocal division (bit-exact, see lerp_np)
__device__ __forceinline__ double lerp_np_rcp(double xq, double x0, double y0, double x1, double y1, double dx,
                                              double rdx, bool dx_safe) {
#pragma clang fp contract(off)
    if (x0 == xq) return y0;
    const double slope = div_shared_rcp(y1 - y0, dx, rdx, dx_safe);
    double r = slope * (xq - x0) + y0;
    if (__builtin_isnan(r)) {
        r = slope * (xq - x1) + y1;
        if (__builtin_isnan(r) && y0 == y1) r = y0;
    }
    return r;
}

// Branch-free fast path of the same formula: `slow` is raised whenever the value must be recomputed by
// lerp_np() (operand outside the safe division range, or a NaN that needs np.interp's fallback rules).
__device__ __forceinline__ double lerp_fast(double xq, double x0, double y0, double y1, double dx, double rdx,
                                            bool& slow) {
#pragma clang fp contract(off)
    const double a = y1 - y0;
    const double q = a * rdx;
    const double rem = __builtin_fma(-q, dx, a);
    const double slope = __builtin_fma(rem, rdx, q);
    const double res = slope * (xq - x0) + y0;
    const double aa = __builtin_fabs(a);
    slow |= !((a == 0.0) || (aa >= 0x1p-500 && aa <= 0x1p500)) || __builtin_isnan(res);
    return (x0 == xq) ? y0 : res;
}

// nearest / zero / from_derivatives on the interval [x0, x1] that holds xq (x0 <= xq < x1, or xq == x1 on the last interval):
// the arithmetic of eval_nearest / eval_zero / eval_bpoly_linear (ivs_device.hpp) with the interval already known
template <int METHOD>
__device__ __forceinline__ double step_eval(double xq, double x0, double x1, double y0, double y1) {
#pragma clang fp contract(off)
    if (METHOD == IVS_NEAREST) return (x0 / 2.0 + x1 / 2.0 >= xq) ? y0 : y1;      // ties to the left knot
    if (METHOD == IVS_ZERO) return (xq == x1) ? y1 : y0;                          // left knot's value; the last knot its own
    const double s = (xq - x0) / (x1 - x0);
    const double a = y0 * (1.0 - s), b = y1 * s;
    return a + b;
}

// ---- 'quadratic' on the fast kernels (scipy make_interp_spline(k = 2); knots and basis: quad_knot / quad_basis, ivs_device.hpp)
// collocation row of site i on n sites X: lo c_{i-1} + di c_i + up c_{i+1} = y_i (rows 0 and n-1: c_i = y_i)
template <class XA>
__device__ __forceinline__ void quad_row(const XA& x, int n, int i, double& lo, double& di, double& up) {
    lo = 0.0; di = 1.0; up = 0.0;
    if (i > 0 && i < n - 1) quad_basis(x, n, i + 1, x(i), lo, di, up);
}
// evaluation at xq in the data interval jj (= min(find_interval, n - 2); the caller handles outside-the-hull): weights of
// c[jj-1], c[jj], c[jj+1], c[jj+2] -- eval_quadratic's window [q, q+2] is [jj-1, jj+1] or [jj, jj+2]
template <class XA>
__device__ __forceinline__ void quad_weights(const XA& x, int n, int j, double xq, double& a0, double& a1, double& a2, double& a3) {
#pragma clang fp contract(off)
    const int jj = j > n - 2 ? n - 2 : j;
    int q = n - 3;
    if (j < n - 1) { q = j - 1 + ((xq >= (x(j + 1) + x(j)) / 2.0) ? 1 : 0); q = q < 0 ? 0 : (q > n - 3 ? n - 3 : q); }
    double h0, h1, h2;
    quad_basis(x, n, q + 2, xq, h0, h1, h2);
    const bool lowwin = q < jj;                          // window starts at jj - 1
    a0 = lowwin ? h0 : 0.0; a1 = lowwin ? h1 : h0; a2 = lowwin ? h2 : h1; a3 = lowwin ? 0.0 : h2;
}

// Inclusive scan over lanes 0..N-1 (N = 16 or 64) of 2x2 matrix products P_i <- P_i * P_{i-1} * ... * P_0.
template <int N>
__device__ __forceinline__ void scan_mat2(double& p00, double& p01, double& p10, double& p11, int lane) {
    // lanes without a valid source (row start, rows masked off) receive the IDENTITY through the DPP
    // `old` operand, so every lane multiplies unconditionally: no compare/select per step
#define IVS_SCAN_STEP(CTRL, MASK)                                                              \
    {                                                                                          \
        const double e = dpp_f64<CTRL, MASK>(1.0, p00), h = dpp_f64<CTRL, MASK>(1.0, p11);     \
        const double f = MASK == 0xF ? dpp0_f64<CTRL>(p01) : dpp_f64<CTRL, MASK>(0.0, p01);    \
        const double g = MASK == 0xF ? dpp0_f64<CTRL>(p10) : dpp_f64<CTRL, MASK>(0.0, p10);    \
        const double n00 = p00 * e + p01 * g, n01 = p00 * f + p01 * h;                         \
        const double n10 = p10 * e + p11 * g, n11 = p10 * f + p11 * h;                         \
        p00 = n00; p01 = n01; p10 = n10; p11 = n11;                                            \
    }
    (void)lane;
    IVS_SCAN_STEP(DPP_ROW_SHR(1), 0xF)
    IVS_SCAN_STEP(DPP_ROW_SHR(2), 0xF)
    IVS_SCAN_STEP(DPP_ROW_SHR(4), 0xF)
    IVS_SCAN_STEP(DPP_ROW_SHR(8), 0xF)
    if (N > 16) {
        IVS_SCAN_STEP(DPP_ROW_BCAST15, 0xA)     // rows 1,3 <- lane 15 of rows 0,2
        IVS_SCAN_STEP(DPP_ROW_BCAST31, 0xC)     // rows 2,3 <- lane 31
    }
#undef IVS_SCAN_STEP
}

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
    const double rdxc = refined_rcp(dxc);
    double a, b, c;
    if (i == 0) { a = 0.0; b = dxp; c = dxc + dxp; }
    else if (i == N - 1) { a = dxmm + dxm; b = dxmm; c = 0.0; }
    else { a = dxc; b = 2.0 * (dxm + dxc); c = dxm; }
    const double rb = refined_rcp(b);
    // g_i = a_i c_{i-1} / (b_i b_{i-1})
    const double crb = c * rb;                                          // c_i / b_i
    const double crb_prev = dpp0_f64<DPP_WAVE_SHR1>(crb);
    const double g = i == 0 ? 0.0 : a * rb * crb_prev;
    // omega_i = 1 - g_i / omega_{i-1}: prefix product of M_i = [[1,-g_i],[1,0]] (identity on lane 0)
    double p00 = 1.0, p01 = i == 0 ? 0.0 : -g, p10 = i == 0 ? 0.0 : 1.0, p11 = i == 0 ? 1.0 : 0.0;
    scan_mat2<N>(p00, p01, p10, p11, lane);
    const double num = p00 + p01, den = p10 + p11;          // omega_i = num / den (applied to omega_0 = 1)
    const double rw = i == 0 ? rb : den * rb * refined_rcp(num);   // 1 / w_i
    al = a * rw;
    cp = c * rw;
    const double rdx_prev = dpp0_f64<DPP_WAVE_SHR1>(rdxc);   // 1/dx[i-1]
    const double rdx_next = dpp0_f64<DPP_WAVE_SHL1>(rdxc);   // 1/dx[i+1]
    // (DPP reads need the SOURCE lane active: keep every cross-lane move outside divergent branches)
    const double rdxmm = dpp0_f64<DPP_WAVE_SHR1>(rdx_prev);  // 1/dx[i-2]
    // edge rows need 1/d with d = x_{i+2}-x_i (row 0) or x_i - x_{i-2} (row N-1)
    const double d = i == 0 ? dxc + dxp : dxmm + dxm;
    const double rd = refined_rcp(d);
    if (i == 0) {
        pp = (dxc + 2.0 * d) * dxp * rdxc * rd * rw;         // * dy_0
        qq = dxc * dxc * rdx_next * rd * rw;                 // * dy_1
    } else if (i == N - 1) {
        pp = dxm * dxm * rdxmm * rd * rw;                    // * dy_{N-3}
        qq = (2.0 * d + dxm) * dxmm * rdx_prev * rd * rw;    // * dy_{N-2}
    } else {
        pp = 3.0 * dxc * rdx_prev * rw;                      // * dy_{i-1}
        qq = 3.0 * dxm * rdxc * rw;                          // * dy_i
    }
    rdx_out = rdxc;
}

// segmented inclusive prefix / suffix products within aligned rows of 16 lanes
__device__ __forceinline__ double seg16_prefix_prod(double v, int lane) {
    (void)lane;                                   // out-of-row sources read as 1.0 (DPP `old`)
    v *= dpp_f64<DPP_ROW_SHR(1)>(1.0, v);
    v *= dpp_f64<DPP_ROW_SHR(2)>(1.0, v);
    v *= dpp_f64<DPP_ROW_SHR(4)>(1.0, v);
    v *= dpp_f64<DPP_ROW_SHR(8)>(1.0, v);
    return v;
}
__device__ __forceinline__ double seg16_suffix_prod(double v, int lane) {
    (void)lane;
    v *= dpp_f64<DPP_ROW_SHL(1)>(1.0, v);
    v *= dpp_f64<DPP_ROW_SHL(2)>(1.0, v);
    v *= dpp_f64<DPP_ROW_SHL(4)>(1.0, v);
    v *= dpp_f64<DPP_ROW_SHL(8)>(1.0, v);
    return v;
}

// ---- local-slope methods (pchip, akima): a knot's slope depends on the 2 (pchip) or 4 (akima) neighbouring
// secants only, so there is no system to solve: uniform per-knot tables + per-lane arithmetic.
// Tables on N knots X[0..N), entry i:  r0 = 1/dx_i (0 for i = N-1);  pchip: (r1, r2) = (w1, w2) =
// (2 dx_i + dx_{i-1}, dx_i + 2 dx_{i-1}) for interior knots, and for the two end knots the coefficients of the
// one-sided three-point formula d = r1*m_near - r2*m_far (scipy _cubic.py _edge_case: ((2h0+h1) m0 - h0 m1)/(h0+h1)).
template <int N>
__device__ __forceinline__ void local_tables(const double* X, int i, double& r0, double& r1, double& r2) {
    const double x0 = X[i];
    const double xp = X[i + 1 < N ? i + 1 : N - 1], xm = X[i > 0 ? i - 1 : 0];
    const double dxc = xp - x0, dxm = x0 - xm;
    r0 = i < N - 1 ? refined_rcp(dxc) : 0.0;
    const bool first = i == 0, last = i == N - 1;
    const double h0 = first ? dxc : dxm;
    const double h1 = first ? X[2] - X[1] : X[N - 2] - X[N - 3];
    const double rs = refined_rcp(h0 + h1);
    r1 = (first || last) ? (2.0 * h0 + h1) * rs : 2.0 * dxc + dxm;
    r2 = (first || last) ? h0 * rs : dxc + 2.0 * dxm;
}
// interior knot: weighted harmonic mean of the two secants, 1/((w1/mp + w2/mk)/(w1+w2)) in one division
__device__ __forceinline__ double pchip_knot(double mp, double mk, double w1, double w2) {
    const double d = (w1 + w2) * mp * mk * refined_rcp(w1 * mk + w2 * mp);
    return (mp * mk > 0.0) ? d : 0.0;
}
__device__ __forceinline__ double pchip_edge(double m0, double m1, double ea, double eb) {
    const double d = ea * m0 - eb * m1;
    const bool clip = !(m0 * m1 > 0.0) && __builtin_fabs(d) > 3.0 * __builtin_fabs(m0);
    const double r = clip ? 3.0 * m0 : d;
    return (d * m0 > 0.0) ? r : 0.0;                       // branch-free on purpose (selects)
}
// akima (scipy _cubic.py:510-541): secants ma..md = m_{i-2}, m_{i-1}, m_i, m_{i+1}
#ifndef IVS_AKIMA_ROLL
#define IVS_AKIMA_ROLL 1      // akima's maturity slopes formed while the output rows are walked (dense_maturity_pass)
#endif
__device__ __forceinline__ double akima_f12(double ma, double mb, double mc, double md) {
    return __builtin_fabs(md - mc) + __builtin_fabs(mb - ma);
}
__device__ __forceinline__ double akima_knot(double ma, double mb, double mc, double md, double thr) {
    const double f1 = __builtin_fabs(md - mc), f2 = __builtin_fabs(mb - ma), f12 = f1 + f2;
    const double t = (f1 * mb + f2 * mc) * refined_rcp(f12);
    return f12 > thr ? t : 0.5 * (md + ma);
}
constexpr int DPP_QUAD_SWAP1 = 0xB1, DPP_QUAD_SWAP2 = 0x4E;    // quad_perm [1,0,3,2] / [2,3,0,1]

// Strike-direction slopes for the local methods, rs-lane layout as in dense_strike_slopes: reads Y and Ksh,
// writes RDX and (last) the S plane; the three tables alias the S plane until then.  All 64 lanes must call it.
template <int METHOD>
__device__ __forceinline__ void dense_strike_slopes_local(const double* Y, double* S, const double* Ksh, double* RDX,
                                                          int lane) {
    constexpr bool AK = METHOD == IVS_AKIMA;
    double* R0 = S;
    double* R1 = S + 80;                // one spare slot behind each table: R0[kp + 18] of segment 3 stays inside
    double* R2 = S + 160;
    {
        double r0, r1, r2;
        local_tables<DK>(Ksh, lane, r0, r1, r2);
        const int kl = d_sl(lane);
        R0[kl] = r0; R1[kl] = r1; R2[kl] = r2; RDX[lane] = r0;
    }
    __syncthreads();
    const int rs_t = lane >> 2, rs_seg = lane & 3;
    const double* yr = Y + rs_t * D_RS;
    const int kp = rs_seg * 18;
    const bool s_first = rs_seg == 0, s_last = rs_seg == 3;
    double y[20];                                   // y[m+2] = y_{kb+m}, m = -2..17
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(yr + kp + 2 * c);
        y[2 + 2 * c] = v.x; y[3 + 2 * c] = v.y;
    }
    y[1] = yr[s_first ? 0 : kp - 3];
    y[18] = yr[s_last ? 0 : kp + 18];
    if (AK) { y[0] = yr[s_first ? 0 : kp - 4]; y[19] = yr[s_last ? 0 : kp + 19]; }
    // secants m_{kb-2+i}, i = 0..18: intervals kb-2 .. kb+16 (pchip needs i = 1..17 only)
    auto sec = [&](int i) {
        const int ti = i == 0 ? kp - 4 : (i == 1 ? kp - 3 : (i == 18 ? kp + 18 : kp + i - 2));
        return (y[i + 1] - y[i]) * R0[(s_first && i < 2) ? 0 : ti];
    };
    double d[16];
    if (!AK) {
        double mp = sec(1), mk = sec(2);            // rolling pair (m_{k-1}, m_k): keeps the live set small
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const double mn = m < 15 ? sec(m + 3) : 0.0;           // m_{k+1}
            const double w1 = R1[kp + m], w2 = R2[kp + m];
            double v = pchip_knot(mp, mk, w1, w2);
            if (m == 0) { const double e = pchip_edge(mk, mn, w1, w2); v = s_first ? e : v; }
            if (m == 15) { const double e = pchip_edge(mp, sec(15), w1, w2); v = s_last ? e : v; }
            d[m] = v;
            mp = mk; mk = mn;
            if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // bound the live range of the division chains
        }
    } else {
        // secants beyond the ends by linear extrapolation of the secant sequence; two rolling passes (the row's
        // largest |second difference| first, then the slopes) instead of a 19-entry secant array
        const double l1 = 2.0 * sec(2) - sec(3), l0 = 2.0 * l1 - sec(2);
        const double r17 = 2.0 * sec(16) - sec(15), r18 = 2.0 * r17 - sec(16);
        auto secx = [&](int i) {
            double v = sec(i);
            if (i == 0) v = s_first ? l0 : v;
            if (i == 1) v = s_first ? l1 : v;
            if (i == 17) v = s_last ? r17 : v;
            if (i == 18) v = s_last ? r18 : v;
            return v;
        };
        double fmax = 0.0;
        {
            double ma = secx(0), mb = secx(1), mc = secx(2);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const double md = secx(m + 3);
                fmax = __builtin_fmax(fmax, akima_f12(ma, mb, mc, md));
                ma = mb; mb = mc; mc = md;
            }
        }
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP1>(fmax, fmax));      // the row's 4 segments = one quad
        fmax = __builtin_fmax(fmax, dpp_f64<DPP_QUAD_SWAP2>(fmax, fmax));
        const double thr = 1e-9 * fmax;
        double ma = secx(0), mb = secx(1), mc = secx(2);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const double md = secx(m + 3);
            d[m] = akima_knot(ma, mb, mc, md, thr);
            ma = mb; mb = mc; mc = md;
            if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();                               // all table reads done: the S plane may be overwritten
    double* srow = S + rs_t * D_RS + kp;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double2 v; v.x = d[2 * c]; v.y = d[2 * c + 1];
        *reinterpret_cast<double2*>(srow + 2 * c) = v;
    }
}

// The same slopes in the maturity direction: 16 knots per lane in registers, tables TT[i] = {r0, r1, r2, -}.
template <int METHOD, class TP>
__device__ __forceinline__ void dense_maturity_slopes_local(const double (&z)[DT], TP TT, double (&s)[DT]) {
    constexpr bool AK = METHOD == IVS_AKIMA;
    auto sec = [&](int i) { return (z[i + 1] - z[i]) * TT[i * 4]; };       // m_i, i = 0..14
    if (!AK) {
        double mp = sec(0), mk = sec(1);
        s[0] = pchip_edge(mp, mk, TT[1], TT[2]);
#pragma unroll
        for (int i = 1; i < DT - 1; ++i) {          // (mp, mk) = (m_{i-1}, m_i)
            s[i] = pchip_knot(mp, mk, TT[i * 4 + 1], TT[i * 4 + 2]);
            if (i < DT - 2) { mp = mk; mk = sec(i + 1); }
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        s[DT - 1] = pchip_edge(mk, mp, TT[(DT - 1) * 4 + 1], TT[(DT - 1) * 4 + 2]);
    } else {
        const double l1 = 2.0 * sec(0) - sec(1), l0 = 2.0 * l1 - sec(0);                       // m_{-1}, m_{-2}
        const double r15 = 2.0 * sec(DT - 2) - sec(DT - 3), r16 = 2.0 * r15 - sec(DT - 2);     // m_15, m_16
        auto secx = [&](int i) {                    // m_i, i = -2..16
            return i == -2 ? l0 : (i == -1 ? l1 : (i == DT - 1 ? r15 : (i == DT ? r16 : sec(i < 0 ? 0 : (i > DT - 2 ? DT - 2 : i)))));
        };
        double fmax = 0.0;
        {
            double ma = secx(-2), mb = secx(-1), mc = secx(0);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const double md = secx(i + 1);
                fmax = __builtin_fmax(fmax, akima_f12(ma, mb, mc, md));
                ma = mb; mb = mc; mc = md;
            }
        }
        const double thr = 1e-9 * fmax;
        double ma = secx(-2), mb = secx(-1), mc = secx(0);
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const double md = secx(i + 1);
            s[i] = akima_knot(ma, mb, mc, md, thr);
            ma = mb; mb = mc; mc = md;
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Run-time maturity count nT (4..16): tables are all-zero beyond nT, so secants beyond the last knot come out 0 and are
// replaced by akima's linear extension F(i) = 2 F(i-1) - F(i-2); pchip's one-sided rule moves to knot nT - 1.
template <int METHOD, class TP>
__device__ __forceinline__ void dense_maturity_slopes_local_rt(const double (&z)[DT], TP TT, double (&s)[DT], int nT) {
    constexpr bool AK = METHOD == IVS_AKIMA;
    double F[DT + 3];                               // F[i] = m_{i-2}
#pragma unroll
    for (int i = 0; i < DT - 1; ++i) F[i + 2] = (z[i + 1] - z[i]) * TT[i * 4];
    F[DT + 1] = 0.0; F[DT + 2] = 0.0; F[0] = 0.0; F[1] = 0.0;
    if (!AK) {
        s[0] = pchip_edge(F[2], F[3], TT[1], TT[2]);
#pragma unroll
        for (int i = 1; i < DT; ++i) {
            const double w1 = TT[i * 4 + 1], w2 = TT[i * 4 + 2];
            if (i == nT - 1) s[i] = pchip_edge(F[i + 1], F[i], w1, w2);      // wave-uniform branch: one of the two is computed
            else s[i] = pchip_knot(F[i + 1], F[i + 2], w1, w2);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        F[1] = 2.0 * F[2] - F[3]; F[0] = 2.0 * F[1] - F[2];
#pragma unroll
        for (int i = 4; i < DT + 3; ++i) {
            const int idx = i - 2;
            F[i] = (idx == nT - 1 || idx == nT) ? 2.0 * F[i - 1] - F[i - 2] : F[i];
        }
        double fmax = 0.0;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            const double f = akima_f12(F[i], F[i + 1], F[i + 2], F[i + 3]);
            fmax = (i < nT) ? __builtin_fmax(fmax, f) : fmax;
        }
        const double thr = 1e-9 * fmax;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            s[i] = akima_knot(F[i], F[i + 1], F[i + 2], F[i + 3], thr);
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Strike-direction slopes of all 16 rows of the staged surface: K-phase (k-lane factorisation) followed by
// the segmented sweeps (rs-lane).  Reads Y and Ksh, writes RDX and (last) the S plane; the factor tables
// live in the S plane until then.  Must be called by all 64 lanes of the workgroup.
template <bool STAMPED, class StampFn>
__device__ __forceinline__ void dense_strike_slopes(const double* Y, double* S, const double* Ksh, double* RDX,
                                                    int lane, StampFn&& stamp) {
    // strike tables alias the S plane (dead before S is written); knot k sits at d_sl(k) so that the four
    // segment-broadcast reads of a sweep step hit four different bank groups
    double* AL = S;
    double* CP = S + 72;
    double* PP = S + 144;
    double* QQ = S + 216;
    double* PI = S + 288;
    double* PSI = S + 360;
    const int rs_t = lane >> 2, rs_seg = lane & 3;
    // ---- K-phase (k-lane)
    double al, cp, pp, qq, rdx;
    factor_tables<DK>(Ksh, lane, al, cp, pp, qq, rdx);
    const int kl = d_sl(lane);
    AL[kl] = al; CP[kl] = cp; PP[kl] = pp; QQ[kl] = qq; RDX[lane] = rdx;
    PI[kl] = seg16_prefix_prod(-al, lane);         // prod_{seg start..i} (-AL)
    PSI[kl] = seg16_suffix_prod(-cp, lane);        // prod_{i..seg end} (-CP)
    __syncthreads();
    if (STAMPED) stamp(1);
    // ---- strike sweeps (rs-lane): row rs_t, knots rs_seg*16 .. +15
    const double* yrow = Y + rs_t * D_RS + rs_seg * 18;
    const int kb = rs_seg * 16, kp = rs_seg * 18;   // first knot of the segment / its padded table index
    double y[18];                                  // y[m+1] = y_{kb+m}, m = -1..16
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double2 v = *reinterpret_cast<const double2*>(yrow + 2 * c);
        y[1 + 2 * c] = v.x; y[2 + 2 * c] = v.y;
    }
    y[0] = rs_seg > 0 ? Y[rs_t * D_RS + d_sl(kb - 1)] : 0.0;
    y[17] = rs_seg < 3 ? Y[rs_t * D_RS + d_sl(kb + 16)] : 0.0;
    // dy(m) = y_{kb+m} - y_{kb+m-1} = y[m+1] - y[m]
    double d[16];
    double prev = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        double dA = y[m + 1] - y[m], dB = y[m + 2] - y[m + 1];         // (dy_{i-1}, dy_i)
        if (m == 0) { const double e = y[3] - y[2]; dA = rs_seg == 0 ? dB : dA; dB = rs_seg == 0 ? e : dB; }
        if (m == 15) { const double e = y[15] - y[14]; dB = rs_seg == 3 ? dA : dB; dA = rs_seg == 3 ? e : dA; }
        const double r = PP[kp + m] * dA + QQ[kp + m] * dB;
        prev = r - AL[kp + m] * prev;
        d[m] = prev;
        if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // carry across the 4 segments of the row (lanes 4t..4t+3 sit in one DPP row)
    const double pie = PI[kp + 15];
    double din = 0.0, tot = d[15];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const double v = dpp0_f64<DPP_ROW_SHR(1)>(tot);
        if (rs_seg == j) { din = v; tot = d[15] + pie * din; }
    }
    double nxt = 0.0;
#pragma unroll
    for (int m = 15; m >= 0; --m) {
        const double dp = d[m] + PI[kp + m] * din;
        nxt = dp - CP[kp + m] * nxt;
        d[m] = nxt;                                // local backward solution
        if ((m & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    const double psb = PSI[kp];
    double sin_ = 0.0;
    tot = d[0];
#pragma unroll
    for (int j = 2; j >= 0; --j) {
        const double v = dpp0_f64<DPP_ROW_SHL(1)>(tot);
        if (rs_seg == j) { sin_ = v; tot = d[0] + psb * sin_; }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) d[m] = d[m] + PSI[kp + m] * sin_;
    __syncthreads();                               // all table reads done: the S plane may be overwritten
    double* srow = S + rs_t * D_RS + rs_seg * 18;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        double2 v; v.x = d[2 * c]; v.y = d[2 * c + 1];
        *reinterpret_cast<double2*>(srow + 2 * c) = v;
    }
}

// STAMP = true is the diagnostic build (ivs_debug_stamps): s_memtime deltas per phase are summed per
// workgroup into a debug buffer; never used for timing claims (the stamps serialise the phases).
constexpr int D_NSTAMP = 8;   // stage, k-phase, sweeps, strike-eval, maturity-solve, maturity-eval+store, (unused), surfaces

// What the T-phase leaves in registers: per-query-row weights in lane = tq, row counts per class (uniform).
struct TqTables {
    double w0, w1, w2, w3;
    double pm_last;                       // run-time nT only: third tap of the last row of the maturity system
    int n_left, n_hold, n_nan, unsorted;
    unsigned long long iv_lo, iv_hi;      // rows per interval 0..7 / 8..14, one byte each
    __device__ __forceinline__ int n_iv(int j) const { return (int)(((j < 8 ? iv_lo : iv_hi) >> (8 * (j & 7))) & 0xffull); }
};

// Batch-wide maturity tables (T and Tq shared by every surface): written once per call by tq_tables_kernel into the
// caller's workspace and read by the surface kernels through the SCALAR cache (constant address space => s_load):
// uniform data costs neither LDS cycles (a broadcast ds_read_b128 occupies the LDS for 4 cycles) nor VGPRs.
struct TqShared {
    double TT[DT * 4];                    // per maturity knot: {PP, QQ, AL, CP} | {r0, r1, r2, -} | {T_j, T_j+1, 1/dt, dt}
    double W[D_MAX_MT * 4];               // per query row: Hermite weights | {Tq, T_j, T_j+1, 1/dt}
    double CP[DT];                        // the not-a-knot tables once more as separate arrays: the register solve requests
    double PP[DT], QQ[DT], AL[DT];        // eight rows of a coefficient with ONE s_load_dwordx16 (three requests in flight, one
                                          // wait per half) instead of waiting for a scalar load in every row
    double pm_last;
    unsigned long long iv_lo, iv_hi;
    int n_left, n_hold, n_nan, unsorted;
    int redo[2];                          // surfaces tagged by the fast kernel / left tagged by the compaction kernel (SurfaceParams::redo)
    int mode;                             // 1 = "missing quotes first" (SurfaceParams::mode): the batch goes to the compaction kernel directly
    int pad_i;
    unsigned long long pad_[6];
    unsigned long long queue[24 * QUEUE_STRIDE];      // work-queue heads of the row-pass kernels (SurfaceParams::queue)
};
static_assert(offsetof(TqShared, W) == DT * 4 * 8 && offsetof(TqShared, CP) == (DT * 4 + D_MAX_MT * 4) * 8 &&
              offsetof(TqShared, AL) == (DT * 4 + D_MAX_MT * 4 + 3 * DT) * 8, "TT, W, CP, PP, QQ, AL back to back");
#define IVS_CONST __attribute__((address_space(4)))
typedef const double IVS_CONST* cdptr;
__device__ __forceinline__ cdptr to_const(const double* p) { return (cdptr)p; }

template <int NKB>
__device__ __forceinline__ void factor_tables_var(const double* X, int n, int lane, double* AL, double* CP, double* PP,
                                                  double* QQ, double* PM, double* PI, double* PSI, double* RDX);
__device__ __forceinline__ void local_tables_rt(const double* X, int n, int i, double& r0, double& r1, double& r2);

// T-phase: maturity-direction factor tables (LDS, TT) + per-query-row weights (registers, lane = tq; also W in LDS
// when WLDS) + row counts per class.  Needs mT <= 64.  All 64 lanes must call it; ends with a barrier.
// NTR: the maturity count is a run-time value nT_rt in [4, 16] (knots beyond it are +inf, the last system row sits at
// nT_rt - 1 and uses the three-tap right-hand side of the variable-shape kernels); `scratch` = 8 x 72 doubles of LDS.
template <int METHOD, bool WLDS, bool NTR = false>
__device__ __forceinline__ void dense_t_phase(const double* Tb, const double* Tqb, int mT, int lane, double* Tsh,
                                              double* TT, double* W, TqTables& tt, int nT_rt = DT,
                                              double* scratch = nullptr) {
    const int nT = NTR ? nT_rt : DT;
    constexpr bool CUB = d_is_hermite(METHOD);
    constexpr bool w_lds = WLDS;
    if (lane < DT) Tsh[lane] = (!NTR || lane < nT) ? Tb[lane] : __builtin_inf();
    __syncthreads();
    tt.pm_last = 0.0;
    if (d_is_nak(METHOD) && !NTR) {
        double al, cp, pp, qq, rdx;
        factor_tables<DT>(Tsh, lane, al, cp, pp, qq, rdx);
        if (lane < DT) { TT[lane * 4 + 0] = pp; TT[lane * 4 + 1] = qq; TT[lane * 4 + 2] = al; TT[lane * 4 + 3] = cp; }
    }
    if (d_is_nak(METHOD) && NTR) {
        double* SC = scratch;            // AL, CP, PP, QQ, PM, PI, PSI (72 each, d_sl(i) = i for i < 16) + RDX
        factor_tables_var<1>(Tsh, nT, lane, SC, SC + 72, SC + 144, SC + 216, SC + 288, SC + 360, SC + 432, SC + 504);
        __syncthreads();
        if (lane < DT) {
            const bool in = lane < nT;
            TT[lane * 4 + 0] = in ? SC[144 + lane] : 0.0; TT[lane * 4 + 1] = in ? SC[216 + lane] : 0.0;
            TT[lane * 4 + 2] = in ? SC[lane] : 0.0; TT[lane * 4 + 3] = in ? SC[72 + lane] : 0.0;
        }
        tt.pm_last = SC[288 + nT - 1];
        __syncthreads();
    }
    if (d_is_quad(METHOD)) {       // once per call (tq_tables_kernel): lane 0 eliminates the 16-knot collocation system serially
        if (lane == 0) {
            const CView tx{Tsh, 1};
            double cprev = 0.0;
            for (int i = 0; i < DT; ++i) {
                double lo = 0.0, di = 1.0, up = 0.0;
                if (i < nT) quad_row(tx, nT, i, lo, di, up);
                const double rw = 1.0 / (di - lo * cprev);
                cprev = up * rw;
                TT[i * 4 + 0] = i < nT ? rw : 0.0; TT[i * 4 + 1] = 0.0; TT[i * 4 + 2] = i < nT ? lo * rw : 0.0; TT[i * 4 + 3] = i < nT ? cprev : 0.0;
            }
        }
        __syncthreads();
    }
    if (!d_is_hermite(METHOD) && !d_is_quad(METHOD) && lane < DT) {          // lerp methods: {T_j, T_j+1, 1/dt, dt} per interval
        const bool in = lane < nT - 1;
        const double t0 = in ? Tsh[lane] : 0.0, t1 = in ? Tsh[lane + 1] : 1.0;
        TT[lane * 4 + 0] = t0; TT[lane * 4 + 1] = t1; TT[lane * 4 + 2] = refined_rcp(t1 - t0); TT[lane * 4 + 3] = t1 - t0;
    }
    if (d_is_local(METHOD) && lane < DT) {
        double r0, r1, r2;
        if (NTR) local_tables_rt(Tsh, nT, lane, r0, r1, r2);
        else local_tables<DT>(Tsh, lane, r0, r1, r2);
        TT[lane * 4 + 0] = r0; TT[lane * 4 + 1] = r1; TT[lane * 4 + 2] = r2; TT[lane * 4 + 3] = 0.0;
    }
    const int tq = lane;
    const bool act = tq < mT;
    const double x = act ? Tqb[tq] : __builtin_inf();
    int j = -1;                                           // largest j with Tsh[j] <= x, or -1
    if (Tsh[0] <= x) {
        j = 0;
#pragma unroll
        for (int st = 8; st >= 1; st >>= 1) if (Tsh[j + st] <= x) j += st;
    }
    int code;
    const double tl = Tsh[nT - 1];
    if (j < 0) code = TQ_LEFT;
    else if (j >= nT - 1) {
        if (METHOD == IVS_LINEAR) code = TQ_HOLD;
        else if (METHOD == IVS_SLINEAR) code = (x == tl) ? TQ_HOLD : TQ_NAN;
        else if (!d_extrap_right(METHOD)) code = (x == tl) ? nT - 2 : TQ_NAN;
        else code = nT - 2;
    } else code = j;
    const int jj = code >= 0 && code <= nT - 2 ? code : 0;
    const double x0 = Tsh[jj], x1 = Tsh[jj + 1];
    if (d_is_quad(METHOD)) {
        const CView tx{Tsh, 1};
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (code >= 0 && code <= nT - 2) quad_weights(tx, nT, (x == tl) ? nT - 1 : code, x, a0, a1, a2, a3);
        tt.w0 = a0; tt.w1 = a1; tt.w2 = a2; tt.w3 = a3;
    } else if (CUB) {
        const double h = x1 - x0, u = x - x0, t = u / h, omt = 1.0 - t;
        tt.w0 = (1.0 + 2.0 * t) * omt * omt;              // h00
        tt.w1 = t * t * (3.0 - 2.0 * t);                  // h01
        tt.w2 = u * omt * omt;                            // h * h10
        tt.w3 = u * t * (t - 1.0);                        // h * h11
    } else {
        tt.w0 = x; tt.w1 = x0; tt.w2 = x1; tt.w3 = refined_rcp(x1 - x0);
    }
    if (w_lds && act) {
        *reinterpret_cast<double2*>(W + tq * 4) = double2{tt.w0, tt.w1};
        *reinterpret_cast<double2*>(W + tq * 4 + 2) = double2{tt.w2, tt.w3};
    }
    tt.n_left = __popcll(__ballot(act && code == TQ_LEFT));
    tt.n_hold = __popcll(__ballot(act && code == TQ_HOLD));
    tt.n_nan = __popcll(__ballot(act && code == TQ_NAN));
    tt.iv_lo = 0; tt.iv_hi = 0;
#pragma unroll
    for (int jv = 0; jv < DT - 1; ++jv) {
        const unsigned long long n = (unsigned long long)__popcll(__ballot(act && code == jv));
        if (jv < 8) tt.iv_lo |= n << (8 * jv); else tt.iv_hi |= n << (8 * (jv - 8));
    }
    // the row loops below walk the classes in order: needs ascending, NaN-free Tq
    const double xprev = dpp_f64<DPP_WAVE_SHR1>(-__builtin_inf(), x);
    tt.unsorted = __ballot(act && ((x < xprev) || !(x == x))) != 0ull;
    __syncthreads();
}

// Maturity direction for one block of 64 output strikes (q-lane): z[t] = strike-pass value of row t at the lane's
// strike.  Solves the lane's 16-knot system in registers (cubic) and walks the output rows class by class
// (left-NaN rows, rows per maturity interval, hold rows, right-NaN rows), storing 512-B rows through a
// wave-uniform base pointer.
// RANGED: only the output rows [row_lo, row_hi) are produced (two-wavefront kernels split the rows).
// NTR: run-time maturity count nT_rt (see dense_t_phase): masked system rows, three-tap last row, hold row = nT_rt - 1.
// SM: TT and W point at the TqShared tables in global memory; every read of them is a scalar load (uniform index),
// the next row's weights are requested while the current row is combined and stored.
// AKROLL: akima's slopes formed while the rows are walked (see there); false where the registers are not the limit.
template <int METHOD, bool WLDS, bool RANGED = false, bool NTR = false, bool SM = false, bool AKROLL = true, class StampFn>
__device__ __forceinline__ void dense_maturity_pass(const double (&z)[DT], const TqTables& tt, const double* TT,
                                                    const double* W, double* outb, int q0, int lane, bool act, int mT,
                                                    int mK, StampFn&& stamp, int row_lo = 0, int row_hi = 0,
                                                    int nT_rt = DT) {
    const int nT = NTR ? nT_rt : DT;
    constexpr bool QUAD = d_is_quad(METHOD);
    constexpr bool CUB = d_is_hermite(METHOD) || QUAD;
    static_assert(!d_is_step(METHOD) || WLDS, "nearest / zero / from_derivatives: per-row weights only");
    static_assert(!QUAD || SM, "quadratic: batch-wide maturity tables only");
    constexpr bool w_lds = WLDS && !SM;
    const double nanv = __builtin_nan("");
    const cdptr cTT = to_const(TT), cW = to_const(W);          // SM only
    const cdptr cCP = to_const(TT + DT * 4 + D_MAX_MT * 4);    // TqShared::CP, PP, QQ, AL (laid out back to back behind TT, W)
    const cdptr cPP = cCP + DT, cQQ = cCP + 2 * DT, cAL = cCP + 3 * DT;
    // ---- maturity direction (q-lane, registers).  Row pointers are wave-uniform (scalar base),
    // the lane contributes only its 32-bit column offset.
    // the packed per-interval row counts stay packed: made opaque here, their extraction (one s_bfe per interval) cannot be
    // hoisted out of the caller's surface loop as 15 separate scalars (which were spilled to VGPR lanes and read back)
    unsigned long long ivl = tt.iv_lo, ivh = tt.iv_hi;
    asm volatile("" : "+s"(ivl), "+s"(ivh));
    auto n_iv = [&](int j) { return (int)(((j < 8 ? ivl : ivh) >> (8 * (j & 7))) & 0xffull); };
    int tq = 0;
    double* rp = outb + q0;                                    // row tq of the output block: uniform running pointer
    cdptr wp = cW;                                             // SM: weights of row tq
#ifdef IVS_PUT_PAIR
    // A/B variant (verdict r02 item 8): 16 bytes per lane -- rows are stored in PAIRS, lane pair (2i, 2i+1)
    // swaps one value through a quad_perm DPP move, the even lane writes strikes (2i, 2i+1) of the even row, the odd lane
    // those of the odd row: half the store instructions, 1 KB per instruction.  Needs mK even and a 16-byte aligned block.
    const bool pair_ok = !RANGED && (mK & 1) == 0 && ((reinterpret_cast<uintptr_t>(outb + q0) & 15) == 0);      // wave-uniform
    const int pair_off = (lane & 1) ? lane - 1 : lane - mK;
    double pend = 0.0;
#endif
    auto put = [&](int row, double v) {                        // row == tq at every call site
        (void)row;
        if (ABL == 4 && (row & 7) != 0) { asm volatile("" :: "v"(v)); return; }
#ifdef IVS_PUT_PAIR
        if (pair_ok) {
            if (!(tq & 1)) {
                pend = v;
                if (tq == mT - 1 && act) rp[lane] = v;         // an odd row count leaves the last row unpaired
            } else {
                const double send = (lane & 1) ? pend : v;
                const double recv = dpp_f64<0xB1>(send, send);       // quad_perm [1, 0, 3, 2]
                double2 o; o.x = (lane & 1) ? recv : pend; o.y = (lane & 1) ? v : recv;
                if (act) *reinterpret_cast<double2*>(rp + pair_off) = o;
            }
            return;
        }
#endif
        if (act) rp[lane] = v;
    };
    auto adv = [&]() { ++tq; rp += mK; wp += 4; };
    auto mine = [&](int row) { return !RANGED || (row >= row_lo && row < row_hi); };      // wave-uniform
    for (int c = 0; c < tt.n_left; ++c, adv()) if (mine(tq)) put(tq, nanv);
    // weights of row tq: LDS broadcast (prefetched one row ahead) when mT <= 16, else readlane
    const double2* W2 = reinterpret_cast<const double2*>(W);
    double2 wa_n = double2{0.0, 0.0}, wb_n = double2{0.0, 0.0};
    if (w_lds) {
        int t0 = tq < mT ? tq : 0;
        if (RANGED && t0 < row_lo) t0 = row_lo < mT ? row_lo : 0;       // first row this wave will ask for
        wa_n = W2[2 * t0]; wb_n = W2[2 * t0 + 1];
    }
    // SM: the next row's weights are requested one row ahead (the table has spare rows behind row mT - 1: TqShared::CP
    // follows W, so the request past the last row needs no clamp).  A wave that skips rows (RANGED) reads at the use.
    double wn0 = 0.0, wn1 = 0.0, wn2 = 0.0, wn3 = 0.0;
    if (SM && !RANGED) { wn0 = wp[0]; wn1 = wp[1]; wn2 = wp[2]; wn3 = wp[3]; }
    auto weights = [&](int row, double& a0, double& a1, double& a2, double& a3) {
        (void)row;
        if (SM && RANGED) {
            a0 = wp[0]; a1 = wp[1]; a2 = wp[2]; a3 = wp[3];
        } else if (SM) {
            a0 = wn0; a1 = wn1; a2 = wn2; a3 = wn3;
            // scalar loads return out of order (only lgkmcnt(0) is meaningful): take this row's values out of the
            // counter BEFORE the next row's request is issued, so that the request flies during the row's arithmetic
            asm volatile("" : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3));
            wn0 = wp[4]; wn1 = wp[5]; wn2 = wp[6]; wn3 = wp[7];
        } else if (w_lds) {
            a0 = wa_n.x; a1 = wa_n.y; a2 = wb_n.x; a3 = wb_n.y;
            const int nx = row + 1 < mT ? row + 1 : row;      // prefetch the next row's weights
            wa_n = W2[2 * nx]; wb_n = W2[2 * nx + 1];
        } else {
            a0 = readlane_f64(tt.w0, row); a1 = readlane_f64(tt.w1, row);
            a2 = readlane_f64(tt.w2, row); a3 = readlane_f64(tt.w3, row);
        }
    };
    if (METHOD == IVS_AKIMA && IVS_AKIMA_ROLL && AKROLL) {
        // akima, slopes formed WHILE the rows are walked: knot j needs the secants m_{j-2} .. m_{j+1} only, so a window of
        // four secants slides along the 16 knots twice -- once for the row maximum of |m_{j+1} - m_j| + |m_{j-1} - m_{j-2}|
        // (the threshold of the equal-secants rule), once in the row loop, where an interval needs just its two end slopes.
        // Live: z[16], the window, two slopes -- not s[16] and the 19 secants a common-subexpression pass keeps between the
        // two loops of dense_maturity_slopes_local (240 VGPRs on the run-time-shape kernels, 2 wavefronts per SIMD).  Same
        // expressions in the same order: bit-identical.
        double zz[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i) zz[i] = z[i];
        auto secz = [&](int i) -> double {                     // m_i, i = 0..14 (run-time nT: the table is 0 from interval nT-1 on)
            const double r0 = SM ? (double)cTT[i * 4] : TT[i * 4];
            return (zz[i + 1] - zz[i]) * r0;
        };
        auto m_next = [&](int idx, double mb_, double mc_) -> double {      // m_idx, idx >= 2, from the two before it
            const double raw = idx <= DT - 2 ? secz(idx < DT - 1 ? idx : DT - 2) : 0.0;
            return (idx == nT - 1 || idx == nT) ? 2.0 * mc_ - mb_ : raw;     // linear extension behind the last knot
        };
        double thr;
        {
            const double m0 = secz(0), m1 = secz(1);
            const double l1 = 2.0 * m0 - m1, l0 = 2.0 * l1 - m0;
            double ma = l0, mb = l1, mc = m0, md = m1, fmax = 0.0;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const double f = akima_f12(ma, mb, mc, md);
                fmax = (!NTR || i < nT) ? __builtin_fmax(fmax, f) : fmax;
                const double nd = m_next(i + 2, mc, md);
                ma = mb; mb = mc; mc = md; md = nd;
            }
            thr = 1e-9 * fmax;
        }
#pragma unroll
        for (int i = 0; i < DT; ++i) asm volatile("" : "+v"(zz[i]));      // the second walk recomputes its secants
        stamp(4);
        const double m0 = secz(0), m1 = secz(1);
        const double l1 = 2.0 * m0 - m1, l0 = 2.0 * l1 - m0;
        double ma = l0, mb = l1, mc = m0, md = m1;
        double s_lo = akima_knot(ma, mb, mc, md, thr);
        { const double nd = m_next(2, mc, md); ma = mb; mb = mc; mc = md; md = nd; }
#pragma unroll
        for (int jv = 0; jv < DT - 1; ++jv) {
            const double s_hi = akima_knot(ma, mb, mc, md, thr);          // knot jv + 1
            { const double nd = m_next(jv + 3, mc, md); ma = mb; mb = mc; mc = md; md = nd; }
            for (int c = 0, n = n_iv(jv); c < n; ++c, adv()) {
                if (!mine(tq)) continue;
                double a0, a1, a2, a3;
                weights(tq, a0, a1, a2, a3);
                put(tq, a0 * zz[jv] + a1 * zz[jv + 1] + a2 * s_lo + a3 * s_hi);
            }
            s_lo = s_hi;
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (CUB) {
        double s[DT];
        double prev = 0.0;
        const double pm_last = NTR ? readlane_f64(tt.pm_last, 0) : 0.0;      // uniform -> SGPRs
        if (d_is_local(METHOD)) {
            if (SM) {
                if (NTR) dense_maturity_slopes_local_rt<METHOD>(z, cTT, s, nT);
                else dense_maturity_slopes_local<METHOD>(z, cTT, s);
            } else {
                if (NTR) dense_maturity_slopes_local_rt<METHOD>(z, TT, s, nT);
                else dense_maturity_slopes_local<METHOD>(z, TT, s);
            }
            // pin the slopes here: otherwise they (and, transitively, the strike evaluation with its 64 gathered
            // operands) are sunk into the row loops below and the live set overflows the register file
#pragma unroll
            for (int i = 0; i < DT; ++i) asm volatile("" : "+v"(s[i]));
        } else if (ABL == 3 || ABL == 6) {
#pragma unroll
            for (int i = 0; i < DT; ++i) s[i] = z[i] * 0.5;
        } else if (SM) {   // forward / backward sweep with the table rows in SGPRs: SMG rows per request group
            constexpr int SMG = 4;
#pragma unroll
            for (int h = 0; h < DT / SMG; ++h) {
                double pp[SMG], qq[SMG], al[SMG];
#pragma unroll
                for (int u = 0; u < SMG; ++u) { pp[u] = cPP[SMG * h + u]; qq[u] = cQQ[SMG * h + u]; al[u] = cAL[SMG * h + u]; }
#pragma unroll
                for (int u = 0; u < SMG; ++u) {
                    const int i = SMG * h + u;
                    if (QUAD) {      // collocation system: right-hand side = the value itself (PP = 1 / pivot; all-zero rows beyond nT)
                        prev = pp[u] * z[i] - al[u] * prev;
                    } else if (!NTR) {
                        const int ia = i == 0 ? 0 : (i == DT - 1 ? DT - 3 : i - 1), ib = i == 0 ? 1 : (i == DT - 1 ? DT - 2 : i);
                        const double dA = z[ia + 1] - z[ia], dB = z[ib + 1] - z[ib];
                        prev = (pp[u] * dA + qq[u] * dB) - al[u] * prev;
                    } else {
                        const int ia = i == 0 ? 0 : i - 1, ib = i == 0 ? 1 : (i == DT - 1 ? DT - 2 : i);
                        const double dA = z[ia + 1] - z[ia], dB = z[ib + 1] - z[ib];
                        double r = pp[u] * dA + qq[u] * dB;
                        if (i >= 3) r += ((i == nT - 1) ? pm_last : 0.0) * (z[i - 1] - z[i - 2]);     // scalar select
                        prev = r - al[u] * prev;
                    }
                    s[i] = prev;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h = 1; h >= 0; --h) {
                double cp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) cp[u] = cCP[8 * h + u];
#pragma unroll
                for (int u = 7; u >= 0; --u) { const int i = 8 * h + u; if (i <= DT - 2) s[i] = s[i] - cp[u] * s[i + 1]; }   // CP = 0 from the last row on
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {   // forward sweep; table rows (broadcast b128 pairs) are loaded 4 steps ahead
            constexpr int LA = 4;
            double2 tpq[4], tac[4];
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                tpq[i] = *reinterpret_cast<const double2*>(TT + i * 4);
                tac[i] = *reinterpret_cast<const double2*>(TT + i * 4 + 2);
            }
            double cpv[DT];                                // CP_i, kept for the backward sweep
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const double2 pq = tpq[i & 3], ac = tac[i & 3];
                __builtin_amdgcn_sched_barrier(0);
                if (i + LA < DT) {
                    tpq[i & 3] = *reinterpret_cast<const double2*>(TT + (i + LA) * 4);
                    tac[i & 3] = *reinterpret_cast<const double2*>(TT + (i + LA) * 4 + 2);
                }
                if (!NTR) {
                    // (dz_{i-1}, dz_i); (dz_0, dz_1) for row 0; (dz_{13}, dz_{14}) for row 15, dz_i = z[i+1] - z[i]
                    const int ia = i == 0 ? 0 : (i == DT - 1 ? DT - 3 : i - 1), ib = i == 0 ? 1 : (i == DT - 1 ? DT - 2 : i);
                    const double dA = z[ia + 1] - z[ia], dB = z[ib + 1] - z[ib];
                    prev = (pq.x * dA + pq.y * dB) - ac.x * prev;
                } else {
                    // three-tap form: PM*dz_{i-2} + PP*dz_{i-1} + QQ*dz_i (PM only on the last row, where QQ = 0).  Rows
                    // beyond nT have all-zero table entries, so they produce 0 without any masking.
                    const int ia = i == 0 ? 0 : i - 1, ib = i == 0 ? 1 : (i == DT - 1 ? DT - 2 : i);
                    const double dA = z[ia + 1] - z[ia], dB = z[ib + 1] - z[ib];
                    double r = pq.x * dA + pq.y * dB;
                    if (i >= 3) r += ((i == nT - 1) ? pm_last : 0.0) * (z[i - 1] - z[i - 2]);     // scalar select
                    prev = r - ac.x * prev;
                }
                s[i] = prev;
                cpv[i] = ac.y;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = DT - 2; i >= 0; --i) s[i] = s[i] - cpv[i] * s[i + 1];      // CP = 0 from the last row on
        }
        stamp(4);
#pragma unroll
        for (int jv = 0; jv < DT - 1; ++jv) {
            for (int c = 0, n = n_iv(jv); c < n; ++c, adv()) {
                if (!mine(tq)) continue;
                double a0, a1, a2, a3;
                weights(tq, a0, a1, a2, a3);
                if (QUAD) put(tq, a0 * s[jv > 0 ? jv - 1 : 0] + a1 * s[jv] + a2 * s[jv + 1] + a3 * s[jv + 2 < DT ? jv + 2 : DT - 1]);
                else put(tq, a0 * z[jv] + a1 * z[jv + 1] + a2 * s[jv] + a3 * s[jv + 1]);
            }
        }
    } else {
#pragma unroll
        for (int jv = 0; jv < DT - 1; ++jv) {
            if (!WLDS) {
#pragma clang fp contract(off)
                // many query rows per interval (mT > 16): np.interp's slope of the interval once per lane, two
                // operations per output row (same arithmetic as lerp_fast, hence the same bits)
                const int n = n_iv(jv);
                if (n == 0) continue;
                double2 tj, tr;                                                             // {T_j, T_j+1}, {1/dt, dt}
                if (SM) { tj = double2{cTT[jv * 4], cTT[jv * 4 + 1]}; tr = double2{cTT[jv * 4 + 2], cTT[jv * 4 + 3]}; }
                else { tj = *reinterpret_cast<const double2*>(TT + jv * 4); tr = *reinterpret_cast<const double2*>(TT + jv * 4 + 2); }
                const double a = z[jv + 1] - z[jv];
                const double q = a * tr.x;
                const double rem = __builtin_fma(-q, tr.y, a);
                const double slope = __builtin_fma(rem, tr.x, q);
                const double aa = __builtin_fabs(a);
                const bool slow_iv = !div_safe(tr.y) || !((a == 0.0) || (aa >= 0x1p-500 && aa <= 0x1p500));
                for (int c = 0; c < n; ++c, adv()) {
                    if (!mine(tq)) continue;
                    const double xt = SM ? wp[0] : readlane_f64(tt.w0, tq);
                    const double res = slope * (xt - tj.x) + z[jv];
                    const bool slow = slow_iv || __builtin_isnan(res);
                    double r = (tj.x == xt) ? z[jv] : res;
                    if (__builtin_expect(__ballot(slow) != 0ull, 0)) {
                        if (slow) r = lerp_np(xt, tj.x, z[jv], tj.y, z[jv + 1]);
                    }
                    put(tq, r);
                }
                continue;
            }
            for (int c = 0, n = n_iv(jv); c < n; ++c, adv()) {
                if (!mine(tq)) continue;
                double xt, t0, t1, rdt;                      // {Tq, T_j, T_j+1, 1/(T_j+1 - T_j)}
                weights(tq, xt, t0, t1, rdt);
                if (d_is_step(METHOD)) { put(tq, step_eval<METHOD>(xt, t0, t1, z[jv], z[jv + 1])); continue; }
                const double dt = t1 - t0;
                bool slow = !div_safe(dt);
                double r = lerp_fast(xt, t0, z[jv], z[jv + 1], dt, rdt, slow);
                if (__builtin_expect(__ballot(slow) != 0ull, 0)) {
                    if (slow) r = lerp_np(xt, t0, z[jv], t1, z[jv + 1]);
                }
                put(tq, r);
            }
        }
    }
    double z_last = z[DT - 1];
    if (NTR) {
#pragma unroll
        for (int i = 3; i < DT - 1; ++i) {
            z_last = (i == nT - 1) ? z[i] : z_last;      // wave-uniform selects
            asm volatile("" : "+v"(z_last));             // keeps LLVM from turning the chain into z[nT - 1] (= scratch array)
        }
    }
    for (int c = 0; c < tt.n_hold; ++c, adv()) if (mine(tq)) put(tq, z_last);
    for (int c = 0; c < tt.n_nan; ++c, adv()) if (mine(tq)) put(tq, nanv);
}

// One wavefront per call: the T-phase of a batch whose T and Tq are shared, published as TqShared (see there).
template <int METHOD, bool NTR>
__global__ __launch_bounds__(64) void tq_tables_kernel(SurfaceParams p, TqShared* o) {
    __shared__ __attribute__((aligned(16))) double sm[8 * 72 + 64 + 64];
    const int lane = threadIdx.x;
    double* scratch = sm;                  // NTR: factor_tables_var scratch
    double* Tsh = sm + 8 * 72;
    double* TT = Tsh + 64;
    TqTables tt;
    // the missing-quote probe's 64 row requests (below) go out FIRST: they fly while the T-phase runs
    const bool probe = !NTR && !p.k_off && p.nK == DK && p.mK <= 64 && p.B >= 4096;
    double q[64];
    if (probe) {
#pragma unroll
        for (int s = 0; s < 64; ++s) q[s] = p.sigma[(int64_t)s * (p.B / 64) * (int64_t)(DT * DK) + (s & (DT - 1)) * DK + lane];
    }
    dense_t_phase<METHOD, false, NTR>(p.T, p.Tq, p.mT, lane, Tsh, TT, nullptr, tt, p.nT, scratch);
    if (lane < DT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) o->TT[lane * 4 + c] = TT[lane * 4 + c];
        o->CP[lane] = TT[lane * 4 + 3]; o->PP[lane] = TT[lane * 4]; o->QQ[lane] = TT[lane * 4 + 1]; o->AL[lane] = TT[lane * 4 + 2];
    }
    if (lane < p.mT) { o->W[lane * 4] = tt.w0; o->W[lane * 4 + 1] = tt.w1; o->W[lane * 4 + 2] = tt.w2; o->W[lane * 4 + 3] = tt.w3; }
    // Missing-quote probe (64 x 16 batches only).  A surface with a missing quote is tagged by the fast kernel after it has
    // streamed its quotes, and the compaction kernel reads them again: with a feed in which most snapshots lack some quote
    // that first pass is pure traffic (1.5 ms per 1 M surfaces).  One row (row s mod 16) of 64 surfaces spread over the batch is
    // sampled here, 4096 quotes; the call runs "missing quotes first" -- the fast kernel returns at once, the compaction
    // kernel takes EVERY surface -- when
    //   * at least half of the sampled rows lack a quote, or
    //   * at least 5 of them do and they lack about one quote each (total <= 1.5 per such row): quotes missing here and
    //     there, independently -- then 5 of 64 rows means >= 0.12 % of the quotes and, 1024 quotes per surface, >= 72 % of
    //     the surfaces: the break-even of the two orders (3.0 + 5.4 a ms against 6.9 ms for a share a of tagged surfaces;
    //     profiles/r03/nan/sparse_probe_rule.txt).  Several quotes missing per sampled row says the gaps cluster (illiquid strikes
    //     of SOME snapshots): the share of surfaces hit is then about the share of rows hit, and tagging stays cheaper.
    int mode = 0;
    if (probe) {
        int hit = 0, tot = 0;
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const int c = __popcll(__ballot(__builtin_isnan(q[s])));
            hit += c != 0 ? 1 : 0; tot += c;
        }
        mode = (hit >= 32 || (hit >= 5 && 2 * tot <= 3 * hit)) ? 1 : 0;
    }
    if (lane == 0) {
        o->pm_last = tt.pm_last; o->iv_lo = tt.iv_lo; o->iv_hi = tt.iv_hi;
        o->n_left = tt.n_left; o->n_hold = tt.n_hold; o->n_nan = tt.n_nan; o->unsorted = tt.unsorted;
        o->redo[0] = 0; o->redo[1] = 0; o->mode = mode;
    }
    if (lane < 24) o->queue[lane * QUEUE_STRIDE] = 0ull;
}
template <bool NTR>
inline void launch_tq_tables(const SurfaceParams& p, TqShared* o, hipStream_t st) {
#define IVS_TQ_CASE(M) case M: hipLaunchKernelGGL((tq_tables_kernel<M, NTR>), dim3(1), dim3(64), 0, st, p, o); break;
    switch (p.method) {
        IVS_TQ_CASE(IVS_LINEAR) IVS_TQ_CASE(IVS_CUBIC) IVS_TQ_CASE(IVS_CUBICSPLINE) IVS_TQ_CASE(IVS_SLINEAR)
        IVS_TQ_CASE(IVS_PCHIP) IVS_TQ_CASE(IVS_AKIMA) IVS_TQ_CASE(IVS_NEAREST) IVS_TQ_CASE(IVS_ZERO) IVS_TQ_CASE(IVS_FROM_DERIVATIVES)
        IVS_TQ_CASE(IVS_QUADRATIC)
        default: break;
    }
#undef IVS_TQ_CASE
}
// fills the uniform part of TqTables from the published tables (scalar loads)
__device__ __forceinline__ void tq_from_shared(const void* tqs, TqTables& tt, const double*& TT, const double*& W) {
    const TqShared* g = static_cast<const TqShared*>(tqs);
    const TqShared IVS_CONST* c = (const TqShared IVS_CONST*)g;
    tt.w0 = tt.w1 = tt.w2 = tt.w3 = 0.0;
    tt.pm_last = c->pm_last; tt.iv_lo = c->iv_lo; tt.iv_hi = c->iv_hi;
    tt.n_left = c->n_left; tt.n_hold = c->n_hold; tt.n_nan = c->n_nan; tt.unsorted = c->unsorted;
    TT = g->TT; W = g->W;
}

template <int METHOD, bool TSHARED, bool WLDS, bool STAMP = false>
__global__ __launch_bounds__(64, 2) void surface_dense_kernel(SurfaceParams p, unsigned long long* dbg = nullptr) {
    if (TSHARED && p.mode && *((const int IVS_CONST*)p.mode)) return;      // "missing quotes first" (tq_tables_kernel)
    constexpr bool CUB = d_is_hermite(METHOD);
    unsigned long long acc[D_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
    auto stamp = [&](int i) {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tn = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (i >= 0) acc[i] += tn - tprev;
            tprev = tn;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int mT = p.mT, mK = p.mK;
    double* Y = reinterpret_cast<double*>(smem);
    double* S = Y + DT * D_RS;
    double* Ksh = S + DT * D_RS;
    double* RDX = Ksh + 64;
    double* TT = RDX + 64;             // [16][4] = {PP, QQ, AL, CP} of the maturity system
    double* W = TT + 64;               // [16][4] query-row weights (only when mT <= 16)
    double* Tsh = TSHARED ? Y : S;     // T-phase scratch: the plane that is free at the time it runs

    constexpr bool t_shared = TSHARED;   // T and Tq shared by the whole batch: T-phase hoisted out of the loop
    const double nanv = __builtin_nan("");

    TqTables tt;
    auto t_phase = [&](const double* Tb, const double* Tqb) { dense_t_phase<METHOD, WLDS>(Tb, Tqb, mT, lane, Tsh, TT, W, tt); };
    const double* TTp = TT;            // tables of the maturity pass: LDS, or the published TqShared (scalar cache)
    const double* Wp = W;
    if (t_shared) tq_from_shared(p.tqs, tt, TTp, Wp);

    // ---- prefetch registers (k-lane layout: chunk i of the surface = 16 B at i*1024 + lane*16)
    double2 pre[8];
    double pre_k;
    auto prefetch = [&](int64_t b) {
        const double2* s2 = reinterpret_cast<const double2*>(p.sigma + b * (int64_t)(DT * DK));
#pragma unroll
        for (int i = 0; i < 8; ++i) pre[i] = s2[i * 64 + lane];
        pre_k = p.K[b * p.k_stride + lane];
    };
    // query strikes of the first 4 column blocks live in registers: loaded once when the grid is shared,
    // else per surface BEFORE the prefetch is issued (vmcnt retires in order: a load issued behind the
    // prefetch would make the evaluation wait for the whole next surface)
    constexpr int XQ_REG = 4;
    double xq_reg[XQ_REG];
    auto load_xq = [&](const double* Kqb) {
#pragma unroll
        for (int i = 0; i < XQ_REG; ++i) xq_reg[i] = (i * 64 + lane < mK) ? Kqb[i * 64 + lane] : nanv;
    };
    const bool kq_shared = p.kq_stride == 0;
    if (kq_shared) load_xq(p.Kq);
    // Surface -> workgroup mapping: the workgroups form R groups (wg % R), group r sweeps the r-th region of the batch
    // with its G/R workgroups interleaved.  With R = 1 all 2048 resident workgroups advance through ONE window of the
    // batch; several distant windows at once stream 4-5 % faster (tools/map_probe.hip: 5.60 -> 5.90 TB/s for this
    // kernel's access pattern, the same effect that separates a persistent grid-stride copy from a wide-grid one).
    // Since round 2 the surfaces of a region are CLAIMED from the group's work queue instead of being dealt out with a
    // fixed stride (WorkQueue, ivs_surface_generic.hpp).
    bool told = false;
    WorkQueue wq;
    wq.init(p.queue, p.map_groups, p.B, WQ_CHUNK, lane);
    int64_t b = wq.take(), b_next = b >= 0 ? wq.take() : -1;
    if (b >= 0) prefetch(b);

    while (b >= 0) {
        const bool more = b_next >= 0;
        const WorkQueue::Pending pend = wq.begin(more);
        auto advance = [&]() { const int64_t n2 = wq.finish(pend, more); b = b_next; b_next = n2; };
        __syncthreads();                                   // everyone is done reading the previous surface's LDS
        stamp(-1);
        // ---- stage quotes: chunk i, lane -> row t = 2i + (lane>>5), k = 2*(lane&31)
        unsigned long long bad = 0ull;                     // wave-level NaN mask, accumulated in scalar registers
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = 2 * i + (lane >> 5), k = 2 * (lane & 31);
            *reinterpret_cast<double2*>(&Y[t * D_RS + d_sl(k)]) = pre[i];
            bad |= __ballot(pre[i].x != pre[i].x) | __ballot(pre[i].y != pre[i].y);
        }
        Ksh[lane] = pre_k;
        double* outb = p.out + b * (int64_t)mT * mK;
        if (!t_shared) t_phase(p.T + b * p.t_stride, p.Tq + b * p.tq_stride);   // contains a barrier
        if (bad != 0ull || tt.unsorted) {                  // wave-uniform: leave it to the generic kernel
            if (lane == 0) { reinterpret_cast<unsigned long long*>(outb)[0] = D_SENTINEL; count_redo(p, told); }
            prefetch(more ? b_next : b);
            advance();
            continue;
        }
        __syncthreads();
        stamp(0);
        const double* Kqb = p.Kq + b * p.kq_stride;
        if (!kq_shared) load_xq(Kqb);
        {   // next surface's loads fly during the whole computation; past the end the last surface is re-read
            // (harmless) so that the prefetch registers are written on every path
            if (ABL != 5) prefetch(more ? b_next : b);
        }
        if (d_is_nak(METHOD) && (ABL == 1 || ABL == 6)) {
            RDX[lane] = 1.0;
            __syncthreads();
        } else if (d_is_nak(METHOD)) {
            dense_strike_slopes<STAMP>(Y, S, Ksh, RDX, lane, stamp);
            __syncthreads();
            stamp(2);
        } else if (d_is_local(METHOD)) {
            dense_strike_slopes_local<METHOD>(Y, S, Ksh, RDX, lane);
            __syncthreads();
            stamp(2);
        }
#pragma unroll 1
        for (int q0 = 0, qb = 0; q0 < mK; q0 += 64, ++qb) {
            const int q = q0 + lane;
            const bool act = q < mK;
            double xq;
            if (qb < XQ_REG) xq = qb == 0 ? xq_reg[0] : (qb == 1 ? xq_reg[1] : (qb == 2 ? xq_reg[2] : xq_reg[3]));
            else xq = act ? Kqb[q] : nanv;
            // ---- strike evaluation (q-lane): j = largest index with K[j] <= xq
            // level 1: seven independent broadcast reads of the pivots K[8m]; level 2: three dependent gathers
            int j = 0;
#pragma unroll
            for (int m = 1; m < 8; ++m) j += (Ksh[8 * m] <= xq) ? 8 : 0;
#pragma unroll
            for (int st = 4; st >= 1; st >>= 1) if (Ksh[j + st] <= xq) j += st;
            const double xl = Ksh[DK - 1];
            const bool left = !(Ksh[0] <= xq);
            const int jj = j > DK - 2 ? DK - 2 : j;
            const double x0 = Ksh[jj], x1 = Ksh[jj + 1];
            const int o0 = d_sl(jj), o1 = d_sl(jj + 1);
            double z[DT];
            if (ABL == 2 || ABL == 6) {
#pragma unroll
                for (int r = 0; r < DT; ++r) z[r] = Y[r * D_RS + d_sl(lane)] + xq;
            } else if (CUB) {
                const bool ok = !left && ((xq <= xl) || d_extrap_right(METHOD));
                const double u = xq - x0, t = u * RDX[jj], omt = 1.0 - t;
                const double w0 = ok ? (1.0 + 2.0 * t) * omt * omt : nanv;
                const double w1 = t * t * (3.0 - 2.0 * t);
                const double w2 = u * omt * omt;
                const double w3 = u * t * (t - 1.0);
                // software pipeline: the four gathers of row r+3 are issued before row r is combined,
                // so ~12 LDS reads stay in flight (lgkmcnt holds 15) instead of waiting row by row
                constexpr int LA = 3;
                double g0[4], g1[4], g2[4], g3[4];
#pragma unroll
                for (int r = 0; r < LA; ++r) {
                    g0[r] = Y[r * D_RS + o0]; g1[r] = Y[r * D_RS + o1]; g2[r] = S[r * D_RS + o0]; g3[r] = S[r * D_RS + o1];
                }
#pragma unroll
                for (int r = 0; r < DT; ++r) {
                    if (r + LA < DT) {
                        const int n = r + LA;
                        g0[n & 3] = Y[n * D_RS + o0]; g1[n & 3] = Y[n * D_RS + o1];
                        g2[n & 3] = S[n * D_RS + o0]; g3[n & 3] = S[n * D_RS + o1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    z[r] = w0 * g0[r & 3] + w1 * g1[r & 3] + w2 * g2[r & 3] + w3 * g3[r & 3];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const bool right = j >= DK - 1;
                const bool hold = right && (METHOD == IVS_LINEAR || xq == xl);
                const double dx = x1 - x0, rdx = refined_rcp(dx);
                bool slow = !div_safe(dx);
                constexpr int LA = 6;                     // 12 gathers in flight
                double g0[8], g1[8];
#pragma unroll
                for (int r = 0; r < LA; ++r) { g0[r] = Y[r * D_RS + o0]; g1[r] = Y[r * D_RS + o1]; }
#pragma unroll
                for (int r = 0; r < DT; ++r) {
                    if (r + LA < DT) { const int n = r + LA; g0[n & 7] = Y[n * D_RS + o0]; g1[n & 7] = Y[n * D_RS + o1]; }
                    __builtin_amdgcn_sched_barrier(0);
                    const double y0 = g0[r & 7], y1 = g1[r & 7];
                    double v = lerp_fast(xq, x0, y0, y1, dx, rdx, slow);
                    if (right) v = hold ? y1 : nanv;      // jj = 62 -> y1 is the last quote
                    if (left) v = nanv;
                    z[r] = v;
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (__builtin_expect(__ballot(slow && !right && !left) != 0ull, 0)) {   // rare: full IEEE path
                    if (slow && !right && !left) {
#pragma unroll
                        for (int r = 0; r < DT; ++r) z[r] = lerp_np(xq, x0, Y[r * D_RS + o0], x1, Y[r * D_RS + o1]);
                    }
                }
            }
            stamp(3);
            if (d_is_local(METHOD)) {
#pragma unroll
                for (int r = 0; r < DT; ++r) asm volatile("" : "+v"(z[r]));      // see dense_maturity_pass
            }
            if (act) dense_maturity_pass<METHOD, WLDS, false, false, TSHARED>(z, tt, TTp, Wp, outb, q0, lane, true, mT, mK, stamp);
            stamp(5);
        }
        if (p.status && lane == 0) p.status[b] = IVS_ST_OK;
        if (STAMP) acc[7] += 1;
        advance();
    }
    if (STAMP && dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < D_NSTAMP; ++i) dbg[(size_t)blockIdx.x * D_NSTAMP + i] = acc[i];
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
// Everything a launcher needs from the entry point: the CURRENT device (index, CU count), the caller's stream and
// workspace, and the tuning override of the surface -> workgroup mapping (flags bits 8..15).  No process-global state:
// the kernel attribute that unlocks > 64 KiB of dynamic LDS is per device and is tracked per device below.
struct LaunchCtx {
    int dev = 0, num_cu = 256;
    hipStream_t st = nullptr;
    unsigned char* ws = nullptr;
    size_t ws_bytes = 0;
    int map_groups = 0;
};
constexpr int IVS_MAX_DEV = 64;
constexpr size_t WS_TQ_BYTES = 8192;                 // TqShared at offset 0
constexpr size_t WS_COUNTS_BYTES = 256;              // ragged: per-class counters
constexpr int V_NCLASS = 4;                          // ragged: work lists (one per size class), B items each
static_assert(sizeof(TqShared) <= WS_TQ_BYTES, "TqShared must fit its workspace slot");
inline size_t surface_ws_bytes(int64_t B, bool ragged) {
    return WS_TQ_BYTES + (ragged ? WS_COUNTS_BYTES + (size_t)V_NCLASS * (size_t)(B < 0 ? 0 : B) * 16 : 0);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set once per (kernel slot, device)
inline void ensure_max_lds(const void* fn, int slot, int dev) {
    static std::atomic<unsigned long long> done[IVS_MAX_DEV];
    const int d = (dev >= 0 && dev < IVS_MAX_DEV) ? dev : 0;
    const unsigned long long bit = 1ull << (slot & 63);
    if (dev < 0 || dev >= IVS_MAX_DEV || !(done[d].load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done[d].fetch_or(bit, std::memory_order_relaxed);
    }
}

// Launch the generic kernel (FILTER selects the "only tagged surfaces" variant).  Returns false if the
// shape does not fit in LDS.
template <bool FILTER>
inline bool launch_surface_generic(const SurfaceParams& p, const LaunchCtx& cx) {
    if (p.nK > 65535) return false;                               // 16-bit strike indices
    const size_t lds = generic_lds_bytes(p.nK, p.nT, method_is_cubic(p.method));
    if (lds > 160 * 1024) return false;
    if (lds > 64 * 1024) ensure_max_lds(reinterpret_cast<const void*>(surface_generic_kernel<FILTER>), FILTER ? 1 : 0, cx.dev);
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
    int64_t grid = (int64_t)cx.num_cu * per_cu * 2;
    const int64_t work = FILTER ? (p.B + 63) / 64 : p.B;
    if (grid > work) grid = work;
    hipLaunchKernelGGL(surface_generic_kernel<FILTER>, dim3((unsigned)grid), dim3(64), lds, cx.st, p);
    return true;
}

// number of workgroup groups (= regions of the batch = work-queue heads; see the mapping comment in surface_dense_kernel);
// `forced` > 0 is the caller's IVS_FLAG_MAP_GROUPS override (experiments).  With static striding small batches had to fall
// back to one group (uneven shares); with the work queues they must NOT: one head for 3072 workgroups runs at the atomic
// unit's pace (a 125 000-surface shard -- config 3 split over 8 GPUs -- ran at 252 instead of 317 M surfaces/s).
inline int dense_map_groups(int64_t grid, int64_t B, int forced) {
    int r = forced > 0 ? forced : 8;
    if (r > 16) r = 16;                                   // queue heads in the workspace
    while (r > 1 && (B < (int64_t)r * 256 || grid < r)) r >>= 1;      // tiny batches / grids: fewer, fuller regions
    return r < 1 ? 1 : r;
}

#ifndef IVS_DIAG_MINIMAL      // diagnostic builds (tools/pass_api.hip) skip the launchers that instantiate every kernel
inline bool launch_surface_masked(const SurfaceParams& p, const LaunchCtx& cx);      // ivs_surface_masked.hpp
// Dense dispatch.  Returns 1 if dispatched (dense kernel + filtered generic redo pass), 0 if the
// shape is not covered by a dense kernel.
inline int launch_surface_dense(const SurfaceParams& p_in, const LaunchCtx& cx, const char** name,
                                unsigned long long* dbg = nullptr, int64_t* grid_out = nullptr) {
    SurfaceParams p = p_in;
    hipStream_t st = cx.st;
    if (p.k_off || p.nK != DK || p.nT != DT) return 0;
    if (p.k_stride != 0 && p.k_stride < DK) return 0;
    if (reinterpret_cast<uintptr_t>(p.sigma) & 15) return 0;
    if (p.mT > D_MAX_MT) return 0;
    const size_t lds = dense_lds_bytes(p.mT);
    if (generic_lds_bytes(p.nK, p.nT) > 160 * 1024) return 0;
    int per_cu = (int)((160 * 1024) / lds);
    per_cu = per_cu > 8 ? 8 : per_cu;
    int64_t grid = (int64_t)cx.num_cu * per_cu;
    if (grid > p.B) grid = p.B;
    if (grid_out) *grid_out = grid;
    p.map_groups = dense_map_groups(grid, p.B, cx.map_groups);
    const bool tsh = p.t_stride == 0 && p.tq_stride == 0;
    const bool wl = p.mT <= D_WLDS_MAX_MT;
    if (tsh) {                                   // batch-wide maturity tables into the workspace, same stream
        TqShared* tq = reinterpret_cast<TqShared*>(cx.ws);
        launch_tq_tables<false>(p, tq, st);
        p.tqs = tq;
        p.redo = tq->redo;
        p.mode = &tq->mode;
    } else {                                     // per-surface maturities: no table kernel runs, the queue heads are zeroed here
        if (hipMemsetAsync(reinterpret_cast<TqShared*>(cx.ws)->queue, 0, sizeof(TqShared::queue), st) != hipSuccess) return -1;
    }
    p.queue = reinterpret_cast<TqShared*>(cx.ws)->queue;
    if (p.map_groups > 16) p.map_groups = 16;
    if (dbg) {   // diagnostic build: cubic and linear, shared T only
        if (!tsh) return 0;
        if (p.method == IVS_CUBIC) {
            if (wl) hipLaunchKernelGGL((surface_dense_kernel<IVS_CUBIC, true, true, true>), dim3((unsigned)grid), dim3(64), lds, st, p, dbg);
            else hipLaunchKernelGGL((surface_dense_kernel<IVS_CUBIC, true, false, true>), dim3((unsigned)grid), dim3(64), lds, st, p, dbg);
        } else {
            if (wl) hipLaunchKernelGGL((surface_dense_kernel<IVS_LINEAR, true, true, true>), dim3((unsigned)grid), dim3(64), lds, st, p, dbg);
            else hipLaunchKernelGGL((surface_dense_kernel<IVS_LINEAR, true, false, true>), dim3((unsigned)grid), dim3(64), lds, st, p, dbg);
        }
        *name = "surface_dense_kernel<stamp>";
        return hipGetLastError() == hipSuccess ? 1 : -1;
    }
#define IVS_DENSE_CASE(M, NAME)                                                                            \
    case M: {                                                                                              \
        if (tsh && wl) hipLaunchKernelGGL((surface_dense_kernel<M, true, true, false>), dim3((unsigned)grid), dim3(64), lds, st, p, nullptr);        \
        else if (tsh) hipLaunchKernelGGL((surface_dense_kernel<M, true, false, false>), dim3((unsigned)grid), dim3(64), lds, st, p, nullptr);        \
        else if (wl) hipLaunchKernelGGL((surface_dense_kernel<M, false, true, false>), dim3((unsigned)grid), dim3(64), lds, st, p, nullptr);         \
        else hipLaunchKernelGGL((surface_dense_kernel<M, false, false, false>), dim3((unsigned)grid), dim3(64), lds, st, p, nullptr);                \
        *name = NAME;                                                                                      \
        break;                                                                                             \
    }
    switch (p.method) {
        IVS_DENSE_CASE(IVS_LINEAR, "surface_dense_kernel<linear>")
        IVS_DENSE_CASE(IVS_CUBIC, "surface_dense_kernel<cubic>")
        IVS_DENSE_CASE(IVS_CUBICSPLINE, "surface_dense_kernel<cubicspline>")
        IVS_DENSE_CASE(IVS_SLINEAR, "surface_dense_kernel<slinear>")
        IVS_DENSE_CASE(IVS_PCHIP, "surface_dense_kernel<pchip>")
        IVS_DENSE_CASE(IVS_AKIMA, "surface_dense_kernel<akima>")
        default: return 0;
    }
#undef IVS_DENSE_CASE
    if (hipGetLastError() != hipSuccess) return -1;
    if (launch_surface_masked(p, cx) && p.redo) ++p.redo;      // tagged surfaces (missing quotes): compaction kernel first (shared T / Tq only) ...
    launch_surface_generic<true>(p, cx);     // ... then whatever is still tagged (returns at once when its counter is 0)
    return 1;
}

#endif  // IVS_DIAG_MINIMAL

}  // namespace ivs
