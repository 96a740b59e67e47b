// Batched 1-D masked-knot interpolation (the three channels of core.py:58-61 for many symbols)
// and the forward-fill gather index of core.py:64-68.
//
// prepare kernel: one 256-thread workgroup per (series, channel) compacts the valid knots into
//   the caller's workspace (ballot + cross-wave prefix) and, for the cubic methods, solves the
//   not-a-knot system for the knot slopes.
// eval kernel:    one thread per output row; binary search in the compacted knots, evaluate,
//   coalesced store.  Queries default to the integer lattice 0..m-1 (the reference's RangeIndex).
#pragma once
#include "ivs_device.hpp"
#include "ivs_greeks.hpp"
#include "ivs_surface_dense.hpp"      // DPP helpers, scan_mat2

namespace ivs {

struct Interp1dParams {
    const double* xk; const double* yk; int64_t yk_stride; const int64_t* knot_off;
    int64_t S; int C; int64_t total_knots;
    const double* xq; const int64_t* q_off; int64_t total_q;
    double* out; int64_t out_stride; int32_t* status; int method;
    double* wx; double* wy; double* ws; double* wcp; int32_t* wn;
    int32_t* wr;     // optional [C][total_knots]: valid knots of the channel among the series' rows 0..i (the fused frame pass turns a
                     // source-row interval into the channel's knot interval with it: no per-channel search); nullptr: not written
    // optional Greeks epilogue (config.interpolation.preserve_greeks; greeks == nullptr: off).  The channel values of a row are
    // still in registers when its delta..rho are formed; strike / interest_rate / callput come from the source rows
    // through the forward-fill gather index (rows fi_* of fidx; -1 = column absent: strike -> NaN Greeks, rate -> 0.0,
    // callput -> call).
    const int32_t* fidx; int64_t fidx_stride; int fi_strike, fi_rate, fi_put;
    const double* strike_src; const double* rate_src; const uint8_t* put_src;      // put_src: 0 call, 1 put, 2 null
    int ch_iv, ch_S, ch_T;
    double* greeks; int64_t greeks_stride;
};

__host__ __device__ inline size_t interp1d_ws_bytes(int64_t total_knots, int64_t S, int C) {
    return (size_t)4 * C * (size_t)total_knots * 8 + (size_t)S * C * 4 + 64;
}

constexpr int P1_STAGE = 512;    // knots per (series, channel) whose slope solve runs out of LDS

// ---- wave-level scans of affine maps f_i(x) = A_i x + B_i over 64 lanes (lanes without a map carry the identity (1, 0)).
// prefix: v_i = (f_i o f_{i-1} o ... o f_0)(x_in); suffix: v_i = (f_i o f_{i+1} o ... o f_63)(x_in).  Inside a row of
// 16 lanes the compositions are formed with DPP shifts, the four rows are then chained through v_readlane.
__device__ __forceinline__ double affine_prefix64(double A, double B, double x_in, int lane) {
#define IVS_AFF_STEP(SH)                                                                            \
    { const double a = dpp_f64<DPP_ROW_SHR(SH)>(1.0, A), b = dpp0_f64<DPP_ROW_SHR(SH)>(B); B = __builtin_fma(A, b, B); A = A * a; }
    IVS_AFF_STEP(1) IVS_AFF_STEP(2) IVS_AFF_STEP(4) IVS_AFF_STEP(8)
#undef IVS_AFF_STEP
    double v = 0.0, x = x_in;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double t = __builtin_fma(A, x, B);
        v = (lane >> 4) == r ? t : v;
        x = readlane_f64(t, 16 * r + 15);
    }
    return v;
}
__device__ __forceinline__ double affine_suffix64(double A, double B, double x_in, int lane) {
#define IVS_AFF_STEP(SH)                                                                            \
    { const double a = dpp_f64<DPP_ROW_SHL(SH)>(1.0, A), b = dpp0_f64<DPP_ROW_SHL(SH)>(B); B = __builtin_fma(A, b, B); A = A * a; }
    IVS_AFF_STEP(1) IVS_AFF_STEP(2) IVS_AFF_STEP(4) IVS_AFF_STEP(8)
#undef IVS_AFF_STEP
    double v = 0.0, x = x_in;
#pragma unroll
    for (int r = 3; r >= 0; --r) {
        const double t = __builtin_fma(A, x, B);
        v = (lane >> 4) == r ? t : v;
        x = readlane_f64(t, 16 * r);
    }
    return v;
}

// Not-a-knot slopes of ONE series-channel (n >= 4 knots staged in LDS px / py) by one wavefront: blocks of 64 knots in
// turn -- pivots by the Moebius scan of the dense kernels (factor_tables), the forward and the backward recurrence as
// affine scans -- instead of a serial Thomas recurrence on one thread.  cp_s / dp_s: LDS scratch of n entries (every
// lane re-reads only what it wrote itself); the slopes end up in dp_s.
__device__ __forceinline__ void nak_slopes_wave(const double* px, const double* py, int n, int lane, double* cp_s, double* dp_s) {
    const int nb = (n + 63) >> 6;
    double c00 = 1.0, c01 = 0.0, c10 = 0.0, c11 = 1.0, carry_crb = 0.0, dp_carry = 0.0;
    for (int blk = 0; blk < nb; ++blk) {
        const int i = blk * 64 + lane;
        const bool in = i < n;
        const int ii = in ? i : n - 1;
        const int ip = ii + 1 < n ? ii + 1 : n - 1, ipp = ii + 2 < n ? ii + 2 : n - 1, im = ii > 0 ? ii - 1 : 0, imm = ii > 1 ? ii - 2 : 0;
        const double x0 = px[ii], xp = px[ip], xpp = px[ipp], xm = px[im], xmm = px[imm];
        const double y0 = py[ii], yp = py[ip], ypp = py[ipp], ym = py[im], ymm = py[imm];
        const double dxc = xp - x0, dxm = x0 - xm, dxp = xpp - xp, dxmm = xm - xmm;
        const bool first = i == 0, last = i == n - 1;
        double a, b, c, r;
        if (first) {
            const double d = dxc + dxp, d0 = (yp - y0) * refined_rcp(dxc), d1 = (ypp - yp) * refined_rcp(dxp);
            a = 0.0; b = dxp; c = d; r = ((dxc + 2.0 * d) * dxp * d0 + dxc * dxc * d1) * refined_rcp(d);
        } else if (last) {
            const double d = dxmm + dxm, dA = (ym - ymm) * refined_rcp(dxmm), dB = (y0 - ym) * refined_rcp(dxm);
            a = d; b = dxmm; c = 0.0; r = (dxm * dxm * dA + (2.0 * d + dxm) * dxmm * dB) * refined_rcp(d);
        } else {
            const double dm = (y0 - ym) * refined_rcp(dxm), dc = (yp - y0) * refined_rcp(dxc);
            a = dxc; b = 2.0 * (dxm + dxc); c = dxm; r = 3.0 * (dxc * dm + dxm * dc);
        }
        if (!in) { a = 0.0; b = 1.0; c = 0.0; r = 0.0; }
        const double rb = refined_rcp(b);
        const double crb = c * rb;
        double crb_prev = dpp0_f64<DPP_WAVE_SHR1>(crb);
        if (blk > 0 && lane == 0) crb_prev = carry_crb;
        const bool ident = first || !in;
        const double g = ident ? 0.0 : a * rb * crb_prev;
        double p00 = 1.0, p01 = ident ? 0.0 : -g, p10 = ident ? 0.0 : 1.0, p11 = ident ? 1.0 : 0.0;
        scan_mat2<64>(p00, p01, p10, p11, lane);
        if (blk > 0) {
            const double n00 = p00 * c00 + p01 * c10, n01 = p00 * c01 + p01 * c11;
            const double n10 = p10 * c00 + p11 * c10, n11 = p10 * c01 + p11 * c11;
            p00 = n00; p01 = n01; p10 = n10; p11 = n11;
        }
        const double rw = first ? rb : (p10 + p11) * rb * refined_rcp(p00 + p01);      // 1 / w_i
        const double al = a * rw, cp = c * rw, rr = r * rw;
        const double dp = affine_prefix64(in ? -al : 1.0, in ? rr : 0.0, dp_carry, lane);     // dp_i = rr_i - al_i dp_{i-1}
        if (in) { cp_s[i] = cp; dp_s[i] = dp; }
        c00 = readlane_f64(p00, 63); c01 = readlane_f64(p01, 63); c10 = readlane_f64(p10, 63); c11 = readlane_f64(p11, 63);
        carry_crb = readlane_f64(crb, 63); dp_carry = readlane_f64(dp, 63);
    }
    double s_carry = 0.0;
    for (int blk = nb - 1; blk >= 0; --blk) {
        const int i = blk * 64 + lane;
        const bool in = i < n;
        const double cp = in ? cp_s[i] : 0.0, dp = in ? dp_s[i] : 0.0;
        const double sv = affine_suffix64(in ? -cp : 1.0, in ? dp : 0.0, s_carry, lane);      // s_i = dp_i - cp_i s_{i+1}
        if (in) dp_s[i] = sv;
        s_carry = readlane_f64(sv, 0);
    }
}

__global__ __launch_bounds__(256) void interp1d_prepare_kernel(Interp1dParams p) {
    __shared__ int wave_cnt[4];
    __shared__ double px[P1_STAGE], py[P1_STAGE], ps[P1_STAGE], pc[P1_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t sc = blockIdx.x;
    const int64_t s = sc / p.C; const int c = (int)(sc % p.C);
    const int64_t a = p.knot_off[s];
    const int64_t n = p.knot_off[s + 1] - a;
    const double* x = p.xk + a;
    const double* y = p.yk + c * p.yk_stride + a;
    double* wx = p.wx + (int64_t)c * p.total_knots + a;
    double* wy = p.wy + (int64_t)c * p.total_knots + a;
    int64_t base = 0;
    for (int64_t c0 = 0; c0 < n; c0 += 256) {
        int64_t i = c0 + tid;
        double v = i < n ? y[i] : qnan();
        bool valid = !__builtin_isnan(v);
        unsigned long long m = __ballot(valid);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
        for (int w = 0; w < 4; ++w) { int cw = wave_cnt[w]; if (w < wave) pre += cw; tot += cw; }
        const int64_t rk = base + pre + __popcll(m & ((1ull << lane) - 1ull));
        if (valid) { wx[rk] = x[i]; wy[rk] = v; }
        if (p.wr && i < n) p.wr[(int64_t)c * p.total_knots + a + i] = (int32_t)(rk + (valid ? 1 : 0));
        base += tot;
        __syncthreads();
    }
    const int minkn = method_min_knots(p.method);
    const bool solve = method_is_cubic(p.method) && base >= minkn && base >= 2;      // block-uniform
    const bool poly = method_is_poly(p.method);
    const bool poly_ok = poly && base >= 1 && base <= IVS_POLY_MAX_KNOTS;
    double* gs = p.ws + (int64_t)c * p.total_knots + a;
    if (tid == 0) {
        // a polynomial through more than IVS_POLY_MAX_KNOTS knots: the eval kernel sees 0 knots (all NaN) and the status
        // tells the host that the reference's (numerically meaningless) values are not reproduced
        p.wn[sc] = (poly && base > IVS_POLY_MAX_KNOTS) ? 0 : (int32_t)base;
        p.status[sc] = (base > 0 && base < minkn) ? IVS_ST_TOO_FEW_KNOTS
                     : ((poly && base > IVS_POLY_MAX_KNOTS) ? IVS_ST_ILL_CONDITIONED : IVS_ST_OK);
    }
    if (poly_ok) {      // per-knot table: barycentric weights (one thread per knot) or Newton coefficients (one thread)
        __syncthreads();                                       // the compacted knots of this block are visible
        for (int i = tid; i < (int)base; i += 256) { px[i] = wx[i]; py[i] = wy[i]; }
        __syncthreads();
        CView xv{px, 1}, yv{py, 1};
        if (p.method == IVS_BARYCENTRIC) {
            if (tid < (int)base) gs[tid] = barycentric_weight(xv, (int)base, tid);
        } else {
            if (tid == 0) { View cv{ps, 1}, vv{pc, 1}; krogh_coeffs(xv, yv, cv, vv, (int)base); }
            __syncthreads();
            if (tid < (int)base) gs[tid] = ps[tid];
        }
        return;
    }
    if (solve && base <= P1_STAGE) {
        // the recurrences read their knots from LDS (one global round trip per step would dominate)
        for (int i = tid; i < (int)base; i += 256) { px[i] = wx[i]; py[i] = wy[i]; }
        __syncthreads();
        const bool nak = p.method == IVS_CUBIC || p.method == IVS_CUBICSPLINE;
        if (nak && base >= 4) {
            if (wave == 0) nak_slopes_wave(px, py, (int)base, lane, pc, ps);     // one wavefront, scans instead of a serial chain
        } else if (tid == 0) {      // local-slope methods, the quadratic B-spline, 2- and 3-knot cubicspline: one thread
            CView xv{px, 1}, yv{py, 1};
            View sv{ps, 1}, cv{pc, 1};
            method_slopes(p.method, xv, yv, sv, cv, (int)base);
        }
        __syncthreads();
        for (int i = tid; i < (int)base; i += 256) gs[i] = ps[i];
    } else if (solve && tid == 0) {
        CView xv{wx, 1}, yv{wy, 1};
        View sv{gs, 1}, cv{p.wcp + (int64_t)c * p.total_knots + a, 1};
        method_slopes(p.method, xv, yv, sv, cv, (int)base);
    }
}

__device__ __forceinline__ int64_t series_of(const int64_t* off, int64_t S, int64_t g) {
    int64_t lo = 0, hi = S;           // largest s with off[s] <= g
    while (lo < hi) {
        int64_t mid = (lo + hi + 1) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Blocks of E1_ROWS consecutive output rows almost always lie inside ONE series (a symbol has thousands of minute
// rows): two lanes find the series of the block's first and last row, and when they agree the series' compacted knots
// (x, y, slopes) are staged in LDS once per channel, so the per-row binary search and the evaluation read LDS instead
// of chasing 6-12 dependent global loads per row.  Blocks that straddle series, or series with more than
// E1_STAGE knots, use the global arrays directly.  Four rows per thread amortise the staging (it cost as much as the
// 256 evaluations it served).
constexpr int E1_STAGE = 512;
constexpr int E1_RPT = 4;                       // rows per thread
constexpr int E1_ROWS = 256 * E1_RPT;           // rows per block

__global__ __launch_bounds__(256) void interp1d_eval_kernel(Interp1dParams p) {
    __shared__ double sx[E1_STAGE], sy[E1_STAGE], ss[E1_STAGE];
    __shared__ int64_t s_edge[2];
    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * E1_ROWS;
    const int64_t g_last = g0 + E1_ROWS - 1 < p.total_q ? g0 + E1_ROWS - 1 : p.total_q - 1;
    if (tid == 0) s_edge[0] = series_of(p.q_off, p.S, g0);
    if (tid == 64) s_edge[1] = series_of(p.q_off, p.S, g_last);
    __syncthreads();
    const int64_t s_first = s_edge[0], s_last = s_edge[1];
    const bool one_series = s_first == s_last;                 // block-uniform
    const bool two_series = s_last == s_first + 1;             // the block straddles one series boundary: both are staged
    int64_t sr[E1_RPT]; double xq[E1_RPT]; bool active[E1_RPT];
#pragma unroll
    for (int u = 0; u < E1_RPT; ++u) {
        const int64_t g = g0 + u * 256 + tid;
        active[u] = g < p.total_q;
        int64_t s = s_first;
        if (!one_series && active[u]) {                        // few candidates: walk
            while (s < s_last && p.q_off[s + 1] <= g) ++s;
        }
        sr[u] = s;
        xq[u] = !active[u] ? qnan() : (p.xq ? p.xq[g] : (double)(g - p.q_off[s]));
    }
    const int method = p.method;
    double gv[3][E1_RPT];                                      // Greeks epilogue: iv, S, T of the thread's rows
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int u = 0; u < E1_RPT; ++u) gv[k][u] = qnan();
    const bool cubic = method_is_cubic(method) || method_is_poly(method);      // a per-knot table rides along with (x, y)
    const bool lerp_method = method == IVS_LINEAR || method == IVS_SLINEAR;
    const int minkn = method_min_knots(method);
    for (int c = 0; c < p.C; ++c) {
        const int64_t a1 = p.knot_off[s_first];
        const int n1 = p.wn[s_first * p.C + c];
        const int n2 = two_series ? p.wn[s_last * p.C + c] : 0;
        // block-uniform.  (Round 3: a block that straddles ONE series boundary -- 27 % of the blocks at 3781 rows per series --
        // stages both series back to back instead of searching the global arrays row by row.)
        const bool staged = (one_series || two_series) && n1 + n2 > 0 && n1 + n2 <= E1_STAGE;
        if (staged) {
            const double* gx = p.wx + (int64_t)c * p.total_knots + a1;
            const double* gy = p.wy + (int64_t)c * p.total_knots + a1;
            const double* gs = p.ws + (int64_t)c * p.total_knots + a1;
            const int64_t d2 = two_series ? p.knot_off[s_last] - a1 : 0;       // the second series' compacted knots start here
            __syncthreads();                                   // previous channel's readers are done
            for (int i = tid; i < n1 + n2; i += 256) {
                const int64_t gi = i < n1 ? i : d2 + (i - n1);
                sx[i] = gx[gi]; sy[i] = gy[gi]; if (cubic) ss[i] = gs[gi];
            }
            __syncthreads();
            if (lerp_method) {     // np.interp's per-interval slope (dy/dx, IEEE division) once per knot instead of per row
                for (int i = tid; i + 1 < n1 + n2; i += 256)
                    if (i + 1 != n1) ss[i] = (sy[i + 1] - sy[i]) / (sx[i + 1] - sx[i]);      // (no interval across the two series)
                __syncthreads();
            }
        }
#pragma unroll
        for (int u = 0; u < E1_RPT; ++u) {
            const int64_t s = sr[u];
            const int64_t a = p.knot_off[s];
            const bool second = two_series && s == s_last;
            const int n = staged ? (second ? n2 : n1) : p.wn[s * p.C + c];
            const int sb = second ? n1 : 0;                    // the series' first staged knot
            const double* gx = p.wx + (int64_t)c * p.total_knots + a;
            const double* gy = p.wy + (int64_t)c * p.total_knots + a;
            const double* gs = p.ws + (int64_t)c * p.total_knots + a;
            double r = qnan();
            if (active[u] && n > 0 && n >= minkn) {
                CView x{staged ? sx + sb : gx, 1}, y{staged ? sy + sb : gy, 1}, sl{staged ? ss + sb : gs, 1};
                const int j = find_interval(x, n, xq[u]);
                if (staged && lerp_method) r = eval_linear_slopes(x, y, sl, n, j, xq[u], method == IVS_LINEAR);
                else r = eval_method(method, x, y, sl, n, j, xq[u]);
                // a row that IS a knot of this channel keeps its source cell in the reference's frame (the interpolant
                // reproduces it anyway, except the Newton form of 'krogh' in the last bits): `out` is the merged column
                if (j >= 0 && x(j) == xq[u]) r = y(j);
            } else if (active[u] && n > 0) {                   // too few knots to interpolate: knots keep their cells
                CView x{gx, 1}, y{gy, 1};
                const int j = find_interval(x, n, xq[u]);
                if (j >= 0 && x(j) == xq[u]) r = y(j);
            }
            if (active[u]) p.out[c * p.out_stride + g0 + u * 256 + tid] = r;
            if (p.greeks) {
                if (c == p.ch_iv) gv[0][u] = r;
                if (c == p.ch_S) gv[1][u] = r;
                if (c == p.ch_T) gv[2][u] = r;
            }
        }
    }
    if (p.greeks) {
#pragma unroll
        for (int u = 0; u < E1_RPT; ++u) {
            if (!active[u]) continue;
            const int64_t g = g0 + u * 256 + tid;
            const double nanv = qnan();
            double K = nanv, rate = 0.0;
            int put = 0;
            if (p.fi_strike >= 0) { const int32_t i = p.fidx[p.fi_strike * p.fidx_stride + g]; if (i >= 0) K = p.strike_src[i]; }
            if (p.fi_rate >= 0) { const int32_t i = p.fidx[p.fi_rate * p.fidx_stride + g]; rate = i >= 0 ? p.rate_src[i] : nanv; }
            if (p.fi_put >= 0) { const int32_t i = p.fidx[p.fi_put * p.fidx_stride + g]; put = i >= 0 ? p.put_src[i] : 2; }
            double de, ga, th, ve, rh;
            bs_greeks_one(gv[1][u], K, gv[2][u], rate, gv[0][u], put == 1, de, ga, th, ve, rh);
            if (put == 2) { de = nanv; ga = nanv; th = nanv; ve = nanv; rh = nanv; }        // null callput: undefined
            p.greeks[0 * p.greeks_stride + g] = de; p.greeks[1 * p.greeks_stride + g] = ga;
            p.greeks[2 * p.greeks_stride + g] = th; p.greeks[3 * p.greeks_stride + g] = ve;
            p.greeks[4 * p.greeks_stride + g] = rh;
        }
    }
}

struct FfillParams {
    const int64_t* src_pos; const int64_t* src_off; const uint8_t* valid; int64_t valid_stride; int n_cols;
    const int64_t* q_off; int64_t S; int64_t total_q; int32_t* idx_out; int64_t out_stride;
};

// Same block structure as the eval kernel (F1_ROWS rows per block, four per thread): the positions (and validity
// bytes) of the series' source rows are staged in LDS when the block lies inside one series with at most F1_STAGE
// source rows.
constexpr int F1_STAGE = 512;
constexpr int F1_COLS = 12;
constexpr int F1_RPT = 4;
constexpr int F1_ROWS = 256 * F1_RPT;

__global__ __launch_bounds__(256) void ffill_index_kernel(FfillParams p) {
    __shared__ int64_t spos[F1_STAGE];
    __shared__ uint8_t sval[F1_COLS][F1_STAGE];
    __shared__ int64_t s_edge[2];
    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * F1_ROWS;
    const int64_t g_last = g0 + F1_ROWS - 1 < p.total_q ? g0 + F1_ROWS - 1 : p.total_q - 1;
    if (tid == 0) s_edge[0] = series_of(p.q_off, p.S, g0);
    if (tid == 64) s_edge[1] = series_of(p.q_off, p.S, g_last);
    __syncthreads();
    const int64_t s_first = s_edge[0], s_last = s_edge[1];
    const int64_t lo1 = p.src_off[s_first], n1 = p.src_off[s_first + 1] - lo1;
    const bool staged = s_first == s_last && n1 <= F1_STAGE && p.n_cols <= F1_COLS;      // block-uniform
    if (staged) {
        for (int i = tid; i < n1; i += 256) {
            spos[i] = p.src_pos[lo1 + i];
            for (int c = 0; c < p.n_cols; ++c) sval[c][i] = p.valid[c * p.valid_stride + lo1 + i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < F1_RPT; ++u) {
        const int64_t g = g0 + u * 256 + tid;
        if (g >= p.total_q) continue;
        int64_t s = s_first;
        if (s_first != s_last) { while (s < s_last && p.q_off[s + 1] <= g) ++s; }
        const int64_t lo0 = p.src_off[s];
        const int64_t nsrc = p.src_off[s + 1] - lo0;
        const int64_t pos = g - p.q_off[s];
        int64_t lo = 0, hi = nsrc;                             // first local j with src_pos[j] > pos
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            const int64_t v = staged ? spos[mid] : p.src_pos[lo0 + mid];
            if (v <= pos) lo = mid + 1; else hi = mid;
        }
        const int64_t jlast = lo - 1;
        for (int c = 0; c < p.n_cols; ++c) {
            int64_t j = jlast;
            if (staged) { while (j >= 0 && !sval[c][j]) --j; }
            else { const uint8_t* v = p.valid + c * p.valid_stride + lo0; while (j >= 0 && !v[j]) --j; }
            p.idx_out[c * p.out_stride + g] = j >= 0 ? (int32_t)(lo0 + j) : -1;
        }
    }
}

// ---- columnar egress: forward-filled columns gathered on the device, so that the host receives the
// finished columns of the long output frame instead of an index to chase through its own memory.
//   out[c][g] = idx[idx_row[c]][g] >= 0 ? src[c][idx] : missing      (missing = NaN for f64 columns, -1 for code columns)
struct GatherParams {
    const void* src; int64_t src_stride; const int32_t* idx; int64_t idx_stride; const int32_t* idx_row; int n_cols;
    int64_t n; void* out; int64_t out_stride;
};
template <class T>
__global__ __launch_bounds__(256) void gather_rows_kernel(GatherParams p, T missing) {
    const T* src = static_cast<const T*>(p.src);
    T* out = static_cast<T*>(p.out);
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < p.n; g += (int64_t)gridDim.x * 256) {
        for (int c = 0; c < p.n_cols; ++c) {
            const int32_t i = p.idx[(int64_t)p.idx_row[c] * p.idx_stride + g];
            out[(int64_t)c * p.out_stride + g] = i >= 0 ? src[(int64_t)c * p.src_stride + i] : missing;
        }
    }
}

// Per output row of the long frame: its timestamp (first timestamp of the symbol + one minute per lattice step; only
// meaningful for symbols without duplicate timestamps, the host falls back otherwise) and whether the row survives the
// reference's dropna (core.py:74: symbol, iv, underlying_price, time_to_maturity all present) in a symbol that did not
// fail (a channel that needed interpolation but had too few knots makes the whole symbol None, core.py:83-85).
struct FrameRowsParams {
    const int64_t* q_off; int64_t S; int64_t total_q; const int64_t* first_ns;
    const double* chan; int64_t chan_stride; int n_chan; const int32_t* sym_code;
    const int32_t* status; const uint8_t* needs;             // [S][n_chan] each
    int64_t* date_ns; uint8_t* keep;
};
__global__ __launch_bounds__(256) void frame_rows_kernel(FrameRowsParams p) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < p.total_q; g += (int64_t)gridDim.x * 256) {
        const int64_t s = series_of(p.q_off, p.S, g);
        p.date_ns[g] = p.first_ns[s] + (g - p.q_off[s]) * 60000000000LL;
        bool ok = p.sym_code ? p.sym_code[g] >= 0 : true;
        for (int c = 0; c < p.n_chan; ++c) {
            ok = ok && !__builtin_isnan(p.chan[(int64_t)c * p.chan_stride + g]);
            ok = ok && !(p.needs[s * p.n_chan + c] && p.status[s * p.n_chan + c] != IVS_ST_OK);
        }
        p.keep[g] = ok ? 1 : 0;
    }
}

}  // namespace ivs
