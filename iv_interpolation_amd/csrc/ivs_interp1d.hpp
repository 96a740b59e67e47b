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

namespace ivs {

struct Interp1dParams {
    const double* xk; const double* yk; int64_t yk_stride; const int64_t* knot_off;
    int64_t S; int C; int64_t total_knots;
    const double* xq; const int64_t* q_off; int64_t total_q;
    double* out; int64_t out_stride; int32_t* status; int method;
    double* wx; double* wy; double* ws; double* wcp; int32_t* wn;
};

__host__ __device__ inline size_t interp1d_ws_bytes(int64_t total_knots, int64_t S, int C) {
    return (size_t)4 * C * (size_t)total_knots * 8 + (size_t)S * C * 4 + 64;
}

__global__ __launch_bounds__(256) void interp1d_prepare_kernel(Interp1dParams p) {
    __shared__ int wave_cnt[4];
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
        if (valid) {
            int64_t r = base + pre + __popcll(m & ((1ull << lane) - 1ull));
            wx[r] = x[i]; wy[r] = v;
        }
        base += tot;
        __syncthreads();
    }
    const int minkn = method_min_knots(p.method);
    if (tid == 0) {
        p.wn[sc] = (int32_t)base;
        p.status[sc] = (base > 0 && base < minkn) ? IVS_ST_TOO_FEW_KNOTS : IVS_ST_OK;
        if (method_is_cubic(p.method) && base >= minkn && base >= 2) {
            CView xv{wx, 1}, yv{wy, 1};
            View sv{p.ws + (int64_t)c * p.total_knots + a, 1}, cv{p.wcp + (int64_t)c * p.total_knots + a, 1};
            method_slopes(p.method, xv, yv, sv, cv, (int)base);
        }
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

__global__ __launch_bounds__(256) void interp1d_eval_kernel(Interp1dParams p) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= p.total_q) return;
    const int64_t s = series_of(p.q_off, p.S, g);
    const int64_t a = p.knot_off[s];
    const double xq = p.xq ? p.xq[g] : (double)(g - p.q_off[s]);
    const int method = p.method;
    const int minkn = method_min_knots(method);
    for (int c = 0; c < p.C; ++c) {
        int n = p.wn[s * p.C + c];
        double r = qnan();
        if (n > 0 && n >= minkn) {
            CView x{p.wx + (int64_t)c * p.total_knots + a, 1}, y{p.wy + (int64_t)c * p.total_knots + a, 1};
            int j = find_interval(x, n, xq);
            CView sl{p.ws + (int64_t)c * p.total_knots + a, 1};
            r = eval_method(method, x, y, sl, n, j, xq);
        }
        p.out[c * p.out_stride + g] = r;
    }
}

struct FfillParams {
    const int64_t* src_pos; const int64_t* src_off; const uint8_t* valid; int64_t valid_stride; int n_cols;
    const int64_t* q_off; int64_t S; int64_t total_q; int32_t* idx_out; int64_t out_stride;
};

__global__ __launch_bounds__(256) void ffill_index_kernel(FfillParams p) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= p.total_q) return;
    const int64_t s = series_of(p.q_off, p.S, g);
    const int64_t pos = g - p.q_off[s];
    const int64_t lo0 = p.src_off[s], hi0 = p.src_off[s + 1];
    int64_t lo = lo0, hi = hi0;       // first j with src_pos[j] > pos
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (p.src_pos[mid] <= pos) lo = mid + 1; else hi = mid;
    }
    const int64_t jlast = lo - 1;
    for (int c = 0; c < p.n_cols; ++c) {
        const uint8_t* v = p.valid + c * p.valid_stride;
        int64_t j = jlast;
        while (j >= lo0 && !v[j]) --j;
        p.idx_out[c * p.out_stride + g] = j >= lo0 ? (int32_t)j : -1;
    }
}

}  // namespace ivs
