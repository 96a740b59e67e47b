// N-minute candle aggregation (reference src/candle_reconstruction/core.py:68-88):
// rows of one symbol sorted by timestamp; bucket = floor(ts / (N min)); open = first non-NaN, high = max,
// low = min, close = last non-NaN, volume = Kahan sum in row order (pandas group_sum), count = rows in the bucket.
// One thread per input row; the thread that sits on the FIRST row of a bucket reduces the whole bucket (buckets hold
// ~N rows) and writes the candle at that row's index; the other rows write count = 0.  Output is sparse (n rows),
// the host keeps rows with count >= N.  HBM traffic: 48 B in + 60 B out per row, coalesced.
#pragma once
#include "ivs_device.hpp"

namespace ivs {

struct CandleParams {
    const int64_t* ts; const double* o; const double* h; const double* l; const double* c; const double* v;
    const int64_t* series_off; int64_t S; int64_t n; int64_t freq_ns;
    int64_t* out_ts; double* out_o; double* out_h; double* out_l; double* out_c; double* out_v; int32_t* out_count;
};

__device__ __forceinline__ int64_t floor_div(int64_t a, int64_t b) {   // b > 0; floors towards -inf like pandas dt.floor
    int64_t q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}

__global__ __launch_bounds__(256) void candle_kernel(CandleParams p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    // series of row i: largest s with series_off[s] <= i
    int64_t lo = 0, hi = p.S;
    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (p.series_off[mid] <= i) lo = mid; else hi = mid - 1; }
    const int64_t a = p.series_off[lo], b = p.series_off[lo + 1];
    const int64_t bucket = floor_div(p.ts[i], p.freq_ns);
    const bool head = i == a || floor_div(p.ts[i - 1], p.freq_ns) != bucket;
    int32_t count = 0;
    if (head) {
        double open = qnan(), high = qnan(), low = qnan(), close = qnan(), sum = 0.0, comp = 0.0;
        for (int64_t j = i; j < b && floor_div(p.ts[j], p.freq_ns) == bucket; ++j) {
            const double oo = p.o[j], hh = p.h[j], ll = p.l[j], cc = p.c[j], vv = p.v[j];
            if (__builtin_isnan(open) && !__builtin_isnan(oo)) open = oo;
            if (!__builtin_isnan(hh) && !(high >= hh)) high = hh;     // first non-NaN or larger
            if (!__builtin_isnan(ll) && !(low <= ll)) low = ll;
            if (!__builtin_isnan(cc)) close = cc;
            if (!__builtin_isnan(vv)) {
#pragma clang fp contract(off)
                const double y = vv - comp;
                const double t = sum + y;
                comp = t - sum - y;
                sum = t;
            }
            ++count;
        }
        p.out_ts[i] = bucket * p.freq_ns;
        p.out_o[i] = open; p.out_h[i] = high; p.out_l[i] = low; p.out_c[i] = close; p.out_v[i] = sum;
    }
    p.out_count[i] = count;
}

}  // namespace ivs
