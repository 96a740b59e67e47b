// IV -> OHLCV bridge (reference src/data_bridge/ohlcv_converter.py:138-369).
//
// The reference builds one synthetic candle per interpolated row from draws of NumPy's process-global legacy
// generator (MT19937: uniform / normal / exponential), row after row, symbol after symbol.  Reproducing its numbers
// therefore means reproducing that STREAM: the kernels below consume the same 32-bit words in the same order.
//   mt19937_words_kernel      the raw stream of np.random.seed(seed): one workgroup, the 624-word state double-buffered in LDS,
//                             a regeneration = one parallel step from the old state (round 3) and one barrier
//   bridge_count_kernel       words drawn by every row (0 for skipped rows) + block-local exclusive scan
//   bridge_scan_blocks_kernel exclusive scan of the block totals (one workgroup)
//   bridge_accept_kernel /    trend_following only: the polar-method normal deviates have data-dependent stream
//   bridge_gauss_walk_kernel  positions (rejection + a cached second deviate).  Acceptance of an attempt at every stream
//                             position is computed in parallel; one wavefront then walks the rows in order doing integer
//                             work only (next accepted position, flags staged through LDS) and leaves each row the
//                             position of its attempt; the deviates themselves are computed lane-parallel later
//   bridge_candles_kernel<S>  one wavefront per symbol, 64 rows per step: everything that does not depend on the
//                             previous candle is lane-parallel; the previous-close chain (spread_simulation,
//                             trend_following) is walked over the valid lanes with v_readlane broadcasts
// Arithmetic order follows the reference statement by statement (fp contraction off); Python's round(x, n) on
// floats (correctly rounded decimal, exact ties to even) is emulated exactly with an FMA residual.
#pragma once
#include "ivs_device.hpp"

namespace ivs {

enum { BR_SPREAD = 0, BR_MIDPOINT = 1, BR_TREND = 2, BR_SIMPLE = 3, BR_PIPELINE = 4 };   // 4: complete_pipeline.py:473-510

struct BridgeParams {
    const double* price; const double* volume; const int64_t* row_off; int64_t S; int64_t total_rows;
    int strategy; double base_spread_pct; double vol_factor;
    const uint32_t* words; int64_t n_words;
    double* out; uint8_t* valid;            // out [6][total_rows]: open, high, low, close, volume, source_price
    int64_t* rng_tail;                      // [0] words consumed (out), [1] has_gauss (in/out), [2] gauss bits (in/out),
                                            // [3] set to 1 when the word buffer was too short (out)
    int32_t* woff; int64_t* bbase; double* gauss; double* expu; uint8_t* acc; double* gauss0;   // workspace
};

constexpr int BR_BLOCK = 1024;              // rows per scan block (256 threads x 4)

__host__ __device__ inline int64_t bridge_blocks(int64_t rows) { return (rows + BR_BLOCK - 1) / BR_BLOCK; }
// doubles of the stream whose polar-method acceptance flag fits in the workspace (trend_following)
__host__ __device__ inline size_t bridge_acc_capacity(int64_t rows) { return (size_t)rows * 6 + 8192; }
__host__ __device__ inline size_t bridge_ws_bytes(int64_t rows) {
    const size_t nb = (size_t)bridge_blocks(rows) + 1;
    // woff, bbase, gauss (gpos), expu (epos), acceptance flags, gauss0 (+ alignment slack)
    return (size_t)rows * 4 + 64 + nb * 8 + 64 + 2 * ((size_t)rows * 8 + 64) + bridge_acc_capacity(rows) + 64 + 64;
}

// ---------------------------------------------------------------- MT19937 (numpy legacy seeding: init_genrand)
constexpr int MT_N = 624, MT_M = 397;
__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

// The in-place recurrence mt[i] = mt[(i + 397) % 624] ^ twist(mt[i], mt[i + 1]) reads, for i >= 227, words it has just
// written -- rounds 1-2 ran it as four dependent phases with a barrier between reading and writing each (nine barriers per
// 624 words, 0.79 G words/s: barrier and LDS latency, nothing else; now 1.48 G words/s).  Substituting the earlier phases gives every new word
// from OLD words only:
//   i <  227:  new[i] = old[i + 397] ^ tw(old[i], old[i + 1])
//   i <  454:  new[i] = old[i + 170] ^ tw(old[i - 227], old[i - 226]) ^ tw(old[i], old[i + 1])
//   i <  623:  new[i] = old[i -  57] ^ tw(old[i - 454], old[i - 453]) ^ tw(old[i - 227], old[i - 226]) ^ tw(old[i], old[i + 1])
//   i == 623:  new[623] = new[396] ^ tw(old[623], new[0]), both by the lines above
// so a regeneration is ONE parallel step into the other of two LDS buffers and one barrier (checked against NumPy's stream in
// Python before it went to the GPU, and by tests/test_bridge.py on it).
constexpr int MT_THREADS = 640;            // one new word per thread: ten wavefronts share the CU's four SIMDs (the step is issue-bound)
__global__ __launch_bounds__(MT_THREADS) void mt19937_words_kernel(uint32_t seed, uint32_t* words, int64_t n) {
    __shared__ uint32_t mtb[2][MT_N];
    const int tid = threadIdx.x;
    if (tid == 0) {
        uint32_t s = seed;
        for (int i = 0; i < MT_N; ++i) { mtb[0][i] = s; s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1); }
    }
    __syncthreads();
    int cur = 0;
    for (int64_t base = 0; base < n; base += MT_N, cur ^= 1) {
        const uint32_t* o = mtb[cur];
        uint32_t* w = mtb[cur ^ 1];
        const int i = tid;
        if (i < MT_N) {
            uint32_t v;
            if (i < 227) v = o[i + 397] ^ mt_twist(o[i], o[i + 1]);
            else if (i < 454) v = o[i + 170] ^ mt_twist(o[i - 227], o[i - 226]) ^ mt_twist(o[i], o[i + 1]);
            else if (i < 623) v = o[i - 57] ^ mt_twist(o[i - 454], o[i - 453]) ^ mt_twist(o[i - 227], o[i - 226]) ^ mt_twist(o[i], o[i + 1]);
            else {
                const uint32_t n0 = o[397] ^ mt_twist(o[0], o[1]);
                const uint32_t n396 = o[566] ^ mt_twist(o[169], o[170]) ^ mt_twist(o[396], o[397]);
                v = n396 ^ mt_twist(o[623], n0);
            }
            w[i] = v;
            if (base + i < n) words[base + i] = mt_temper(v);
        }
        __syncthreads();                                        // the new state is complete; the old buffer is free to be overwritten
    }
}

// legacy_double: two words -> [0, 1)
__device__ __forceinline__ double mt_double(uint32_t w0, uint32_t w1) {
    return ((double)(w0 >> 5) * 67108864.0 + (double)(w1 >> 6)) / 9007199254740992.0;
}

// ---------------------------------------------------------------- helpers
// Python's round(x, nd) for a float x (floatobject.c double_round: correctly rounded decimal, exact ties to even),
// p10 = 10^nd.  x * p10 = p + e exactly (FMA residual); only an exact tie of p can be tipped by e.
__device__ __forceinline__ double py_round(double x, double p10) {
#pragma clang fp contract(off)
    if (!(__builtin_fabs(x) < __builtin_inf())) return x;
    const double p = x * p10;
    const double e = __builtin_fma(x, p10, -p);
    double r = __builtin_rint(p);
    const double t = p - r;
    if (t == 0.5 && e > 0.0) r += 1.0;
    if (t == -0.5 && e < 0.0) r -= 1.0;
    return r / p10;
}
__device__ __forceinline__ bool row_valid(double base) { return !(__builtin_isnan(base) || base <= 0.0); }   // :156-157
__device__ __forceinline__ bool vol_missing(double v) { return __builtin_isnan(v) || v <= 0.0; }             // :364

__host__ __device__ inline int bridge_uniforms(int strategy) {      // doubles drawn by the candle builder itself
    return strategy == BR_SPREAD ? 5 : (strategy == BR_PIPELINE ? 4 : (strategy == BR_MIDPOINT ? 2 : (strategy == BR_SIMPLE ? 1 : 0)));
}

// ---------------------------------------------------------------- per-row word counts + scan
__global__ __launch_bounds__(256) void bridge_count_kernel(BridgeParams p) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * BR_BLOCK + tid * 4;
    const int per_row = 2 * bridge_uniforms(p.strategy);
    int c[4]; int tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + k;
        int n = 0;
        if (r < p.total_rows && row_valid(p.price[r])) n = per_row + ((p.volume == nullptr || vol_missing(p.volume[r])) ? 2 : 0);
        c[k] = n; tot += n;
    }
    // exclusive scan of the 256 thread totals
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int pre = 0, all = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) pre += wsum[w]; all += wsum[w]; }
    int run = pre + incl - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (r0 + k < p.total_rows) p.woff[r0 + k] = run; run += c[k]; }
    if (tid == 0) p.bbase[blockIdx.x] = all;        // block total; turned into an exclusive prefix by the next kernel
}

__global__ __launch_bounds__(256) void bridge_scan_blocks_kernel(BridgeParams p, int64_t nblocks) {
    __shared__ int64_t wsum[4];
    __shared__ int64_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblocks; b0 += 256) {
        const int64_t b = b0 + tid;
        const int64_t v = b < nblocks ? p.bbase[b] : 0;
        int64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int64_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int64_t pre = carry_s, all = 0;
        for (int w = 0; w < 4; ++w) { if (w < wave) pre += wsum[w]; all += wsum[w]; }
        if (b < nblocks) p.bbase[b] = pre + incl - v;
        __syncthreads();
        if (tid == 0) carry_s += all;
        __syncthreads();
    }
    if (tid == 0) {
        p.bbase[nblocks] = carry_s;
        p.rng_tail[0] = carry_s;                     // words consumed; the gauss cache is untouched by these strategies
        if (carry_s > p.n_words) p.rng_tail[3] = 1;
    }
}

// ---------------------------------------------------------------- trend_following: where the normal deviates sit
// legacy_gauss (polar method): an ATTEMPT at double position P uses doubles P, P+1; it is accepted iff
// 0 < x1^2 + x2^2 < 1; an accepted attempt yields two deviates (f*x2 now, f*x1 cached for the next call).  Whether an
// attempt is accepted depends on the stream only -> computed for every position in parallel; the order-dependent
// part that remains is a walk "next accepted position from here", a few integer steps per row.
__device__ __forceinline__ void polar_pair(const uint32_t* words, int64_t P, double& x1, double& x2, double& r2) {
#pragma clang fp contract(off)
    x1 = 2.0 * mt_double(words[2 * P], words[2 * P + 1]) - 1.0;
    x2 = 2.0 * mt_double(words[2 * P + 2], words[2 * P + 3]) - 1.0;
    r2 = x1 * x1 + x2 * x2;
}

__global__ __launch_bounds__(256) void bridge_accept_kernel(BridgeParams p, int64_t nd) {
    for (int64_t P = (int64_t)blockIdx.x * 256 + threadIdx.x; P < nd; P += (int64_t)gridDim.x * 256) {
        uint8_t a = 0;
        if (P + 1 < nd) { double x1, x2, r2; polar_pair(p.words, P, x1, x2, r2); a = !(r2 >= 1.0 || r2 == 0.0); }
        p.acc[P] = a;
    }
}

constexpr int GW_WIN = 16384;                          // acceptance flags staged per refill
constexpr int64_t G_CACHED = (int64_t)1 << 62;        // gpos flag: the row takes the cached deviate (f*x1) of that attempt
constexpr int64_t G_INITIAL = -1;                     // gpos value: the row takes the deviate cached before this call

__global__ __launch_bounds__(64) void bridge_gauss_walk_kernel(BridgeParams p, int64_t nd) {
#pragma clang fp contract(off)
    __shared__ uint8_t win[GW_WIN];
    const int lane = threadIdx.x;
    int64_t pos = 0;                                  // next draw, in doubles
    int64_t win_base = 0, win_end = 0;
    bool overflow = false;
    auto stage = [&](int64_t P) {                     // uniform call: the window of acceptance flags from position P on
        win_base = P & ~(int64_t)3;                      // keeps the 4-byte loads below aligned
        __syncthreads();
        for (int i = lane * 4; i < GW_WIN; i += 256) {
            uint32_t v = 0;
            if (win_base + i + 3 < nd) v = *reinterpret_cast<const uint32_t*>(p.acc + win_base + i);
            else for (int k = 0; k < 4; ++k) if (win_base + i + k < nd) v |= (uint32_t)p.acc[win_base + i + k] << (8 * k);
            *reinterpret_cast<uint32_t*>(win + i) = v;
        }
        __syncthreads();
        win_end = win_base + GW_WIN;
    };
    int64_t mbase = -64;                              // the mask covers the positions [mbase, mbase + 64)
    unsigned long long amask = 0ull;                  // bit i: an attempt at mbase + i is accepted (or runs off the stream: overflow)
    auto load_mask = [&](int64_t P) {                 // uniform call
        if (P + 64 > win_end) stage(P);
        const int64_t q = P + lane;
        const bool a = (q + 1 >= nd) || win[q - win_base] != 0;      // an attempt that would run off the stream counts as accepted: overflow
        amask = __ballot(a);
        mbase = P;
    };
    int64_t* gpos = reinterpret_cast<int64_t*>(p.gauss);
    int64_t* epos = reinterpret_cast<int64_t*>(p.expu);
    bool has = p.rng_tail[1] != 0;
    int64_t last = G_INITIAL;                         // attempt whose f*x1 is cached
    for (int64_t r0 = 0; r0 < p.total_rows; r0 += 64) {
        const int64_t r = r0 + lane;
        const double base = r < p.total_rows ? p.price[r] : __builtin_nan("");
        const double vol = (p.volume && r < p.total_rows) ? p.volume[r] : 0.0;
        // Valid rows alternate between SEARCH rows (a new attempt: the row's deviate is f*x2 of the next accepted attempt) and
        // CACHED rows (f*x1 of the same attempt), so the order-dependent chain runs over PAIRS only:
        //     P_k = next accepted attempt at or behind pos,   pos <- P_k + 2 + (exponentials drawn by the pair's two rows)
        // -- a dozen scalar instructions per pair.  Which row is which, the pair's exponential count and, afterwards, every
        // row's positions from its pair's P are lane-parallel.  (Rounds 1-2 walked row by row, ~60 instructions each with the
        // per-lane selects in the loop: 2.9 s per 15 M rows; now 1.53 s.)
        const unsigned long long vm = __ballot(row_valid(base));
        const unsigned long long emv = __ballot(p.volume == nullptr || vol_missing(vol)) & vm;      // valid rows that draw an exponential
        const unsigned long long lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
        const bool v_l = (vm >> lane) & 1ull, e_l = (emv >> lane) & 1ull;
        const int t0 = has ? 1 : 0;                                   // uniform: 1 = the next valid row takes a cached deviate
        const bool cached_l = v_l && (((t0 + __popcll(vm & lt)) & 1) != 0);
        const bool search_l = v_l && !cached_l;
        const unsigned long long after = vm & ~le, before = vm & lt;
        const int nl = after ? __builtin_ctzll(after) : 0, pl = before ? 63 - __builtin_clzll(before) : 0;
        const bool partner_e = after != 0ull && ((emv >> nl) & 1ull);     // search row: its cached partner (next valid row, if in this group) draws too
        unsigned long long S = __ballot(search_l);
        const unsigned long long C0 = __ballot(search_l && e_l), C1 = __ballot(search_l && partner_e);
        // a leading cached row (its search partner sat in an earlier group -- or before this call) draws its exponential first
        const int64_t last_in = last, pos_in = pos;
        if (t0 && vm) {
            if ((emv >> __builtin_ctzll(vm)) & 1ull) { pos += 1; if (pos > nd) overflow = true; }
        }
        int p_lo = 0, p_hi = 0;                                       // P of the pair, deposited in the search row's lane
        while (S) {
            const int l = __builtin_ctzll(S);
            S &= S - 1ull;
            int64_t P = pos;
            for (;;) {
                if (P < mbase || P >= mbase + 64) load_mask(P);
                const unsigned long long m = (amask >> (P - mbase)) & 0x5555555555555555ull;      // positions of P's parity
                if (m) { P += __builtin_ctzll(m); break; }
                P += ((mbase + 64 - P) + 1) & ~(int64_t)1;                                           // ... beyond the mask
            }
            if (P + 1 >= nd) overflow = true;
            p_lo = lane == l ? (int)(uint32_t)P : p_lo;
            p_hi = lane == l ? (int)(uint32_t)((uint64_t)P >> 32) : p_hi;
            const int c = (int)((C0 >> l) & 1ull) + (int)((C1 >> l) & 1ull);
            last = P;
            pos = P + 2 + c;
            if (c && pos > nd) overflow = true;
        }
        has = ((t0 + __popcll(vm)) & 1) != 0;
        // every row's positions from its pair's P
        const int q_lo = __shfl(p_lo, pl), q_hi = __shfl(p_hi, pl);   // the previous valid row's deposit (a cached row's search partner)
        const int64_t P_own = (int64_t)(((uint64_t)(uint32_t)p_hi << 32) | (uint32_t)p_lo);
        const int64_t P_prev = (int64_t)(((uint64_t)(uint32_t)q_hi << 32) | (uint32_t)q_lo);
        int64_t my_g = 0, my_e = -1;
        if (search_l) { my_g = P_own; my_e = e_l ? P_own + 2 : -1; }
        else if (cached_l && before != 0ull) {
            my_g = P_prev | G_CACHED;
            my_e = e_l ? P_prev + 2 + (int64_t)((emv >> pl) & 1ull) : -1;
        } else if (cached_l) {                                        // the leading cached row
            my_g = last_in == G_INITIAL ? G_INITIAL : (last_in | G_CACHED);
            my_e = e_l ? pos_in : -1;
        }
        if (r < p.total_rows) { gpos[r] = my_g; epos[r] = my_e; }        // coalesced; rows that draw nothing are never read
    }
    if (lane == 0) {
        double cached = __longlong_as_double(p.rng_tail[2]);
        p.gauss0[0] = cached;                         // the deviate cached BEFORE this call (rows tagged G_INITIAL)
        if (has && last != G_INITIAL && !overflow) {
            double x1, x2, r2; polar_pair(p.words, last, x1, x2, r2);
            cached = __builtin_sqrt(-2.0 * __ocml_log_f64(r2) / r2) * x1;
        }
        if (!has) cached = 0.0;
        p.rng_tail[0] = 2 * pos; p.rng_tail[1] = has ? 1 : 0; p.rng_tail[2] = __double_as_longlong(cached);
        if (overflow || 2 * pos > p.n_words) p.rng_tail[3] = 1;
    }
}

// ---------------------------------------------------------------- candles
__device__ __forceinline__ double rl_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int STRAT>
__global__ __launch_bounds__(64) void bridge_candles_kernel(BridgeParams p) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x;
    const double nanv = __builtin_nan("");
    for (int64_t s = blockIdx.x; s < p.S; s += gridDim.x) {
        const int64_t a = p.row_off[s], b = p.row_off[s + 1];
        // chain state, identical in every lane
        bool has_prev = false; double prev = 0.0;                 // spread_simulation: previous rounded close
        double ring[5] = {0, 0, 0, 0, 0}; int cnt = 0;            // trend_following: last five rounded closes (ring[4] newest)
        for (int64_t c0 = a; c0 < b; c0 += 64) {
            const int64_t row = c0 + lane;
            const bool in = row < b;
            const double base = in ? p.price[row] : nanv;
            const double vol = (in && p.volume) ? p.volume[row] : 0.0;
            const bool valid = in && row_valid(base);
            const bool vmiss = p.volume == nullptr || vol_missing(vol);
            const int64_t w = (valid && STRAT != BR_TREND) ? p.bbase[row >> 10] + (int64_t)p.woff[row] : 0;
            auto U = [&](int k) {                                  // k-th double of this row's draws
                if (STRAT == BR_TREND) {                           // only the exponential draw, at the position left by the walker
                    const int64_t e = valid ? reinterpret_cast<const int64_t*>(p.expu)[row] : -1;
                    return (e < 0 || 2 * e + 1 >= p.n_words) ? 0.0 : mt_double(p.words[2 * e], p.words[2 * e + 1]);
                }
                if (!valid || w + 2 * k + 1 >= p.n_words) return 0.0;
                return mt_double(p.words[w + 2 * k], p.words[w + 2 * k + 1]);
            };
            auto uniform = [](double lo, double hi, double d) { return lo + (hi - lo) * d; };
            double o = nanv, h = nanv, lw = nanv, c = nanv;
            double spread = 0.0, uo = 0.0, uc = 0.0, noise = 0.0;
            constexpr int NU = STRAT == BR_SPREAD ? 5 : (STRAT == BR_PIPELINE ? 4 : (STRAT == BR_MIDPOINT ? 2 : (STRAT == BR_SIMPLE ? 1 : 0)));
            if (STRAT == BR_SPREAD) {                              // ohlcv_converter.py:209-231
                const double vmul = uniform(0.5, p.vol_factor, U(0));
                spread = base * p.base_spread_pct * vmul;
                uo = uniform(-spread / 3, spread / 3, U(1));
                uc = uniform(-spread / 3, spread / 3, U(2));
            } else if (STRAT == BR_TREND) {
                double g = 0.0;
                if (valid) {
                    const int64_t gp = reinterpret_cast<const int64_t*>(p.gauss)[row];
                    if (gp == G_INITIAL) g = p.gauss0[0];
                    else {
                        const int64_t P = gp & ~G_CACHED;
                        if (2 * P + 3 < p.n_words) {
                            double x1, x2, r2; polar_pair(p.words, P, x1, x2, r2);
                            const double f = __builtin_sqrt(-2.0 * __ocml_log_f64(r2) / r2);
                            g = (gp & G_CACHED) ? f * x1 : f * x2;
                        }
                    }
                }
                noise = 0.0 + (base * 0.001) * g;                  // normal(0, sd) = loc + scale * gauss
            }
            // ---- the chain over the valid rows of this chunk (uniform loop, state replicated in all lanes)
            double tb = 0.0, trend = 0.0;
            if (STRAT == BR_SPREAD || STRAT == BR_TREND) {
                unsigned long long m = __ballot(valid);
                while (m) {
                    const int l = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const double bl = rl_f64(base, l);
                    if (STRAT == BR_SPREAD) {
                        const double tbl = has_prev ? (bl - prev) * 0.3 : 0.0;         // :225-228
                        const double cl = bl + (rl_f64(uc, l) + tbl * 0.5);             // :234-235
                        prev = py_round(cl, 1e4); has_prev = true;
                        if (lane == l) tb = tbl;
                    } else {
                        const int lookback = cnt < 5 ? cnt : 5;                         // :296-302
                        double tr = 0.0;
                        if (lookback > 1) {
                            const double first = lookback == 5 ? ring[0] : (lookback == 4 ? ring[1] : (lookback == 3 ? ring[2] : ring[3]));
                            tr = (ring[4] - first) / (double)lookback;
                        }
                        const double cl = bl + tr * 0.6 * 1.2 + rl_f64(noise, l);       // :309
                        ring[0] = ring[1]; ring[1] = ring[2]; ring[2] = ring[3]; ring[3] = ring[4];
                        ring[4] = py_round(cl, 1e4);
                        cnt += 1;
                        if (lane == l) trend = tr;
                    }
                }
            }
            // ---- lane-parallel completion
            if (STRAT == BR_SPREAD) {
                o = base + (uo + tb * 0.2);
                c = base + (uc + tb * 0.5);
                const double mid = (o + c) / 2;
                const double he = uniform(0.0, spread / 2, U(3));
                const double lr = uniform(0.0, spread / 2, U(4));
                h = (c > o ? c : o) + he;
                lw = (c < o ? c : o) - lr;
                if (h - lw < base * 0.0005) { h = mid + base * 0.00025; lw = mid - base * 0.00025; }
            } else if (STRAT == BR_MIDPOINT) {                     // :265-275
                const double sp = base * 0.001;
                o = base + uniform(-sp / 4, sp / 4, U(0));
                c = base + uniform(-sp / 4, sp / 4, U(1));
                h = base + sp / 2; lw = base - sp / 2;
            } else if (STRAT == BR_TREND) {                        // :304-318
                o = base + trend * 0.6 + noise;
                c = base + trend * 0.6 * 1.2 + noise;
                const double at = __builtin_fabs(trend);
                const double mx = c > o ? c : o, mn = c < o ? c : o;
                if (trend > 0) { h = mx + at * 0.5; lw = mn - at * 0.2; }
                else { h = mx + at * 0.2; lw = mn - at * 0.5; }
            } else if (STRAT == BR_PIPELINE) {                     // complete_pipeline.py:484-494
                const double sp = base * 0.001;
                o = base + uniform(-sp / 3, sp / 3, U(0));
                c = base + uniform(-sp / 3, sp / 3, U(1));
                h = (c > o ? c : o) + __builtin_fabs(uniform(0.0, sp / 2, U(2)));
                lw = (c < o ? c : o) - __builtin_fabs(uniform(0.0, sp / 2, U(3)));
            } else {                                               // :334-343
                const double sp = base * 0.001;
                o = base;
                c = base + uniform(-sp / 2, sp / 2, U(0));
                h = base + sp / 2; lw = base - sp / 2;
            }
            double v = vol;                                        // :359-369
            if (vmiss) v = 50.0 * (-__ocml_log_f64(1.0 - U(NU)));
            v = v > 0.0 ? v : 0.0;
            if (in) {
                const int64_t n = p.total_rows;
                p.valid[row] = valid ? 1 : 0;
                p.out[row] = valid ? py_round(o, 1e4) : nanv;
                p.out[n + row] = valid ? py_round(h, 1e4) : nanv;
                p.out[2 * n + row] = valid ? py_round(lw, 1e4) : nanv;
                p.out[3 * n + row] = valid ? py_round(c, 1e4) : nanv;
                p.out[4 * n + row] = valid ? py_round(v, 1e6) : nanv;
                p.out[5 * n + row] = valid ? base : nanv;
            }
        }
    }
}

}  // namespace ivs
