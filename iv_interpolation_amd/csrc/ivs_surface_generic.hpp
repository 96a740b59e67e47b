// Generic surface kernel: any strike count (ragged CSR), NaN-masked quotes, all four methods.
//
// One 64-lane workgroup (= one wavefront) per surface, grid-stride over the batch.
//   1. quotes of row t are loaded coalesced (lane = strike) and compacted with a wave ballot
//      into LDS (strike index and value [t][rank]); rows keep their own knot sets (NaN = missing quote).
//   2. cubic methods: lane t runs the not-a-knot Thomas recurrence of row t in LDS.
//   3. per block of 64 query strikes (lane = query): evaluate every row at the lane's strike,
//      compacting non-NaN results into a lane-private LDS column; solve the column's own
//      not-a-knot system; evaluate at every query maturity and store out[tq][q] coalesced.
// This is the correctness-first path; the dense fast path lives in ivs_surface_dense.hpp.
#pragma once
#include "ivs_device.hpp"

namespace ivs {

struct SurfaceParams {
    const double* K; const int64_t* k_off; int64_t k_stride; int nK;
    int64_t k_total;     // ragged batches: number of strikes K holds (k_off spans are checked against it), 0 = not checked
    const double* T; int64_t t_stride; int nT;
    const double* sigma; int64_t B;
    const double* Kq; int64_t kq_stride; int mK;
    const double* Tq; int64_t tq_stride; int mT;
    double* out; int32_t* status; int method;
    int map_groups;      // dense kernels: workgroups are split into this many groups, group r sweeps region r of the batch
    const void* tqs;     // dense kernels, T and Tq shared by the batch: TqShared tables in the caller's workspace (written by
                         // tq_tables_kernel on the same stream, read through the scalar cache)
    unsigned long long* queue;   // row-pass kernels: work-queue heads in the caller's workspace (zeroed by tq_tables_kernel), or nullptr
    const int* mode;     // nullptr, or TqShared::mode (written by tq_tables_kernel): 1 = the 64 x 16 fast kernels return at once and
                         // the compaction kernel takes every surface of the batch ("missing quotes first")
    int* redo;           // nullptr, or a counter in the caller's workspace (zeroed by tq_tables_kernel): the fast kernels count
                         // the surfaces they tag for a redo pass, and a FILTER pass whose counter is 0 returns at once instead
                         // of scanning B tags (two such scans were 1.7 % of a config-3 call with nothing to redo)
};
constexpr int QUEUE_STRIDE = 16;     // work-queue heads sit 128 B apart (one per workgroup group / size class)

// Work distribution of the persistent fast kernels.  Static striding (workgroup w takes surfaces w, w + G, ...) left 15 %
// of the workgroup time idle at the tail of a launch: the workgroups of one launch finished between 2.23 and 3.13 ms
// (tools/ends_probe.py; the memory system does not serve them evenly).  Surfaces are claimed from work queues instead:
// one head per workgroup group (group g = blockIdx % R sweeps region g of the batch front to back), a
// workgroup whose own region is exhausted helps the next group.  One claim = CH consecutive surfaces (a single head
// claimed surface by surface ran at the atomic unit's pace).  Single-wavefront workgroups only (the ticket is broadcast
// with readfirstlane).  Use: take() twice for the current and the next item; in the loop, while the current item is
// processed, `pend = begin()` requests the chunk the item after next may need and `finish(pend)` returns that item at
// the end of the iteration -- the atomic's latency stays off the critical path.
struct WorkQueue {
    unsigned long long* q; int R, g0, goff, gcur, ch, lane; int64_t region, limit, cur_i, cur_end;
    struct Pending { unsigned long long t; int g; bool need; };
    __device__ __forceinline__ void init(unsigned long long* heads, int groups, int64_t n_items, int chunk, int lane_) {
        q = heads; R = groups; limit = n_items; ch = chunk; lane = lane_;
        g0 = blockIdx.x % R; goff = 0; gcur = g0; region = (limit + R - 1) / R; cur_i = 0; cur_end = 0;
    }
    __device__ __forceinline__ unsigned long long claim(int g) const {       // lane 0's return value is the ticket
        unsigned long long t = 0ull;
        if (lane == 0) t = atomicAdd(q + g * QUEUE_STRIDE, 1ull);
        return t;
    }
    __device__ __forceinline__ bool resolve(unsigned long long t0, int g) {   // ticket -> chunk; false: group g is exhausted
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)t0), hi = __builtin_amdgcn_readfirstlane((unsigned)(t0 >> 32));
        return resolve_ticket((int64_t)(((unsigned long long)hi << 32) | lo), g);
    }
    __device__ __forceinline__ void skip_group() { ++goff; gcur = (g0 + goff) % R; }
    __device__ __forceinline__ bool resolve_ticket(int64_t t, int g) {        // uniform ticket value -> chunk
        int64_t gend = (int64_t)(g + 1) * region;
        gend = gend < limit ? gend : limit;
        const int64_t s0 = (int64_t)g * region + t * ch;
        if (t * ch >= region || s0 >= gend) return false;
        cur_i = s0; cur_end = s0 + ch < gend ? s0 + ch : gend;
        return true;
    }
    __device__ __forceinline__ int64_t take() {                               // next item, claiming synchronously if needed; -1: none left
        while (cur_i >= cur_end) {
            if (goff >= R) return -1;
            if (resolve(claim(gcur), gcur)) break;
            ++goff; gcur = (g0 + goff) % R;
        }
        return cur_i++;
    }
    __device__ __forceinline__ Pending begin(bool more) const {
        Pending pd; pd.g = gcur; pd.need = more && cur_i >= cur_end && goff < R; pd.t = pd.need ? claim(pd.g) : 0ull;
        return pd;
    }
    __device__ __forceinline__ int64_t finish(const Pending& pd, bool more) {
        if (!more) return -1;
        if (pd.need && !resolve(pd.t, pd.g)) { ++goff; gcur = (g0 + goff) % R; }
        return take();
    }
};
constexpr int WQ_CHUNK = 4;

// "something was tagged": a plain store of 1 (an atomic counter serialised a batch in which EVERY surface is tagged --
// 10 % missing quotes -- on one address: 83 -> 45 M surfaces/s); `told` makes it once per workgroup
__device__ __forceinline__ void count_redo(const SurfaceParams& p) { if (p.redo) *reinterpret_cast<volatile int*>(p.redo) = 1; }
__device__ __forceinline__ void count_redo(const SurfaceParams& p, bool& told) { if (!told) { count_redo(p); told = true; } }

constexpr int GEN_NTMAX = 32;

// LDS carve (per workgroup); LK = nKmax + 1 (row pad).  `slopes`: the method needs knot slopes (cubic, cubicspline,
// pchip, akima); the others (linear, slinear, nearest, zero, from_derivatives) carry neither slope planes nor
// Thomas scratch, which takes 64x16 surfaces from 51 KB (3 workgroups per CU) to 19 KB (8 per CU).
// Without slopes the compacted strikes are kept as 16-bit indices into Ksh; the cubic methods keep a copy of the
// values (their serial Thomas recurrence would pay the index indirection on every step: measured -8 %).
__host__ __device__ inline size_t generic_lds_bytes(int nKmax, int nT, bool slopes = true) {
    size_t LK = (size_t)nKmax + 1;
    size_t rows = (size_t)nT * LK;
    size_t scratch = rows > (size_t)nT * 64 ? rows : (size_t)nT * 64;   // row cp[] aliases column cp[]
    size_t doubles = (size_t)nKmax + GEN_NTMAX      // Ksh, Tsh
                     + rows                          // ys
                     + (size_t)nT * 64;              // cz
    if (slopes) doubles += 2 * rows + scratch + (size_t)nT * 64;       // xs, ss, cps / ccp, cs
    return doubles * 8 + (slopes ? 0 : ((rows * 2 + 15) & ~(size_t)15)) /*xi*/ + (size_t)nT * 64 /*ctidx*/ + GEN_NTMAX * 4 /*nrow*/;
}

struct RowX {   // row knot coordinate = Ksh[xi[i]]
    const uint16_t* idx; const double* K;
    __device__ __forceinline__ double operator()(int i) const { return K[idx[i]]; }
};
struct ColX {   // column knot coordinate = Tsh[ctidx[r][lane]]
    const uint8_t* idx; const double* T;
    __device__ __forceinline__ double operator()(int i) const { return T[idx[i * 64]]; }
};

// FILTER = true: second launch behind the dense kernel; only surfaces whose out[b][0] carries the
// "redo" sentinel are processed.
constexpr unsigned long long REDO_SENTINEL = 0x7ff8dead00000001ull;

template <bool FILTER>
__global__ __launch_bounds__(64) void surface_generic_kernel(SurfaceParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int nT = p.nT, nKmax = p.nK, LK = nKmax + 1;
    const int method = p.method;
    const bool cubic = method_is_cubic(method);
    const int minkn = method_min_knots(method);

    const size_t rows = (size_t)nT * LK;
    const size_t scratch = rows > (size_t)nT * 64 ? rows : (size_t)nT * 64;
    double* Ksh = reinterpret_cast<double*>(smem);
    double* Tsh = Ksh + nKmax;
    double* ys = Tsh + GEN_NTMAX;
    double* cz = ys + rows;
    double* xs = cz + (size_t)nT * 64;               // xs, ss, cps, cs exist only for the cubic methods
    double* ss = xs + (cubic ? rows : 0);
    double* cps = ss + (cubic ? rows : 0);
    double* cs = cps + (cubic ? scratch : 0);
    uint16_t* xi = reinterpret_cast<uint16_t*>(cs + (cubic ? (size_t)nT * 64 : 0));      // only without slopes
    uint8_t* ctidx = reinterpret_cast<uint8_t*>(xi) + (cubic ? 0 : ((rows * 2 + 15) & ~(size_t)15));
    int* nrow = reinterpret_cast<int*>(ctidx + (size_t)nT * 64);

    // FILTER: a wave inspects 64 surfaces per load (lane i reads the tag of surface base+i) and only walks the
    // tagged ones, so the redo pass costs microseconds when nothing was tagged.
    if (FILTER && p.redo && *p.redo == 0) return;      // nothing was tagged (wave-uniform)
    const int64_t n_outer = FILTER ? (p.B + 63) / 64 : p.B;
    for (int64_t ob = blockIdx.x; ob < n_outer; ob += gridDim.x) {
      unsigned long long todo = 1ull;
      if (FILTER) {
          const int64_t bi = ob * 64 + lane;
          const bool tagged = bi < p.B &&
              reinterpret_cast<const unsigned long long*>(p.out + bi * (int64_t)p.mT * p.mK)[0] == REDO_SENTINEL;
          todo = __ballot(tagged);
      }
      while (todo) {
        const int bit = FILTER ? __builtin_ctzll(todo) : 0;
        todo &= todo - 1;
        const int64_t b = FILTER ? ob * 64 + bit : ob;
        int64_t koff; int nKb;
        if (p.k_off) {
            koff = p.k_off[b];
            const int64_t span = p.k_off[b + 1] - koff;
            if (span < 0 || span > nKmax || koff < 0 || (p.k_total > 0 && koff + span > p.k_total)) {      // wave-uniform: offsets that the LDS carve / the caller's arrays cannot hold
                if (p.status && lane == 0) p.status[b] = IVS_ST_BAD_SHAPE;
                continue;
            }
            nKb = (int)span;
        } else { koff = b * p.k_stride; nKb = nKmax; }
        const double* Kb = p.K + koff;
        const double* sb = p.k_off ? p.sigma + (int64_t)nT * koff : p.sigma + b * (int64_t)nT * nKmax;
        const double* Tb = p.T + b * p.t_stride;
        const double* Kqb = p.Kq + b * p.kq_stride;
        const double* Tqb = p.Tq + b * p.tq_stride;
        double* outb = p.out + b * (int64_t)p.mT * p.mK;
        int st = 0;

        __syncthreads();   // previous surface's readers are done with LDS
        for (int k = lane; k < nKb; k += 64) Ksh[k] = Kb[k];
        if (lane < nT) Tsh[lane] = Tb[lane];
        // 1. load + compact rows.  All rows of a 64-strike chunk are loaded back to back (one memory round trip per
        //    chunk instead of one per row), then compacted row by row with a wave ballot.
        if (lane < nT) nrow[lane] = 0;
        __syncthreads();
        for (int c0 = 0; c0 < nKb; c0 += 64) {
            const int k = c0 + lane;
            const bool in = k < nKb;
            double v[GEN_NTMAX];
            const double kx = in ? Kb[k] : 0.0;
#pragma unroll
            for (int t = 0; t < GEN_NTMAX; ++t) v[t] = (t < nT && in) ? sb[(int64_t)t * nKb + k] : qnan();
#pragma unroll
            for (int t = 0; t < GEN_NTMAX; ++t) {
                if (t < nT) {                                  // wave-uniform
                    const bool valid = !__builtin_isnan(v[t]);
                    const unsigned long long m = __ballot(valid);
                    const int cnt = nrow[t];
                    const int rank = cnt + __popcll(m & ((1ull << lane) - 1ull));
                    if (valid) {
                        if (cubic) xs[t * LK + rank] = kx; else xi[t * LK + rank] = (uint16_t)k;
                        ys[t * LK + rank] = v[t];
                    }
                    if (lane == 0) nrow[t] = cnt + __popcll(m);
                }
            }
            __syncthreads();
        }
        __syncthreads();
        // 2. strike-direction slopes, lane = row
        if (cubic && lane < nT) {
            int n = nrow[lane];
            if (n >= minkn && n >= 2) {
                CView x{xs + lane * LK, 1}, y{ys + lane * LK, 1};
                View s{ss + lane * LK, 1}, c{cps + lane * LK, 1};
                method_slopes(method, x, y, s, c, n);
            }
        }
        __syncthreads();
        // 3. per query block
        for (int q0 = 0; q0 < p.mK; q0 += 64) {
            int q = q0 + lane;
            bool active = q < p.mK;
            double xq = active ? Kqb[q] : qnan();
            CView kfull{Ksh, 1};
            int jfull = find_interval(kfull, nKb, xq);
            int cn = 0;   // valid knots in this lane's column
#pragma unroll 4
            for (int t = 0; t < nT; ++t) {
                int n = nrow[t];
                double z = qnan();
                if (n > 0 && n < minkn) {
                    st |= IVS_ST_TOO_FEW_KNOTS;
                } else if (n > 0) {
                    CView y{ys + t * LK, 1}, s{ss + t * LK, 1};
                    if (cubic) {
                        CView x{xs + t * LK, 1};
                        int j = (n == nKb) ? jfull : find_interval(x, n, xq);
                        z = eval_method(method, x, y, s, n, j, xq);
                    } else {
                        RowX x{xi + t * LK, Ksh};
                        int j = (n == nKb) ? jfull : find_interval(x, n, xq);
                        z = eval_method(method, x, y, s, n, j, xq);
                    }
                }
                if (!__builtin_isnan(z)) {
                    cz[cn * 64 + lane] = z;
                    ctidx[cn * 64 + lane] = (uint8_t)t;
                    ++cn;
                }
            }
            // maturity direction, lane-private column
            ColX cx{ctidx + lane, Tsh};
            CView cy{cz + lane, 64};
            View csv{cs + lane, 64}, ccp{cps + lane, 64};
            bool few = cn > 0 && cn < minkn;
            if (few && active) st |= IVS_ST_TOO_FEW_KNOTS;
            bool solvable = cn > 0 && !few;
            if (cubic && solvable && cn >= 2) method_slopes(method, cx, cy, csv, ccp, cn);
            // query maturities are normally ascending: march the interval pointer instead of a binary search per row
            // (each search step costs two dependent LDS reads through the index column)
            int jc = -1;
            double xprev = -__builtin_inf();
#pragma unroll 4
            for (int tq = 0; tq < p.mT; ++tq) {
                double r = qnan();
                if (solvable) {
                    const double x = Tqb[tq];
                    if (x >= xprev) { while (jc + 1 < cn && cx(jc + 1) <= x) ++jc; }
                    else jc = find_interval(cx, cn, x);
                    xprev = x;
                    CView csr{cs + lane, 64};
                    r = eval_method(method, cx, cy, csr, cn, jc, x);
                }
                if (active) outb[(int64_t)tq * p.mK + q] = r;
            }
        }
        if (p.status) {
            // OR over lanes
            unsigned long long any = __ballot(st != 0);
            if (lane == 0) p.status[b] = any ? IVS_ST_TOO_FEW_KNOTS : IVS_ST_OK;
        }
      }
    }
}

}  // namespace ivs
