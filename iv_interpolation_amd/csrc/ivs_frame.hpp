// ONE pass over the output rows of a long frame (SURVEY 8 rows a4-a6 + 8f rank 1): the reference's own shape.
//
// core.py:58-74 per symbol: three interpolated channels, nine forward-filled columns, the `date` column, the dropna.
// Rounds 1-2 ran this as five kernels (eval, ffill index, two gathers, frame rows) with the forward-fill index -- 36 B per
// output row -- written to HBM and read back, one thread per row with a binary search per row and channel, 8-byte stores:
// 0.17-0.19 of the HBM roofline.  Here a block of 256 threads owns FR_ROWS = 4096 consecutive output rows:
//   * the source rows of every symbol the block touches are ONE contiguous range of the packed arrays.  Everything a row
//     needs is staged in LDS once per block: the source positions as output-row numbers (one ascending sequence across
//     the symbols), validity bytes, the compacted knots / slopes and the knot-rank table of every channel (rank = valid
//     knots among the symbol's rows 0..i, written by the prepare kernel), the forward-filled source columns themselves
//     and the per-symbol scalars -- after the staging a row touches no global memory except its stores;
//   * a thread takes PAIRS of consecutive rows (FR_UNITS pairs, 512 rows apart): one binary search per pair finds the
//     source interval, the rank table turns it into each channel's knot interval (no per-channel search), the second row
//     walks one step;
//   * every f64 column leaves as 16-byte stores, 1 KiB contiguous per wave instruction.
// Blocks whose symbols exceed the staging limits (FR_CAP source rows, FR_MAXSYM symbols, FR_MAXV / FR_MAXF / FR_MAXCC
// columns) take a per-row path on the global arrays: same results, the old cost.
#pragma once
#include "ivs_interp1d.hpp"

namespace ivs {

constexpr int FR_UNITS = 8;               // row pairs per thread
constexpr int FR_ROWS = 256 * 2 * FR_UNITS;      // rows per block
// Staged source rows per block (<= 256: one source row per thread).  The LDS image is 179 B per staged row; measured on the
// reference's shape (tests/bench/bench_symbols.py, 64 hourly rows -> 3781 minute rows, a block touches 2-3 symbols = 128-192
// source rows; profiles/r03/ab_frame_cap.txt): 256 rows = 48.9 KB = 3 blocks per CU 0.404 of HBM; 224 (3 per CU) 0.418;
// 208 = 39.7 KB = 4 per CU **0.466**; 192 (4 per CU, rows of the column arrays 1536 B apart = the same banks again) 0.44;
// 176 / 160 (4 / 5 per CU, but the three-symbol blocks fall to the window form) 0.394 / 0.409.
#ifndef IVS_FR_CAP
#define IVS_FR_CAP 208
#endif
constexpr int FR_CAP = IVS_FR_CAP;
static_assert(FR_CAP <= 256 && FR_CAP % 16 == 0, "one staged source row per thread; 16-byte aligned column rows");
constexpr int FR_MAXSYM = 64;             // symbols per block on the staged path
constexpr int FR_MAXV = 16;               // validity rows (forward-filled columns + the three Greek inputs)
constexpr int FR_MAXC = 3;                // channels
constexpr int FR_MAXF = 8;                // forward-filled f64 columns
constexpr int FR_MAXCC = 4;               // forward-filled code columns

struct FrameParams {
    const int64_t* src_pos; const int64_t* src_off; const int64_t* q_off; int64_t S, total_src, total_q;
    // forward fill
    const uint8_t* valid; int64_t valid_stride; int n_valid;
    const double* fsrc; int64_t fsrc_stride; const int32_t* f_rows; int n_f; double* f_out; int64_t f_stride;
    const int32_t* csrc; int64_t csrc_stride; const int32_t* c_rows; int n_c; int32_t* c_out; int64_t c_stride;
    const int32_t* idx_rows; int n_idx; int32_t* idx_out; int64_t idx_stride;
    // frame rows
    const int64_t* first_ns; const uint8_t* needs; int sym_col; int64_t* date_ns; uint8_t* keep;
    // Greeks epilogue (greeks == nullptr: off): validity rows of strike / interest_rate / callput or -1
    int g_strike, g_rate, g_put; const double* strike_src; const double* rate_src; const uint8_t* put_src;
    int ch_iv, ch_S, ch_T; double* greeks; int64_t greeks_stride;
};

typedef double v2d_u8 __attribute__((ext_vector_type(2), aligned(8)));
typedef int v2i_u4 __attribute__((ext_vector_type(2), aligned(4)));
typedef long long v2l_u8 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned short u16_u1 __attribute__((aligned(1)));

// ---- the per-row path on the global arrays (blocks beyond the staging limits): one output row g of symbol s
__device__ __forceinline__ void frame_row_global(const FrameParams& f, const Interp1dParams& p, int64_t g, int64_t s) {
    const int C = p.C, method = p.method;
    const double nanv = qnan();
    const int minkn = method_min_knots(method);
    const int64_t a = f.src_off[s], nsrc = f.src_off[s + 1] - a;
    const int64_t pos = g - f.q_off[s];
    const double xq = (double)pos;
    bool ok = true;
    double gv[3] = {nanv, nanv, nanv};
    for (int c = 0; c < C; ++c) {
        const int n = p.wn[s * C + c];
        CView x{p.wx + (int64_t)c * p.total_knots + a, 1}, y{p.wy + (int64_t)c * p.total_knots + a, 1}, sl{p.ws + (int64_t)c * p.total_knots + a, 1};
        double r = nanv;
        if (n > 0) {
            const int j = find_interval(x, n, xq);
            if (n >= minkn) r = eval_method(method, x, y, sl, n, j, xq);
            if (j >= 0 && x(j) == xq) r = y(j);
        }
        p.out[(int64_t)c * p.out_stride + g] = r;
        if (c < 3) { if (c == f.ch_iv) gv[0] = r; if (c == f.ch_S) gv[1] = r; if (c == f.ch_T) gv[2] = r; }
        ok = ok && !__builtin_isnan(r) && !(f.needs && f.needs[s * C + c] && p.status[s * C + c] != IVS_ST_OK);
    }
    int64_t x0 = 0, x1 = nsrc;                                     // source interval: last source row at or before the row
    while (x0 < x1) { const int64_t m = (x0 + x1) >> 1; if (f.src_pos[a + m] <= pos) x0 = m + 1; else x1 = m; }
    const int64_t jsrc = x0 - 1;
    auto fill = [&](int v) -> int64_t {
        const uint8_t* col = f.valid + (int64_t)v * f.valid_stride + a;
        int64_t j = jsrc;
        while (j >= 0 && !col[j]) --j;
        return j >= 0 ? a + j : -1;
    };
    for (int c = 0; c < f.n_f; ++c) { const int64_t i = fill(f.f_rows[c]); f.f_out[(int64_t)c * f.f_stride + g] = i >= 0 ? f.fsrc[(int64_t)c * f.fsrc_stride + i] : nanv; }
    bool sym_ok = true;
    for (int c = 0; c < f.n_c; ++c) {
        const int64_t i = fill(f.c_rows[c]);
        const int code = i >= 0 ? f.csrc[(int64_t)c * f.csrc_stride + i] : -1;
        f.c_out[(int64_t)c * f.c_stride + g] = code;
        if (c == f.sym_col) sym_ok = code >= 0;
    }
    for (int c = 0; c < f.n_idx; ++c) f.idx_out[(int64_t)c * f.idx_stride + g] = (int32_t)fill(f.idx_rows[c]);
    if (f.date_ns) { f.date_ns[g] = f.first_ns[s] + pos * 60000000000LL; f.keep[g] = (ok && sym_ok) ? 1 : 0; }
    if (f.greeks) {
        double K = nanv, rate = 0.0; int put = 0;
        if (f.g_strike >= 0) { const int64_t i = fill(f.g_strike); if (i >= 0) K = f.strike_src[i]; }
        if (f.g_rate >= 0) { const int64_t i = fill(f.g_rate); rate = i >= 0 ? f.rate_src[i] : nanv; }
        if (f.g_put >= 0) { const int64_t i = fill(f.g_put); put = i >= 0 ? f.put_src[i] : 2; }
        double de, ga, th, ve, rh;
        bs_greeks_one(gv[1], K, gv[2], rate, gv[0], put == 1, de, ga, th, ve, rh);
        if (put == 2) { de = nanv; ga = nanv; th = nanv; ve = nanv; rh = nanv; }
        f.greeks[0 * f.greeks_stride + g] = de; f.greeks[1 * f.greeks_stride + g] = ga; f.greeks[2 * f.greeks_stride + g] = th;
        f.greeks[3 * f.greeks_stride + g] = ve; f.greeks[4 * f.greeks_stride + g] = rh;
    }
}

// accessor of a channel's compacted knot array (index = knot number within the symbol): the knots that lie inside the
// block's staged window come from LDS, the few outside it (the knot just before / after the window, a symbol's last
// knot for the hull test, the whole array for 'nearest') from the compacted global array
struct WinView {
    const double* lds; const double* glob; int q0, q1;       // lds[q - q0] holds knot q for q0 <= q <= q1 (`lds` itself points INTO
                                                             // the staged array: a pointer offset by -q0 would leave the LDS object,
                                                             // and as a flat address its 64-bit arithmetic carries out of the aperture)
    __device__ __forceinline__ double operator()(int q) const { return (q >= q0 && q <= q1) ? lds[q - q0] : glob[q]; }
};

struct FrameShared {
    alignas(16) double cx[FR_MAXC][FR_CAP], cy[FR_MAXC][FR_CAP], cs[FR_MAXC][FR_CAP];     // 14.6 KB
    alignas(16) double s_f[FR_MAXF][FR_CAP];                                               // 13 KB
    int32_t s_c[FR_MAXCC][FR_CAP];                                                                         //  4 KB
    int32_t s_rel[FR_CAP];                 // output row of a staged source row, relative to the block's first row
    uint16_t s_rank[FR_MAXC][FR_CAP];      // valid knots of the channel among the symbol's rows 0..i
    uint8_t s_symk[FR_CAP];                // symbol of a staged source row, relative to the block's first symbol
    uint8_t s_val[FR_MAXV][FR_CAP];
    int64_t y_firstns[FR_MAXSYM];
    int32_t y_qrel[FR_MAXSYM + 1];         // first output row of the symbol, relative to the block's first row
    int32_t y_first[FR_MAXSYM + 1];        // the symbol's first source row, relative to the window (negative: before it)
    int32_t y_q0[FR_MAXC][FR_MAXSYM], y_q1[FR_MAXC][FR_MAXSYM];      // staged knots of (channel, symbol): q0..q1
    uint16_t y_n[FR_MAXC][FR_MAXSYM];
    uint8_t y_bad[FR_MAXSYM];              // bit c: channel c needed interpolation and failed (the symbol is None)
    int64_t s_edge[4];
};

// one staged block.  WHOLE: every source row of the block's symbols is staged (flat copy of the compacted knot arrays, no
// access outside LDS); !WHOLE: a window of the source rows, knots and fill sources outside it come from global memory
// MC: method class, a compile-time cut of the method switch -- 0 = linear / slinear (np.interp slopes), 1 = the Hermite family
// (cubic, cubicspline, pchip, akima: one eval_cubic), 2 = everything else (eval_method's full switch: nearest, zero,
// quadratic, the polynomial forms, the fill methods).  One body for all methods was 26 k instructions, twelve copies of the switch.
template <bool WHOLE, int MC>
__device__ __forceinline__ void frame_block_staged(const FrameParams& f, const Interp1dParams& p, FrameShared& sh, int64_t g0,
                                                   int64_t s_first, int nsym, int64_t W0, int64_t W1x) {
    const int tid = threadIdx.x;
    const int C = p.C, method = p.method;
    const int NS = (int)(W1x - W0);
    const double nanv = qnan();
    const bool has_table = MC == 1 || (MC == 2 && (method_is_cubic(method) || method_is_poly(method)));
    constexpr bool lerp_method = MC == 0;
    const int minkn = method_min_knots(method);

    // ---- staging: per symbol scalars, then everything per source row of the window
    if (tid < nsym) {
        const int64_t s = s_first + tid;
        const int64_t first_k = f.src_off[s] - W0, next_k = f.src_off[s + 1] - W0;
        sh.y_qrel[tid] = (int32_t)(f.q_off[s] - g0);
        sh.y_first[tid] = (int32_t)first_k;
        sh.y_firstns[tid] = f.first_ns ? f.first_ns[s] : 0;
        const int64_t ik = first_k > 0 ? first_k : 0, ie = (next_k < NS ? next_k : NS) - 1;      // the symbol's staged rows ik..ie
        unsigned bad = 0;
        for (int c = 0; c < C; ++c) {
            sh.y_n[c][tid] = (uint16_t)p.wn[s * C + c];
            if (f.needs && f.needs[s * C + c] && p.status[s * C + c] != IVS_ST_OK) bad |= 1u << c;
            int q0 = 0, q1 = WHOLE ? (int)p.wn[s * C + c] - 1 : -1;
            if (!WHOLE && ie >= ik) {
                const int64_t o = (int64_t)c * p.total_knots + W0;
                q0 = p.wr[o + ik] - (__builtin_isnan(p.yk[(int64_t)c * p.yk_stride + W0 + ik]) ? 0 : 1);
                q1 = p.wr[o + ie] - 1;
            }
            sh.y_q0[c][tid] = q0; sh.y_q1[c][tid] = q1;
        }
        sh.y_bad[tid] = (uint8_t)bad;
    }
    if (tid == 0) { sh.y_qrel[nsym] = 0x7fffffff; sh.y_first[nsym] = 0x7fffffff; }
    __syncthreads();
    {   // one staged source row per thread (NS <= FR_CAP <= 256 = the block's threads): every column load of the row is issued before the first
        // LDS store waits for any of them (a load -> store loop with run-time trip counts serialised ~30 global round
        // trips per block: 45 of the block's 105 us in the first version of this kernel)
        const int i = tid;
        const bool in = i < NS;
        const int64_t gi = W0 + (in ? i : 0);
        uint8_t vv[FR_MAXV]; double vy[FR_MAXC]; int vr[FR_MAXC]; double vf[FR_MAXF]; int vc[FR_MAXCC];
        const int64_t vpos = in ? f.src_pos[gi] : 0;
#pragma unroll
        for (int v = 0; v < FR_MAXV; ++v) vv[v] = (in && v < f.n_valid) ? f.valid[(int64_t)v * f.valid_stride + gi] : 0;
#pragma unroll
        for (int c = 0; c < FR_MAXC; ++c) {
            const bool on = in && c < C;
            vy[c] = on ? p.yk[(int64_t)c * p.yk_stride + gi] : nanv; vr[c] = on ? p.wr[(int64_t)c * p.total_knots + gi] : 0;
        }
#pragma unroll
        for (int c = 0; c < FR_MAXF; ++c) vf[c] = (in && c < f.n_f) ? f.fsrc[(int64_t)c * f.fsrc_stride + gi] : 0.0;
#pragma unroll
        for (int c = 0; c < FR_MAXCC; ++c) vc[c] = (in && c < f.n_c) ? f.csrc[(int64_t)c * f.csrc_stride + gi] : 0;
        int k = 0;
        if (in) { while (sh.y_first[k + 1] <= i) ++k; }                // symbols without source rows share a start: the last one owns it
        // a valid cell is knot q = rank - 1 of its symbol: slot = the symbol's first staged row + (q - q0); its slope (or
        // B-spline coefficient / polynomial weight) sits at the compacted index in the prepare kernel's table
        const int ik = in ? (sh.y_first[k] > 0 ? sh.y_first[k] : 0) : 0;
        const int64_t ak = in ? W0 + sh.y_first[k] : 0;               // flat index of the symbol's first source row
        // whole symbols staged: a flat copy of the compacted arrays (slot = row); window: see above
        double vs[FR_MAXC], wxv[FR_MAXC], wyv[FR_MAXC];
#pragma unroll
        for (int c = 0; c < FR_MAXC; ++c) {
            const bool on = in && c < C;
            const int64_t o = (int64_t)c * p.total_knots;
            if (WHOLE) {
                wxv[c] = on ? p.wx[o + gi] : 0.0; wyv[c] = on ? p.wy[o + gi] : 0.0; vs[c] = (on && has_table) ? p.ws[o + gi] : 0.0;
            } else {
                wxv[c] = (double)vpos; wyv[c] = vy[c];
                vs[c] = (on && has_table && !__builtin_isnan(vy[c])) ? p.ws[o + ak + vr[c] - 1] : 0.0;
            }
        }
        if (in) {
            sh.s_symk[i] = (uint8_t)k;
            sh.s_rel[i] = sh.y_qrel[k] + (int32_t)vpos;
#pragma unroll
            for (int v = 0; v < FR_MAXV; ++v) sh.s_val[v][i] = vv[v];
#pragma unroll
            for (int c = 0; c < FR_MAXC; ++c) {
                sh.s_rank[c][i] = (uint16_t)vr[c];
                if (c < C && (WHOLE || !__builtin_isnan(vy[c]))) {
                    const int slot = WHOLE ? i : ik + (vr[c] - 1 - sh.y_q0[c][k]);
                    sh.cx[c][slot] = wxv[c]; sh.cy[c][slot] = wyv[c]; sh.cs[c][slot] = vs[c];
                }
            }
#pragma unroll
            for (int c = 0; c < FR_MAXF; ++c) sh.s_f[c][i] = vf[c];
#pragma unroll
            for (int c = 0; c < FR_MAXCC; ++c) sh.s_c[c][i] = vc[c];
        }
        if (lerp_method) {     // np.interp's per-interval slope (IEEE division) once per knot
            __syncthreads();
            if (in && WHOLE) {      // flat arrays: a symbol's last entry pairs with the next symbol's first and is never read
                if (i + 1 < NS) {
#pragma unroll
                    for (int c = 0; c < FR_MAXC; ++c) if (c < C) sh.cs[c][i] = (sh.cy[c][i + 1] - sh.cy[c][i]) / (sh.cx[c][i + 1] - sh.cx[c][i]);
                }
            } else if (in) {        // window: the right neighbour of the last staged knot comes through the window accessor
#pragma unroll
                for (int c = 0; c < FR_MAXC; ++c) {
                    if (c < C && !__builtin_isnan(vy[c])) {
                        const int q = vr[c] - 1, n = sh.y_n[c][k];
                        const WinView X{sh.cx[c] + ik, p.wx + (int64_t)c * p.total_knots + ak, sh.y_q0[c][k], sh.y_q1[c][k]};
                        const WinView Y{sh.cy[c] + ik, p.wy + (int64_t)c * p.total_knots + ak, sh.y_q0[c][k], sh.y_q1[c][k]};
                        if (q + 1 < n) sh.cs[c][ik + q - sh.y_q0[c][k]] = (Y(q + 1) - vy[c]) / (X(q + 1) - (double)vpos);
                    }
                }
            }
        }
    }
    __syncthreads();
    int fv[FR_MAXF], cv[FR_MAXCC];                                 // validity row of every forward-filled column (uniform)
#pragma unroll
    for (int c = 0; c < FR_MAXF; ++c) fv[c] = c < f.n_f ? f.f_rows[c] : 0;
#pragma unroll
    for (int c = 0; c < FR_MAXCC; ++c) cv[c] = c < f.n_c ? f.c_rows[c] : 0;
    constexpr int NONE = -0x40000000;                              // "no valid source row": local indices may be negative (before the window)

    for (int unit = 0; unit < FR_UNITS; ++unit) {
        const int r0 = unit * 512 + 2 * tid;                      // the pair's first row, relative to the block
        const int64_t g = g0 + r0;
        if (g >= f.total_q) break;
        const bool two = g + 1 < f.total_q;
        // source interval of the first row by binary search (the sequence ascends across the block's symbols); the symbol
        // comes from q_off, not from the interval: a row in front of its symbol's first source row has no source row yet
        int j;
        { int a = 0, b = NS; while (a < b) { const int m = (a + b) >> 1; if (sh.s_rel[m] <= r0) a = m + 1; else b = m; } j = a - 1; }
        int k = j >= 0 ? (int)sh.s_symk[j] : 0;
        int jr[2], kr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = r0 + u;
            if (u) { while (j + 1 < NS && sh.s_rel[j + 1] <= r) ++j; }
            while (sh.y_qrel[k + 1] <= r) ++k;                        // the sentinel behind the last symbol ends the walk
            kr[u] = k;
            jr[u] = (j >= 0 && (int)sh.s_symk[j] == k) ? j : -1;
        }
        // ---- channels (core.py:58-61): the merged column = the interpolant, knot rows keep their source cell
        // the channel values the Greeks epilogue picks from, as SCALARS: as an array they were turned into a scratch array by
        // the compiler (its select chain over a uniform channel number became a dynamically indexed load), 96 B of scratch
        // stores per row pair whether Greeks were asked for or not -- 0.5 GB of HBM writes per 15 M rows by the counters
        // (and so was a select chain over six scalars at the point of use: the choice is made where the value is produced)
        double gS0 = nanv, gS1 = nanv, gT0 = nanv, gT1 = nanv, gI0 = nanv, gI1 = nanv;      // underlying, maturity, iv of the two rows
        bool row_ok[2] = {true, true};
#pragma unroll
        for (int c = 0; c < FR_MAXC; ++c) {
            if (c >= C) break;
            double r2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kk = kr[u], n = sh.y_n[c][kk];
                const double xq = (double)(r0 + u - sh.y_qrel[kk]);
                double r = nanv;
                if (n > 0) {
                    const int jc = jr[u] >= 0 ? (int)sh.s_rank[c][jr[u]] - 1 : -1;      // knots at or before the row's source interval
                    auto eval_row = [&](const auto& x, const auto& y, const auto& sl, int q0) {
                        double v = nanv;
                        if (n >= minkn) {
                            // the staged slope table holds np.interp's slopes; a knot in front of the window has none there
                            if (lerp_method) v = jc >= q0 ? eval_linear_slopes(x, y, sl, n, jc, xq, method == IVS_LINEAR)
                                                          : eval_linear(x, y, n, jc, xq, method == IVS_LINEAR);
                            else if (MC == 1) v = eval_cubic(x, y, sl, n, jc, xq, method_extrapolates_right(method));
                            else v = eval_method(method, x, y, sl, n, jc, xq);
                        }
                        if (jc >= 0 && x(jc) == xq) v = y(jc);     // a knot of this channel keeps its source cell
                        return v;
                    };
                    if constexpr (WHOLE) {
                        const int base = sh.y_first[kk];
                        r = eval_row(CView{sh.cx[c] + base, 1}, CView{sh.cy[c] + base, 1}, CView{sh.cs[c] + base, 1}, 0);
                    } else {
                        const int q0 = sh.y_q0[c][kk], q1 = sh.y_q1[c][kk];
                        const int ikk = sh.y_first[kk] > 0 ? sh.y_first[kk] : 0;          // the symbol's first staged row
                        const int64_t ga = (int64_t)c * p.total_knots + W0 + sh.y_first[kk];
                        r = eval_row(WinView{sh.cx[c] + ikk, p.wx + ga, q0, q1}, WinView{sh.cy[c] + ikk, p.wy + ga, q0, q1},
                                     WinView{sh.cs[c] + ikk, p.ws + ga, q0, q1}, q0);
                    }
                }
                r2[u] = r;
                row_ok[u] = row_ok[u] && !__builtin_isnan(r) && !((sh.y_bad[kk] >> c) & 1);
            }
            if (f.greeks) {                                          // uniform selects, c is a constant of the unrolled loop
                gS0 = c == f.ch_S ? r2[0] : gS0; gS1 = c == f.ch_S ? r2[1] : gS1;
                gT0 = c == f.ch_T ? r2[0] : gT0; gT1 = c == f.ch_T ? r2[1] : gT1;
                gI0 = c == f.ch_iv ? r2[0] : gI0; gI1 = c == f.ch_iv ? r2[1] : gI1;
            }
            double* o = p.out + (int64_t)c * p.out_stride + g;
            if (two) { v2d_u8 a; a.x = r2[0]; a.y = r2[1]; *reinterpret_cast<v2d_u8*>(o) = a; } else o[0] = r2[0];
        }
        // ---- forward fill (core.py:64-68): the last valid source row of the symbol at or before the row's source interval
        auto fill_index = [&](int v, int u) -> int {              // index relative to the window (negative: in front of it) or NONE
            const uint8_t* col = sh.s_val[v];
            const uint8_t* gcol = f.valid + (int64_t)v * f.valid_stride + W0;
            int jj = jr[u];
            const int first = sh.y_first[kr[u]];
            while (jj >= first && !((WHOLE || jj >= 0) ? col[jj] : gcol[jj])) --jj;
            return jj >= first ? jj : NONE;
        };
        const bool same = two && jr[0] == jr[1] && kr[0] == kr[1];   // both rows in one source interval: one lookup per column
#pragma unroll
        for (int c = 0; c < FR_MAXF; ++c) {
            if (c >= f.n_f) break;
            const int i0 = fill_index(fv[c], 0), i1 = same ? i0 : (two ? fill_index(fv[c], 1) : NONE);
            const double* gsrc = f.fsrc + (int64_t)c * f.fsrc_stride + W0;
            double* o = f.f_out + (int64_t)c * f.f_stride + g;
            v2d_u8 a;
            a.x = i0 != NONE ? ((WHOLE || i0 >= 0) ? sh.s_f[c][i0] : gsrc[i0]) : nanv;
            a.y = i1 != NONE ? ((WHOLE || i1 >= 0) ? sh.s_f[c][i1] : gsrc[i1]) : nanv;
            if (two) *reinterpret_cast<v2d_u8*>(o) = a; else o[0] = a.x;
        }
        bool sym_ok[2] = {true, true};
#pragma unroll
        for (int c = 0; c < FR_MAXCC; ++c) {
            if (c >= f.n_c) break;
            const int i0 = fill_index(cv[c], 0), i1 = same ? i0 : (two ? fill_index(cv[c], 1) : NONE);
            const int32_t* gsrc = f.csrc + (int64_t)c * f.csrc_stride + W0;
            int32_t* o = f.c_out + (int64_t)c * f.c_stride + g;
            v2i_u4 a;
            a.x = i0 != NONE ? ((WHOLE || i0 >= 0) ? sh.s_c[c][i0] : gsrc[i0]) : -1;
            a.y = i1 != NONE ? ((WHOLE || i1 >= 0) ? sh.s_c[c][i1] : gsrc[i1]) : -1;
            if (two) *reinterpret_cast<v2i_u4*>(o) = a; else o[0] = a.x;
            if (c == f.sym_col) { sym_ok[0] = a.x >= 0; sym_ok[1] = a.y >= 0; }
        }
        for (int c = 0; c < f.n_idx; ++c) {                      // raw index rows the host asked for (flat source row or -1)
            const int v = f.idx_rows[c];
            const int i0 = fill_index(v, 0), i1 = same ? i0 : (two ? fill_index(v, 1) : NONE);
            int32_t* o = f.idx_out + (int64_t)c * f.idx_stride + g;
            v2i_u4 a; a.x = i0 != NONE ? (int)(W0 + i0) : -1; a.y = i1 != NONE ? (int)(W0 + i1) : -1;
            if (two) *reinterpret_cast<v2i_u4*>(o) = a; else o[0] = a.x;
        }
        // ---- date column and the dropna keep flag (core.py:42-46, 74)
        if (f.date_ns) {
            v2l_u8 dd;
            dd.x = sh.y_firstns[kr[0]] + (long long)(r0 - sh.y_qrel[kr[0]]) * 60000000000LL;
            dd.y = sh.y_firstns[kr[1]] + (long long)(r0 + 1 - sh.y_qrel[kr[1]]) * 60000000000LL;
            const unsigned short kp = (unsigned short)(((row_ok[0] && sym_ok[0]) ? 1u : 0u) | ((row_ok[1] && sym_ok[1]) ? 0x100u : 0u));
            if (two) { *reinterpret_cast<v2l_u8*>(f.date_ns + g) = dd; *reinterpret_cast<u16_u1*>(f.keep + g) = kp; }
            else { f.date_ns[g] = dd.x; f.keep[g] = (uint8_t)(kp & 1u); }
        }
        // ---- Greeks epilogue (config.py:46 preserve_greeks; greeks.py:12-43), from the channel values still in registers
        if (f.greeks) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u && !two) break;
                double K = nanv, rate = 0.0; int put = 0;
                if (f.g_strike >= 0) { const int i = fill_index(f.g_strike, u); if (i != NONE) K = f.strike_src[W0 + i]; }
                if (f.g_rate >= 0) { const int i = fill_index(f.g_rate, u); rate = i != NONE ? f.rate_src[W0 + i] : nanv; }
                if (f.g_put >= 0) { const int i = fill_index(f.g_put, u); put = i != NONE ? f.put_src[W0 + i] : 2; }
                double de, ga, th, ve, rh;
                bs_greeks_one(u ? gS1 : gS0, K, u ? gT1 : gT0, rate, u ? gI1 : gI0, put == 1, de, ga, th, ve, rh);
                if (put == 2) { de = nanv; ga = nanv; th = nanv; ve = nanv; rh = nanv; }
                const int64_t gg = g + u;
                f.greeks[0 * f.greeks_stride + gg] = de; f.greeks[1 * f.greeks_stride + gg] = ga;
                f.greeks[2 * f.greeks_stride + gg] = th; f.greeks[3 * f.greeks_stride + gg] = ve;
                f.greeks[4 * f.greeks_stride + gg] = rh;
            }
        }
    }
}

template <int MC>
__global__ __launch_bounds__(256) void frame_fused_kernel(FrameParams f, Interp1dParams p) {
    __shared__ FrameShared sh;
    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * FR_ROWS;
    const int64_t g_last = g0 + FR_ROWS - 1 < f.total_q ? g0 + FR_ROWS - 1 : f.total_q - 1;
    if (tid == 0) sh.s_edge[0] = series_of(f.q_off, f.S, g0);
    if (tid == 64) sh.s_edge[1] = series_of(f.q_off, f.S, g_last);
    __syncthreads();
    const int64_t s_first = sh.s_edge[0], s_last = sh.s_edge[1];
    const int C = p.C;
    const int nsym = (int)(s_last - s_first + 1 < FR_MAXSYM + 1 ? s_last - s_first + 1 : FR_MAXSYM + 1);
    // the WINDOW of source rows the block needs: whole symbols when they fit, else from the last source row at or before the
    // block's first row (in the first symbol) to the last source row at or before its last row (in the last symbol)
    int64_t W0 = f.src_off[s_first], W1x = f.src_off[s_last + 1];
    const bool whole = W1x - W0 <= FR_CAP;                         // block-uniform: every source row of the block's symbols is staged
    if (!whole) {
        if (tid == 0 || tid == 64) {
            const int64_t s = tid ? s_last : s_first, a = f.src_off[s], n = f.src_off[s + 1] - a;
            const int64_t pos = (tid ? g_last : g0) - f.q_off[s];
            int64_t x0 = 0, x1 = n;
            while (x0 < x1) { const int64_t m = (x0 + x1) >> 1; if (f.src_pos[a + m] <= pos) x0 = m + 1; else x1 = m; }
            sh.s_edge[2 + (tid ? 1 : 0)] = tid ? a + x0 : a + (x0 > 0 ? x0 - 1 : 0);      // W1x (exclusive) / W0
        }
        __syncthreads();
        W0 = sh.s_edge[2]; W1x = sh.s_edge[3];
    }
    const bool staged = W1x - W0 <= FR_CAP && nsym <= FR_MAXSYM && f.n_valid <= FR_MAXV && C <= FR_MAXC && f.n_f <= FR_MAXF &&
                        f.n_c <= FR_MAXCC && (!f.greeks || C == 3);                                      // block-uniform
    if (!staged) {
        for (int64_t g = g0 + tid; g <= g_last; g += 256) {
            int64_t s = s_first;                                   // largest s in [s_first, s_last] with q_off[s] <= g
            { int64_t a = s_first, b = s_last; while (a < b) { const int64_t m = (a + b + 1) >> 1; if (f.q_off[m] <= g) a = m; else b = m - 1; } s = a; }
            frame_row_global(f, p, g, s);
        }
        return;
    }
    if (whole) frame_block_staged<true, MC>(f, p, sh, g0, s_first, nsym, W0, W1x);
    else frame_block_staged<false, MC>(f, p, sh, g0, s_first, nsym, W0, W1x);
}
__host__ __device__ constexpr int frame_method_class(int m) {
    return (m == IVS_LINEAR || m == IVS_SLINEAR) ? 0 : ((m == IVS_CUBIC || m == IVS_CUBICSPLINE || m == IVS_PCHIP || m == IVS_AKIMA) ? 1 : 2);
}

}  // namespace ivs
