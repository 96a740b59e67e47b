/*
 * ivs.h -- C ABI of the MI355X-native IV interpolation engine (libivs.so).
 *
 * This is the drop-in boundary for the reference hot path
 *   /root/reference/src/interpolation/core.py:16-85  IVInterpolator.interpolate_symbol
 * The reference has no FFI today (plain Python calling pandas); these entry points are
 * what a ctypes binding inside that class binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C linkage, no exceptions, no torch types; every pointer is a DEVICE pointer
 *     owned by the caller (e.g. torch tensor .data_ptr()) unless marked host;
 *   - fp64, row-major, contiguous; NaN in a value array means "missing quote" (not a knot);
 *   - asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *   - no allocation, no synchronisation inside a call (safe to capture in a hipGraph): every entry point
 *     that needs device scratch takes a caller-owned `workspace` sized by its *_workspace_bytes() function;
 *   - return 0 on success or a negative IVS_E* code; ivs_last_error() gives the text
 *     (thread local).  Numerical conditions (too few knots) are reported per item in
 *     the `status` array, never through the return code.
 */
#ifndef IVS_H
#define IVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IVS_ABI_VERSION 3

/* interpolation methods: the pandas method names that core.py:61 forwards
 * (`merged[col].interpolate(method=self.method)`) and that this engine implements */
enum {
    IVS_LINEAR      = 0, /* 'linear','index','values': np.interp; NaN left of the first knot, hold-last on the right */
    IVS_CUBIC       = 1, /* 'cubic': interp1d(kind=3) = not-a-knot spline; NaN outside the hull; needs >= 4 knots */
    IVS_CUBICSPLINE = 2, /* 'cubicspline': CubicSpline(not-a-knot); 2 knots = line, 3 = parabola; NaN left, extrapolates right */
    IVS_SLINEAR     = 3, /* 'slinear': interp1d(kind=1); NaN outside the hull; needs >= 2 knots */
    IVS_NEAREST     = 4, /* 'nearest': interp1d nearest, ties to the left knot; NaN outside the hull; >= 1 knot */
    IVS_ZERO        = 5, /* 'zero': order-0 spline = left knot's value; NaN outside the hull; >= 1 knot */
    IVS_PCHIP       = 6, /* 'pchip': PchipInterpolator; NaN left, extrapolates right; >= 2 knots */
    IVS_AKIMA       = 7, /* 'akima': Akima1DInterpolator; NaN outside the hull; >= 3 knots (scipy's 2-knot case is undefined) */
    IVS_FROM_DERIVATIVES = 8, /* 'from_derivatives' / 'piecewise_polynomial': BPoly on values = Bernstein-form lines; NaN outside; >= 2 */
    IVS_QUADRATIC   = 9, /* 'quadratic': interp1d(kind=2) = make_interp_spline(k=2), knots at the midpoints of the sites; NaN outside; >= 3 */
    IVS_BARYCENTRIC = 10,/* 'barycentric': scipy barycentric_interpolate, ONE polynomial through all valid knots; NaN left of the
                            first knot, the polynomial continues on the right; 1..IVS_POLY_MAX_KNOTS knots; 1-D kernels only */
    IVS_KROGH       = 11,/* 'krogh': scipy krogh_interpolate, the same polynomial in Newton form (divided differences); 1-D only */
    IVS_PAD         = 12,/* 'pad' / 'ffill': pandas pad_or_backfill forward (core.py:61 forwards the name; pandas 2.x still runs
                            it, with a FutureWarning): the last valid knot at or before the row; NaN left of the first knot,
                            hold-last on the right; never raises (>= 0 knots) */
    IVS_BFILL       = 13 /* 'bfill' / 'backfill': the mirror -- the first valid knot at or after the row; NaN right of the last
                            knot, the first knot's value on the left */
};
#define IVS_POLY_MAX_KNOTS 32   /* above this the reference's own polynomial is numerical noise: IVS_ST_ILL_CONDITIONED */
/* surfaces: methods 0-9 run on the fast (row-pass / one-pass / masked) kernels, 12 and 13 on the generic surface kernel,
 * 10 and 11 are rejected (EINVAL); the 1-D kernels take all of 0-13 */

/* return codes */
enum {
    IVS_OK            = 0,
    IVS_EINVAL        = -22, /* bad argument (null pointer, negative size, unknown method) */
    IVS_ERANGE        = -34, /* shape outside what the kernels support (see ivs_last_error) */
    IVS_ENOMEM        = -12, /* workspace too small */
    IVS_ELAUNCH       = -5   /* HIP launch error */
};

/* per-item status bits written to `status` arrays */
enum {
    IVS_ST_OK            = 0,
    IVS_ST_TOO_FEW_KNOTS = 1, /* the reference's scipy call raises here -> interpolate_symbol returns None */
    IVS_ST_BAD_SHAPE     = 2, /* ragged surface whose k_off span is negative, exceeds nK or leaves the strike array: skipped, outputs untouched */
    IVS_ST_ILL_CONDITIONED = 4 /* 'barycentric' / 'krogh' with more than IVS_POLY_MAX_KNOTS valid knots: values are NaN */
};

int         ivs_version(void);        /* IVS_ABI_VERSION of the loaded library */
const char* ivs_last_error(void);     /* host pointer, valid until the next call on this thread */
int         ivs_device_count(void);   /* number of visible HIP devices (0 if none) */

/*
 * Batch of 1-D series, C channels sharing knot coordinates: the arithmetic of
 * core.py:58-61 (three channels of one symbol) for S symbols at once.
 *
 *   xk  [total_knots]            knot coordinates of all series back to back, ascending within a series
 *   yk  [C][yk_stride]           channel c of knot i at yk[c*yk_stride + i]; NaN = not a knot for that channel
 *   knot_off [S+1]               CSR offsets into xk / yk
 *   xq  [total_queries] or NULL  query coordinates; NULL = 0,1,..,m_s-1 per series (the reference's RangeIndex)
 *   q_off [S+1]                  CSR offsets into xq / out
 *   out [C][out_stride]          result of channel c, query i at out[c*out_stride + i]; a query that coincides with a
 *                                knot of the channel returns that knot's value (the merged column of the reference's
 *                                frame), also when status reports too few knots to interpolate between them
 *   status [S*C]                 IVS_ST_* per (series, channel)
 *   workspace                    ivs_interp1d_workspace_bytes(total_knots, S, C) bytes of device memory
 */
size_t ivs_interp1d_workspace_bytes(int64_t total_knots, int64_t n_series, int32_t n_channels);
int ivs_interp1d_batch_f64(const double* xk, const double* yk, int64_t yk_stride, const int64_t* knot_off,
                           int64_t n_series, int32_t n_channels, int64_t total_knots,
                           const double* xq, const int64_t* q_off, int64_t total_queries,
                           double* out, int64_t out_stride, int32_t* status, int32_t method,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same call with the Black-Scholes Greeks as an EPILOGUE (reference config.py:46 `preserve_greeks`: "Recalculate Greeks
 * after interpolation"; columns delta, gamma, theta, vega, rho of src/database/schema.py:36-40; formulas
 * src/interpolation/greeks.py:12-43).  The reference never wires the flag up; here the eval kernel forms the five values
 * of an output row from the channel values it has just produced (a row that is a knot of a channel uses the source cell,
 * as the reference's frame does) plus the forward-filled strike / interest_rate / callput of that row:
 *   ch_iv, ch_underlying, ch_ttm     channel numbers of sigma, S and T inside yk
 *   fill_idx [rows][fill_stride]     output of ivs_ffill_index_batch for this batch (same q_off); row_strike / row_rate /
 *                                    row_callput = its row for that column, or -1 when the frame has no such column
 *                                    (no strike: NaN Greeks; no interest_rate: 0.0, the schema default; no callput: call)
 *   strike_src, rate_src [total_src] source-row values of those columns; is_put_src: 0 call, 1 put, 2 null (-> NaN Greeks)
 *   greeks [5][greeks_stride]        delta, gamma, theta (per day), vega and rho (per 1 %), put rho unsigned like the reference
 */
int ivs_interp1d_greeks_batch_f64(const double* xk, const double* yk, int64_t yk_stride, const int64_t* knot_off,
                                  int64_t n_series, int32_t n_channels, int64_t total_knots,
                                  const double* xq, const int64_t* q_off, int64_t total_queries,
                                  double* out, int64_t out_stride, int32_t* status, int32_t method,
                                  int32_t ch_iv, int32_t ch_underlying, int32_t ch_ttm,
                                  const int32_t* fill_idx, int64_t fill_stride, int32_t row_strike, int32_t row_rate,
                                  int32_t row_callput, const double* strike_src, const double* rate_src,
                                  const uint8_t* is_put_src, double* greeks, int64_t greeks_stride,
                                  void* workspace, size_t workspace_bytes, void* stream);

/*
 * Forward-fill gather index: the "index of the last valid source row at or before
 * output row i" of core.py:64-68 (nine `fillna(method='ffill')` columns), for S symbols.
 *
 *   src_pos [total_src]          merged-frame row position of every source row, ascending within a series
 *   src_off [S+1]                CSR offsets into src_pos / valid
 *   valid   [n_cols][valid_stride]  1 = source cell is non-null in that column
 *   q_off   [S+1]                CSR offsets of the output rows (row i of series s is position i - q_off[s])
 *   idx_out [n_cols][out_stride] flat source-row index to gather from, or -1 (stays null)
 */
int ivs_ffill_index_batch(const int64_t* src_pos, const int64_t* src_off, const uint8_t* valid, int64_t valid_stride,
                          int32_t n_cols, const int64_t* q_off, int64_t n_series, int64_t total_queries,
                          int32_t* idx_out, int64_t out_stride, void* stream);

/*
 * Columnar egress of the forward-filled columns (core.py:64-68 for every symbol at once): column c of the long output
 * frame is gathered on the device with the index ivs_ffill_index_batch produced,
 *   out[c][g] = idx[idx_row[c]][g] >= 0 ? src[c][idx[idx_row[c]][g]] : missing     (missing: NaN / -1)
 * f64 for numeric columns, i32 for the codes of object columns (symbol, callput); idx_row [n_cols] (device) names the
 * row of idx each column uses.
 */
int ivs_gather_rows_f64(const double* src, int64_t src_stride, const int32_t* idx, int64_t idx_stride, const int32_t* idx_row,
                        int32_t n_cols, int64_t n, double* out, int64_t out_stride, void* stream);
int ivs_gather_rows_i32(const int32_t* src, int64_t src_stride, const int32_t* idx, int64_t idx_stride, const int32_t* idx_row,
                        int32_t n_cols, int64_t n, int32_t* out, int64_t out_stride, void* stream);

/*
 * Row bookkeeping of the long output frame on the device: date_ns[g] = first_ns[s] + (g - q_off[s]) minutes (s = the
 * symbol of row g; valid for symbols without duplicate timestamps) and keep[g] = the row survives the reference's dropna
 * (core.py:74) in a symbol that did not fail (status[s][c] != 0 on a channel with needs[s][c] -> the symbol is None).
 * chan [n_channels][chan_stride] = the merged channel columns (out of ivs_interp1d_batch_f64), sym_code [total_queries] =
 * gathered symbol codes (negative = null) or NULL.
 */
int ivs_frame_rows(const int64_t* q_off, int64_t n_series, int64_t total_queries, const int64_t* first_ns,
                   const double* chan, int64_t chan_stride, int32_t n_channels, const int32_t* sym_code,
                   const int32_t* status, const uint8_t* needs, int64_t* date_ns, uint8_t* keep, void* stream);

/*
 * The whole long output frame in ONE pass over its rows (core.py:54-74 for S symbols at once; ABI 3): what the five calls
 * above do together -- ivs_interp1d[_greeks]_batch_f64 on the integer lattice, ivs_ffill_index_batch, ivs_gather_rows_f64 /
 * _i32 and ivs_frame_rows -- without the forward-fill index ever reaching memory: a block owns 1024 consecutive output rows,
 * stages the source rows of its symbols in LDS, a thread walks four consecutive rows (one interval search per thread, not per
 * row and channel) and every column leaves as 16-byte stores.  Same results as the separate calls, bit for bit.
 *
 *   src_pos [total_src], src_off [S+1], q_off [S+1]   as in ivs_ffill_index_batch; src_off is also the CSR of the knots
 *   yk [C][yk_stride], chan_out [C][chan_stride], status [S*C], method          as in ivs_interp1d_batch_f64 (xq = the lattice)
 *   valid [n_valid][valid_stride]                   validity rows (1 = source cell non-null), n_valid <= 16 on the fast path
 *   fsrc [n_f][fsrc_stride] f64 / csrc [n_c][csrc_stride] i32   source columns; f_rows [n_f] / c_rows [n_c] (device) = the
 *                                                   validity row each column forward-fills by; f_out [n_f][f_stride],
 *                                                   c_out [n_c][c_stride] (missing: NaN / -1)
 *   idx_rows [n_idx] (device), idx_out [n_idx][idx_stride]      raw gather index (flat source row or -1) of these validity
 *                                                   rows, for columns the host gathers itself; n_idx may be 0
 *   first_ns [S], needs [S*C], sym_col, date_ns [total_queries], keep [total_queries]   as in ivs_frame_rows; sym_col = the
 *                                                   code column that holds the symbol (its -1 drops the row) or -1;
 *                                                   date_ns == NULL skips both
 *   g_strike / g_rate / g_put, strike_src, rate_src, put_src, ch_*, greeks [5][greeks_stride]   as in
 *                                                   ivs_interp1d_greeks_batch_f64 with validity rows in place of fill_idx
 *                                                   rows; greeks == NULL: no epilogue
 *   workspace   ivs_frame_workspace_bytes(total_src, S, C) bytes
 */
typedef struct ivs_frame_args {
    const int64_t* src_pos; const int64_t* src_off; const int64_t* q_off;
    int64_t n_series, total_src, total_queries;
    const double* yk; int64_t yk_stride; int32_t n_channels; int32_t method;
    double* chan_out; int64_t chan_stride; int32_t* status;
    const uint8_t* valid; int64_t valid_stride; int32_t n_valid;
    const double* fsrc; int64_t fsrc_stride; const int32_t* f_rows; int32_t n_f; double* f_out; int64_t f_stride;
    const int32_t* csrc; int64_t csrc_stride; const int32_t* c_rows; int32_t n_c; int32_t* c_out; int64_t c_stride;
    const int32_t* idx_rows; int32_t n_idx; int32_t* idx_out; int64_t idx_stride;
    const int64_t* first_ns; const uint8_t* needs; int32_t sym_col; int64_t* date_ns; uint8_t* keep;
    int32_t g_strike, g_rate, g_put; const double* strike_src; const double* rate_src; const uint8_t* put_src;
    int32_t ch_iv, ch_underlying, ch_ttm; double* greeks; int64_t greeks_stride;
} ivs_frame_args;
size_t ivs_frame_workspace_bytes(int64_t total_src, int64_t n_series, int32_t n_channels);
int ivs_frame_columns_f64(const ivs_frame_args* args /* host */, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Batch of (strike x maturity) surfaces: strike pass then maturity pass, each pass the
 * 1-D operator above (this repository's documented composition).
 *
 *   K      strikes.  k_off == NULL: surface b at K + b*k_stride (k_stride 0 = one shared grid), nK each.
 *                    k_off != NULL: ragged, surface b has k_off[b+1]-k_off[b] strikes at K + k_off[b];
 *                                   nK is then the maximum count in the batch and k_stride the TOTAL number of strikes
 *                                   K holds (sigma: nT times that): a surface whose span is negative, exceeds nK or
 *                                   leaves [0, k_stride] gets IVS_ST_BAD_SHAPE and is skipped -- no host-side validation
 *                                   of the offsets is needed (0 = total unknown: only the span is checked).
 *   T      maturities of surface b at T + b*t_stride (0 = shared), nT each (nT <= 32)
 *   sigma  surface b at sigma + nT*nK*b (uniform) or sigma + nT*k_off[b] (ragged); [nT][nK_b] row-major
 *   Kq/Tq  query grids of surface b at Kq + b*kq_stride / Tq + b*tq_stride (0 = shared), mK / mT points
 *   out    [B][mT][mK]
 *   status [B] or NULL; IVS_ST_* OR-ed over every 1-D solve of the surface
 *   NaN in sigma = missing quote: the row keeps its own knot set.  64 x 16 batches are probed by the call itself (one row of
 *          64 surfaces spread over the batch): when half of the sampled rows lack a quote -- or at least 5 of them lack about
 *          one quote each: sparse independent gaps, which hit most SURFACES -- every surface goes to the compaction kernel
 *          directly, otherwise the fast kernel tags the few that do and a second pass redoes them -- no flag, same results
 *   flags  0, or IVS_FLAG_FORCE_GENERIC to bypass the dense fast path (testing / A-B timing); bits 8..15 =
 *          IVS_FLAG_MAP_GROUPS(n): tuning override of the surface -> workgroup mapping of the 64x16 kernel (0 = default)
 *   workspace  ivs_surface_workspace_bytes(B, ragged) bytes of device scratch (ragged = k_off != NULL): the
 *          batch-wide maturity tables (read by the kernels through the scalar cache), the work-queue heads and redo
 *          flags of the persistent kernels (zeroed by the call itself, on the caller's stream) and, for ragged batches,
 *          the per-size-class work lists.  Contents are undefined after the call; calls that may overlap in time (other
 *          streams, other threads) need distinct workspaces, calls on one stream may share one.
 */
enum { IVS_FLAG_FORCE_GENERIC = 1, IVS_FLAG_ONE_PASS = 2 /* testing / A-B timing: skip the row-pass kernels (one-pass dense kernels instead) */ };
#define IVS_FLAG_MAP_GROUPS(n) (((n) & 0xff) << 8)
size_t ivs_surface_workspace_bytes(int64_t B, int32_t ragged);
int ivs_surface_batch_f64(const double* K, const int64_t* k_off, int64_t k_stride, int32_t nK,
                          const double* T, int64_t t_stride, int32_t nT,
                          const double* sigma, int64_t B,
                          const double* Kq, int64_t kq_stride, int32_t mK,
                          const double* Tq, int64_t tq_stride, int32_t mT,
                          double* out, int32_t* status, int32_t method, int32_t flags,
                          void* workspace, size_t workspace_bytes, void* stream);

/*
 * Black-Scholes Greeks epilogue (reference src/interpolation/greeks.py:12-43, BlackScholesGreeks.calculate_greeks):
 * elementwise over n options.  is_put [n] (1 = put) or NULL -> every option uses default_is_put.
 * theta is per day (/365), vega and rho per 1 % (/100), put rho without sign flip -- all as the reference.
 */
int ivs_bs_greeks_f64(const double* S, const double* K, const double* T, const double* r, const double* sigma,
                      const uint8_t* is_put, int32_t default_is_put, int64_t n, double* delta, double* gamma,
                      double* theta, double* vega, double* rho, void* stream);

/*
 * N-minute candle aggregation (reference src/candle_reconstruction/core.py:68-88) for S symbols (CSR series_off [S+1],
 * rows sorted by timestamp within a symbol).  For every input row i: if it is the first row of its
 * floor(ts / freq_ns) bucket, out_*[i] holds the bucket's candle (open = first non-NaN, high = max, low = min,
 * close = last non-NaN, volume = Kahan sum in row order, like pandas) and out_count[i] its row count; otherwise
 * out_count[i] = 0.  The caller keeps the rows with out_count >= N (incomplete groups are dropped, core.py:86-88).
 */
int ivs_candle_aggregate_f64(const int64_t* ts_ns, const double* open, const double* high, const double* low,
                             const double* close, const double* volume, const int64_t* series_off, int64_t n_series,
                             int64_t n_rows, int64_t freq_ns, int64_t* out_ts, double* out_open, double* out_high,
                             double* out_low, double* out_close, double* out_volume, int32_t* out_count, void* stream);

/*
 * IV -> OHLCV bridge (reference src/data_bridge/ohlcv_converter.py:138-369, InterpolatedToOHLCVConverter.
 * _generate_ohlcv_from_interpolated and its four candle builders).  The reference draws from NumPy's process-global
 * legacy generator; its numbers are reproduced by consuming the SAME MT19937 stream in the same order.
 *
 * ivs_mt19937_words_u32: the first n_words raw 32-bit outputs of np.random.seed(seed) into words (device).
 *
 * ivs_bridge_candles_f64: S symbols (CSR row_off [S+1]) over total_rows interpolated rows, processed in row order like
 * the reference's sequential loops.  price [total_rows] = the selected price column, volume [total_rows] or NULL
 * (column absent).  strategy: 0 spread_simulation (:209-263; base_spread_pct / vol_factor = its spread_parameters),
 * 1 price_as_midpoint (:265-290), 2 trend_following (:292-332), 3 simple_spread (:334-357, also the reference's
 * fallback for unknown names), 4 the inline variant of CompleteOptimizedPipeline._generate_ohlcv_candles (reference
 * complete_pipeline.py:473-510; price = its per-row `underlying or mark or index` choice, made by the caller).  words [n_words]: the stream positioned at this call's first draw.
 * out [6][total_rows]: open, high, low, close (Python round(x, 4)), volume (round(x, 6)), source_price; rows the
 * reference skips (price NaN or <= 0, :156-157) get valid = 0 and NaN.  rng_tail (device int64[4]):
 * [0] out: words consumed; [1],[2] in/out: the legacy generator's cached normal deviate (has_gauss, gauss bits; pass
 * zeros after a fresh seed); [3] out: 1 if n_words was too small (results invalid: regenerate more words and retry).
 * workspace: ivs_bridge_workspace_bytes(total_rows) bytes of device scratch.
 */
size_t  ivs_bridge_workspace_bytes(int64_t total_rows);
int     ivs_mt19937_words_u32(uint32_t seed, uint32_t* words, int64_t n_words, void* stream);
int     ivs_bridge_candles_f64(const double* price, const double* volume, const int64_t* row_off, int64_t S,
                               int64_t total_rows, int32_t strategy, double base_spread_pct, double vol_factor,
                               const uint32_t* words, int64_t n_words, double* out, uint8_t* valid, int64_t* rng_tail,
                               void* workspace, size_t workspace_bytes, void* stream);

/* name of the kernel the last ivs_surface_batch_f64 call on this thread dispatched to (host string) */
const char* ivs_last_kernel(void);

/*
 * Diagnostics (not part of the drop-in surface).  ivs_debug_stamps(buf, n): while `buf` (device,
 * n uint64, n >= 8 * 8 * CU count) is set, ivs_surface_batch_f64 runs the STAMPED build of the dense
 * kernel, which sums s_memtime deltas per phase and per workgroup into buf[wg*8 + phase]
 * (phase 7 = surfaces processed).  Pass NULL to return to the production kernel.  Returns the
 * number of slots per workgroup.  Stamped runs are for phase SHARES only, never for timing claims.
 */
int     ivs_debug_stamps(void* device_buf, int64_t n_u64);
int64_t ivs_debug_last_grid(void);   /* workgroups of the last dense launch */
/* byte offset, in the workspace a 64 x 16 surface call was given, of the int32 'missing quotes first' flag that call's
 * probe left there (1 = the compaction kernel took every surface); tests read it back after the call */
int64_t ivs_debug_mode_offset(void);

#ifdef __cplusplus
}
#endif
#endif /* IVS_H */
