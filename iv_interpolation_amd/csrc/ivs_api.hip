// extern "C" entry points of libivs.so (declared in include/ivs.h).
// Argument validation happens here on the host; kernels assume validated shapes.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/ivs.h"
#include "ivs_bridge.hpp"
#include "ivs_candles.hpp"
#include "ivs_interp1d.hpp"
#include "ivs_frame.hpp"
#include "ivs_surface_dense.hpp"
#include "ivs_surface_dense_var2.hpp"
#include "ivs_surface_pass.hpp"
#include "ivs_surface_generic.hpp"

namespace {

thread_local char g_err[512] = "";
thread_local const char* g_last_kernel = "";
// diagnostics (ivs_debug_stamps): per calling thread, like the error string
thread_local unsigned long long* g_stamp_buf = nullptr;   // device buffer for the diagnostic (stamped) dense kernel
thread_local int64_t g_stamp_cap = 0;
thread_local int64_t g_last_grid = 0;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(IVS_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return IVS_OK;
}

bool valid_method(int m) { return m >= IVS_LINEAR && m <= IVS_BFILL; }

// CU count of the CURRENT device, cached per device index (a process may drive several devices)
std::atomic<int> g_cu[ivs::IVS_MAX_DEV];
void current_device(int& dev, int& cus) {
    dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const bool cached = dev >= 0 && dev < ivs::IVS_MAX_DEV;
    cus = cached ? g_cu[dev].load(std::memory_order_relaxed) : 0;
    if (cus <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); v = 0; }
        cus = v > 0 ? v : 256;
        if (cached) g_cu[dev].store(cus, std::memory_order_relaxed);
    }
}
int num_cu() { int d, c; current_device(d, c); return c; }

}  // namespace

extern "C" {

int ivs_version(void) { return IVS_ABI_VERSION; }

const char* ivs_last_error(void) { return g_err; }

const char* ivs_last_kernel(void) { return g_last_kernel; }

int ivs_debug_stamps(void* device_buf, int64_t n_u64) {
    g_stamp_buf = static_cast<unsigned long long*>(device_buf);
    g_stamp_cap = device_buf ? n_u64 : 0;
    return (int)ivs::D_NSTAMP;
}

int64_t ivs_debug_last_grid(void) { return g_last_grid; }
int64_t ivs_debug_mode_offset(void) { return (int64_t)offsetof(ivs::TqShared, mode); }

int ivs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

size_t ivs_interp1d_workspace_bytes(int64_t total_knots, int64_t n_series, int32_t n_channels) {
    if (total_knots < 0 || n_series < 0 || n_channels < 0) return 0;
    return ivs::interp1d_ws_bytes(total_knots, n_series, n_channels);
}

namespace {
int interp1d_impl(const char* fn, const double* xk, const double* yk, int64_t yk_stride, const int64_t* knot_off,
                  int64_t n_series, int32_t n_channels, int64_t total_knots, const double* xq, const int64_t* q_off,
                  int64_t total_queries, double* out, int64_t out_stride, int32_t* status, int32_t method, void* workspace,
                  size_t workspace_bytes, void* stream, ivs::Interp1dParams& p) {
    g_err[0] = 0;
    if (!valid_method(method)) return fail(IVS_EINVAL, "%s: unknown method %d", fn, method);
    if (n_series < 0 || n_channels < 0 || total_knots < 0 || total_queries < 0) return fail(IVS_EINVAL, "%s: negative size", fn);
    if (n_series == 0 || n_channels == 0) return IVS_OK;
    if (!knot_off || !q_off || !status) return fail(IVS_EINVAL, "%s: null offsets/status", fn);
    if (total_knots > 0 && (!xk || !yk)) return fail(IVS_EINVAL, "%s: null knots", fn);
    if (total_queries > 0 && !out) return fail(IVS_EINVAL, "%s: null out", fn);
    if (yk_stride < total_knots || out_stride < total_queries)
        return fail(IVS_EINVAL, "%s: channel stride smaller than row length", fn);
    if (n_series * (int64_t)n_channels > 0x7fffffffLL || (total_queries + 255) / 256 > 0x7fffffffLL)
        return fail(IVS_ERANGE, "%s: batch too large for one launch", fn);
    size_t need = ivs::interp1d_ws_bytes(total_knots, n_series, n_channels);
    if (!workspace || workspace_bytes < need) return fail(IVS_ENOMEM, "%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
    p.xk = xk; p.yk = yk; p.yk_stride = yk_stride; p.knot_off = knot_off;
    p.S = n_series; p.C = n_channels; p.total_knots = total_knots;
    p.xq = xq; p.q_off = q_off; p.total_q = total_queries;
    p.out = out; p.out_stride = out_stride; p.status = status; p.method = method;
    double* w = static_cast<double*>(workspace);
    size_t plane = (size_t)n_channels * (size_t)total_knots;
    p.wx = w; p.wy = w + plane; p.ws = w + 2 * plane; p.wcp = w + 3 * plane;
    p.wn = reinterpret_cast<int32_t*>(w + 4 * plane);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ivs::interp1d_prepare_kernel, dim3((unsigned)(n_series * n_channels)), dim3(256), 0, st, p);
    int rc = check_launch("interp1d_prepare_kernel");
    if (rc) return rc;
    if (total_queries > 0) {
        hipLaunchKernelGGL(ivs::interp1d_eval_kernel, dim3((unsigned)((total_queries + ivs::E1_ROWS - 1) / ivs::E1_ROWS)), dim3(256), 0, st, p);
        rc = check_launch("interp1d_eval_kernel");
    }
    return rc;
}
}  // namespace

int ivs_interp1d_batch_f64(const double* xk, const double* yk, int64_t yk_stride, const int64_t* knot_off,
                           int64_t n_series, int32_t n_channels, int64_t total_knots,
                           const double* xq, const int64_t* q_off, int64_t total_queries,
                           double* out, int64_t out_stride, int32_t* status, int32_t method,
                           void* workspace, size_t workspace_bytes, void* stream) {
    ivs::Interp1dParams p{};
    p.greeks = nullptr;
    return interp1d_impl("ivs_interp1d_batch_f64", xk, yk, yk_stride, knot_off, n_series, n_channels, total_knots, xq, q_off,
                         total_queries, out, out_stride, status, method, workspace, workspace_bytes, stream, p);
}

int ivs_interp1d_greeks_batch_f64(const double* xk, const double* yk, int64_t yk_stride, const int64_t* knot_off,
                                  int64_t n_series, int32_t n_channels, int64_t total_knots,
                                  const double* xq, const int64_t* q_off, int64_t total_queries,
                                  double* out, int64_t out_stride, int32_t* status, int32_t method,
                                  int32_t ch_iv, int32_t ch_underlying, int32_t ch_ttm,
                                  const int32_t* fill_idx, int64_t fill_stride, int32_t row_strike, int32_t row_rate,
                                  int32_t row_callput, const double* strike_src, const double* rate_src,
                                  const uint8_t* is_put_src, double* greeks, int64_t greeks_stride,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "ivs_interp1d_greeks_batch_f64";
    g_err[0] = 0;
    if (n_series > 0 && n_channels > 0 && total_queries > 0) {
        if (!greeks || greeks_stride < total_queries) return fail(IVS_EINVAL, "%s: null / short greeks output", fn);
        if (ch_iv < 0 || ch_iv >= n_channels || ch_underlying < 0 || ch_underlying >= n_channels || ch_ttm < 0 || ch_ttm >= n_channels)
            return fail(IVS_EINVAL, "%s: channel numbers outside [0, %d)", fn, n_channels);
        if ((row_strike >= 0 || row_rate >= 0 || row_callput >= 0) && (!fill_idx || fill_stride < total_queries))
            return fail(IVS_EINVAL, "%s: null / short fill index", fn);
        if ((row_strike >= 0 && !strike_src) || (row_rate >= 0 && !rate_src) || (row_callput >= 0 && !is_put_src))
            return fail(IVS_EINVAL, "%s: a present column needs its source array", fn);
    }
    ivs::Interp1dParams p{};
    p.fidx = fill_idx; p.fidx_stride = fill_stride; p.fi_strike = row_strike; p.fi_rate = row_rate; p.fi_put = row_callput;
    p.strike_src = strike_src; p.rate_src = rate_src; p.put_src = is_put_src;
    p.ch_iv = ch_iv; p.ch_S = ch_underlying; p.ch_T = ch_ttm;
    p.greeks = greeks; p.greeks_stride = greeks_stride;
    return interp1d_impl(fn, xk, yk, yk_stride, knot_off, n_series, n_channels, total_knots, xq, q_off, total_queries, out,
                         out_stride, status, method, workspace, workspace_bytes, stream, p);
}

int ivs_ffill_index_batch(const int64_t* src_pos, const int64_t* src_off, const uint8_t* valid, int64_t valid_stride,
                          int32_t n_cols, const int64_t* q_off, int64_t n_series, int64_t total_queries,
                          int32_t* idx_out, int64_t out_stride, void* stream) {
    g_err[0] = 0;
    if (n_cols < 0 || n_series < 0 || total_queries < 0) return fail(IVS_EINVAL, "ivs_ffill_index_batch: negative size");
    if (n_cols == 0 || n_series == 0 || total_queries == 0) return IVS_OK;
    if (!src_pos || !src_off || !valid || !q_off || !idx_out) return fail(IVS_EINVAL, "ivs_ffill_index_batch: null pointer");
    if (out_stride < total_queries) return fail(IVS_EINVAL, "ivs_ffill_index_batch: out_stride < total_queries");
    if ((total_queries + 255) / 256 > 0x7fffffffLL) return fail(IVS_ERANGE, "ivs_ffill_index_batch: too many rows");
    if (valid_stride > 0x7fffffffLL) return fail(IVS_ERANGE, "ivs_ffill_index_batch: %lld source rows exceed the int32 gather index", (long long)valid_stride);
    ivs::FfillParams p{src_pos, src_off, valid, valid_stride, n_cols, q_off, n_series, total_queries, idx_out, out_stride};
    hipLaunchKernelGGL(ivs::ffill_index_kernel, dim3((unsigned)((total_queries + ivs::F1_ROWS - 1) / ivs::F1_ROWS)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
    return check_launch("ffill_index_kernel");
}

extern "C++" {
namespace {
template <class T>
int gather_impl(const char* fn, const void* src, int64_t src_stride, const int32_t* idx, int64_t idx_stride, const int32_t* idx_row,
                int32_t n_cols, int64_t n, void* out, int64_t out_stride, T missing, void* stream) {
    g_err[0] = 0;
    if (n_cols < 0 || n < 0) return fail(IVS_EINVAL, "%s: negative size", fn);
    if (n_cols == 0 || n == 0) return IVS_OK;
    if (!src || !idx || !idx_row || !out) return fail(IVS_EINVAL, "%s: null pointer", fn);
    if (idx_stride < n || out_stride < n) return fail(IVS_EINVAL, "%s: stride smaller than row count", fn);
    if (src_stride > 0x7fffffffLL) return fail(IVS_ERANGE, "%s: %lld source rows exceed the int32 gather index", fn, (long long)src_stride);
    ivs::GatherParams p{src, src_stride, idx, idx_stride, idx_row, n_cols, n, out, out_stride};
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)num_cu() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ivs::gather_rows_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, missing);
    return check_launch(fn);
}
}  // namespace
}  // extern "C++"

int ivs_gather_rows_f64(const double* src, int64_t src_stride, const int32_t* idx, int64_t idx_stride, const int32_t* idx_row,
                        int32_t n_cols, int64_t n, double* out, int64_t out_stride, void* stream) {
    return gather_impl<double>("ivs_gather_rows_f64", src, src_stride, idx, idx_stride, idx_row, n_cols, n, out, out_stride,
                               __builtin_nan(""), stream);
}

int ivs_gather_rows_i32(const int32_t* src, int64_t src_stride, const int32_t* idx, int64_t idx_stride, const int32_t* idx_row,
                        int32_t n_cols, int64_t n, int32_t* out, int64_t out_stride, void* stream) {
    return gather_impl<int32_t>("ivs_gather_rows_i32", src, src_stride, idx, idx_stride, idx_row, n_cols, n, out, out_stride,
                                (int32_t)-1, stream);
}

int ivs_frame_rows(const int64_t* q_off, int64_t n_series, int64_t total_queries, const int64_t* first_ns,
                   const double* chan, int64_t chan_stride, int32_t n_channels, const int32_t* sym_code,
                   const int32_t* status, const uint8_t* needs, int64_t* date_ns, uint8_t* keep, void* stream) {
    g_err[0] = 0;
    if (n_series < 0 || total_queries < 0 || n_channels < 0) return fail(IVS_EINVAL, "ivs_frame_rows: negative size");
    if (n_series == 0 || total_queries == 0) return IVS_OK;
    if (!q_off || !first_ns || !date_ns || !keep || (n_channels > 0 && (!chan || !status || !needs)))
        return fail(IVS_EINVAL, "ivs_frame_rows: null pointer");
    if (chan_stride < total_queries) return fail(IVS_EINVAL, "ivs_frame_rows: chan_stride < total_queries");
    ivs::FrameRowsParams p{q_off, n_series, total_queries, first_ns, chan, chan_stride, n_channels, sym_code, status, needs, date_ns, keep};
    int64_t blocks = (total_queries + 255) / 256;
    const int64_t cap = (int64_t)num_cu() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ivs::frame_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("frame_rows_kernel");
}

namespace {
__global__ __launch_bounds__(256) void i64_to_f64_kernel(const int64_t* a, double* o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) o[i] = (double)a[i];
}
}  // namespace

size_t ivs_frame_workspace_bytes(int64_t total_src, int64_t n_series, int32_t n_channels) {
    if (total_src < 0 || n_series < 0 || n_channels < 0) return 0;
    // the 1-D workspace, the positions as doubles, the channels' knot-rank tables
    return ((ivs::interp1d_ws_bytes(total_src, n_series, n_channels) + 63) & ~(size_t)63) + (size_t)total_src * 8 +
           (size_t)n_channels * (size_t)total_src * 4 + 64;
}

int ivs_frame_columns_f64(const ivs_frame_args* a, void* workspace, size_t workspace_bytes, void* stream) {
    const char* fn = "ivs_frame_columns_f64";
    g_err[0] = 0;
    if (!a) return fail(IVS_EINVAL, "%s: null args", fn);
    if (!valid_method(a->method)) return fail(IVS_EINVAL, "%s: unknown method %d", fn, a->method);
    if (a->n_series < 0 || a->total_src < 0 || a->total_queries < 0 || a->n_channels < 0 || a->n_valid < 0 || a->n_f < 0 ||
        a->n_c < 0 || a->n_idx < 0)
        return fail(IVS_EINVAL, "%s: negative size", fn);
    if (a->n_series == 0 || a->total_queries == 0) return IVS_OK;
    if (!a->src_pos || !a->src_off || !a->q_off) return fail(IVS_EINVAL, "%s: null offsets / positions", fn);
    if (a->n_channels > 0 && (!a->yk || !a->chan_out || !a->status)) return fail(IVS_EINVAL, "%s: null channel arrays", fn);
    if (a->n_channels > 0 && (a->yk_stride < a->total_src || a->chan_stride < a->total_queries))
        return fail(IVS_EINVAL, "%s: channel stride smaller than row length", fn);
    if (a->n_valid > 0 && (!a->valid || a->valid_stride < a->total_src)) return fail(IVS_EINVAL, "%s: null / short validity rows", fn);
    if (a->n_f > 0 && (!a->fsrc || !a->f_rows || !a->f_out || a->f_stride < a->total_queries || a->fsrc_stride < a->total_src))
        return fail(IVS_EINVAL, "%s: bad f64 column arguments", fn);
    if (a->n_c > 0 && (!a->csrc || !a->c_rows || !a->c_out || a->c_stride < a->total_queries || a->csrc_stride < a->total_src))
        return fail(IVS_EINVAL, "%s: bad code column arguments", fn);
    if (a->n_idx > 0 && (!a->idx_rows || !a->idx_out || a->idx_stride < a->total_queries)) return fail(IVS_EINVAL, "%s: bad index row arguments", fn);
    if (a->date_ns && (!a->keep || !a->first_ns)) return fail(IVS_EINVAL, "%s: date_ns needs keep and first_ns", fn);
    if (a->sym_col >= a->n_c) return fail(IVS_EINVAL, "%s: sym_col outside the code columns", fn);
    if (a->greeks) {
        if (a->greeks_stride < a->total_queries) return fail(IVS_EINVAL, "%s: short greeks output", fn);
        if (a->ch_iv < 0 || a->ch_iv >= a->n_channels || a->ch_underlying < 0 || a->ch_underlying >= a->n_channels || a->ch_ttm < 0 ||
            a->ch_ttm >= a->n_channels || a->n_channels > 3)
            return fail(IVS_EINVAL, "%s: Greeks need the three channels iv / underlying / ttm", fn);
        if (a->g_strike >= a->n_valid || a->g_rate >= a->n_valid || a->g_put >= a->n_valid) return fail(IVS_EINVAL, "%s: Greek validity row outside valid", fn);
        if ((a->g_strike >= 0 && !a->strike_src) || (a->g_rate >= 0 && !a->rate_src) || (a->g_put >= 0 && !a->put_src))
            return fail(IVS_EINVAL, "%s: a present column needs its source array", fn);
    }
    if (a->total_src > 0x7fffffffLL || a->n_series * (int64_t)(a->n_channels > 0 ? a->n_channels : 1) > 0x7fffffffLL ||
        (a->total_queries + ivs::FR_ROWS - 1) / ivs::FR_ROWS > 0x7fffffffLL)
        return fail(IVS_ERANGE, "%s: batch too large for one launch", fn);
    const size_t need = ivs_frame_workspace_bytes(a->total_src, a->n_series, a->n_channels);
    if (!workspace || workspace_bytes < need) return fail(IVS_ENOMEM, "%s: workspace %zu < %zu bytes", fn, workspace ? workspace_bytes : (size_t)0, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    ivs::Interp1dParams p{};
    double* w = static_cast<double*>(workspace);
    const size_t plane = (size_t)a->n_channels * (size_t)a->total_src;
    p.wx = w; p.wy = w + plane; p.ws = w + 2 * plane; p.wcp = w + 3 * plane;
    p.wn = reinterpret_cast<int32_t*>(w + 4 * plane);
    double* xk = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(workspace) + ((ivs::interp1d_ws_bytes(a->total_src, a->n_series, a->n_channels) + 63) & ~(size_t)63));
    p.wr = reinterpret_cast<int32_t*>(xk + a->total_src);
    p.xk = xk; p.yk = a->yk; p.yk_stride = a->yk_stride; p.knot_off = a->src_off;
    p.S = a->n_series; p.C = a->n_channels; p.total_knots = a->total_src;
    p.xq = nullptr; p.q_off = a->q_off; p.total_q = a->total_queries;
    p.out = a->chan_out; p.out_stride = a->chan_stride; p.status = a->status; p.method = a->method;
    p.greeks = nullptr;
    if (a->total_src > 0) {
        int64_t cb = (a->total_src + 255) / 256;
        const int64_t cap = (int64_t)num_cu() * 16;
        if (cb > cap) cb = cap;
        hipLaunchKernelGGL(i64_to_f64_kernel, dim3((unsigned)cb), dim3(256), 0, st, a->src_pos, xk, a->total_src);
    }
    if (a->n_channels > 0) {
        hipLaunchKernelGGL(ivs::interp1d_prepare_kernel, dim3((unsigned)(a->n_series * a->n_channels)), dim3(256), 0, st, p);
        const int rc = check_launch("interp1d_prepare_kernel");
        if (rc) return rc;
    }
    ivs::FrameParams f{};
    f.src_pos = a->src_pos; f.src_off = a->src_off; f.q_off = a->q_off; f.S = a->n_series; f.total_src = a->total_src; f.total_q = a->total_queries;
    f.valid = a->valid; f.valid_stride = a->valid_stride; f.n_valid = a->n_valid;
    f.fsrc = a->fsrc; f.fsrc_stride = a->fsrc_stride; f.f_rows = a->f_rows; f.n_f = a->n_f; f.f_out = a->f_out; f.f_stride = a->f_stride;
    f.csrc = a->csrc; f.csrc_stride = a->csrc_stride; f.c_rows = a->c_rows; f.n_c = a->n_c; f.c_out = a->c_out; f.c_stride = a->c_stride;
    f.idx_rows = a->idx_rows; f.n_idx = a->n_idx; f.idx_out = a->idx_out; f.idx_stride = a->idx_stride;
    f.first_ns = a->first_ns; f.needs = a->needs; f.sym_col = a->sym_col; f.date_ns = a->date_ns; f.keep = a->keep;
    f.g_strike = a->g_strike; f.g_rate = a->g_rate; f.g_put = a->g_put; f.strike_src = a->strike_src; f.rate_src = a->rate_src; f.put_src = a->put_src;
    f.ch_iv = a->ch_iv; f.ch_S = a->ch_underlying; f.ch_T = a->ch_ttm; f.greeks = a->greeks; f.greeks_stride = a->greeks_stride;
    const dim3 fgrid((unsigned)((a->total_queries + ivs::FR_ROWS - 1) / ivs::FR_ROWS));
    switch (ivs::frame_method_class(p.method)) {
        case 0: hipLaunchKernelGGL(ivs::frame_fused_kernel<0>, fgrid, dim3(256), 0, st, f, p); break;
        case 1: hipLaunchKernelGGL(ivs::frame_fused_kernel<1>, fgrid, dim3(256), 0, st, f, p); break;
        default: hipLaunchKernelGGL(ivs::frame_fused_kernel<2>, fgrid, dim3(256), 0, st, f, p); break;
    }
    return check_launch("frame_fused_kernel");
}

int ivs_candle_aggregate_f64(const int64_t* ts_ns, const double* open, const double* high, const double* low,
                             const double* close, const double* volume, const int64_t* series_off, int64_t n_series,
                             int64_t n_rows, int64_t freq_ns, int64_t* out_ts, double* out_open, double* out_high,
                             double* out_low, double* out_close, double* out_volume, int32_t* out_count, void* stream) {
    g_err[0] = 0;
    if (n_rows < 0 || n_series < 0 || freq_ns <= 0) return fail(IVS_EINVAL, "ivs_candle_aggregate_f64: bad size / frequency");
    if (n_rows == 0 || n_series == 0) return IVS_OK;
    if (!ts_ns || !open || !high || !low || !close || !volume || !series_off || !out_ts || !out_open || !out_high ||
        !out_low || !out_close || !out_volume || !out_count)
        return fail(IVS_EINVAL, "ivs_candle_aggregate_f64: null pointer");
    if ((n_rows + 255) / 256 > 0x7fffffffLL) return fail(IVS_ERANGE, "ivs_candle_aggregate_f64: too many rows");
    ivs::CandleParams p{ts_ns, open, high, low, close, volume, series_off, n_series, n_rows, freq_ns,
                        out_ts, out_open, out_high, out_low, out_close, out_volume, out_count};
    hipLaunchKernelGGL(ivs::candle_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), p);
    return check_launch("candle_kernel");
}

size_t ivs_bridge_workspace_bytes(int64_t total_rows) { return ivs::bridge_ws_bytes(total_rows < 0 ? 0 : total_rows); }

int ivs_mt19937_words_u32(uint32_t seed, uint32_t* words, int64_t n_words, void* stream) {
    g_err[0] = 0;
    if (n_words < 0) return fail(IVS_EINVAL, "ivs_mt19937_words_u32: negative size");
    if (n_words == 0) return IVS_OK;
    if (!words) return fail(IVS_EINVAL, "ivs_mt19937_words_u32: null pointer");
    hipLaunchKernelGGL(ivs::mt19937_words_kernel, dim3(1), dim3(ivs::MT_THREADS), 0, static_cast<hipStream_t>(stream), seed, words, n_words);
    return check_launch("mt19937_words_kernel");
}

int ivs_bridge_candles_f64(const double* price, const double* volume, const int64_t* row_off, int64_t S,
                           int64_t total_rows, int32_t strategy, double base_spread_pct, double vol_factor,
                           const uint32_t* words, int64_t n_words, double* out, uint8_t* valid, int64_t* rng_tail,
                           void* workspace, size_t workspace_bytes, void* stream) {
    g_err[0] = 0;
    if (S < 0 || total_rows < 0 || n_words < 0) return fail(IVS_EINVAL, "ivs_bridge_candles_f64: negative size");
    if (strategy < 0 || strategy > 4) return fail(IVS_EINVAL, "ivs_bridge_candles_f64: unknown strategy %d", strategy);
    if (!rng_tail) return fail(IVS_EINVAL, "ivs_bridge_candles_f64: null rng_tail");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (S == 0 || total_rows == 0) {
        if (hipMemsetAsync(rng_tail, 0, 8, st) != hipSuccess || hipMemsetAsync(rng_tail + 3, 0, 8, st) != hipSuccess)
            return fail(IVS_ELAUNCH, "ivs_bridge_candles_f64: memset failed");
        return IVS_OK;
    }
    if (!price || !row_off || !words || !out || !valid || !workspace)
        return fail(IVS_EINVAL, "ivs_bridge_candles_f64: null pointer");
    if (workspace_bytes < ivs::bridge_ws_bytes(total_rows)) return fail(IVS_ENOMEM, "ivs_bridge_candles_f64: workspace too small");
    const int64_t nb = ivs::bridge_blocks(total_rows);
    if (nb > 0x7fffffffLL) return fail(IVS_ERANGE, "ivs_bridge_candles_f64: too many rows");
    if (n_words > 0x7fffffffLL) return fail(IVS_ERANGE, "ivs_bridge_candles_f64: %lld words exceed the int32 per-row stream offsets", (long long)n_words);
    auto align64 = [](uintptr_t a) { return (a + 63) & ~(uintptr_t)63; };
    uintptr_t w = align64(reinterpret_cast<uintptr_t>(workspace));
    ivs::BridgeParams p;
    p.price = price; p.volume = volume; p.row_off = row_off; p.S = S; p.total_rows = total_rows;
    p.strategy = strategy; p.base_spread_pct = base_spread_pct; p.vol_factor = vol_factor;
    p.words = words; p.n_words = n_words; p.out = out; p.valid = valid; p.rng_tail = rng_tail;
    p.woff = reinterpret_cast<int32_t*>(w); w = align64(w + (size_t)total_rows * 4);
    p.bbase = reinterpret_cast<int64_t*>(w); w = align64(w + (size_t)(nb + 1) * 8);
    p.gauss = reinterpret_cast<double*>(w); w = align64(w + (size_t)total_rows * 8);
    p.expu = reinterpret_cast<double*>(w); w = align64(w + (size_t)total_rows * 8);
    p.acc = reinterpret_cast<uint8_t*>(w); w = align64(w + ivs::bridge_acc_capacity(total_rows));
    p.gauss0 = reinterpret_cast<double*>(w);
    if (hipMemsetAsync(rng_tail + 3, 0, 8, st) != hipSuccess) return fail(IVS_ELAUNCH, "ivs_bridge_candles_f64: memset failed");
    if (strategy == ivs::BR_TREND) {
        int64_t nd = n_words / 2;                                   // doubles of the stream this call may look at
        const int64_t cap = (int64_t)ivs::bridge_acc_capacity(total_rows);
        if (nd > cap) { nd = cap; p.n_words = 2 * nd; }
        int64_t ab = (nd + 255) / 256;
        const int64_t abcap = (int64_t)num_cu() * 16;
        if (ab > abcap) ab = abcap;
        if (ab < 1) ab = 1;
        hipLaunchKernelGGL(ivs::bridge_accept_kernel, dim3((unsigned)ab), dim3(256), 0, st, p, nd);
        hipLaunchKernelGGL(ivs::bridge_gauss_walk_kernel, dim3(1), dim3(64), 0, st, p, nd);
    } else {
        hipLaunchKernelGGL(ivs::bridge_count_kernel, dim3((unsigned)nb), dim3(256), 0, st, p);
        hipLaunchKernelGGL(ivs::bridge_scan_blocks_kernel, dim3(1), dim3(256), 0, st, p, nb);
    }
    int64_t grid = (int64_t)num_cu() * 8;
    if (grid > S) grid = S;
    switch (strategy) {
        case ivs::BR_SPREAD: hipLaunchKernelGGL(ivs::bridge_candles_kernel<ivs::BR_SPREAD>, dim3((unsigned)grid), dim3(64), 0, st, p); break;
        case ivs::BR_MIDPOINT: hipLaunchKernelGGL(ivs::bridge_candles_kernel<ivs::BR_MIDPOINT>, dim3((unsigned)grid), dim3(64), 0, st, p); break;
        case ivs::BR_TREND: hipLaunchKernelGGL(ivs::bridge_candles_kernel<ivs::BR_TREND>, dim3((unsigned)grid), dim3(64), 0, st, p); break;
        case ivs::BR_PIPELINE: hipLaunchKernelGGL(ivs::bridge_candles_kernel<ivs::BR_PIPELINE>, dim3((unsigned)grid), dim3(64), 0, st, p); break;
        default: hipLaunchKernelGGL(ivs::bridge_candles_kernel<ivs::BR_SIMPLE>, dim3((unsigned)grid), dim3(64), 0, st, p); break;
    }
    return check_launch("bridge_candles_kernel");
}

int ivs_bs_greeks_f64(const double* S, const double* K, const double* T, const double* r, const double* sigma,
                      const uint8_t* is_put, int32_t default_is_put, int64_t n, double* delta, double* gamma,
                      double* theta, double* vega, double* rho, void* stream) {
    g_err[0] = 0;
    if (n < 0) return fail(IVS_EINVAL, "ivs_bs_greeks_f64: negative size");
    if (n == 0) return IVS_OK;
    if (!S || !K || !T || !r || !sigma || !delta || !gamma || !theta || !vega || !rho)
        return fail(IVS_EINVAL, "ivs_bs_greeks_f64: null pointer");
    ivs::GreeksParams p{S, K, T, r, sigma, is_put, default_is_put, n, delta, gamma, theta, vega, rho};
    int64_t blocks = (n + 255) / 256;
    int64_t cap = (int64_t)num_cu() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ivs::greeks_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("greeks_kernel");
}

size_t ivs_surface_workspace_bytes(int64_t B, int32_t ragged) { return ivs::surface_ws_bytes(B, ragged != 0); }

int ivs_surface_batch_f64(const double* K, const int64_t* k_off, int64_t k_stride, int32_t nK,
                          const double* T, int64_t t_stride, int32_t nT,
                          const double* sigma, int64_t B,
                          const double* Kq, int64_t kq_stride, int32_t mK,
                          const double* Tq, int64_t tq_stride, int32_t mT,
                          double* out, int32_t* status, int32_t method, int32_t flags,
                          void* workspace, size_t workspace_bytes, void* stream) {
    g_err[0] = 0;
    g_last_kernel = "";
    if (!valid_method(method)) return fail(IVS_EINVAL, "ivs_surface_batch_f64: unknown method %d", method);
    if (ivs::method_is_poly(method)) return fail(IVS_EINVAL, "ivs_surface_batch_f64: method %d ('barycentric' / 'krogh') is 1-D only", method);
    if (B < 0 || nK < 0 || nT < 0 || mK < 0 || mT < 0) return fail(IVS_EINVAL, "ivs_surface_batch_f64: negative size");
    if (B == 0 || mK == 0 || mT == 0) return IVS_OK;
    if (!K || !T || !sigma || !Kq || !Tq || !out) return fail(IVS_EINVAL, "ivs_surface_batch_f64: null pointer");
    if (nT < 1 || nT > ivs::GEN_NTMAX) return fail(IVS_ERANGE, "ivs_surface_batch_f64: nT=%d outside [1,%d]", nT, ivs::GEN_NTMAX);
    if (nK < 1) return fail(IVS_ERANGE, "ivs_surface_batch_f64: nK=%d < 1", nK);
    if (!k_off && k_stride != 0 && k_stride < nK) return fail(IVS_EINVAL, "ivs_surface_batch_f64: k_stride < nK");
    if (t_stride < 0 || kq_stride < 0 || tq_stride < 0 || k_stride < 0)
        return fail(IVS_EINVAL, "ivs_surface_batch_f64: negative stride");
    if (ivs::generic_lds_bytes(nK, nT, ivs::method_is_cubic(method)) > 160 * 1024)
        return fail(IVS_ERANGE, "ivs_surface_batch_f64: nK=%d x nT=%d needs %zu B of LDS (> 160 KiB)", nK, nT,
                    ivs::generic_lds_bytes(nK, nT, ivs::method_is_cubic(method)));
    const size_t need = ivs::surface_ws_bytes(B, k_off != nullptr);
    if (!workspace || workspace_bytes < need)
        return fail(IVS_ENOMEM, "ivs_surface_batch_f64: workspace %zu < %zu bytes (ivs_surface_workspace_bytes)",
                    workspace ? workspace_bytes : (size_t)0, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 255)
        return fail(IVS_EINVAL, "ivs_surface_batch_f64: workspace must be 256-byte aligned");

    ivs::SurfaceParams p;
    p.K = K; p.k_off = k_off; p.k_stride = k_off ? 0 : k_stride; p.nK = nK; p.k_total = k_off ? k_stride : 0;
    p.T = T; p.t_stride = t_stride; p.nT = nT;
    p.sigma = sigma; p.B = B; p.map_groups = 1; p.tqs = nullptr; p.redo = nullptr; p.queue = nullptr; p.mode = nullptr;
    p.Kq = Kq; p.kq_stride = kq_stride; p.mK = mK;
    p.Tq = Tq; p.tq_stride = tq_stride; p.mT = mT;
    p.out = out; p.status = status; p.method = method;
    ivs::LaunchCtx cx;
    current_device(cx.dev, cx.num_cu);
    cx.st = static_cast<hipStream_t>(stream);
    cx.ws = static_cast<unsigned char*>(workspace);
    cx.ws_bytes = workspace_bytes;
    cx.map_groups = (flags >> 8) & 0xff;

    if (!(flags & IVS_FLAG_FORCE_GENERIC)) {
        const char* name = nullptr;
        int64_t need_blocks = (int64_t)cx.num_cu * 8;
        unsigned long long* dbg = (g_stamp_buf && g_stamp_cap >= need_blocks * ivs::D_NSTAMP) ? g_stamp_buf : nullptr;
        int rc = (dbg || (flags & IVS_FLAG_ONE_PASS)) ? 0 : ivs::launch_surface_pass(p, cx, &name);
        if (rc == 1) {
            g_last_kernel = name;
            return check_launch(name);
        }
        if (rc < 0) return fail(IVS_ELAUNCH, "ivs_surface_batch_f64: row-pass dispatch failed");
        rc = ivs::launch_surface_dense(p, cx, &name, dbg, &g_last_grid);
        if (rc == 1) {   // dispatched
            g_last_kernel = name;
            return check_launch(name);
        }
        if (rc < 0) return fail(IVS_ELAUNCH, "ivs_surface_batch_f64: dense dispatch failed");
        if (!g_stamp_buf) {
            rc = ivs::launch_surface_dense_var(p, cx, &name);
            if (rc == 1) {
                g_last_kernel = name;
                return check_launch(name);
            }
            if (rc < 0) return fail(IVS_ELAUNCH, "ivs_surface_batch_f64: dense-var dispatch failed");
        }
    }

    if (!ivs::launch_surface_generic<false>(p, cx))
        return fail(IVS_ERANGE, "ivs_surface_batch_f64: nK=%d x nT=%d needs %zu B of LDS (> 160 KiB)", nK, nT,
                    ivs::generic_lds_bytes(nK, nT, ivs::method_is_cubic(method)));
    g_last_kernel = "surface_generic_kernel";
    return check_launch("surface_generic_kernel");
}

}  // extern "C"
