// Diagnostic library for tools/ab_bench.py: ONLY the row-pass kernel for uniform 64x16 batches (cubic).  Fast to build.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "../iv_interpolation_amd/csrc/ivs_surface_pass.hpp"
#ifdef IVS_PASS_ENDSTAMP
extern "C" int ivs_diag_ends(unsigned long long* host, int n_wg) {      // copies {start, end} of the last launch's workgroups
    static unsigned long long* buf = nullptr;
    if (!buf) {
        if (hipMalloc(&buf, 2 * 8 * 65536) != hipSuccess) return -1;
        hipMemset(buf, 0, 2 * 8 * 65536);
        hipMemcpyToSymbol(HIP_SYMBOL(ivs::d_pass_ends), &buf, sizeof(buf));
        return 1;                                                         // first call: armed
    }
    return hipMemcpy(host, buf, (size_t)n_wg * 16, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
extern "C" {
int ivs_version(void) { return 2; }
size_t ivs_surface_workspace_bytes(int64_t B, int32_t ragged) { return ivs::surface_ws_bytes(B, ragged != 0); }
int ivs_surface_batch_f64(const double* K, const int64_t* k_off, int64_t k_stride, int32_t nK, const double* T, int64_t t_stride,
                          int32_t nT, const double* sigma, int64_t B, const double* Kq, int64_t kq_stride, int32_t mK,
                          const double* Tq, int64_t tq_stride, int32_t mT, double* out, int32_t* status, int32_t method,
                          int32_t flags, void* workspace, size_t workspace_bytes, void* stream) {
    ivs::SurfaceParams p;
    p.K = K; p.k_off = k_off; p.k_stride = k_off ? 0 : k_stride; p.k_total = k_off ? k_stride : 0; p.nK = nK; p.T = T; p.t_stride = t_stride; p.nT = nT;
    p.sigma = sigma; p.B = B; p.map_groups = 1; p.tqs = nullptr; p.redo = nullptr; p.queue = nullptr; p.mode = nullptr; p.Kq = Kq; p.kq_stride = kq_stride; p.mK = mK;
    p.Tq = Tq; p.tq_stride = tq_stride; p.mT = mT; p.out = out; p.status = status; p.method = method;
    ivs::LaunchCtx cx;
    cx.st = static_cast<hipStream_t>(stream); cx.ws = static_cast<unsigned char*>(workspace); cx.ws_bytes = workspace_bytes;
    cx.map_groups = (flags >> 8) & 0xff;
    if (const char* e = getenv("IVS_PASS_CUS")) cx.num_cu = atoi(e);      // diagnostic: grid = this x 12 workgroups
    const char* name = nullptr;
    return ivs::launch_surface_pass(p, cx, &name) == 1 ? 0 : -5;
}
}
