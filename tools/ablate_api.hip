// Diagnostic library for tools/ab_bench.py: ONLY the 64x16 dense kernel (cubic / linear, shared T, mT <= 16), built with
// -DIVS_ABLATE=n to leave one phase out (see ivs_surface_dense.hpp).  ABI-2 entry points, no validation.  Never shipped.
#include <hip/hip_runtime.h>
#include "../iv_interpolation_amd/csrc/ivs_surface_dense.hpp"
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t grid = 2048;
    p.map_groups = ivs::dense_map_groups(grid, B, (flags >> 8) & 0xff);
    ivs::TqShared* tq = reinterpret_cast<ivs::TqShared*>(workspace);
    ivs::launch_tq_tables<false>(p, tq, st);
    p.tqs = tq; p.queue = tq->queue;
    const size_t lds = ivs::dense_lds_bytes(mT);
    if (method == IVS_CUBIC) hipLaunchKernelGGL((ivs::surface_dense_kernel<IVS_CUBIC, true, true, false>), dim3((unsigned)grid), dim3(64), lds, st, p, nullptr);
    else hipLaunchKernelGGL((ivs::surface_dense_kernel<IVS_LINEAR, true, true, false>), dim3((unsigned)grid), dim3(64), lds, st, p, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
}
