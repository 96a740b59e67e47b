// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: N back-to-back MFMAs on 1, 2, 4 independent accumulators, one wave per SIMD
// and three waves per SIMD; cycles from s_memtime.  (Run on the GPU box.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[64];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 64; ++i) avg += h[i]; avg /= 64;
    printf("accumulators %d, %d wave(s)/SIMD: %.1f cycles per MFMA per wave  (%.1f per SIMD)\n", NACC, waves_per_simd,
           avg / (16.0 * iters), avg / (16.0 * iters) / waves_per_simd);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<4>(2); run<4>(3); run<1>(3);
    return 0;
}
