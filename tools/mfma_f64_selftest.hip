// Layout probe of v_mfma_f64_16x16x4_f64 on gfx950 (run on the GPU box).  Operands are loaded with the layout the
// kernels use (A[i][k] in lane i + 16k, B[k][j] in lane j + 16k); the probe prints which D element every (lane, register)
// holds and checks the formula the kernels rely on.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* raw) {      // A [16][4], B [4][16]; raw [64][4]
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];
    const double b = B[(l / 16) * 16 + l % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) raw[l * 4 + v] = c[v];
}
int main() {
    double hA[64], hB[64], raw[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = 1.0 + 0.37 * i + 1.91 * kk + 0.013 * i * kk;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 3.0 + 0.53 * j - 0.29 * kk * (j + 1) + 0.007 * j * j;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s = fma(hA[i * 4 + kk], hB[kk * 16 + j], s); ref[i * 16 + j] = s; }
    double *dA, *dB, *dR;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dR, sizeof raw);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dR);
    (void)hipMemcpy(raw, dR, sizeof raw, hipMemcpyDeviceToHost);
    int bad = 0, unknown = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        int fi = -1, fj = -1;
        for (int i = 0; i < 16 && fi < 0; ++i) for (int j = 0; j < 16; ++j) if (fabs(raw[l * 4 + v] - ref[i * 16 + j]) < 1e-9) { fi = i; fj = j; break; }
        if (fi < 0) ++unknown;
        if (l % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d]\n", l, v, fi, fj);
        if (fi != 4 * v + l / 16 || fj != l % 16) ++bad;
    }
    printf("mfma_f64_16x16x4 layout D[4*v + l/16][l%%16] in register v of lane l: %s (%d mismatches, %d unmatched)\n", bad ? "MISMATCH" : "OK", bad, unknown);
    return bad != 0;
}
