// Plain device copy ceilings on this box (float4 grid-stride copy, read-only sum, write-only fill) for context.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void copy4(const double2* a, double2* b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void copy4x4(const double2* a, double2* b, long n) {   // 4 loads in flight per lane
    long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3;
    }
}
__global__ __launch_bounds__(256) void readsum(const double2* a, double* out, long n) {
    double s = 0; for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void fill(double2* b, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) b[i] = double2{1.0, 2.0};
}
int main() {
    const long bytes = 8L << 30; const long n = bytes / 16; double2 *a, *b; double* o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 8); hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 65536}) for (int k = 0; k < 4; ++k) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) { hipEventRecord(e0);
            if (k == 0) copy4<<<grid, 256>>>(a, b, n); else if (k == 1) copy4x4<<<grid, 256>>>(a, b, n); else if (k == 2) readsum<<<grid, 256>>>(a, o, n); else fill<<<grid, 256>>>(b, n);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
        const char* nm[] = {"copy float4", "copy float4 x4", "read only", "write only"};
        double moved = (k < 2 ? 2.0 : 1.0) * bytes;
        printf("grid %6d %-15s %.3f ms  %.0f GB/s\n", grid, nm[k], best, moved / best / 1e6);
    }
    return 0;
}
