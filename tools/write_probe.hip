// Write-side ceilings: bytes per lane per store instruction (4 / 8 / 16), plain vs non-temporal, persistent vs wide
// grid; and a copy whose stores are 4 B/lane.  (MI355X_MICROARCH.md quotes 6.0-6.2 TB/s for dword-per-lane stores.)
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T, bool NT> __global__ __launch_bounds__(256) void fill(T* b, long n, T v) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if constexpr (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
// each wave writes 512-B rows (like the surface kernel's output rows): lane -> 8 B, rows strided by `row_bytes`
template <int WIDTH> __global__ __launch_bounds__(64) void fill_rows(double* b, long rows_total, int rows_per_wg) {
    const int lane = threadIdx.x;
    for (long r0 = (long)blockIdx.x * rows_per_wg; r0 < rows_total; r0 += (long)gridDim.x * rows_per_wg)
        for (int r = 0; r < rows_per_wg; ++r) {
            double* row = b + (r0 + r) * 64;
            if (WIDTH == 8) row[lane] = 1.0;
            else { float* f = reinterpret_cast<float*>(row); f[lane] = 1.f; f[64 + lane] = 2.f; }
        }
}
template <int SW> __global__ __launch_bounds__(256) void copy(const float4* a, float* b, long n4) {   // 16-B loads, SW-byte stores
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = a[i];
        if (SW == 16) reinterpret_cast<float4*>(b)[i] = v;
        else {   // four dword stores, each wave-instruction covering 256 contiguous bytes
            const long w = (long)blockIdx.x * 256 + (threadIdx.x & ~63);      // first lane's element of this wave in this step
            const long base = (i - (threadIdx.x & 63)) * 4;                    // float index of the wave's 1-KB chunk
            (void)w;
            const int l = threadIdx.x & 63;
            // lane l holds floats 4l..4l+3 of the chunk; write them so that instruction k stores floats [64k, 64k+64)
            // -> needs a transpose; cheap trick: store element k of lane l at base + 4l + k (strided dword stores) instead
            b[base + 4 * l + 0] = v.x; b[base + 4 * l + 1] = v.y; b[base + 4 * l + 2] = v.z; b[base + 4 * l + 3] = v.w;
        }
    }
}
int main() {
    const long bytes = 8L << 30; char *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto&& launch, const char* nm, double moved) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
        printf("%-44s %.3f ms  %.0f GB/s\n", nm, best, moved / best / 1e6);
    };
    for (int grid : {2048, 16384, 262144}) {
        printf("grid %d\n", grid);
        time([&] { fill<float, false><<<grid, 256>>>((float*)b, bytes / 4, 1.f); }, "  fill  4 B/lane", bytes);
        time([&] { fill<float2, false><<<grid, 256>>>((float2*)b, bytes / 8, float2{1.f, 2.f}); }, "  fill  8 B/lane", bytes);
        time([&] { fill<float4, false><<<grid, 256>>>((float4*)b, bytes / 16, float4{1.f, 2.f, 3.f, 4.f}); }, "  fill 16 B/lane", bytes);
        time([&] { fill<float, true><<<grid, 256>>>((float*)b, bytes / 4, 1.f); }, "  fill  4 B/lane nontemporal", bytes);
        time([&] { fill<double, true><<<grid, 256>>>((double*)b, bytes / 8, 1.0); }, "  fill  8 B/lane nontemporal", bytes);
        time([&] { copy<16><<<grid, 256>>>((const float4*)a, (float*)b, bytes / 16); }, "  copy 16-B loads, 16-B stores", 2.0 * bytes);
        time([&] { copy<4><<<grid, 256>>>((const float4*)a, (float*)b, bytes / 16); }, "  copy 16-B loads, 4 x 4-B strided stores", 2.0 * bytes);
    }
    for (int rows : {16, 64}) {
        time([&] { fill_rows<8><<<2048, 64>>>((double*)b, bytes / 512, rows); }, rows == 16 ? "rows of 512 B, 8-B stores, 16 rows/step" : "rows of 512 B, 8-B stores, 64 rows/step", bytes);
        time([&] { fill_rows<4><<<2048, 64>>>((double*)b, bytes / 512, rows); }, rows == 16 ? "rows of 512 B, 2 x 4-B stores, 16 rows/step" : "rows of 512 B, 2 x 4-B stores, 64 rows/step", bytes);
    }
    return 0;
}
