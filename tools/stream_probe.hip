// What HBM rate does the dense kernel's ACCESS PATTERN allow at its occupancy (8 single-wave workgroups per CU,
// 8 x 1 KiB loads + 512 B K per surface prefetched one surface ahead, 16 x 512 B row stores)?  No math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int STORE_W, bool NT = false>   // 1: dwordx2 rows of 512 B (as the kernel), 2: dwordx4 (two rows per instruction); NT: non-temporal hints
__global__ __launch_bounds__(64) void probe(const double* sig, const double* K, double* out, long B) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    double2 pre[8]; double pk;
    auto pf = [&](long b) { const double2* s2 = (const double2*)(sig + b * 1024);
        for (int i = 0; i < 8; ++i) { if (NT) { pre[i].x = __builtin_nontemporal_load(&s2[i * 64 + lane].x); pre[i].y = __builtin_nontemporal_load(&s2[i * 64 + lane].y); } else pre[i] = s2[i * 64 + lane]; } pk = K[b * 64 + lane]; };
    long b = blockIdx.x; if (b < B) pf(b);
    for (; b < B; b += gridDim.x) {
        double2 cur[8]; for (int i = 0; i < 8; ++i) cur[i] = pre[i]; double ck = pk;
        if (b + gridDim.x < B) pf(b + gridDim.x);
        double* ob = out + b * 1024;
        if (STORE_W == 1) { for (int r = 0; r < 16; ++r) { double v = (r & 1 ? cur[r >> 1].y : cur[r >> 1].x) + ck; if (NT) __builtin_nontemporal_store(v, &ob[r * 64 + lane]); else ob[r * 64 + lane] = v; } }
        else { double2* o2 = (double2*)ob; for (int i = 0; i < 8; ++i) { double2 v = cur[i]; v.x += ck; o2[i * 64 + lane] = v; } }
    }
}
int main() {
    const long B = 1000000; double *sig, *K, *out;
    hipMalloc(&sig, B * 1024 * 8); hipMalloc(&K, B * 64 * 8); hipMalloc(&out, B * 1024 * 8);
    hipMemset(sig, 0, B * 1024 * 8); hipMemset(K, 0, B * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = B * (1024.0 * 8 * 2 + 512);
    for (int lds : {20480, 10240, 1024}) for (int per_cu : {8, 16}) for (int sw : {1, 2}) {
        if (per_cu * lds > 163840) continue;
        int grid = 256 * per_cu; float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0);
            if (sw == 1) probe<1><<<grid, 64, lds>>>(sig, K, out, B); else probe<2><<<grid, 64, lds>>>(sig, K, out, B);
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        printf("lds %6d B  wg/CU %2d  store %s : %.3f ms  %.0f GB/s  (%.1f M surfaces/s)\n", lds, per_cu, sw == 1 ? "dwordx2" : "dwordx4", best, bytes / best / 1e6, B / best / 1e3);
    }
    { int grid = 256 * 8; float best = 1e9;
      for (int it = 0; it < 6; ++it) { hipEventRecord(e0); probe<1, true><<<grid, 64, 20480>>>(sig, K, out, B); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms; }
      printf("lds  20480 B  wg/CU  8  dwordx2 + NON-TEMPORAL loads/stores : %.3f ms  %.0f GB/s  (%.1f M surfaces/s)\n", best, bytes / best / 1e6, B / best / 1e3); }
    return 0;
}
