// Does the surface -> workgroup MAPPING change the streaming rate of the dense kernel's access pattern?
// (tools/write_probe.hip: a float4 copy runs at 4.85 TB/s as a persistent grid-stride loop over 2048 workgroups but
// at 5.93 TB/s as 262144 short workgroups, i.e. when the resident workgroups work in several distant regions at once.)
//   mode 0  interleaved   b = wg + k*G                       (the shipped mapping: all workgroups sweep one window)
//   mode 1  blocked       b = wg*chunk + k                   (every workgroup its own contiguous region)
//   mode 2  R regions     b = r*RS + wg + j*G, k = j*R + r   (R distant windows swept in turn by every workgroup)
//   mode 3  R groups      workgroups split into R groups, group r sweeps region r interleaved: b = r*RS + (wg/R) + j*(G/R)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void probe(const double* sig, const double* K, double* out, long B, int mode, int R) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long G = gridDim.x, wg = blockIdx.x;
    const long chunk = (B + G - 1) / G, RS = (B + R - 1) / R;
    long n_it;
    if (mode == 0) n_it = (B - wg + G - 1) / G;
    else if (mode == 1) n_it = chunk;
    else if (mode == 2) n_it = ((RS + G - 1) / G) * R;
    else n_it = (RS + (G / R) - 1) / (G / R);
    auto surf = [&](long k) -> long {
        long b;
        if (mode == 0) b = wg + k * G;
        else if (mode == 1) { b = wg * chunk + k; if (k >= chunk) b = B; }
        else if (mode == 2) { const long r = k % R, j = k / R; const long o = wg + j * G; b = o < RS ? r * RS + o : B; }
        else { const long r = wg % R, o = wg / R + k * (G / R); b = o < RS ? r * RS + o : B; }
        return b < B ? b : -1;
    };
    double2 pre[8]; double pk;
    auto pf = [&](long b) { const double2* s2 = (const double2*)(sig + b * 1024);
        for (int i = 0; i < 8; ++i) pre[i] = s2[i * 64 + lane]; pk = K[b * 64 + lane]; };
    long b = surf(0); if (b >= 0) pf(b);
    for (long k = 0; k < n_it; ++k) {
        double2 cur[8]; for (int i = 0; i < 8; ++i) cur[i] = pre[i]; double ck = pk;
        const long bn = k + 1 < n_it ? surf(k + 1) : -1;
        if (bn >= 0) pf(bn);
        if (b >= 0) {
            double* ob = out + b * 1024;
            for (int r = 0; r < 16; ++r) { double v = (r & 1 ? cur[r >> 1].y : cur[r >> 1].x) + ck; ob[r * 64 + lane] = v; }
        }
        b = bn;
    }
}
int main() {
    const long B = 1000000; double *sig, *K, *out;
    hipMalloc(&sig, B * 1024 * 8); hipMalloc(&K, B * 64 * 8); hipMalloc(&out, B * 1024 * 8);
    hipMemset(sig, 0, B * 1024 * 8); hipMemset(K, 0, B * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = B * (1024.0 * 8 * 2 + 512);
    struct Cfg { int mode, R; const char* nm; };
    const Cfg cfgs[] = {{0, 1, "interleaved (shipped)"}, {1, 1, "blocked"}, {2, 2, "2 regions in turn"}, {2, 4, "4 regions in turn"},
                        {2, 8, "8 regions in turn"}, {2, 16, "16 regions in turn"}, {2, 64, "64 regions in turn"},
                        {3, 2, "2 groups"}, {3, 4, "4 groups"}, {3, 8, "8 groups (wg % 8 = XCD)"}, {3, 16, "16 groups"}, {3, 64, "64 groups"}, {3, 256, "256 groups"}};
    for (const Cfg& c : cfgs) {
        float best = 1e9;
        for (int it = 0; it < 6; ++it) {
            hipEventRecord(e0); probe<<<2048, 64, 20480>>>(sig, K, out, B, c.mode, c.R); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it && ms < best) best = ms;
        }
        printf("%-28s %.3f ms  %.0f GB/s  (%.1f M surfaces/s)\n", c.nm, best, bytes / best / 1e6, B / best / 1e3);
    }
    return 0;
}
