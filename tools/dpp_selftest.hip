// Standalone check of the cross-lane primitives the dense kernel relies on (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../iv_interpolation_amd/csrc/ivs_surface_dense.hpp"
using namespace ivs;

__global__ void k_dpp(double* out) {
    int lane = threadIdx.x; double v = 100.0 + lane;
    out[0 * 64 + lane] = dpp_f64<DPP_WAVE_SHR1>(-1.0, v);
    out[1 * 64 + lane] = dpp_f64<DPP_WAVE_SHL1>(-1.0, v);
    out[2 * 64 + lane] = dpp_f64<DPP_ROW_SHR(1)>(-1.0, v);
    out[3 * 64 + lane] = dpp_f64<DPP_ROW_SHL(1)>(-1.0, v);
    out[4 * 64 + lane] = dpp_f64<DPP_ROW_BCAST15, 0xA>(-1.0, v);
    out[5 * 64 + lane] = dpp_f64<DPP_ROW_BCAST31, 0xC>(-1.0, v);
    out[6 * 64 + lane] = dpp_f64<DPP_ROW_SHR(4)>(-1.0, v);
}
__global__ void k_factor(const double* X, double* out) {
    __shared__ double Xs[64];
    int lane = threadIdx.x; Xs[lane] = X[lane]; __syncthreads();
    double al, cp, pp, qq, rdx;
    factor_tables<64>(Xs, lane, al, cp, pp, qq, rdx);
    out[0 * 64 + lane] = al; out[1 * 64 + lane] = cp; out[2 * 64 + lane] = pp; out[3 * 64 + lane] = qq;
    out[4 * 64 + lane] = seg16_prefix_prod(-al, lane); out[5 * 64 + lane] = seg16_suffix_prod(-cp, lane);
}
__global__ void k_slopes(const double* K, const double* sig, double* out) {
    __shared__ __attribute__((aligned(16))) double Y[DT * D_RS], S[DT * D_RS], Ksh[64], RDX[64];
    int lane = threadIdx.x;
    for (int t = 0; t < DT; ++t) Y[t * D_RS + d_sl(lane)] = sig[t * 64 + lane];
    Ksh[lane] = K[lane];
    __syncthreads();
    auto nostamp = [](int) {};
    dense_strike_slopes<false>(Y, S, Ksh, RDX, lane, nostamp);
    __syncthreads();
    for (int t = 0; t < DT; ++t) out[t * 64 + lane] = S[t * D_RS + d_sl(lane)];
}
static void thomas(const double* x, const double* y, int n, double* s) {
    double dx[64], dl[64], lo[64], di[64], up[64], rhs[64], cp[64], dp[64];
    for (int i = 0; i < n - 1; ++i) { dx[i] = x[i + 1] - x[i]; dl[i] = (y[i + 1] - y[i]) / dx[i]; }
    for (int i = 1; i < n - 1; ++i) { lo[i] = dx[i]; di[i] = 2 * (dx[i - 1] + dx[i]); up[i] = dx[i - 1]; rhs[i] = 3 * (dx[i] * dl[i - 1] + dx[i - 1] * dl[i]); }
    double d = x[2] - x[0]; lo[0] = 0; di[0] = dx[1]; up[0] = d; rhs[0] = ((dx[0] + 2 * d) * dx[1] * dl[0] + dx[0] * dx[0] * dl[1]) / d;
    d = x[n - 1] - x[n - 3]; lo[n - 1] = d; di[n - 1] = dx[n - 3]; up[n - 1] = 0;
    rhs[n - 1] = (dx[n - 2] * dx[n - 2] * dl[n - 3] + (2 * d + dx[n - 2]) * dx[n - 3] * dl[n - 2]) / d;
    cp[0] = up[0] / di[0]; dp[0] = rhs[0] / di[0];
    for (int i = 1; i < n; ++i) { double w = di[i] - lo[i] * cp[i - 1]; cp[i] = up[i] / w; dp[i] = (rhs[i] - lo[i] * dp[i - 1]) / w; }
    s[n - 1] = dp[n - 1]; for (int i = n - 2; i >= 0; --i) s[i] = dp[i] - cp[i] * s[i + 1];
}
int main() {
    double *d, h[7 * 64];
    hipMalloc(&d, sizeof(h));
    k_dpp<<<1, 64>>>(d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"wave_shr1", "wave_shl1", "row_shr1", "row_shl1", "row_bcast15(A)", "row_bcast31(C)", "row_shr4"};
    for (int r = 0; r < 7; ++r) { printf("%-14s:", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
    // factor tables vs serial Thomas
    double X[64], *dX; for (int i = 0; i < 64; ++i) X[i] = 0.7 + 0.6 * i / 63.0 + 0.003 * sin(7.0 * i);
    hipMalloc(&dX, sizeof(X)); hipMemcpy(dX, X, sizeof(X), hipMemcpyHostToDevice);
    k_factor<<<1, 64>>>(dX, d); hipMemcpy(h, d, 6 * 64 * 8, hipMemcpyDeviceToHost);
    const int n = 64; double dx[63]; for (int i = 0; i < 63; ++i) dx[i] = X[i + 1] - X[i];
    double a[64], b[64], c[64], w[64], al[64], cp[64];
    for (int i = 1; i < n - 1; ++i) { a[i] = dx[i]; b[i] = 2 * (dx[i - 1] + dx[i]); c[i] = dx[i - 1]; }
    a[0] = 0; b[0] = dx[1]; c[0] = dx[0] + dx[1]; a[n - 1] = dx[n - 3] + dx[n - 2]; b[n - 1] = dx[n - 3]; c[n - 1] = 0;
    w[0] = b[0]; for (int i = 1; i < n; ++i) w[i] = b[i] - a[i] * c[i - 1] / w[i - 1];
    double eal = 0, ecp = 0, epi = 0, epsi = 0;
    for (int i = 0; i < n; ++i) { al[i] = a[i] / w[i]; cp[i] = c[i] / w[i]; eal = fmax(eal, fabs(al[i] - h[i])); ecp = fmax(ecp, fabs(cp[i] - h[64 + i])); }
    for (int s = 0; s < 4; ++s) { double p = 1; for (int i = 16 * s; i < 16 * s + 16; ++i) { p *= -al[i]; epi = fmax(epi, fabs(p - h[4 * 64 + i])); }
        p = 1; for (int i = 16 * s + 15; i >= 16 * s; --i) { p *= -cp[i]; epsi = fmax(epsi, fabs(p - h[5 * 64 + i])); } }
    printf("factor: max|AL err| %.3e  max|CP err| %.3e  max|PI err| %.3e  max|PSI err| %.3e\n", eal, ecp, epi, epsi);
    {
        double sig[16 * 64], *dsig, *dout, got[16 * 64], ref[64];
        for (int t = 0; t < 16; ++t) for (int k = 0; k < 64; ++k) sig[t * 64 + k] = 0.5 + 0.2 * sin(0.3 * k + t) + 0.01 * cos(1.7 * k * t);
        hipMalloc(&dsig, sizeof(sig)); hipMalloc(&dout, sizeof(sig)); hipMemcpy(dsig, sig, sizeof(sig), hipMemcpyHostToDevice);
        k_slopes<<<1, 64>>>(dX, dsig, dout); hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost);
        for (int t = 0; t < 16; ++t) { thomas(X, sig + t * 64, 64, ref); double e = 0; int wk = 0;
            for (int k = 0; k < 64; ++k) { double dd = fabs(ref[k] - got[t * 64 + k]); if (dd > e) { e = dd; wk = k; } }
            printf("row %2d: max|slope err| %.3e at k=%d (ref %.6f got %.6f)\n", t, e, wk, ref[wk], got[t * 64 + wk]); }
    }
    return 0;
}
