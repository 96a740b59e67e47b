// Black-Scholes Greeks epilogue: elementwise on (S, K, T, r, sigma) -> delta, gamma,
// theta, vega, rho.  Formulas and their quirks follow the reference src/interpolation/greeks.py:21-35 (put rho
// has no sign flip there).  Grid-stride, one element per lane per step, SoA arrays -> fully coalesced
// 8-byte accesses; 80 B per element move through HBM, ~150 fp64 instructions of math per element.
#pragma once
#include "ivs_device.hpp"

namespace ivs {

struct GreeksParams {
    const double* S; const double* K; const double* T; const double* r; const double* sigma;
    const uint8_t* is_put; int default_put; int64_t n;
    double* delta; double* gamma; double* theta; double* vega; double* rho;
};

__device__ __forceinline__ double norm_cdf(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }
__device__ __forceinline__ double norm_pdf(double x) { return exp(-x * x * 0.5) * 0.39894228040143267794; }

// one option: the reference's formulas statement by statement (greeks.py:21-35)
__device__ __forceinline__ void bs_greeks_one(double S, double K, double T, double r, double sg, bool put, double& delta,
                                              double& gamma, double& theta, double& vega, double& rho) {
    const double sq = sqrt(T);
    const double d1 = (log(S / K) + (r + 0.5 * sg * sg) * T) / (sg * sq);
    const double d2 = d1 - sg * sq;
    const double pdf1 = norm_pdf(d1);
    const double cdf1 = norm_cdf(d1);
    const double cdf2 = norm_cdf(put ? -d2 : d2);
    const double disc = exp(-r * T);
    const double common = -S * pdf1 * sg / (2.0 * sq);
    delta = put ? cdf1 - 1.0 : cdf1;
    gamma = pdf1 / (S * sg * sq);
    theta = (put ? common + r * K * disc * cdf2 : common - r * K * disc * cdf2) / 365.0;
    vega = S * pdf1 * sq / 100.0;
    rho = K * T * disc * cdf2 / 100.0;
}

__global__ __launch_bounds__(256) void greeks_kernel(GreeksParams p) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * 256) {
        const bool put = p.is_put ? p.is_put[i] != 0 : p.default_put != 0;
        double de, ga, th, ve, rh;
        bs_greeks_one(p.S[i], p.K[i], p.T[i], p.r[i], p.sigma[i], put, de, ga, th, ve, rh);
        p.delta[i] = de; p.gamma[i] = ga; p.theta[i] = th; p.vega[i] = ve; p.rho[i] = rh;
    }
}

}  // namespace ivs
