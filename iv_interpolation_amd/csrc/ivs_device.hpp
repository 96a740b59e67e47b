// Device-side building blocks shared by the gfx950 kernels.
//
// Arithmetic follows oracle/ivs_oracle.py (which restates numpy's arr_interp and scipy's
// not-a-knot system, cited there).  The linear evaluator is written to reproduce np.interp
// bit for bit: IEEE division, then a separately rounded multiply and add (contraction off).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ivs.h"

namespace ivs {

constexpr int WAVE = 64;

__device__ __forceinline__ double qnan() { return __builtin_nan(""); }

// Largest j in [0, n) with x(j) <= xq, or -1.  NaN xq -> -1.
template <class XA>
__device__ __forceinline__ int find_interval(const XA& x, int n, double xq) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (x(mid) <= xq) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// np.interp's interior formula (numpy compiled_base.c arr_interp), bit-exact:
// slope = (y1-y0)/(x1-x0); r = slope*(x-x0) + y0 with two roundings.
__device__ __forceinline__ double lerp_np(double xq, double x0, double y0, double x1, double y1) {
#pragma clang fp contract(off)
    if (x0 == xq) return y0;
    double slope = (y1 - y0) / (x1 - x0);
    double r = slope * (xq - x0) + y0;
    if (__builtin_isnan(r)) {            // "if we get nan in one direction, try the other"
        r = slope * (xq - x1) + y1;
        if (__builtin_isnan(r) && y0 == y1) r = y0;
    }
    return r;
}

// One piecewise-linear evaluation on n valid knots (n >= 1), knots via accessors.
template <class XA, class YA>
__device__ __forceinline__ double eval_linear(const XA& x, const YA& y, int n, int j, double xq, bool right_hold) {
    if (j < 0) return qnan();
    if (j >= n - 1) return (right_hold || xq == x(n - 1)) ? y(n - 1) : qnan();
    return lerp_np(xq, x(j), y(j), x(j + 1), y(j + 1));
}

// One cubic-Hermite evaluation from knot slopes s (scipy PPoly coefficient build + Horner).
template <class XA, class YA, class SA>
__device__ __forceinline__ double eval_cubic(const XA& x, const YA& y, const SA& s, int n, int j, double xq,
                                             bool extrapolate_right) {
    if (j < 0) return qnan();
    if (!extrapolate_right && !(xq <= x(n - 1))) return qnan();
    int jj = j > n - 2 ? n - 2 : j;
    double x0 = x(jj), y0 = y(jj), s0 = s(jj), s1 = s(jj + 1);
    double h = x(jj + 1) - x0;
    double rh = 1.0 / h;
    double delta = (y(jj + 1) - y0) * rh;
    double t = (s0 + s1 - 2.0 * delta) * rh;
    double c0 = t * rh;
    double c1 = (delta - s0) * rh - t;
    double u = xq - x0;
    return ((c0 * u + c1) * u + s0) * u + y0;
}

// Not-a-knot slopes by the Thomas recurrence, one thread per system (oracle nak_slopes()).
// x, y: knot accessors; s, cp: writable accessors (cp is scratch of n entries).
// Requires n >= 2.  n == 2 -> secant; n == 3 -> parabola; n >= 4 -> tridiagonal solve.
template <class XA, class YA, class SW, class CW>
__device__ __forceinline__ void nak_slopes(const XA& x, const YA& y, SW& s, CW& cp, int n) {
    if (n == 2) {
        double d = (y(1) - y(0)) / (x(1) - x(0));
        s.set(0, d); s.set(1, d);
        return;
    }
    if (n == 3) {
        double dx0 = x(1) - x(0), dx1 = x(2) - x(1);
        double d0 = (y(1) - y(0)) / dx0, d1 = (y(2) - y(1)) / dx1;
        double b0 = 2.0 * d0, b1 = 3.0 * (dx0 * d1 + dx1 * d0), b2 = 2.0 * d1;
        double s1 = (b1 - dx1 * b0 - dx0 * b2) / (2.0 * (dx0 + dx1) - dx1 - dx0);
        s.set(0, b0 - s1); s.set(1, s1); s.set(2, b2 - s1);
        return;
    }
    // row 0 (not-a-knot): [dx1, x2-x0], rhs = ((dx0 + 2d) dx1 d0 + dx0^2 d1) / d
    double xm = x(0), xc = x(1), xp = x(2);
    double ym = y(0), yc = y(1), yp = y(2);
    double dxm = xc - xm, dxc = xp - xc;             // dx[i-1], dx[i]
    double dlm = (yc - ym) / dxm, dlc = (yp - yc) / dxc;
    double d = xp - xm;
    double rhs = ((dxm + 2.0 * d) * dxc * dlm + dxm * dxm * dlc) / d;
    double cprev = d / dxc;                          // up/di
    double dprev = rhs / dxc;
    cp.set(0, cprev); s.set(0, dprev);
    // interior rows i = 1..n-2: [dx_i, 2(dx_{i-1}+dx_i), dx_{i-1}]
    for (int i = 1; i <= n - 2; ++i) {
        rhs = 3.0 * (dxc * dlm + dxm * dlc);
        double w = 2.0 * (dxm + dxc) - dxc * cprev;
        cprev = dxm / w;
        dprev = (rhs - dxc * dprev) / w;
        cp.set(i, cprev); s.set(i, dprev);
        if (i < n - 2) {
            xm = xc; xc = xp; xp = x(i + 2);
            ym = yc; yc = yp; yp = y(i + 2);
            dxm = dxc; dxc = xp - xc;
            dlm = dlc; dlc = (yp - yc) / dxc;
        }
    }
    // last row (not-a-knot): [x_{n-1}-x_{n-3}, dx_{n-3}]; here dxm = dx[n-3], dxc = dx[n-2]
    d = xp - xm;
    rhs = (dxc * dxc * dlm + (2.0 * d + dxc) * dxm * dlc) / d;
    {
        double w = dxm - d * cprev;
        dprev = (rhs - d * dprev) / w;
    }
    double sn = dprev;
    s.set(n - 1, sn);
    for (int i = n - 2; i >= 0; --i) {
        sn = s(i) - cp(i) * sn;
        s.set(i, sn);
    }
}

// strided views over LDS / global arrays
struct View {
    double* p; int stride;
    __device__ __forceinline__ double operator()(int i) const { return p[i * stride]; }
    __device__ __forceinline__ void set(int i, double v) { p[i * stride] = v; }
};
struct CView {
    const double* p; int stride;
    __device__ __forceinline__ double operator()(int i) const { return p[i * stride]; }
};

__device__ __forceinline__ bool method_is_cubic(int m) { return m == IVS_CUBIC || m == IVS_CUBICSPLINE; }
__device__ __forceinline__ int method_min_knots(int m) {
    return m == IVS_LINEAR ? 0 : (m == IVS_CUBIC ? 4 : 2);
}

}  // namespace ivs
