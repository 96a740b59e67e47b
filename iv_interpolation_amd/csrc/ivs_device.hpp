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

// 1/b to <= 1 ulp: v_rcp_f64 + two Newton steps (the reciprocal the IEEE division expansion itself uses);
// a third of the instructions of `1.0 / b`, no v_div_scale/fmas/fixup chain through VCC
__device__ __forceinline__ double refined_rcp(double b) {
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}

// The same with the interval slopes dy/dx precomputed (np.interp does this too when there are fewer knots than
// queries: identical values, identical rounding).
template <class XA, class YA, class SA>
__device__ __forceinline__ double eval_linear_slopes(const XA& x, const YA& y, const SA& slope, int n, int j, double xq,
                                                     bool right_hold) {
#pragma clang fp contract(off)
    if (j < 0) return qnan();
    if (j >= n - 1) return (right_hold || xq == x(n - 1)) ? y(n - 1) : qnan();
    const double x0 = x(j), y0 = y(j);
    if (x0 == xq) return y0;
    const double sl = slope(j);
    double r = sl * (xq - x0) + y0;
    if (__builtin_isnan(r)) {
        r = sl * (xq - x(j + 1)) + y(j + 1);
        if (__builtin_isnan(r) && y0 == y(j + 1)) r = y0;
    }
    return r;
}

// One cubic-Hermite evaluation from knot slopes s (scipy PPoly coefficient build + Horner).
__device__ __forceinline__ double hermite_eval(double x0, double x1, double y0, double y1, double s0, double s1, double xq) {
    double h = x1 - x0;
    double rh = refined_rcp(h);
    double delta = (y1 - y0) * rh;
    double t = (s0 + s1 - 2.0 * delta) * rh;
    double c0 = t * rh;
    double c1 = (delta - s0) * rh - t;
    double u = xq - x0;
    return ((c0 * u + c1) * u + s0) * u + y0;
}
template <class XA, class YA, class SA>
__device__ __forceinline__ double eval_cubic(const XA& x, const YA& y, const SA& s, int n, int j, double xq,
                                             bool extrapolate_right) {
    if (j < 0) return qnan();
    if (!extrapolate_right && !(xq <= x(n - 1))) return qnan();
    int jj = j > n - 2 ? n - 2 : j;
    return hermite_eval(x(jj), x(jj + 1), y(jj), y(jj + 1), s(jj), s(jj + 1), xq);
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
    double dlm = (yc - ym) * refined_rcp(dxm), dlc = (yp - yc) * refined_rcp(dxc);
    double d = xp - xm;
    double rhs = ((dxm + 2.0 * d) * dxc * dlm + dxm * dxm * dlc) * refined_rcp(d);
    const double rdx1 = refined_rcp(dxc);
    double cprev = d * rdx1;                         // up/di
    double dprev = rhs * rdx1;
    cp.set(0, cprev); s.set(0, dprev);
    // interior rows i = 1..n-2: [dx_i, 2(dx_{i-1}+dx_i), dx_{i-1}]
    for (int i = 1; i <= n - 2; ++i) {
        rhs = 3.0 * (dxc * dlm + dxm * dlc);
        const double rw = refined_rcp(2.0 * (dxm + dxc) - dxc * cprev);
        cprev = dxm * rw;
        dprev = (rhs - dxc * dprev) * rw;
        cp.set(i, cprev); s.set(i, dprev);
        if (i < n - 2) {
            xm = xc; xc = xp; xp = x(i + 2);
            ym = yc; yc = yp; yp = y(i + 2);
            dxm = dxc; dxc = xp - xc;
            dlm = dlc; dlc = (yp - yc) * refined_rcp(dxc);
        }
    }
    // last row (not-a-knot): [x_{n-1}-x_{n-3}, dx_{n-3}]; here dxm = dx[n-3], dxc = dx[n-2]
    d = xp - xm;
    rhs = (dxc * dxc * dlm + (2.0 * d + dxc) * dxm * dlc) * refined_rcp(d);
    dprev = (rhs - d * dprev) * refined_rcp(dxm - d * cprev);
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

// methods that carry one precomputed value per knot besides y: the knot slopes of a C1 piecewise cubic (not-a-knot
// solve, pchip or akima local rules) or the B-spline coefficients of the quadratic spline
__host__ __device__ __forceinline__ bool method_is_cubic(int m) {
    return m == IVS_CUBIC || m == IVS_CUBICSPLINE || m == IVS_PCHIP || m == IVS_AKIMA || m == IVS_QUADRATIC;
}
__device__ __forceinline__ bool method_extrapolates_right(int m) { return m == IVS_CUBICSPLINE || m == IVS_PCHIP; }
// one polynomial through all knots: the per-knot table holds barycentric weights / Newton coefficients
__host__ __device__ __forceinline__ bool method_is_poly(int m) { return m == IVS_BARYCENTRIC || m == IVS_KROGH; }
__device__ __forceinline__ int method_min_knots(int m) {
    switch (m) {
        case IVS_LINEAR: case IVS_PAD: case IVS_BFILL: return 0;
        case IVS_CUBIC: return 4;
        case IVS_NEAREST: case IVS_ZERO: case IVS_BARYCENTRIC: case IVS_KROGH: return 1;
        case IVS_AKIMA: return 3;
        case IVS_QUADRATIC: return 3;
        default: return 2;
    }
}

// ---- step / Bernstein evaluators (oracle nearest_eval, zero_eval, bpoly_linear_eval)
template <class XA, class YA>
__device__ __forceinline__ double eval_nearest(const XA& x, const YA& y, int n, double xq) {
    if (!(xq >= x(0) && xq <= x(n - 1))) return qnan();
    int lo = 0, hi = n - 1;                    // first i with x_i/2 + x_{i+1}/2 >= xq (searchsorted side='left')
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x(mid) / 2.0 + x(mid + 1) / 2.0 < xq) lo = mid + 1; else hi = mid;
    }
    return y(lo);
}
template <class XA, class YA>
__device__ __forceinline__ double eval_zero(const XA& x, const YA& y, int n, int j, double xq) {
    if (j < 0 || !(xq <= x(n - 1))) return qnan();
    return y(j > n - 1 ? n - 1 : j);
}
// 'pad' / 'ffill' and 'bfill' / 'backfill' (pandas pad_or_backfill through Series.interpolate): the last valid knot at or
// before xq (hold-last on the right) / the first valid knot at or after xq (the first knot's value on the left)
template <class XA, class YA>
__device__ __forceinline__ double eval_pad(const XA& x, const YA& y, int n, int j) {
    return j < 0 ? qnan() : y(j > n - 1 ? n - 1 : j);
}
template <class XA, class YA>
__device__ __forceinline__ double eval_bfill(const XA& x, const YA& y, int n, int j, double xq) {
    const int k = (j >= 0 && x(j) == xq) ? j : j + 1;
    return (k < n && !__builtin_isnan(xq)) ? y(k) : qnan();
}
template <class XA, class YA>
__device__ __forceinline__ double eval_bpoly_linear(const XA& x, const YA& y, int n, int j, double xq) {
#pragma clang fp contract(off)
    if (j < 0 || !(xq <= x(n - 1))) return qnan();
    const int jj = j > n - 2 ? n - 2 : j;
    const double x0 = x(jj);
    const double s = (xq - x0) / (x(jj + 1) - x0);
    const double a = y(jj) * (1.0 - s), b = y(jj + 1) * s;
    return a + b;
}

__device__ __forceinline__ double sgn(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }

// PchipInterpolator._find_derivatives + _edge_case (scipy _cubic.py:248-309), one thread per system, n >= 2
template <class XA, class YA, class SW>
__device__ __forceinline__ void pchip_slopes(const XA& x, const YA& y, SW& s, int n) {
    if (n == 2) {
        const double mk = (y(1) - y(0)) / (x(1) - x(0));
        s.set(0, mk); s.set(1, mk);
        return;
    }
    auto edge = [](double h0, double h1, double m0, double m1) {
        const double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
        if (sgn(d) != sgn(m0)) return 0.0;
        if (sgn(m0) != sgn(m1) && __builtin_fabs(d) > 3.0 * __builtin_fabs(m0)) return 3.0 * m0;
        return d;
    };
    double hp = x(1) - x(0), mp = (y(1) - y(0)) / hp;          // h_{k-1}, m_{k-1}
    double h0 = hp, m0 = mp;
    for (int k = 1; k <= n - 2; ++k) {
        const double hk = x(k + 1) - x(k), mk = (y(k + 1) - y(k)) / hk;
        if (k == 1) s.set(0, edge(h0, hk, m0, mk));
        const bool cond = sgn(mk) != sgn(mp) || mk == 0.0 || mp == 0.0;
        const double w1 = 2.0 * hk + hp, w2 = hk + 2.0 * hp;
        const double whmean = (w1 / mp + w2 / mk) / (w1 + w2);
        s.set(k, cond ? 0.0 : 1.0 / whmean);
        if (k == n - 2) s.set(n - 1, edge(hk, hp, mk, mp));
        hp = hk; mp = mk;
    }
}

// Akima1DInterpolator slopes (scipy _cubic.py:510-541), one thread per system, n >= 3
template <class XA, class YA, class SW>
__device__ __forceinline__ void akima_slopes(const XA& x, const YA& y, SW& s, int n) {
    auto m_in = [&](int k) { return (y(k + 1) - y(k)) / (x(k + 1) - x(k)); };       // 0 <= k <= n-2
    auto m_ext = [&](int k) -> double {                                             // -2 <= k <= n
        if (k >= 0 && k <= n - 2) return m_in(k);
        if (k == -1) return 2.0 * m_in(0) - m_in(1);
        if (k == -2) { const double a = 2.0 * m_in(0) - m_in(1); return 2.0 * a - m_in(0); }
        if (k == n - 1) return 2.0 * m_in(n - 2) - m_in(n - 3);
        const double a = 2.0 * m_in(n - 2) - m_in(n - 3);
        return 2.0 * a - m_in(n - 2);
    };
    double fmax = -__builtin_inf();
    for (int i = 0; i < n; ++i) {
        const double f1 = __builtin_fabs(m_ext(i + 1) - m_ext(i)), f2 = __builtin_fabs(m_ext(i - 1) - m_ext(i - 2));
        const double f12 = f1 + f2;
        if (f12 > fmax) fmax = f12;
    }
    for (int i = 0; i < n; ++i) {
        const double ma = m_ext(i - 2), mb = m_ext(i - 1), mc = m_ext(i), md = m_ext(i + 1);
        const double f1 = __builtin_fabs(md - mc), f2 = __builtin_fabs(mb - ma), f12 = f1 + f2;
        double t = 0.5 * (md + ma);
        if (f12 > 1e-9 * fmax) t = (f1 * mb + f2 * mc) / f12;
        s.set(i, t);
    }
}

// ---- 'quadratic': interp1d(kind=2) -> scipy make_interp_spline(k=2) (_bsplines.py): knot vector _not_a_knot for even
// k = the midpoints of the data sites without the first and the last one, triple end knots.  Every site then sees
// exactly three consecutive basis functions: the collocation matrix is tridiagonal (and totally positive: elimination
// without pivoting is stable; scipy calls LAPACK gbsv).  Oracle: ivs_oracle.py quadratic_coeffs / quadratic_eval.
template <class XA>
__device__ __forceinline__ double quad_knot(const XA& x, int n, int j) {      // t_j, j = 0..n+2
    if (j <= 2) return x(0);
    if (j >= n) return x(n - 1);
    return (x(j - 1) + x(j - 2)) / 2.0;                                       // mid_{j-2}
}
// the three quadratic B-splines that are non-zero on [t_ell, t_ell+1) at xv (de Boor's recurrence, scipy _deBoor_D)
// (from the four knots t_{ell-1} .. t_{ell+2}; callers that hold the data sites in registers form the knots themselves)
__device__ __forceinline__ void quad_basis_t(double tm1, double t0, double t1, double t2, double xv, double& h0, double& h1, double& h2);
template <class XA>
__device__ __forceinline__ void quad_basis(const XA& x, int n, int ell, double xv, double& h0, double& h1, double& h2) {
    const double tm1 = quad_knot(x, n, ell - 1), t0 = quad_knot(x, n, ell), t1 = quad_knot(x, n, ell + 1),
                 t2 = quad_knot(x, n, ell + 2);
    quad_basis_t(tm1, t0, t1, t2, xv, h0, h1, h2);
}
__device__ __forceinline__ void quad_basis_t(double tm1, double t0, double t1, double t2, double xv, double& h0, double& h1, double& h2) {
    // the three divisions of _deBoor_D as reciprocal (<= 1 ulp) times numerator: a third of the instructions of the IEEE
    // division expansion, and this function runs per site AND per output value on the kernels whose rows have their own knots
    const double w = refined_rcp(t1 - t0);
    const double a0 = w * (t1 - xv), a1 = w * (xv - t0);                      // order 1
    const double w1 = a0 * refined_rcp(t1 - tm1);
    h0 = 0.0 + w1 * (t1 - xv);
    h1 = w1 * (xv - tm1);
    const double w2 = a1 * refined_rcp(t2 - t0);
    h1 += w2 * (t2 - xv);
    h2 = w2 * (xv - t0);
}
// B-spline coefficients into s (cp = scratch), n >= 3
template <class XA, class YA, class SW, class CW>
__device__ __forceinline__ void quadratic_coeffs(const XA& x, const YA& y, SW& s, CW& cp, int n) {
    cp.set(0, 0.0); s.set(0, y(0));                                           // row 0 = [1, 0, ...]
    for (int i = 1; i < n; ++i) {
        double lo = 0.0, di = 1.0, up = 0.0;
        if (i < n - 1) quad_basis(x, n, i + 1, x(i), lo, di, up);             // site i lies in knot interval ell = i + 1
        const double w = di - lo * cp(i - 1);
        cp.set(i, up / w);
        s.set(i, (y(i) - lo * s(i - 1)) / w);
    }
    for (int i = n - 2; i >= 0; --i) s.set(i, s(i) - cp(i) * s(i + 1));
}
template <class XA, class SA>
__device__ __forceinline__ double eval_quadratic(const XA& x, const SA& c, int n, int j, double xq) {
    if (j < 0 || !(xq <= x(n - 1))) return qnan();
    int q = n - 3;                                   // interior knots mid_1..mid_{n-3} that are <= xq
    if (j < n - 1) { q = j - 1 + ((xq >= (x(j + 1) + x(j)) / 2.0) ? 1 : 0); q = q < 0 ? 0 : (q > n - 3 ? n - 3 : q); }
    const int ell = q + 2;
    double h0, h1, h2;
    quad_basis(x, n, ell, xq, h0, h1, h2);
    return h0 * c(ell - 2) + h1 * c(ell - 1) + h2 * c(ell);
}

// ---- 'barycentric' / 'krogh' (oracle barycentric_weights / barycentric_eval / krogh_coeffs / krogh_eval; scipy _polyint.py)
// weight of knot j: 1 / prod_{k != j} (c (x_j - x_k)), c = 4 / (x_max - x_min); factors in index order
template <class XA>
__device__ __forceinline__ double barycentric_weight(const XA& x, int n, int j) {
    if (n == 1) return 1.0;
    const double cap = 4.0 / (x(n - 1) - x(0));
    const double xj = x(j);
    double prod = 1.0;
    for (int k = 0; k < n; ++k) prod *= (k == j) ? 1.0 : cap * (xj - x(k));
    return 1.0 / prod;
}
template <class XA, class YA, class WA>
__device__ __forceinline__ double eval_barycentric(const XA& x, const YA& y, const WA& w, int n, double xq) {
    if (!(xq >= x(0))) return qnan();                          // pandas keeps NaN left of the first knot
    double num = 0.0, den = 0.0, hit = 0.0;
    bool on_node = false;
    for (int k = 0; k < n; ++k) {
        const double c = xq - x(k);
        if (c == 0.0) { on_node = true; hit = y(k); break; }
        const double t = w(k) / c;
        num += t * y(k); den += t;
    }
    return on_node ? hit : num / den;
}
// Newton coefficients by scipy's recurrence (one thread; n <= IVS_POLY_MAX_KNOTS); vk = scratch of n entries
template <class XA, class YA, class CW, class VW>
__device__ __forceinline__ void krogh_coeffs(const XA& x, const YA& y, CW& c, VW& vk, int n) {
    c.set(0, y(0));
    for (int k = 1; k < n; ++k) {
        vk.set(0, y(k));
        for (int i = 0; i < k; ++i) vk.set(i + 1, (c(i) - vk(i)) / (x(i) - x(k)));
        c.set(k, vk(k));
    }
}
template <class XA, class CA>
__device__ __forceinline__ double eval_krogh(const XA& x, const CA& c, int n, double xq) {
#pragma clang fp contract(off)
    if (!(xq >= x(0))) return qnan();
    double pi = 1.0, p = c(0);
    for (int k = 1; k < n; ++k) {
        pi = (xq - x(k - 1)) * pi;
        p = p + pi * c(k);
    }
    return p;
}

// knot slopes of the method's interpolant (needs n >= method_min_knots and n >= 2)
template <class XA, class YA, class SW, class CW>
__device__ __forceinline__ void method_slopes(int method, const XA& x, const YA& y, SW& s, CW& scratch, int n) {
    if (method == IVS_PCHIP) pchip_slopes(x, y, s, n);
    else if (method == IVS_AKIMA) akima_slopes(x, y, s, n);
    else if (method == IVS_QUADRATIC) quadratic_coeffs(x, y, s, scratch, n);
    else nak_slopes(x, y, s, scratch, n);
}

// one evaluation of any method on n valid knots (n >= 1 and n >= method_min_knots), j = find_interval(x, n, xq)
template <class XA, class YA, class SA>
__device__ __forceinline__ double eval_method(int method, const XA& x, const YA& y, const SA& s, int n, int j, double xq) {
    switch (method) {
        case IVS_LINEAR: return eval_linear(x, y, n, j, xq, true);
        case IVS_SLINEAR: return eval_linear(x, y, n, j, xq, false);
        case IVS_NEAREST: return eval_nearest(x, y, n, xq);
        case IVS_ZERO: return eval_zero(x, y, n, j, xq);
        case IVS_FROM_DERIVATIVES: return eval_bpoly_linear(x, y, n, j, xq);
        case IVS_QUADRATIC: return eval_quadratic(x, s, n, j, xq);
        case IVS_BARYCENTRIC: return eval_barycentric(x, y, s, n, xq);
        case IVS_KROGH: return eval_krogh(x, s, n, xq);
        case IVS_PAD: return eval_pad(x, y, n, j);
        case IVS_BFILL: return eval_bfill(x, y, n, j, xq);
        default: return eval_cubic(x, y, s, n, j, xq, method_extrapolates_right(method));
    }
}

}  // namespace ivs
