/* CPU oracle in C.  TEST INFRASTRUCTURE ONLY (checker for large batches and the cpu_baseline leg of bench.py).
 *
 * Same arithmetic, in the same order, as oracle/ivs_oracle.py, which restates
 *   numpy  arr_interp (numpy/_core/src/multiarray/compiled_base.c)          -> lerp()
 *   pandas _interpolate_1d NaN rules (pandas/core/missing.py:435-546)       -> interp1d()
 *   scipy  not-a-knot slopes (scipy/interpolate/_cubic.py:748-884)          -> nak_slopes()
 *   scipy  PPoly coefficient build + Horner (_cubic.py:170-180)             -> hermite()
 *   scipy  PchipInterpolator._find_derivatives/_edge_case (_cubic.py:248-309) -> pchip_slopes()
 *   scipy  Akima1DInterpolator.__init__ (_cubic.py:510-541)                 -> akima_slopes()
 *   scipy  interp1d kind nearest / zero (_interpolate.py:327-328, 486-501)  -> nearest(), zero()
 *   scipy  BPoly.from_derivatives on values (_ppoly.pyx evaluate_bpoly1)    -> bpoly_linear()
 *   scipy  make_interp_spline(k=2) (_bsplines.py; knots at the site midpoints) -> quadratic_coeffs(), quadratic_eval()
 *   pandas pad / backfill (generic.py NDFrame.interpolate -> pad_or_backfill) -> pad(), bfill()
 * reached by the reference through Series.interpolate (reference src/interpolation/core.py:61).
 * Pinned by tests/test_c_oracle.py against the NumPy oracle and the reference's golden vectors.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp; contraction off keeps 'linear' bit-equal to numpy)
 * Never linked or loaded by the product path. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { LINEAR = 0, CUBIC = 1, CUBICSPLINE = 2, SLINEAR = 3, NEAREST = 4, ZERO = 5, PCHIP = 6, AKIMA = 7, FROM_DERIVATIVES = 8,
       QUADRATIC = 9, PAD = 12, BFILL = 13 };
#define NMAX 1024

static int supported(int m) { return (m >= LINEAR && m <= QUADRATIC) || m == PAD || m == BFILL; }
static int min_knots(int m) {
    switch (m) {
        case LINEAR: case PAD: case BFILL: return 0;
        case CUBIC: return 4;
        case NEAREST: case ZERO: return 1;
        case AKIMA: case QUADRATIC: return 3;
        default: return 2;
    }
}

static int interval(const double* x, int n, double xq) { /* largest j with x[j] <= xq, or -1 */
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (x[mid] <= xq) lo = mid + 1; else hi = mid; }
    return lo - 1;
}

static double lerp(const double* x, const double* y, int n, double xq, int right_hold) {
    int j = interval(x, n, xq);
    if (j < 0) return NAN;
    if (j >= n - 1) return (right_hold || xq == x[n - 1]) ? y[n - 1] : NAN;
    if (x[j] == xq) return y[j];
    double slope = (y[j + 1] - y[j]) / (x[j + 1] - x[j]);
    double r = slope * (xq - x[j]) + y[j];
    if (isnan(r)) { r = slope * (xq - x[j + 1]) + y[j + 1]; if (isnan(r) && y[j] == y[j + 1]) r = y[j]; }
    return r;
}

static void nak_slopes(const double* x, const double* y, int n, double* s) {
    double dx[NMAX], dl[NMAX], lo[NMAX], di[NMAX], up[NMAX], rhs[NMAX], cp[NMAX], dp[NMAX];
    for (int i = 0; i < n - 1; ++i) { dx[i] = x[i + 1] - x[i]; dl[i] = (y[i + 1] - y[i]) / dx[i]; }
    if (n == 2) { s[0] = s[1] = dl[0]; return; }
    if (n == 3) {
        double b0 = 2.0 * dl[0], b1 = 3.0 * (dx[0] * dl[1] + dx[1] * dl[0]), b2 = 2.0 * dl[1];
        double s1 = (b1 - dx[1] * b0 - dx[0] * b2) / (2.0 * (dx[0] + dx[1]) - dx[1] - dx[0]);
        s[0] = b0 - s1; s[1] = s1; s[2] = b2 - s1; return;
    }
    for (int i = 1; i < n - 1; ++i) {
        lo[i] = dx[i]; di[i] = 2.0 * (dx[i - 1] + dx[i]); up[i] = dx[i - 1];
        rhs[i] = 3.0 * (dx[i] * dl[i - 1] + dx[i - 1] * dl[i]);
    }
    double d = x[2] - x[0];
    lo[0] = 0.0; di[0] = dx[1]; up[0] = d;
    rhs[0] = ((dx[0] + 2.0 * d) * dx[1] * dl[0] + dx[0] * dx[0] * dl[1]) / d;
    d = x[n - 1] - x[n - 3];
    lo[n - 1] = d; di[n - 1] = dx[n - 3]; up[n - 1] = 0.0;
    rhs[n - 1] = (dx[n - 2] * dx[n - 2] * dl[n - 3] + (2.0 * d + dx[n - 2]) * dx[n - 3] * dl[n - 2]) / d;
    cp[0] = up[0] / di[0]; dp[0] = rhs[0] / di[0];
    for (int i = 1; i < n; ++i) {
        double w = di[i] - lo[i] * cp[i - 1];
        cp[i] = up[i] / w; dp[i] = (rhs[i] - lo[i] * dp[i - 1]) / w;
    }
    s[n - 1] = dp[n - 1];
    for (int i = n - 2; i >= 0; --i) s[i] = dp[i] - cp[i] * s[i + 1];
}

static double sgn(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }

static double pchip_edge(double h0, double h1, double m0, double m1) {
    double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
    if (sgn(d) != sgn(m0)) return 0.0;
    if (sgn(m0) != sgn(m1) && fabs(d) > 3.0 * fabs(m0)) return 3.0 * m0;
    return d;
}

static void pchip_slopes(const double* x, const double* y, int n, double* s) {
    double hk[NMAX], mk[NMAX];
    for (int i = 0; i < n - 1; ++i) { hk[i] = x[i + 1] - x[i]; mk[i] = (y[i + 1] - y[i]) / hk[i]; }
    if (n == 2) { s[0] = s[1] = mk[0]; return; }
    for (int k = 1; k < n - 1; ++k) {
        int cond = sgn(mk[k]) != sgn(mk[k - 1]) || mk[k] == 0.0 || mk[k - 1] == 0.0;
        double w1 = 2.0 * hk[k] + hk[k - 1], w2 = hk[k] + 2.0 * hk[k - 1];
        double whmean = (w1 / mk[k - 1] + w2 / mk[k]) / (w1 + w2);
        s[k] = cond ? 0.0 : 1.0 / whmean;
    }
    s[0] = pchip_edge(hk[0], hk[1], mk[0], mk[1]);
    s[n - 1] = pchip_edge(hk[n - 2], hk[n - 3], mk[n - 2], mk[n - 3]);
}

static void akima_slopes(const double* x, const double* y, int n, double* s) {   /* n >= 3 */
    double mb[NMAX + 3];
    double* m = mb;                                   /* m[i + 2] = secant i */
    for (int i = 0; i < n - 1; ++i) m[i + 2] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
    m[1] = 2.0 * m[2] - m[3];
    m[0] = 2.0 * m[1] - m[2];
    m[n + 1] = 2.0 * m[n] - m[n - 1];
    m[n + 2] = 2.0 * m[n + 1] - m[n];
    double fmax = -INFINITY;
    for (int i = 0; i < n; ++i) {
        double f12 = fabs(m[i + 3] - m[i + 2]) + fabs(m[i + 1] - m[i]);
        if (f12 > fmax) fmax = f12;
    }
    for (int i = 0; i < n; ++i) {
        double f1 = fabs(m[i + 3] - m[i + 2]), f2 = fabs(m[i + 1] - m[i]), f12 = f1 + f2;
        s[i] = f12 > 1e-9 * fmax ? (f1 * m[i + 1] + f2 * m[i + 2]) / f12 : 0.5 * (m[i + 3] + m[i]);
    }
}

static double hermite(const double* x, const double* y, const double* s, int n, double xq, int extrap) {
    int j = interval(x, n, xq);
    if (j < 0) return NAN;
    if (!extrap && !(xq <= x[n - 1])) return NAN;
    int jj = j > n - 2 ? n - 2 : j;
    double h = x[jj + 1] - x[jj];
    double dl = (y[jj + 1] - y[jj]) / h;
    double t = (s[jj] + s[jj + 1] - 2.0 * dl) / h;
    double c0 = t / h, c1 = (dl - s[jj]) / h - t, u = xq - x[jj];
    return ((c0 * u + c1) * u + s[jj]) * u + y[jj];
}

/* ---- step / fill / Bernstein rules (oracle nearest_eval, zero_eval, pad_eval, bfill_eval, bpoly_linear_eval) */
static double nearest(const double* x, const double* y, int n, double xq) {
    if (!(xq >= x[0] && xq <= x[n - 1])) return NAN;
    int lo = 0, hi = n - 1;            /* searchsorted(x/2 + x_next/2, xq, side='left') */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (x[mid] / 2.0 + x[mid + 1] / 2.0 < xq) lo = mid + 1; else hi = mid; }
    return y[lo];
}
static double zero(const double* x, const double* y, int n, double xq) {
    int j = interval(x, n, xq);
    if (j < 0 || !(xq <= x[n - 1])) return NAN;
    return y[j > n - 1 ? n - 1 : j];
}
static double pad(const double* x, const double* y, int n, double xq) {
    int j = interval(x, n, xq);
    return j < 0 ? NAN : y[j > n - 1 ? n - 1 : j];
}
static double bfill(const double* x, const double* y, int n, double xq) {
    if (isnan(xq)) return NAN;
    int lo = 0, hi = n;                /* first knot >= xq */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (x[mid] < xq) lo = mid + 1; else hi = mid; }
    return lo < n ? y[lo] : NAN;
}
static double bpoly_linear(const double* x, const double* y, int n, double xq) {
    int j = interval(x, n, xq);
    if (j < 0 || !(xq <= x[n - 1])) return NAN;
    int jj = j > n - 2 ? n - 2 : j;
    double s = (xq - x[jj]) / (x[jj + 1] - x[jj]);
    double a = y[jj] * (1.0 - s), b = y[jj + 1] * s;
    return a + b;
}

/* ---- quadratic B-spline (oracle quadratic_coeffs / quadratic_eval) */
static double quad_knot(const double* x, int n, int j) {      /* t_j, j = 0..n+2 */
    if (j <= 2) return x[0];
    if (j >= n) return x[n - 1];
    return (x[j - 1] + x[j - 2]) / 2.0;
}
static void quad_basis(const double* x, int n, int ell, double xv, double* h0, double* h1, double* h2) {
    double tm1 = quad_knot(x, n, ell - 1), t0 = quad_knot(x, n, ell), t1 = quad_knot(x, n, ell + 1), t2 = quad_knot(x, n, ell + 2);
    double w = 1.0 / (t1 - t0);
    double a0 = w * (t1 - xv), a1 = w * (xv - t0);
    double w1 = a0 / (t1 - tm1);
    *h0 = 0.0 + w1 * (t1 - xv);
    *h1 = w1 * (xv - tm1);
    double w2 = a1 / (t2 - t0);
    *h1 = *h1 + w2 * (t2 - xv);
    *h2 = w2 * (xv - t0);
}
static void quadratic_coeffs(const double* x, const double* y, int n, double* c) {
    double cp[NMAX], dp[NMAX];
    cp[0] = 0.0 / 1.0; dp[0] = y[0] / 1.0;
    for (int i = 1; i < n; ++i) {
        double lo = 0.0, di = 1.0, up = 0.0;
        if (i < n - 1) quad_basis(x, n, i + 1, x[i], &lo, &di, &up);
        double w = di - lo * cp[i - 1];
        cp[i] = up / w; dp[i] = (y[i] - lo * dp[i - 1]) / w;
    }
    c[n - 1] = dp[n - 1];
    for (int i = n - 2; i >= 0; --i) c[i] = dp[i] - cp[i] * c[i + 1];
}
static double quadratic_eval(const double* x, const double* c, int n, double xq) {
    if (!(xq >= x[0] && xq <= x[n - 1])) return NAN;
    int q = 0;                         /* interior knots mid_1..mid_{n-3} that are <= xq (searchsorted side='right') */
    for (int i = 1; i <= n - 3; ++i) if ((x[i + 1] + x[i]) / 2.0 <= xq) q = i; else break;
    int ell = q + 2;
    double h0, h1, h2;
    quad_basis(x, n, ell, xq, &h0, &h1, &h2);
    return h0 * c[ell - 2] + h1 * c[ell - 1] + h2 * c[ell];
}

/* one masked-knot 1-D op: xs/ys stride, NaN = missing; returns status (1 = too few knots) */
static int interp1d(const double* xk, const double* yk, int ystride, int n, const double* xq, int m, double* out,
                    int ostride, int method) {
    double xv[NMAX], yv[NMAX], s[NMAX];
    int nv = 0;
    for (int i = 0; i < n; ++i) { double v = yk[(size_t)i * ystride]; if (!isnan(v)) { xv[nv] = xk[i]; yv[nv] = v; ++nv; } }
    if (nv == 0 || nv < min_knots(method)) {
        for (int q = 0; q < m; ++q) out[(size_t)q * ostride] = NAN;
        return nv == 0 ? 0 : 1;
    }
    if (method == LINEAR || method == SLINEAR) {
        for (int q = 0; q < m; ++q) out[(size_t)q * ostride] = lerp(xv, yv, nv, xq[q], method == LINEAR);
        return 0;
    }
    if (method == NEAREST || method == ZERO || method == FROM_DERIVATIVES || method == PAD || method == BFILL) {
        for (int q = 0; q < m; ++q) {
            double v;
            if (method == NEAREST) v = nearest(xv, yv, nv, xq[q]);
            else if (method == ZERO) v = zero(xv, yv, nv, xq[q]);
            else if (method == PAD) v = pad(xv, yv, nv, xq[q]);
            else if (method == BFILL) v = bfill(xv, yv, nv, xq[q]);
            else v = bpoly_linear(xv, yv, nv, xq[q]);
            out[(size_t)q * ostride] = v;
        }
        return 0;
    }
    if (method == QUADRATIC) {
        quadratic_coeffs(xv, yv, nv, s);
        for (int q = 0; q < m; ++q) out[(size_t)q * ostride] = quadratic_eval(xv, s, nv, xq[q]);
        return 0;
    }
    if (method == PCHIP) pchip_slopes(xv, yv, nv, s);
    else if (method == AKIMA) akima_slopes(xv, yv, nv, s);
    else nak_slopes(xv, yv, nv, s);
    for (int q = 0; q < m; ++q)
        out[(size_t)q * ostride] = hermite(xv, yv, s, nv, xq[q], method == CUBICSPLINE || method == PCHIP);
    return 0;
}

void ivs_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int ivs_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* surfaces: strike pass then maturity pass (oracle surface()).  k_off NULL = uniform nK.  Returns 0, or -1 on
 * sizes beyond NMAX. */
int ivs_oracle_surface_batch(const double* K, const int64_t* k_off, int64_t k_stride, int nK, const double* T,
                             int64_t t_stride, int nT, const double* sigma, int64_t B, const double* Kq,
                             int64_t kq_stride, int mK, const double* Tq, int64_t tq_stride, int mT, double* out,
                             int32_t* status, int method) {
    if (nK > NMAX || nT > NMAX || mK > 65536) return -1;
    if (!supported(method)) return -2;
#pragma omp parallel
    {
        double* Z = (double*)malloc(sizeof(double) * (size_t)nT * mK);
#pragma omp for schedule(dynamic, 64)
        for (int64_t b = 0; b < B; ++b) {
            int64_t ko = k_off ? k_off[b] : b * k_stride;
            int n = k_off ? (int)(k_off[b + 1] - k_off[b]) : nK;
            const double* Kb = K + ko;
            const double* sb = k_off ? sigma + (int64_t)nT * ko : sigma + b * (int64_t)nT * nK;
            const double* Tb = T + b * t_stride;
            const double* Kqb = Kq + b * kq_stride;
            const double* Tqb = Tq + b * tq_stride;
            double* ob = out + b * (int64_t)mT * mK;
            int st = 0;
            for (int t = 0; t < nT; ++t) st |= interp1d(Kb, sb + (size_t)t * n, 1, n, Kqb, mK, Z + (size_t)t * mK, 1, method);
            for (int q = 0; q < mK; ++q) st |= interp1d(Tb, Z + q, mK, nT, Tqb, mT, ob + q, mK, method);
            if (status) status[b] = st;
        }
        free(Z);
    }
    return 0;
}

/* CSR batch of 1-D series, C channels (oracle interp1d_batch()); xq NULL = 0..m-1 */
int ivs_oracle_interp1d_batch(const double* xk, const double* yk, int64_t yk_stride, const int64_t* knot_off, int64_t S,
                              int C, const double* xq, const int64_t* q_off, double* out, int64_t out_stride,
                              int32_t* status, int method) {
    int rc = 0;
    if (!supported(method)) return -2;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t s = 0; s < S; ++s) {
        int64_t a = knot_off[s], n = knot_off[s + 1] - a, qa = q_off[s], m = q_off[s + 1] - qa;
        if (n > NMAX) { rc = -1; continue; }
        double* q = (double*)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
        for (int64_t i = 0; i < m; ++i) q[i] = xq ? xq[qa + i] : (double)i;
        for (int c = 0; c < C; ++c)
            status[s * C + c] = interp1d(xk + a, yk + c * yk_stride + a, 1, (int)n, q, (int)m, out + c * out_stride + qa, 1, method);
        free(q);
    }
    return rc;
}
