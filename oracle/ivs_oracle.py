"""CPU oracle for the IV-interpolation hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy restatement of the arithmetic the reference reaches
through ``IVInterpolator.interpolate_symbol`` (reference
``src/interpolation/core.py:16-85``).  The reference holds no numerics of its
own: ``core.py:61`` calls ``pandas.Series.interpolate(method=...)`` which lands
in third-party code that is NOT under /root/reference:

* pandas 2.3.3  ``pandas/core/missing.py:435-546`` (``_interpolate_1d``: knot
  mask, leading-NaN preservation, ``np.interp`` for 'linear'/'index'/'values',
  scipy for everything else) and ``:549-617`` (``_interpolate_scipy_wrapper``);
* numpy 2.2.6   ``np.interp`` (``numpy/_core/src/multiarray/compiled_base.c``,
  ``arr_interp``): ``slope*(x - xp[j]) + fp[j]``, exact knot hit returns
  ``fp[j]``, right of the last knot returns ``fp[-1]``;
* scipy 1.15.3  ``scipy/interpolate/_cubic.py:728-890`` (not-a-knot system in
  the knot slopes; n==2 line, n==3 parabola), ``_interpolate.py:277-279,397``
  (``interp1d(kind='cubic')`` -> ``make_interp_spline(k=3)``, NaN outside the
  knot hull, >= 4 knots required).

Pinning: every function here is checked in ``tests/test_oracle.py`` against
golden vectors produced by importing the REAL reference in the authoring
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz|json``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file.  The product path (``iv_interpolation_amd``) never
does: it fails loudly when the HIP library is missing.

The 2-D surface composition (strike pass, then maturity pass) is this
repository's own documented extension: each pass is the 1-D
operator above, so each pass is checkable against the reference's 1-D
semantics.
"""
from __future__ import annotations

import numpy as np

# method codes shared with include/ivs.h
LINEAR = 0       # pandas 'linear' / 'index' / 'values'  (np.interp; leading NaN, trailing hold)
CUBIC = 1        # pandas 'cubic'      (interp1d kind=3: not-a-knot, NaN outside hull, >=4 knots)
CUBICSPLINE = 2  # pandas 'cubicspline' (CubicSpline not-a-knot: n==2 line, n==3 parabola,
                 #                       leading NaN, trailing extrapolated)
SLINEAR = 3      # pandas 'slinear'    (interp1d linear: np.interp inside, NaN outside hull)
NEAREST = 4      # pandas 'nearest'    (interp1d nearest: ties go to the left knot, NaN outside hull)
ZERO = 5         # pandas 'zero'       (interp1d kind=0: left knot's value, NaN outside hull)
PCHIP = 6        # pandas 'pchip'      (PchipInterpolator: leading NaN, trailing extrapolated)
AKIMA = 7        # pandas 'akima'      (Akima1DInterpolator: NaN outside hull; >= 3 knots, scipy's n == 2 reads
                 #                       uninitialised memory)
FROM_DERIVATIVES = 8   # pandas 'from_derivatives' / 'piecewise_polynomial' (BPoly, values only = linear in
                       #                       Bernstein form, NaN outside hull)

QUADRATIC = 9    # pandas 'quadratic'  (interp1d kind=2 -> make_interp_spline(k=2): quadratic B-spline with knots at the
                 #                       midpoints of the data sites, NaN outside hull, >= 3 knots)

BARYCENTRIC = 10  # pandas 'barycentric' (scipy barycentric_interpolate: ONE polynomial through all valid knots;
                  #                       leading NaN kept, trailing NaN extrapolated by the polynomial, >= 1 knot)
KROGH = 11        # pandas 'krogh'       (scipy krogh_interpolate: the same polynomial in Newton form)
PAD = 12          # pandas 'pad' / 'ffill'   (NDFrame.interpolate -> BlockManager.pad_or_backfill, pandas/core/generic.py:
                  #                       `if method.lower() in fillna_methods`; limit_direction defaults to 'forward'):
                  #                       the last valid knot at or before the row; leading NaN kept, trailing held
BFILL = 13        # pandas 'bfill' / 'backfill' (same path, limit_direction 'backward'): the first valid knot at or after
                  #                       the row; trailing NaN kept, leading filled with the first knot
POLY_MAX_KNOTS = 32   # beyond this the interpolating polynomial on an (almost) equispaced grid is numerical noise in
                      # the reference itself (scipy's two routes disagree at 1e-8 of the curve's scale at 32 knots, and
                      # its barycentric weights even change from run to run: random node permutation); the engine
                      # reports ST_ILL_CONDITIONED there and the symbol becomes None (documented deviation)

METHOD_CODES = {
    "linear": LINEAR, "index": LINEAR, "values": LINEAR, "quadratic": QUADRATIC, "barycentric": BARYCENTRIC, "krogh": KROGH,
    "cubic": CUBIC, "cubicspline": CUBICSPLINE, "slinear": SLINEAR,
    "nearest": NEAREST, "zero": ZERO, "pchip": PCHIP, "akima": AKIMA,
    "from_derivatives": FROM_DERIVATIVES, "piecewise_polynomial": FROM_DERIVATIVES,
    "pad": PAD, "ffill": PAD, "bfill": BFILL, "backfill": BFILL,
}



def method_code(name):
    """Code of a pandas method name or None: the fill methods match case-insensitively (pandas: `method.lower() in
    fillna_methods`), every other name exactly."""
    if name in METHOD_CODES:
        return METHOD_CODES[name]
    if isinstance(name, str) and name.lower() in ("pad", "ffill", "bfill", "backfill"):
        return METHOD_CODES[name.lower()]
    return None


# status codes shared with include/ivs.h
ST_OK = 0
ST_TOO_FEW_KNOTS = 1   # the reference raises inside scipy -> interpolate_symbol returns None
ST_ILL_CONDITIONED = 4 # 'barycentric' / 'krogh' with more than POLY_MAX_KNOTS valid knots


def min_knots(method: int) -> int:
    """Fewest valid knots the reference accepts before scipy raises."""
    return {LINEAR: 0, CUBIC: 4, CUBICSPLINE: 2, SLINEAR: 2, NEAREST: 1, ZERO: 1, PCHIP: 2, AKIMA: 3,
            FROM_DERIVATIVES: 2, QUADRATIC: 3, BARYCENTRIC: 1, KROGH: 1, PAD: 0, BFILL: 0}[method]


# --------------------------------------------------------------------------- linear
def _interval(xv: np.ndarray, xq: np.ndarray) -> np.ndarray:
    """Largest j with xv[j] <= xq (or -1): np.interp's binary_search_with_guess result."""
    return np.searchsorted(xv, xq, side="right") - 1


def lerp_eval(xv, yv, xq, *, right_hold: bool):
    """np.interp restated (compiled_base.c arr_interp) on valid knots xv/yv.

    Left of the first knot -> NaN (pandas preserves leading NaNs, missing.py:494-495;
    interp1d's bounds fill for 'slinear').  Right of the last knot -> yv[-1] when
    ``right_hold`` (np.interp's rval) else NaN ('slinear').
    """
    xv = np.asarray(xv, np.float64); yv = np.asarray(yv, np.float64)
    xq = np.asarray(xq, np.float64)
    n = xv.size
    out = np.full(xq.shape, np.nan)
    if n == 0:
        return out
    j = _interval(xv, xq)
    inside = (j >= 0) & (j < n - 1)
    jj = np.clip(j, 0, max(n - 2, 0))
    if n >= 2:
        x0 = xv[jj]; y0 = yv[jj]
        with np.errstate(invalid="ignore", divide="ignore"):
            slope = (yv[jj + 1] - y0) / (xv[jj + 1] - x0)
            r = slope * (xq - x0) + y0          # two roundings, no FMA (matches numpy's C)
            # "If we get nan in one direction, try the other" (arr_interp)
            bad = np.isnan(r)
            r2 = slope * (xq - xv[jj + 1]) + yv[jj + 1]
            r = np.where(bad, r2, r)
            r = np.where(np.isnan(r) & (yv[jj + 1] == y0), y0, r)
        r = np.where(x0 == xq, y0, r)           # exact knot hit returns fp[j]
        out = np.where(inside, r, out)
    at_last = (j == n - 1) & (xq == xv[n - 1])
    out = np.where(at_last, yv[n - 1], out)
    if right_hold:
        out = np.where(j >= n - 1, yv[n - 1], out)
    return out


# --------------------------------------------------------------------------- cubic
def nak_slopes(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Knot slopes of the not-a-knot cubic spline (scipy _cubic.py:748-884).

    n == 2 -> both slopes = the secant (:748-752); n == 3 -> the parabola through
    the three points (:761-779); n >= 4 -> tridiagonal system, solved here by the
    Thomas recurrence without pivoting (scipy calls LAPACK gtsv; same solution).
    """
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    n = x.size
    if n < 2:
        raise ValueError("need at least 2 knots")
    dx = np.diff(x)
    delta = np.diff(y) / dx
    if n == 2:
        return np.array([delta[0], delta[0]])
    if n == 3:
        # A = [[1,1,0],[dx1, 2(dx0+dx1), dx0],[0,1,1]], b = [2 d0, 3(dx0 d1 + dx1 d0), 2 d1]
        b0 = 2.0 * delta[0]; b1 = 3.0 * (dx[0] * delta[1] + dx[1] * delta[0]); b2 = 2.0 * delta[1]
        # eliminate s0 = b0 - s1 and s2 = b2 - s1
        s1 = (b1 - dx[1] * b0 - dx[0] * b2) / (2.0 * (dx[0] + dx[1]) - dx[1] - dx[0])
        return np.array([b0 - s1, s1, b2 - s1])
    lo = np.empty(n); di = np.empty(n); up = np.empty(n); rhs = np.empty(n)
    # interior rows i = 1..n-2 (:787-803)
    lo[1:-1] = dx[1:]
    di[1:-1] = 2.0 * (dx[:-1] + dx[1:])
    up[1:-1] = dx[:-1]
    rhs[1:-1] = 3.0 * (dx[1:] * delta[:-1] + dx[:-1] * delta[1:])
    # not-a-knot rows (:857-876)
    d = x[2] - x[0]
    lo[0] = 0.0; di[0] = dx[1]; up[0] = d
    rhs[0] = ((dx[0] + 2.0 * d) * dx[1] * delta[0] + dx[0] ** 2 * delta[1]) / d
    d = x[-1] - x[-3]
    lo[-1] = d; di[-1] = dx[-2]; up[-1] = 0.0
    rhs[-1] = (dx[-1] ** 2 * delta[-2] + (2.0 * d + dx[-1]) * dx[-2] * delta[-1]) / d
    # Thomas
    cp = np.empty(n); dp = np.empty(n)
    cp[0] = up[0] / di[0]; dp[0] = rhs[0] / di[0]
    for i in range(1, n):
        w = di[i] - lo[i] * cp[i - 1]
        cp[i] = up[i] / w
        dp[i] = (rhs[i] - lo[i] * dp[i - 1]) / w
    s = np.empty(n)
    s[-1] = dp[-1]
    for i in range(n - 2, -1, -1):
        s[i] = dp[i] - cp[i] * s[i + 1]
    return s


def hermite_eval(xv, yv, s, xq, *, extrapolate_right: bool):
    """Evaluate the C1 piecewise cubic with knot slopes ``s`` as scipy's PPoly does
    (_cubic.py:170-180 coefficient build + Horner in ``x - x[j]``).

    Left of the hull -> NaN (pandas leading-NaN rule / interp1d fill).  Right of the
    hull -> NaN, or the last piece extended when ``extrapolate_right``
    ('cubicspline': CubicSpline(extrapolate=True) and pandas keeps trailing fills).
    """
    xv = np.asarray(xv, np.float64); yv = np.asarray(yv, np.float64); s = np.asarray(s, np.float64)
    xq = np.asarray(xq, np.float64)
    n = xv.size
    j = _interval(xv, xq)
    jj = np.clip(j, 0, n - 2)
    h = xv[jj + 1] - xv[jj]
    delta = (yv[jj + 1] - yv[jj]) / h
    t = (s[jj] + s[jj + 1] - 2.0 * delta) / h
    c0 = t / h
    c1 = (delta - s[jj]) / h - t
    c2 = s[jj]
    c3 = yv[jj]
    u = xq - xv[jj]
    r = ((c0 * u + c1) * u + c2) * u + c3
    ok = (j >= 0) & ((xq <= xv[n - 1]) | extrapolate_right)
    return np.where(ok, r, np.nan)


# --------------------------------------------------------------------------- local-slope methods / step methods
def pchip_slopes(x, y):
    """PchipInterpolator._find_derivatives (scipy _cubic.py:263-309) + _edge_case (:248-260)."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    n = x.size
    hk = x[1:] - x[:-1]
    mk = (y[1:] - y[:-1]) / hk
    if n == 2:
        return np.array([mk[0], mk[0]])
    smk = np.sign(mk)
    cond = (smk[1:] != smk[:-1]) | (mk[1:] == 0) | (mk[:-1] == 0)
    w1 = 2 * hk[1:] + hk[:-1]
    w2 = hk[1:] + 2 * hk[:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        whmean = (w1 / mk[:-1] + w2 / mk[1:]) / (w1 + w2)
    dk = np.zeros(n)
    dk[1:-1] = np.where(cond, 0.0, 1.0 / whmean)

    def edge(h0, h1, m0, m1):
        d = ((2 * h0 + h1) * m0 - h0 * m1) / (h0 + h1)
        if np.sign(d) != np.sign(m0):
            return 0.0
        if np.sign(m0) != np.sign(m1) and abs(d) > 3.0 * abs(m0):
            return 3.0 * m0
        return d
    dk[0] = edge(hk[0], hk[1], mk[0], mk[1])
    dk[-1] = edge(hk[-1], hk[-2], mk[-1], mk[-2])
    return dk


def akima_slopes(x, y):
    """Akima1DInterpolator.__init__ (scipy _cubic.py:510-541), method='akima'.  n >= 3."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    n = x.size
    m = np.empty(n + 3)
    m[2:-2] = np.diff(y) / np.diff(x)
    m[1] = 2.0 * m[2] - m[3]
    m[0] = 2.0 * m[1] - m[2]
    m[-2] = 2.0 * m[-3] - m[-4]
    m[-1] = 2.0 * m[-2] - m[-3]
    t = 0.5 * (m[3:] + m[:-3])
    dm = np.abs(np.diff(m))
    f1 = dm[2:]; f2 = dm[:-2]
    f12 = f1 + f2
    ind = f12 > 1e-9 * np.max(f12, initial=-np.inf)
    with np.errstate(invalid="ignore", divide="ignore"):
        tt = (f1 * m[1:-2] + f2 * m[2:-1]) / f12
    return np.where(ind, tt, t)


def nearest_eval(xv, yv, xq):
    """interp1d kind='nearest' (scipy _interpolate.py:327-328, 486-501): x_bds = x/2 + x/2 shifted, ties to the left."""
    xv = np.asarray(xv, np.float64); xq = np.asarray(xq, np.float64)
    n = xv.size
    h = xv / 2.0
    bds = h[1:] + h[:-1]
    idx = np.clip(np.searchsorted(bds, xq, side="left"), 0, n - 1)
    out = np.asarray(yv, np.float64)[idx]
    return np.where((xq >= xv[0]) & (xq <= xv[-1]), out, np.nan)


def zero_eval(xv, yv, xq):
    """interp1d kind='zero' (order-0 spline): the left knot's value on [x_j, x_j+1), y[-1] at the last knot."""
    xv = np.asarray(xv, np.float64); xq = np.asarray(xq, np.float64)
    j = _interval(xv, xq)
    out = np.asarray(yv, np.float64)[np.clip(j, 0, xv.size - 1)]
    return np.where((j >= 0) & (xq <= xv[-1]), out, np.nan)


def pad_eval(xv, yv, xq):
    """pandas pad (forward fill) restated on knot coordinates: the value of the last knot at or before xq; NaN left of
    the first knot (pandas/_libs/algos.pyx pad_inplace walks the column once and carries the last valid value)."""
    xv = np.asarray(xv, np.float64); xq = np.asarray(xq, np.float64)
    j = _interval(xv, xq)
    out = np.asarray(yv, np.float64)[np.clip(j, 0, xv.size - 1)]
    return np.where(j >= 0, out, np.nan)


def bfill_eval(xv, yv, xq):
    """pandas backfill restated: the value of the first knot at or after xq; NaN right of the last knot
    (algos.pyx backfill_inplace = pad_inplace on the reversed column)."""
    xv = np.asarray(xv, np.float64); xq = np.asarray(xq, np.float64)
    k = np.searchsorted(xv, xq, side="left")             # first knot >= xq
    out = np.asarray(yv, np.float64)[np.clip(k, 0, xv.size - 1)]
    return np.where((k < xv.size) & ~np.isnan(xq), out, np.nan)


def bpoly_linear_eval(xv, yv, xq):
    """BPoly.from_derivatives with function values only (pandas 'from_derivatives'): degree-1 Bernstein form
    c0*(1-s) + c1*s, s = (x - xa)/(xb - xa) (scipy _ppoly.pyx evaluate_bpoly1, k == 1); NaN outside the hull."""
    xv = np.asarray(xv, np.float64); yv = np.asarray(yv, np.float64); xq = np.asarray(xq, np.float64)
    n = xv.size
    j = _interval(xv, xq)
    jj = np.clip(j, 0, n - 2)
    s = (xq - xv[jj]) / (xv[jj + 1] - xv[jj])
    r = yv[jj] * (1.0 - s) + yv[jj + 1] * s
    return np.where((j >= 0) & (xq <= xv[-1]), r, np.nan)


# --------------------------------------------------------------------------- quadratic B-spline
def quadratic_coeffs(x, y):
    """make_interp_spline(x, y, k=2) (scipy _bsplines.py: knot vector _not_a_knot for even k = the midpoints of the data
    sites with the first and the last one dropped, triple end knots; collocation system solved by LAPACK gbsv).
    With those knots every data site sees exactly three consecutive basis functions: the collocation matrix is
    TRIDIAGONAL (row 0 = [1, 0, ...], row n-1 = [..., 0, 1]) and totally positive, so plain elimination without
    pivoting is stable.  Returns the n B-spline coefficients."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    n = x.size
    lo = np.zeros(n); di = np.ones(n); up = np.zeros(n)
    if n > 2:                                   # interior rows, vectorised (_quad_row per site: same operations)
        t = quadratic_knots(x)
        ell = np.arange(1, n - 1) + 1
        v = x[1:n - 1]
        tm1, t0, t1, t2 = t[ell - 1], t[ell], t[ell + 1], t[ell + 2]
        w = 1.0 / (t1 - t0)
        a0 = w * (t1 - v); a1 = w * (v - t0)
        w1 = a0 / (t1 - tm1)
        lo[1:n - 1] = 0.0 + w1 * (t1 - v)
        h1 = w1 * (v - tm1)
        w2 = a1 / (t2 - t0)
        di[1:n - 1] = h1 + w2 * (t2 - v)
        up[1:n - 1] = w2 * (v - t0)
    cp = np.zeros(n); dp = np.zeros(n)
    cp[0] = up[0] / di[0]; dp[0] = y[0] / di[0]
    for i in range(1, n):
        w = di[i] - lo[i] * cp[i - 1]
        cp[i] = up[i] / w
        dp[i] = (y[i] - lo[i] * dp[i - 1]) / w
    c = np.zeros(n)
    c[n - 1] = dp[n - 1]
    for i in range(n - 2, -1, -1):
        c[i] = dp[i] - cp[i] * c[i + 1]
    return c


def _quad_knot(x, n, j):
    """Knot t_j of the quadratic spline, j = 0..n+2: x_0 three times, mid_1..mid_{n-3}, x_{n-1} three times."""
    if j <= 2:
        return x[0]
    if j >= n:
        return x[n - 1]
    i = j - 2                                   # interior knot mid_i, 1 <= i <= n-3
    return (x[i + 1] + x[i]) / 2


def _quad_basis(x, n, ell, xv):
    """The three quadratic B-splines that are non-zero on [t_ell, t_ell+1) at xv (de Boor's recurrence as in scipy's
    _deBoor_D): values of B_{ell-2}, B_{ell-1}, B_{ell}."""
    t = lambda j: _quad_knot(x, n, j)           # noqa: E731
    h = [1.0, 0.0, 0.0]
    for j in range(1, 3):
        hh = h[:j]
        h[0] = 0.0
        for m in range(1, j + 1):
            ind = ell + m
            xb = t(ind); xa = t(ind - j)
            if xb == xa:
                h[m] = 0.0
                continue
            w = hh[m - 1] / (xb - xa)
            h[m - 1] += w * (xb - xv)
            h[m] = w * (xv - xa)
    return h


def _quad_interval(x, n, xv):
    """ell with t_ell <= xv < t_ell+1 (ell in 2..n-1; the last interval is closed on the right)."""
    q = 0
    for i in range(1, n - 2):                   # interior knots mid_1..mid_{n-3}
        if (x[i + 1] + x[i]) / 2 <= xv:
            q = i
    return q + 2


def _quad_row(x, n, i):
    """Row i of the collocation matrix: (sub-diagonal, diagonal, super-diagonal)."""
    if i == 0:
        return 0.0, 1.0, 0.0
    if i == n - 1:
        return 0.0, 1.0, 0.0
    ell = _quad_interval(x, n, x[i])
    h = _quad_basis(x, n, ell, x[i])            # B_{ell-2..ell}; the row's columns are i-1, i, i+1 = ell-2.. for interior sites
    assert ell - 2 == i - 1, (ell, i)
    return h[0], h[1], h[2]


def quadratic_knots(x):
    """Full knot vector t_0..t_{n+2} (see _quad_knot)."""
    x = np.asarray(x, np.float64)
    n = x.size
    mid = (x[1:] + x[:-1]) / 2
    return np.r_[(x[0],) * 3, mid[1:n - 2], (x[n - 1],) * 3]


def quadratic_eval(xv, yv, c, xq):
    """Vectorised over the queries; same operation order per query as _quad_basis."""
    xv = np.asarray(xv, np.float64); xq = np.asarray(xq, np.float64)
    n = xv.size
    t = quadratic_knots(xv)
    v = xq.ravel()
    inside = (v >= xv[0]) & (v <= xv[-1])                       # interp1d bounds_error=False: NaN outside the hull
    vv = np.where(inside, v, xv[0])
    ell = np.clip(np.searchsorted(t[3:n], vv, side="right"), 0, n - 3) + 2      # interior knots <= v
    tm1, t0, t1, t2 = t[ell - 1], t[ell], t[ell + 1], t[ell + 2]
    w = 1.0 / (t1 - t0)
    a0 = w * (t1 - vv); a1 = w * (vv - t0)
    w1 = a0 / (t1 - tm1)
    h0 = 0.0 + w1 * (t1 - vv)
    h1 = w1 * (vv - tm1)
    w2 = a1 / (t2 - t0)
    h1 = h1 + w2 * (t2 - vv)
    h2 = w2 * (vv - t0)
    r = h0 * c[ell - 2] + h1 * c[ell - 1] + h2 * c[ell]
    return np.where(inside, r, np.nan).reshape(xq.shape)


# --------------------------------------------------------------------------- 1-D operator
# --------------------------------------------------------------------------- one polynomial through all knots
def barycentric_weights(xv):
    """scipy/interpolate/_polyint.py BarycentricInterpolator.__init__: w_j = 1 / prod_{k != j} (c (x_j - x_k)) with the
    capacity scaling c = 4 / (max - min).  scipy multiplies the factors in a RANDOM order (rng.permutation, unseeded):
    here in index order -- the reference's own values move in the last digits from run to run."""
    n = xv.size
    if n == 1:
        return np.ones(1)
    cap = 4.0 / (xv.max() - xv.min())
    w = np.empty(n)
    for j in range(n):
        d = cap * (xv[j] - xv)
        d[j] = 1.0
        w[j] = 1.0 / np.prod(d)
    return w


def barycentric_eval(xv, yv, xq):
    """BarycentricInterpolator._evaluate (second barycentric form; a query on a node returns that node's value).
    pandas' forward fill direction keeps NaN left of the first knot and fills everything to its right (the
    polynomial extrapolates)."""
    w = barycentric_weights(xv)
    out = np.full(xq.shape, np.nan)
    for i, x in enumerate(xq):
        if not (x >= xv[0]):
            continue
        c = x - xv
        hit = np.flatnonzero(c == 0)
        if hit.size:
            out[i] = yv[hit[0]]
            continue
        c = w / c
        out[i] = np.dot(c, yv) / np.sum(c)
    return out


def krogh_coeffs(xv, yv):
    """scipy KroghInterpolator (distinct nodes): divided differences by the recurrence of _polyint.py."""
    n = xv.size
    c = np.zeros(n); c[0] = yv[0]
    Vk = np.zeros(n)
    for k in range(1, n):
        Vk[0] = yv[k]
        for i in range(k):
            Vk[i + 1] = (c[i] - Vk[i]) / (xv[i] - xv[k])
        c[k] = Vk[k]
    return c


def krogh_eval(xv, yv, xq):
    """KroghInterpolator._evaluate: p = c_0 + sum_k c_k prod_{i<k} (x - x_i)."""
    c = krogh_coeffs(xv, yv)
    out = np.full(xq.shape, np.nan)
    for i, x in enumerate(xq):
        if not (x >= xv[0]):
            continue
        pi = 1.0; p = c[0]
        for k in range(1, xv.size):
            pi = (x - xv[k - 1]) * pi
            p = p + pi * c[k]
        out[i] = p
    return out


def interp1d(xk, yk, xq, method: int):
    """One masked-knot 1-D interpolation (one channel of core.py:58-61, generalised to
    real-valued knot/query coordinates).  NaN in ``yk`` = missing quote (not a knot).

    Returns (values at xq, status).  status == ST_TOO_FEW_KNOTS where the reference's
    scipy call raises (-> interpolate_symbol returns None); values are then all NaN.
    """
    xk = np.asarray(xk, np.float64); yk = np.asarray(yk, np.float64)
    xq = np.asarray(xq, np.float64)
    valid = ~np.isnan(yk)
    xv = xk[valid]; yv = yk[valid]
    n = xv.size
    if n == 0:
        # pandas: `if not valid.any(): return` -> column stays all-NaN, no scipy call
        return np.full(xq.shape, np.nan), ST_OK
    if n < min_knots(method):
        return np.full(xq.shape, np.nan), ST_TOO_FEW_KNOTS
    if method in (BARYCENTRIC, KROGH):
        if n > POLY_MAX_KNOTS:
            return np.full(xq.shape, np.nan), ST_ILL_CONDITIONED
        return (barycentric_eval if method == BARYCENTRIC else krogh_eval)(xv, yv, xq), ST_OK
    if method == PAD:
        return pad_eval(xv, yv, xq), ST_OK
    if method == BFILL:
        return bfill_eval(xv, yv, xq), ST_OK
    if method == LINEAR:
        return lerp_eval(xv, yv, xq, right_hold=True), ST_OK
    if method == SLINEAR:
        return lerp_eval(xv, yv, xq, right_hold=False), ST_OK
    if method == NEAREST:
        return nearest_eval(xv, yv, xq), ST_OK
    if method == ZERO:
        return zero_eval(xv, yv, xq), ST_OK
    if method == FROM_DERIVATIVES:
        return bpoly_linear_eval(xv, yv, xq), ST_OK
    if method == QUADRATIC:
        return quadratic_eval(xv, yv, quadratic_coeffs(xv, yv), xq), ST_OK
    if method == PCHIP:
        return hermite_eval(xv, yv, pchip_slopes(xv, yv), xq, extrapolate_right=True), ST_OK
    if method == AKIMA:
        return hermite_eval(xv, yv, akima_slopes(xv, yv), xq, extrapolate_right=False), ST_OK
    s = nak_slopes(xv, yv)
    return hermite_eval(xv, yv, s, xq, extrapolate_right=(method == CUBICSPLINE)), ST_OK


def interp1d_batch(xk, yk, knot_off, xq, q_off, method: int):
    """CSR batch of 1-D series with C channels sharing the knot coordinates.

    xk[total_knots], yk[C, total_knots], knot_off[S+1]; xq[total_q] or None (then the
    queries of series s are 0,1,..,m_s-1 as in the reference's RangeIndex), q_off[S+1].
    Returns out[C, total_q], status[S, C].
    """
    yk = np.atleast_2d(np.asarray(yk, np.float64))
    C = yk.shape[0]; S = len(knot_off) - 1
    out = np.full((C, int(q_off[-1])), np.nan)
    status = np.zeros((S, C), np.int32)
    for s_ in range(S):
        a, b = int(knot_off[s_]), int(knot_off[s_ + 1])
        qa, qb = int(q_off[s_]), int(q_off[s_ + 1])
        q = np.arange(qb - qa, dtype=np.float64) if xq is None else xq[qa:qb]
        for c in range(C):
            out[c, qa:qb], status[s_, c] = interp1d(xk[a:b], yk[c, a:b], q, method)
    return out, status


# --------------------------------------------------------------------------- surfaces
def surface(K, T, sigma, Kq, Tq, method: int):
    """One surface: strike pass then maturity pass, each the 1-D operator.

    K[nK], T[nT], sigma[nT, nK] (NaN = missing quote), Kq[mK], Tq[mT] -> out[mT, mK].
    status = OR over every 1-D solve of the surface.
    """
    sigma = np.asarray(sigma, np.float64)
    nT = sigma.shape[0]
    mK = len(Kq); mT = len(Tq)
    Z = np.empty((nT, mK)); st = 0
    for t in range(nT):
        Z[t], s1 = interp1d(K, sigma[t], Kq, method); st |= s1
    out = np.empty((mT, mK))
    for q in range(mK):
        out[:, q], s2 = interp1d(T, Z[:, q], Tq, method); st |= s2
    return out, st


def _batched_interval(xv, xq):
    """j[b, q] = (number of xv[b, :] <= xq[b|:, q]) - 1, rows sorted ascending."""
    if xq.ndim == 1:
        xq = np.broadcast_to(xq, (xv.shape[0], xq.shape[0]))
    B = xv.shape[0]
    j = np.empty(xq.shape, np.int64)
    step = max(1, (1 << 24) // (xv.shape[1] * xq.shape[1]))
    for a in range(0, B, step):
        j[a:a + step] = (xv[a:a + step, None, :] <= xq[a:a + step, :, None]).sum(-1) - 1
    return j, xq


def _take(a, j):
    return np.take_along_axis(a, j, axis=-1)


def _dense_pass(x, y, xq, method):
    """Vectorised 1-D pass over a leading batch axis, all knots valid.
    x[N, n], y[N, R, n] (R rows share x), xq[N|1.., m] -> [N, R, m].  Same arithmetic,
    in the same order, as interp1d() row by row."""
    N, R, n = y.shape
    j, xq = _batched_interval(x, xq)             # [N, m]
    m = xq.shape[1]
    if method in (LINEAR, SLINEAR):
        jj = np.clip(j, 0, n - 2)
        x0 = _take(x, jj); x1 = _take(x, jj + 1)
        jj3 = np.broadcast_to(jj[:, None, :], (N, R, m))
        y0 = _take(y, jj3); y1 = _take(y, jj3 + 1)
        slope = (y1 - y0) / (x1 - x0)[:, None, :]
        r = slope * (xq - x0)[:, None, :] + y0
        r = np.where((x0 == xq)[:, None, :], y0, r)
        last = y[:, :, -1:]
        out = np.where(((j >= 0) & (j < n - 1))[:, None, :], r, np.nan)
        hold = (j >= n - 1) if method == LINEAR else ((j == n - 1) & (xq == x[:, -1:]))
        return np.where(hold[:, None, :], last, out)
    # cubic: batched Thomas (n >= 4), arithmetic identical to nak_slopes()
    assert n >= 4
    dx = np.diff(x, axis=1)                      # [N, n-1]
    delta = np.diff(y, axis=2) / dx[:, None, :]  # [N, R, n-1]
    lo = np.empty((N, n)); di = np.empty((N, n)); up = np.empty((N, n)); rhs = np.empty((N, R, n))
    lo[:, 1:-1] = dx[:, 1:]; di[:, 1:-1] = 2.0 * (dx[:, :-1] + dx[:, 1:]); up[:, 1:-1] = dx[:, :-1]
    rhs[:, :, 1:-1] = 3.0 * (dx[:, None, 1:] * delta[:, :, :-1] + dx[:, None, :-1] * delta[:, :, 1:])
    d = x[:, 2] - x[:, 0]
    lo[:, 0] = 0.0; di[:, 0] = dx[:, 1]; up[:, 0] = d
    rhs[:, :, 0] = ((dx[:, 0] + 2.0 * d)[:, None] * dx[:, 1, None] * delta[:, :, 0]
                    + (dx[:, 0] ** 2)[:, None] * delta[:, :, 1]) / d[:, None]
    d = x[:, -1] - x[:, -3]
    lo[:, -1] = d; di[:, -1] = dx[:, -2]; up[:, -1] = 0.0
    rhs[:, :, -1] = ((dx[:, -1] ** 2)[:, None] * delta[:, :, -2]
                     + (2.0 * d + dx[:, -1])[:, None] * dx[:, -2, None] * delta[:, :, -1]) / d[:, None]
    cp = np.empty((N, n)); dp = np.empty((N, R, n))
    cp[:, 0] = up[:, 0] / di[:, 0]; dp[:, :, 0] = rhs[:, :, 0] / di[:, 0, None]
    for i in range(1, n):
        w = di[:, i] - lo[:, i] * cp[:, i - 1]
        cp[:, i] = up[:, i] / w
        dp[:, :, i] = (rhs[:, :, i] - lo[:, i, None] * dp[:, :, i - 1]) / w[:, None]
    s = np.empty((N, R, n)); s[:, :, -1] = dp[:, :, -1]
    for i in range(n - 2, -1, -1):
        s[:, :, i] = dp[:, :, i] - cp[:, i, None] * s[:, :, i + 1]
    jj = np.clip(j, 0, n - 2)
    x0 = _take(x, jj); h = _take(x, jj + 1) - x0
    jj3 = np.broadcast_to(jj[:, None, :], (N, R, m))
    y0 = _take(y, jj3); y1 = _take(y, jj3 + 1); s0 = _take(s, jj3); s1 = _take(s, jj3 + 1)
    h3 = h[:, None, :]
    dl = (y1 - y0) / h3
    t = (s0 + s1 - 2.0 * dl) / h3
    c0 = t / h3; c1 = (dl - s0) / h3 - t
    u = (xq - x0)[:, None, :]
    r = ((c0 * u + c1) * u + s0) * u + y0
    ok = (j >= 0) & ((xq <= x[:, -1:]) | (method == CUBICSPLINE))
    return np.where(ok[:, None, :], r, np.nan)


def surface_batch(K, T, sigma, Kq, Tq, method: int, k_off=None):
    """Batch of surfaces.  Dense fast route (uniform nK, no NaN, >= 4 knots) is
    vectorised over the batch with arithmetic identical to surface(); anything
    else (ragged ``k_off``, NaNs, few knots) loops over surface().

    K[B, nK] (or flat [total] with k_off[B+1]), T[nT] or [B, nT], sigma[B, nT, nK]
    (or flat [nT * total] per-surface row-major), Kq[mK] or [B, mK], Tq[mT] or [B, mT].
    Returns out[B, mT, mK], status[B].
    """
    Kq = np.asarray(Kq, np.float64); Tq = np.asarray(Tq, np.float64)
    T = np.asarray(T, np.float64)
    mK = Kq.shape[-1]; mT = Tq.shape[-1]
    if k_off is None:
        K = np.asarray(K, np.float64); sigma = np.asarray(sigma, np.float64)
        B, nT, nK = sigma.shape
        dense = (not np.isnan(sigma).any()) and nK >= 4 and nT >= 4 and method <= SLINEAR
        if dense:
            Tb = np.broadcast_to(T, (B, nT)) if T.ndim == 1 else T
            Z = _dense_pass(K, sigma, Kq, method)                       # [B, nT, mK]
            if np.isnan(Z).any():
                dense = False
            else:
                # maturity pass: systems along nT for each of the mK columns
                Zt = np.ascontiguousarray(Z.transpose(0, 2, 1))         # [B, mK, nT]
                out = _dense_pass(Tb, Zt, Tq, method)                   # [B, mK, mT]
                return np.ascontiguousarray(out.transpose(0, 2, 1)), np.zeros(B, np.int32)
    else:
        B = len(k_off) - 1
        nT = T.shape[-1]
    out = np.empty((B, mT, mK)); status = np.zeros(B, np.int32)
    for b in range(B):
        if k_off is None:
            Kb, sb = K[b], sigma[b]
        else:
            a, e = int(k_off[b]), int(k_off[b + 1])
            Kb = K[a:e]; sb = np.asarray(sigma[nT * a:nT * e]).reshape(nT, e - a)
        Tb_ = T if T.ndim == 1 else T[b]
        Kqb = Kq if Kq.ndim == 1 else Kq[b]
        Tqb = Tq if Tq.ndim == 1 else Tq[b]
        out[b], status[b] = surface(Kb, Tb_, sb, Kqb, Tqb, method)
    return out, status
