"""CPU: the oracle (oracle/) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ivs_oracle as O          # noqa: E402
import ref_symbol               # noqa: E402
from golden_io import GOLDEN, SymbolCases, assert_symbol_frame, method_tolerances   # noqa: E402

CASES = SymbolCases()
METHODS = {"linear": O.LINEAR, "cubic": O.CUBIC, "cubicspline": O.CUBICSPLINE, "slinear": O.SLINEAR,
           "nearest": O.NEAREST, "zero": O.ZERO, "pchip": O.PCHIP, "akima": O.AKIMA, "from_derivatives": O.FROM_DERIVATIVES,
           "quadratic": O.QUADRATIC, "pad": O.PAD, "bfill": O.BFILL}
EXACT = ("linear", "nearest", "zero", "from_derivatives", "pad", "bfill")
# fp64 tolerance for the spline methods on the golden shapes (measured <= 1e-15)
RTOL, ATOL = 1e-12, 1e-13


@pytest.mark.parametrize("name", CASES.names())
def test_symbol_contract(name):
    c = CASES.cases[name]
    got = ref_symbol.interpolate_symbol(CASES.input(name), c["method"], c["min_points"])
    assert_symbol_frame(got, CASES.expected(name), name=name, **method_tolerances(c["method"]))


def test_real1d_against_pandas_vectors():
    g = np.load(os.path.join(GOLDEN, "real1d.npz"))
    for k in range(int(g["n_cases"])):
        xk, yk, xq = g[f"c{k}/xk"], g[f"c{k}/yk"], g[f"c{k}/xq"]
        for m, code in METHODS.items():
            if m == "akima" and int((~np.isnan(yk)).sum()) == 2:
                continue          # undefined in the reference (scipy reads an uninitialised slope), see golden_io.py
            got, st = O.interp1d(xk, yk, xq, code)
            if bool(g[f"c{k}/{m}_raised"]):
                assert st == O.ST_TOO_FEW_KNOTS
                continue
            assert st == O.ST_OK
            exp = g[f"c{k}/{m}"]
            assert np.array_equal(np.isnan(got), np.isnan(exp)), (k, m)
            if m in EXACT:
                assert np.array_equal(got, exp, equal_nan=True), (k, m)      # bit-exact vs np.interp / step methods
            else:
                assert np.allclose(got, exp, rtol=RTOL, atol=ATOL, equal_nan=True), (k, m)


def test_surfaces_against_pandas_two_pass_vectors():
    g = np.load(os.path.join(GOLDEN, "surfaces.npz"))
    for k in range(int(g["n_cases"])):
        K, T, s, Kq, Tq = [g[f"s{k}/{n}"] for n in ("K", "T", "sigma", "Kq", "Tq")]
        for m, code in METHODS.items():
            got, st = O.surface(K, T, s, Kq, Tq, code)
            if bool(g[f"s{k}/{m}_raised"]):
                assert st == O.ST_TOO_FEW_KNOTS
                continue
            exp = g[f"s{k}/{m}"]
            assert np.array_equal(np.isnan(got), np.isnan(exp)), (k, m)
            if m in EXACT:
                assert np.array_equal(got, exp, equal_nan=True), (k, m)
            else:
                assert np.allclose(got, exp, rtol=RTOL, atol=ATOL, equal_nan=True), (k, m)
            gb, sb = O.surface_batch(K[None], T, s[None], Kq, Tq, code)
            assert np.array_equal(gb[0], got, equal_nan=True)


def test_batch_route_equals_loop():
    r = np.random.default_rng(5)
    B, nT, nK = 7, 16, 64
    K = np.sort(r.uniform(.7, 1.3, (B, nK)), 1); K[:, 0] = .69; K[:, -1] = 1.31
    T = np.array([1, 2, 3, 7, 14, 21, 30, 45, 60, 90, 120, 150, 180, 270, 365, 540]) / 365
    sig = r.uniform(.2, 1, (B, nT, nK)); Kq = np.linspace(.72, 1.28, 64); Tq = np.geomspace(2 / 365, 1.4, 16)
    for code in METHODS.values():
        a, _ = O.surface_batch(K, T, sig, Kq, Tq, code)
        b = np.stack([O.surface(K[i], T, sig[i], Kq, Tq, code)[0] for i in range(B)])
        assert np.array_equal(a, b, equal_nan=True)
