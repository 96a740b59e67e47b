"""Rebuild DataFrames from the committed golden vectors (numpy + json only; no pickle)."""
import json
import os

import numpy as np
import pandas as pd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dec(arrs, prefix, col, tag):
    v = arrs[f"{prefix}/{col}/v"]
    if tag.startswith("datetime64"):
        idx = pd.DatetimeIndex(v.astype("datetime64[ns]"))
        if "UTC" in tag:
            idx = idx.tz_localize("UTC")
        return pd.Series(idx)
    if tag == "object:str":
        null = arrs[f"{prefix}/{col}/null"]
        o = v.astype(object)
        o[null] = None
        return pd.Series(o, dtype=object)
    if tag == "object:num":
        return pd.Series(v.astype(object), dtype=object)
    if tag == "Float64":
        return pd.Series(pd.arrays.FloatingArray(v.astype(np.float64), arrs[f"{prefix}/{col}/null"].astype(bool)))
    return pd.Series(v)


class SymbolCases:
    def __init__(self):
        with open(os.path.join(GOLDEN, "symbol_cases.json")) as f:
            self.manifest = json.load(f)
        self.arrs = np.load(os.path.join(GOLDEN, "symbol_cases.npz"))
        self.cases = {c["name"]: c for c in self.manifest["cases"]}

    # scipy 1.15.3's Akima1DInterpolator with exactly 2 knots reads an uninitialised slope (m[3] of np.empty,
    # _cubic.py:511-516): the reference's output for it is garbage that changes from run to run, so it pins nothing.
    UNDEFINED_IN_REFERENCE = {"x9_few2_akima"}

    def names(self):
        return [n for n in self.cases if n not in self.UNDEFINED_IN_REFERENCE]

    def input(self, name):
        c = self.cases[name]
        return pd.DataFrame({col: _dec(self.arrs, f"{name}/in", col, tag) for col, tag in c["in_columns"]})

    def expected(self, name):
        c = self.cases[name]
        if c["none"]:
            return None
        df = pd.DataFrame({col: _dec(self.arrs, f"{name}/out", col, tag) for col, tag in c["out_columns"]})
        df.index = self.arrs[f"{name}/out_index"]
        return df


EXACT_METHODS = ("linear", "index", "values", "nearest", "zero", "from_derivatives", "piecewise_polynomial",
                 "pad", "ffill", "bfill", "backfill", "PAD", "BFILL", "Ffill")
POLY_METHODS = ("barycentric", "krogh")


def method_tolerances(method: str) -> dict:
    """Tolerances against the REAL reference's golden outputs, written where they are used:
    np.interp / step methods: bit-exact; splines: 1e-12 relative + 1e-13 absolute (measured <= 1e-15);
    'barycentric' / 'krogh' (one polynomial through <= 32 knots): 1e-9 of the column's largest magnitude -- scipy's
    own barycentric weights use an unseeded random node permutation, so the reference itself moves by up to ~4e-12 of
    that scale from run to run at 24 knots (2e-9 at 32), and its two routes differ by 1e-11 (measured <= 1.2e-11)."""
    if method in EXACT_METHODS:
        return dict(rtol=0.0, atol=0.0)
    if method in POLY_METHODS:
        return dict(rtol=1e-12, atol=1e-13, scale_rtol=1e-9)
    return dict(rtol=1e-12, atol=1e-13)


def assert_symbol_frame(got, exp, *, exact_cols=(), rtol=0.0, atol=0.0, scale_rtol=0.0, name=""):
    """Column order, dtypes, index and values; float columns to (rtol, atol [+ scale_rtol * max|column|]), NaN positions equal."""
    assert (got is None) == (exp is None), f"{name}: None-ness differs (got {type(got)}, expected {type(exp)})"
    if exp is None:
        return
    assert list(got.columns) == list(exp.columns), f"{name}: columns {list(got.columns)} != {list(exp.columns)}"
    assert len(got) == len(exp), f"{name}: {len(got)} rows != {len(exp)}"
    assert np.array_equal(np.asarray(got.index), np.asarray(exp.index)), f"{name}: index differs"
    for c in exp.columns:
        g, e = got[c], exp[c]
        assert str(g.dtype) == str(e.dtype), f"{name}: dtype of {c}: {g.dtype} != {e.dtype}"
        if e.dtype.kind == "f":
            gv, ev = g.to_numpy(), e.to_numpy()
            assert np.array_equal(np.isnan(gv), np.isnan(ev)), f"{name}: NaN pattern of {c}"
            if c in exact_cols or (rtol == 0 and atol == 0 and scale_rtol == 0):
                assert np.array_equal(gv, ev, equal_nan=True), \
                    f"{name}: {c} not bit-exact (max abs diff {np.nanmax(np.abs(gv - ev))})"
            else:
                a_col = atol + (scale_rtol * float(np.nanmax(np.abs(ev))) if scale_rtol and np.isfinite(ev).any() else 0.0)
                ok = np.isclose(gv, ev, rtol=rtol, atol=a_col, equal_nan=True)
                assert ok.all(), f"{name}: {c} max abs diff {np.nanmax(np.abs(gv - ev))}"
        elif e.dtype == object:
            gn, en = g.isna().to_numpy(), e.isna().to_numpy()
            assert np.array_equal(gn, en), f"{name}: null pattern of {c}"
            assert (g[~gn].to_numpy() == e[~en].to_numpy()).all(), f"{name}: values of {c}"
        else:
            assert (g.to_numpy() == e.to_numpy()).all(), f"{name}: values of {c}"


def golden_frame_groups(cases: "SymbolCases", min_cases: int = 2):
    """Long-frame view of the golden symbol cases (what the reference's callers do per symbol,
    batch_processor.py:166-173, done for one frame holding many symbols).  Cases that share (method, min_points, input
    column set and dtypes) become ONE long input frame -- every case under its own symbol name, symbols interleaved in
    reverse order -- and the expected result is the concatenation, in symbol order, of the REAL reference's per-symbol
    outputs.  Yields (method, min_points, long_input, expected_long, case_names).  Cases with a null symbol cell or a
    non-datetime date column are left to the per-symbol tests (a long frame cannot attribute such rows to a symbol)."""
    groups = {}
    for name in cases.names():
        c = cases.cases[name]
        cols = dict((k, t) for k, t in c["in_columns"])
        if "symbol" not in cols or not str(cols.get("date", "")).startswith("datetime64"):
            continue
        if not all(k in cols for k in ("iv", "underlying_price", "time_to_maturity")):
            continue
        df = cases.input(name)
        if df["symbol"].isna().any() or len(df) == 0:
            continue
        groups.setdefault((c["method"], c["min_points"], tuple(map(tuple, c["in_columns"]))), []).append(name)
    for (method, min_points, _), names in sorted(groups.items(), key=lambda kv: (kv[0][0], kv[0][1], len(kv[0][2]))):
        if len(names) < min_cases:
            continue
        ins, exps = [], []
        for i, name in enumerate(names):
            sym = f"s{i:03d}-{name}"
            df = cases.input(name).copy()
            df["symbol"] = sym
            ins.append(df)
            e = cases.expected(name)
            if e is not None:
                e = e.copy()
                e["symbol"] = e["symbol"].where(e["symbol"].isna(), sym)
                exps.append(e)
        long_in = pd.concat(ins[::-1], ignore_index=True)
        exp = pd.concat(exps, ignore_index=True) if exps else None
        yield method, min_points, long_in, exp, names


def assert_long_frame(got, exp, *, rtol=0.0, atol=0.0, scale_rtol=0.0, name=""):
    """Values, column order and dtypes of a long result frame against the concatenated golden frames (index ignored:
    interpolate_frame returns a fresh RangeIndex, the per-symbol frames keep the surviving merged-frame positions)."""
    if exp is None:
        assert got is None or len(got) == 0, f"{name}: expected an empty result, got {0 if got is None else len(got)} rows"
        return
    assert list(got.columns) == list(exp.columns), f"{name}: columns {list(got.columns)} != {list(exp.columns)}"
    assert len(got) == len(exp), f"{name}: {len(got)} rows != {len(exp)}"
    for c in exp.columns:
        g, e = got[c], exp[c]
        assert str(g.dtype) == str(e.dtype), f"{name}: dtype of {c}: {g.dtype} != {e.dtype}"
        if e.dtype.kind == "f":
            gv, ev = g.to_numpy(), e.to_numpy()
            assert np.array_equal(np.isnan(gv), np.isnan(ev)), f"{name}: NaN pattern of {c}"
            if rtol == 0 and atol == 0 and scale_rtol == 0:
                assert np.array_equal(gv, ev, equal_nan=True), f"{name}: {c} not bit-exact (max abs diff {np.nanmax(np.abs(gv - ev))})"
            else:
                a_col = atol + (scale_rtol * float(np.nanmax(np.abs(ev))) if scale_rtol and np.isfinite(ev).any() else 0.0)
                assert np.isclose(gv, ev, rtol=rtol, atol=a_col, equal_nan=True).all(), f"{name}: {c} max abs diff {np.nanmax(np.abs(gv - ev))}"
        elif e.dtype == object:
            gn, en = g.isna().to_numpy(), e.isna().to_numpy()
            assert np.array_equal(gn, en), f"{name}: null pattern of {c}"
            assert (g[~gn].to_numpy() == e[~en].to_numpy()).all(), f"{name}: values of {c}"
        else:
            assert (g.to_numpy() == e.to_numpy()).all(), f"{name}: values of {c}"
