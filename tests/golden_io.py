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


def assert_symbol_frame(got, exp, *, exact_cols=(), rtol=0.0, atol=0.0, name=""):
    """Column order, dtypes, index and values; float columns to (rtol, atol), NaN positions equal."""
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
            if c in exact_cols or (rtol == 0 and atol == 0):
                assert np.array_equal(gv, ev, equal_nan=True), \
                    f"{name}: {c} not bit-exact (max abs diff {np.nanmax(np.abs(gv - ev))})"
            else:
                ok = np.isclose(gv, ev, rtol=rtol, atol=atol, equal_nan=True)
                assert ok.all(), f"{name}: {c} max abs diff {np.nanmax(np.abs(gv - ev))}"
        elif e.dtype == object:
            gn, en = g.isna().to_numpy(), e.isna().to_numpy()
            assert np.array_equal(gn, en), f"{name}: null pattern of {c}"
            assert (g[~gn].to_numpy() == e[~en].to_numpy()).all(), f"{name}: values of {c}"
        else:
            assert (g.to_numpy() == e.to_numpy()).all(), f"{name}: values of {c}"
