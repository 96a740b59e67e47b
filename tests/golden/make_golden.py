#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference in the authoring container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports ``interpolation.core.IVInterpolator`` from /root/reference/src (read-only,
never copied) and records inputs + outputs of ``interpolate_symbol`` for the cases
SURVEY.md section 8c lists (G1..G9 + seeded fuzz), plus 1-D / surface vectors with
real-valued coordinates produced by the very pandas routine the reference calls
(``Series.interpolate``, reference core.py:61) on a float index.

Outputs (data only; no reference source text):
    tests/golden/symbol_cases.npz  + symbol_cases.json   (DataFrame-level contract R1-R14)
    tests/golden/real1d.npz                              (1-D operator, real coordinates)
    tests/golden/surfaces.npz                            (two-pass composition)
Versions used are written into symbol_cases.json.
"""
import json
import os
import sys
import warnings

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
warnings.simplefilter("ignore")
import logging

import numpy as np
import pandas as pd
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
from interpolation.core import IVInterpolator  # noqa: E402  (the real reference)

logging.getLogger("interpolation.core").setLevel(logging.CRITICAL)

COLS13 = ["symbol", "date", "iv", "underlying_price", "time_to_maturity", "strike", "callput",
          "interest_rate", "mark_price", "index_price", "volume", "quote_volume", "record_time"]


def frame(n, seed, freq="1h", start="2023-03-01 00:00:00", cols=COLS13):
    r = np.random.default_rng(seed)
    dates = pd.date_range(start, periods=n, freq=freq)
    S = 25000 + np.cumsum(r.normal(0, 40, n))
    d = {
        "symbol": np.array(["btc-20mar23-25000-c"] * n, dtype=object),
        "date": dates,
        "iv": 0.6 + 0.05 * np.sin(np.arange(n) / 3.0) + r.normal(0, 0.01, n),
        "underlying_price": S,
        "time_to_maturity": 0.05 - np.arange(n) / (24 * 365.0),
        "strike": np.full(n, 25000.0),
        "callput": np.array(["c"] * n, dtype=object),
        "interest_rate": np.full(n, 0.01),
        "mark_price": 0.03 + r.normal(0, 0.001, n),
        "index_price": S + r.normal(0, 5, n),
        "volume": r.uniform(0, 10, n).round(3),
        "quote_volume": r.uniform(0, 1e5, n).round(1),
        "record_time": (dates.asi8 // 10**6).astype(np.float64),
    }
    return pd.DataFrame({c: d[c] for c in cols})


CASES = []   # (name, df, method, min_points)


def add(name, df, method="linear", min_points=10):
    CASES.append((name, df, method, min_points))


# G1/G2: 12 hourly knots, all 13 columns
for m in ("linear", "cubic", "cubicspline", "slinear", "index", "values"):
    add(f"g1_{m}", frame(12, 1), m)
# G3: off-grid row (+30 s) disappears (R6)
df = frame(12, 3); df.loc[5, "date"] += pd.Timedelta(seconds=30)
add("g3_offgrid_linear", df); add("g3_offgrid_cubic", df.copy(), "cubic")
# G4: duplicate timestamp (R7, R8)
df = frame(12, 4); df.loc[6, "date"] = df.loc[5, "date"]
add("g4_dup_linear", df); add("g4_dup_cubic", df.copy(), "cubic")
df = frame(14, 41); df.loc[6, "date"] = df.loc[5, "date"]; df.loc[7, "date"] = df.loc[5, "date"]; df.loc[13, "date"] = df.loc[12, "date"]
add("g4_dup3_linear", df); add("g4_dup3_cubicspline", df.copy(), "cubicspline")
# G5: NaN patterns per channel (R9-R11)
for m in ("linear", "cubic", "cubicspline", "slinear"):
    df = frame(14, 5)
    df.loc[4:6, "iv"] = np.nan                     # interior
    df.loc[0:1, "underlying_price"] = np.nan       # leading
    df.loc[12:13, "time_to_maturity"] = np.nan     # trailing
    df.loc[3, "volume"] = np.nan; df.loc[0, "strike"] = np.nan; df.loc[7, "callput"] = None
    add(f"g5_nan_{m}", df, m)
df = frame(12, 51); df["iv"] = np.nan
add("g5_allnan_iv", df)
df = frame(12, 52); df.loc[1:, "iv"] = np.nan
add("g5_oneknot_linear", df); add("g5_oneknot_cubic", df.copy(), "cubic")
df = frame(12, 53); df.loc[0, "symbol"] = None
add("g5_symbol_none", df)
# G6: guards
add("g6_too_few", frame(9, 6)); add("g6_exactly_min", frame(10, 6))
df = frame(11, 61); df.loc[10, "date"] = df.loc[0, "date"] + pd.Timedelta(days=30, minutes=1)
add("g6_span_gt30d", df)
df = frame(11, 62); df.loc[10, "date"] = df.loc[0, "date"] + pd.Timedelta(days=30)
add("g6_span_eq30d", df)
for m in ("time", "spline", "polynomial", "bogus"):
    add(f"g6_method_{m}", frame(12, 63), m)
add("g6_missing_iv", frame(12, 64).drop(columns=["iv"]))
add("g6_missing_symbol", frame(12, 64).drop(columns=["symbol"]))
add("g6_missing_date", frame(12, 64).drop(columns=["date"]))
add("g6_core_cols_only", frame(12, 65, cols=["symbol", "date", "iv", "underlying_price", "time_to_maturity"]))
df = frame(12, 66); df["extra"] = np.arange(12.0); df["volume"] = np.arange(12)
add("g6_extra_and_int_cols", df)
# G7: shuffled + string dates (R2, R3)
df = frame(12, 7).sample(frac=1.0, random_state=7).reset_index(drop=True)
add("g7_shuffled", df)
df = frame(12, 71); df["date"] = df["date"].dt.strftime("%Y-%m-%d %H:%M:%S")
df = df.sample(frac=1.0, random_state=8).reset_index(drop=True)
add("g7_string_dates", df); add("g7_string_dates_cubic", df.copy(), "cubic")
df = frame(12, 72); df["date"] = df["date"].dt.tz_localize("UTC")
add("g7_tz_utc", df)


# G8: BASELINE-shaped 1-D: knots at integer minute positions
def lattice_frame(pos, seed):
    r = np.random.default_rng(seed)
    n = len(pos)
    df = frame(n, seed, cols=["symbol", "date", "iv", "underlying_price", "time_to_maturity", "strike"])
    df["date"] = pd.Timestamp("2023-03-01") + pd.to_timedelta(np.asarray(pos) * 60, unit="s")
    df["iv"] = 0.6 + 0.2 * r.standard_normal(n)
    return df


r8 = np.random.default_rng(8)
p64 = np.concatenate([[0], np.sort(r8.choice(np.arange(1, 255), 62, replace=False)), [255]])
p16 = np.concatenate([[0], np.sort(r8.choice(np.arange(1, 63), 14, replace=False)), [63]])
for m in ("linear", "cubic", "cubicspline"):
    add(f"g8_64to256_{m}", lattice_frame(p64, 80), m, 2)
    add(f"g8_16to64_{m}", lattice_frame(p16, 81), m, 2)
# G9: ragged knot counts + few-knot rule (R13)
for n in (8, 10, 16, 33, 64, 128):
    pos = np.concatenate([[0], np.sort(r8.choice(np.arange(1, 4 * n), n - 2, replace=False)), [4 * n]])
    for m in ("linear", "cubic"):
        add(f"g9_n{n}_{m}", lattice_frame(pos, 90 + n), m, 2)
add("g9_n8_default_minpoints", lattice_frame(np.arange(8) * 7, 98), "linear", 10)
for n in (2, 3, 4, 5):
    for m in ("linear", "cubic", "cubicspline", "slinear"):
        add(f"g9_few{n}_{m}", lattice_frame(np.arange(n) * 20, 99 + n), m, 2)
# few-knot in ONE channel only
df = lattice_frame(np.arange(8) * 10, 120); df.loc[3:, "iv"] = np.nan
for m in ("cubic", "cubicspline", "linear"):
    add(f"g9_chan_few3_{m}", df.copy(), m, 2)

# further pandas methods (appended after the original cases so their vectors stay byte-identical)
EXTRA = ("nearest", "zero", "pchip", "akima", "from_derivatives", "piecewise_polynomial")
for m in EXTRA:
    add(f"x1_{m}", frame(12, 1), m)
    dfx = frame(14, 5)
    dfx.loc[4:6, "iv"] = np.nan; dfx.loc[0:1, "underlying_price"] = np.nan; dfx.loc[12:13, "time_to_maturity"] = np.nan
    add(f"x5_nan_{m}", dfx, m)
    add(f"x8_64to256_{m}", lattice_frame(p64, 80), m, 2)
    for n in (2, 3, 4):
        add(f"x9_few{n}_{m}", lattice_frame(np.arange(n) * 20, 99 + n), m, 2)
    dfc = lattice_frame(np.arange(8) * 10, 120); dfc.loc[1:, "iv"] = np.nan
    add(f"x9_chan_one_knot_{m}", dfc, m, 2)

# seeded fuzz
rf = np.random.default_rng(20230320)
for k in range(40):
    n = int(rf.integers(10, 60))
    df = frame(n, 1000 + k, freq=rf.choice(["1h", "30min", "7min", "2h"]))
    for c in ("iv", "underlying_price", "time_to_maturity", "volume", "strike"):
        msk = rf.random(n) < rf.choice([0.0, 0.1, 0.4])
        df.loc[msk, c] = np.nan
    if rf.random() < 0.3:
        i = int(rf.integers(1, n)); df.loc[i, "date"] = df.loc[i - 1, "date"]
    if rf.random() < 0.3:
        i = int(rf.integers(1, n - 1)); df.loc[i, "date"] += pd.Timedelta(seconds=int(rf.integers(1, 59)))
    if rf.random() < 0.5:
        df = df.sample(frac=1.0, random_state=k).reset_index(drop=True)
    add(f"fuzz{k:02d}", df, str(rf.choice(["linear", "cubic", "cubicspline", "slinear"])), int(rf.choice([2, 10])))


# appended later, after the fuzz block, so that every earlier vector stays byte-identical
for m in ("quadratic",):
    add(f"x1_{m}", frame(12, 1), m)
    dfx = frame(14, 5)
    dfx.loc[4:6, "iv"] = np.nan; dfx.loc[0:1, "underlying_price"] = np.nan; dfx.loc[12:13, "time_to_maturity"] = np.nan
    add(f"x5_nan_{m}", dfx, m)
    add(f"x8_64to256_{m}", lattice_frame(p64, 80), m, 2)
    for n in (2, 3, 4):
        add(f"x9_few{n}_{m}", lattice_frame(np.arange(n) * 20, 99 + n), m, 2)
    dfc = lattice_frame(np.arange(8) * 10, 120); dfc.loc[1:, "iv"] = np.nan
    add(f"x9_chan_one_knot_{m}", dfc, m, 2)


# appended in round 2 (after every earlier block, so that the earlier vectors stay byte-identical):
# dtype of a numeric channel -- pandas leaves an object column alone (Series.interpolate on object dtype is a deprecated
# no-op: only the source rows survive the dropna), computes a float32 column in float64 and stores float32, keeps the
# nullable Float64 dtype
for m in ("linear", "cubic"):
    dft = frame(12, 1); dft["iv"] = dft["iv"].astype(object)
    add(f"t1_object_iv_{m}", dft, m)
    dft = frame(12, 1); dft["iv"] = dft["iv"].astype(np.float32); dft.loc[3, "iv"] = np.nan
    add(f"t2_float32_iv_{m}", dft, m)
    dft = frame(12, 1); dft["iv"] = dft["iv"].astype("Float64"); dft.loc[4, "iv"] = pd.NA
    add(f"t3_Float64_iv_{m}", dft, m)
dft = frame(12, 1); dft["underlying_price"] = dft["underlying_price"].astype(object); dft["time_to_maturity"] = dft["time_to_maturity"].astype(np.float32)
add("t1_object_underlying_float32_ttm", dft)
# duplicate timestamps in a symbol with MORE than 16 rows: sort_values('date') is numpy's quicksort (introsort), which is
# not stable beyond 16 elements -- the tie order of the duplicates (hence which value sits at which merged-frame
# position) is whatever that routine leaves
for k_, (n_, m) in enumerate([(24, "linear"), (24, "cubic"), (40, "linear"), (33, "cubicspline")]):
    dft = frame(n_, 700 + k_)
    for i in (5, 11, 12, 20):
        dft.loc[i, "date"] = dft.loc[i - 1, "date"]
    if k_ % 2:
        dft = dft.sample(frac=1.0, random_state=70 + k_).reset_index(drop=True)
    add(f"t4_dup_gt16_{n_}_{m}", dft, m, 2)
# polynomial interpolation through ALL knots ('barycentric', 'krogh': scipy barycentric_interpolate / krogh_interpolate).
# scipy's barycentric weights use a random permutation of the nodes (results move in the last digits from run to run);
# with more than ~30 knots the polynomial itself is numerical noise.  Pinned for small knot counts.
for m in ("barycentric", "krogh"):
    add(f"y1_{m}", frame(12, 1), m)
    dfx = frame(14, 5)
    dfx.loc[4:6, "iv"] = np.nan; dfx.loc[0:1, "underlying_price"] = np.nan; dfx.loc[12:13, "time_to_maturity"] = np.nan
    add(f"y5_nan_{m}", dfx, m)
    add(f"y8_16to64_{m}", lattice_frame(p16, 81), m, 2)
    for n in (1, 2, 3, 5):
        add(f"y9_few{n}_{m}", lattice_frame(np.arange(max(n, 2)) * 20, 99 + n) if n > 1 else lattice_frame(np.arange(2) * 20, 100).assign(iv=[0.5, np.nan]), m, 2)
    add(f"y9_n24_{m}", lattice_frame(np.arange(24) * 9, 124), m, 2)


# appended in round 3: the FILL methods.  core.py:61 forwards self.method to Series.interpolate, and pandas 2.x still executes
# 'pad' / 'ffill' / 'bfill' / 'backfill' there (NDFrame.interpolate -> BlockManager.pad_or_backfill, FutureWarning): forward
# fill keeps leading NaN and holds the last value, backward fill keeps trailing NaN; object channels ARE filled (and soft-
# converted to float64), float32 / Float64 keep their dtype
for m in ("pad", "ffill", "bfill", "backfill"):
    add(f"z1_{m}", frame(12, 1), m)
    dfz = frame(14, 5)
    dfz.loc[4:6, "iv"] = np.nan; dfz.loc[0:1, "underlying_price"] = np.nan; dfz.loc[12:13, "time_to_maturity"] = np.nan
    dfz.loc[3, "volume"] = np.nan; dfz.loc[0, "strike"] = np.nan; dfz.loc[7, "callput"] = None
    add(f"z5_nan_{m}", dfz, m)
for m in ("pad", "bfill"):
    dfz = frame(14, 41); dfz.loc[6, "date"] = dfz.loc[5, "date"]; dfz.loc[7, "date"] = dfz.loc[5, "date"]
    dfz.loc[13, "date"] = dfz.loc[12, "date"]; dfz.loc[5, "iv"] = np.nan; dfz.loc[12, "underlying_price"] = np.nan
    add(f"z4_dup3_{m}", dfz, m)
    dfz = frame(12, 3); dfz.loc[5, "date"] += pd.Timedelta(seconds=30); dfz.loc[4, "iv"] = np.nan
    add(f"z3_offgrid_{m}", dfz, m)
    dfz = frame(12, 51); dfz["iv"] = np.nan
    add(f"z5_allnan_iv_{m}", dfz, m)
    dfz = frame(12, 52); dfz.loc[1:, "iv"] = np.nan
    add(f"z5_oneknot_first_{m}", dfz, m)
    dfz = frame(12, 52); dfz.loc[:10, "iv"] = np.nan
    add(f"z5_oneknot_last_{m}", dfz, m)
    dfz = frame(12, 1); dfz["iv"] = dfz["iv"].astype(object); dfz.loc[3, "iv"] = np.nan
    add(f"z6_object_iv_{m}", dfz, m)
    dfz = frame(12, 1); dfz["iv"] = dfz["iv"].astype(np.float32); dfz.loc[3, "iv"] = np.nan
    add(f"z6_float32_iv_{m}", dfz, m)
    dfz = frame(12, 1); dfz["iv"] = dfz["iv"].astype("Float64"); dfz.loc[4, "iv"] = pd.NA
    add(f"z6_Float64_iv_{m}", dfz, m)
    dfz = frame(12, 1); dfz["time_to_maturity"] = np.arange(12)
    add(f"z6_int_ttm_{m}", dfz, m)
    add(f"z8_64to256_{m}", lattice_frame(p64, 80), m, 2)
    for n in (2, 3):
        add(f"z9_few{n}_{m}", lattice_frame(np.arange(n) * 20, 99 + n), m, 2)
    dfz = frame(24, 700)
    for i in (5, 11, 12, 20):
        dfz.loc[i, "date"] = dfz.loc[i - 1, "date"]
    dfz.loc[11, "iv"] = np.nan
    add(f"z4_dup_gt16_{m}", dfz.sample(frac=1.0, random_state=71).reset_index(drop=True), m, 2)
add("z7_upper_PAD", frame(12, 1), "PAD")
add("z7_upper_BFILL", frame(12, 1), "BFILL"); add("z7_mixed_Ffill", frame(12, 1), "Ffill"); add("z7_upper_LINEAR", frame(12, 1), "LINEAR")
rz = np.random.default_rng(20261005)
for k in range(24):
    n = int(rz.integers(10, 60))
    dfz = frame(n, 3000 + k, freq=rz.choice(["1h", "30min", "7min", "2h"]))
    for c in ("iv", "underlying_price", "time_to_maturity", "volume", "strike"):
        msk = rz.random(n) < rz.choice([0.0, 0.1, 0.4])
        dfz.loc[msk, c] = np.nan
    if rz.random() < 0.3:
        i = int(rz.integers(1, n)); dfz.loc[i, "date"] = dfz.loc[i - 1, "date"]
    if rz.random() < 0.3:
        i = int(rz.integers(1, n - 1)); dfz.loc[i, "date"] += pd.Timedelta(seconds=int(rz.integers(1, 59)))
    if rz.random() < 0.5:
        dfz = dfz.sample(frac=1.0, random_state=k).reset_index(drop=True)
    add(f"zfuzz{k:02d}", dfz, str(rz.choice(["pad", "ffill", "bfill", "backfill"])), int(rz.choice([2, 10])))


def enc(col: pd.Series):
    """Encode a column without pickling: returns dict of arrays + dtype tag."""
    dt = str(col.dtype)
    if dt.startswith("datetime64"):
        v = col
        if getattr(col.dt, "tz", None) is not None:
            v = col.dt.tz_convert("UTC").dt.tz_localize(None)
        return {"v": v.astype("datetime64[ns]").astype("int64").to_numpy()}, dt
    if dt == "Float64":                      # nullable extension dtype: values + mask
        return {"v": col.to_numpy(np.float64, na_value=np.nan), "null": col.isna().to_numpy()}, "Float64"
    if col.dtype == object:
        null = col.isna().to_numpy()
        isstr = np.array([isinstance(x, str) for x in col], bool)
        if (isstr | null).all():
            return {"v": np.array(["" if n_ else x for x, n_ in zip(col, null)], dtype="U"), "null": null}, "object:str"
        return {"v": pd.to_numeric(col, errors="coerce").to_numpy(np.float64), "null": null}, "object:num"
    return {"v": col.to_numpy()}, dt


def main():
    arrays = {}
    manifest = {"versions": {"pandas": pd.__version__, "numpy": np.__version__, "scipy": scipy.__version__,
                             "python": sys.version.split()[0]}, "cases": []}
    for name, df, method, mp in CASES:
        out = IVInterpolator(method=method, min_points=mp).interpolate_symbol(df.copy())
        entry = {"name": name, "method": method, "min_points": mp, "none": out is None,
                 "in_columns": [], "out_columns": []}
        for c in df.columns:
            a, tag = enc(df[c])
            entry["in_columns"].append([c, tag])
            for k, v in a.items():
                arrays[f"{name}/in/{c}/{k}"] = v
        if out is not None:
            arrays[f"{name}/out_index"] = out.index.to_numpy()
            for c in out.columns:
                a, tag = enc(out[c])
                entry["out_columns"].append([c, tag])
                for k, v in a.items():
                    arrays[f"{name}/out/{c}/{k}"] = v
        manifest["cases"].append(entry)
    np.savez_compressed(os.path.join(HERE, "symbol_cases.npz"), **arrays)
    with open(os.path.join(HERE, "symbol_cases.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    n_none = sum(c["none"] for c in manifest["cases"])
    print(f"symbol cases: {len(CASES)} ({n_none} -> None)")

    # ---- real-valued 1-D vectors through pandas Series.interpolate on a float index
    def pandas_1d(xk, yk, xq, method):
        ux = np.union1d(xk, xq)
        s = pd.Series(np.nan, index=ux)
        s.loc[xk] = yk
        pm = {"linear": "index"}.get(method, method)   # 'linear' ignores the index; 'index' is the same np.interp path
        try:
            r = s.interpolate(method=pm)
        except Exception:
            return None
        return r.loc[xq].to_numpy()

    ALL_METHODS = ("linear", "cubic", "cubicspline", "slinear", "nearest", "zero", "pchip", "akima", "from_derivatives",
                   "quadratic", "pad", "bfill")      # the last two since round 3 (deterministic: earlier vectors unchanged)
    r1 = np.random.default_rng(11)
    real = {}
    k = 0
    for n, m in [(64, 64), (64, 256), (16, 16), (16, 64), (8, 64), (128, 64), (33, 50), (4, 20), (5, 20), (3, 9), (2, 9)]:
        for nanfrac in (0.0, 0.25):
            S = 25000.0
            xk = np.sort(S * (np.linspace(0.7, 1.3, n) + r1.uniform(-0.2, 0.2, n) * (0.6 / max(n - 1, 1))))
            yk = 0.6 + 0.3 * (np.log(xk / S)) ** 2 + r1.normal(0, 0.002, n)
            if nanfrac:
                yk[r1.random(n) < nanfrac] = np.nan
                yk[0] = np.nan if k % 2 else yk[0]
                yk[-1] = np.nan if k % 3 == 0 else yk[-1]
            xq = S * np.linspace(0.68, 1.32, m)
            if n > 4:
                xq[m // 3] = xk[n // 2]; xq = np.sort(xq)      # one exact knot hit
            real[f"c{k}/xk"] = xk; real[f"c{k}/yk"] = yk; real[f"c{k}/xq"] = xq
            for meth in ALL_METHODS:
                v = pandas_1d(xk, yk, xq, meth)
                real[f"c{k}/{meth}"] = np.array([]) if v is None else v
                real[f"c{k}/{meth}_raised"] = np.array(v is None)
            k += 1
    real["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "real1d.npz"), **real)
    print("real1d cases:", k)

    # ---- surfaces: two pandas passes (strike then maturity)
    TEN = np.array([1, 2, 3, 7, 14, 21, 30, 45, 60, 90, 120, 150, 180, 270, 365, 540]) / 365.0
    surf = {}
    rs = np.random.default_rng(12)
    k = 0
    for (nK, nT, mK, mT, nanfrac) in [(64, 16, 64, 16, 0.0), (64, 16, 64, 16, 0.1), (16, 8, 16, 8, 0.0),
                                      (64, 16, 256, 64, 0.0), (8, 16, 64, 16, 0.0), (128, 16, 64, 16, 0.0),
                                      (37, 16, 64, 16, 0.15), (5, 5, 12, 7, 0.0)]:
        S = float(rs.uniform(20000, 30000))
        mny = np.linspace(0.7, 1.3, nK) + rs.uniform(-0.2, 0.2, nK) * (0.6 / (nK - 1))
        K = S * mny
        T = TEN[:nT]
        kk = np.log(K / S)[None, :] / np.sqrt(T)[:, None]
        a, b, rho, mu, s_ = rs.uniform(.15, .5), rs.uniform(.05, .3), rs.uniform(-.7, .1), rs.normal(0, .05), rs.uniform(.1, .4)
        sig = np.sqrt(a + b * (rho * (kk - mu) + np.sqrt((kk - mu) ** 2 + s_ ** 2))) + rs.normal(0, 0.002, kk.shape)
        sig = np.clip(sig, 0.05, 3.0)
        if nanfrac:
            sig[rs.random(sig.shape) < nanfrac] = np.nan
        Kq = S * np.linspace(0.72, 1.28, mK)
        Tq = np.geomspace(2 / 365.0, min(1.4, T[-1]), mT)
        surf[f"s{k}/K"] = K; surf[f"s{k}/T"] = T; surf[f"s{k}/sigma"] = sig; surf[f"s{k}/Kq"] = Kq; surf[f"s{k}/Tq"] = Tq
        for meth in ALL_METHODS:
            raised = False
            Z = np.full((nT, mK), np.nan)
            for t in range(nT):
                v = pandas_1d(K, sig[t], Kq, meth) if (~np.isnan(sig[t])).any() else np.full(mK, np.nan)
                if v is None:
                    raised = True; break
                Z[t] = v
            out = np.full((mT, mK), np.nan)
            if not raised:
                for q in range(mK):
                    v = pandas_1d(T, Z[:, q], Tq, meth) if (~np.isnan(Z[:, q])).any() else np.full(mT, np.nan)
                    if v is None:
                        raised = True; break
                    out[:, q] = v
            surf[f"s{k}/{meth}"] = out
            surf[f"s{k}/{meth}_raised"] = np.array(raised)
        k += 1
    surf["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "surfaces.npz"), **surf)
    print("surface cases:", k)


if __name__ == "__main__":
    main()
