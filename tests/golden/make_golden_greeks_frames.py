#!/usr/bin/env python3
"""Golden vectors for the Greeks EPILOGUE of interpolated frames, from the REAL reference (data only).
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_greeks_frames.py

The reference declares `preserve_greeks` ("Recalculate Greeks after interpolation", config.py:46) and the delta..rho
columns (src/database/schema.py:36-40) without wiring them up.  What the flag can only mean is pinned here with the
reference's own two routines: IVInterpolator.interpolate_symbol (src/interpolation/core.py) on a frame, then
BlackScholesGreeks.calculate_greeks (src/interpolation/greeks.py) on the result's rows -- iv / underlying_price /
time_to_maturity as interpolated, strike / interest_rate / callput as forward-filled, option type per row."""
import os
import sys
import warnings

sys.dont_write_bytecode = True
warnings.simplefilter("ignore")
import logging

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src"); sys.path.insert(0, HERE)
from interpolation.core import IVInterpolator          # noqa: E402  the real reference
from interpolation.greeks import BlackScholesGreeks    # noqa: E402
import make_golden as MG                                 # noqa: E402  frame() / enc(): the same synthetic symbols and encoding

logging.getLogger("interpolation.core").setLevel(logging.CRITICAL)
GREEKS = ["delta", "gamma", "theta", "vega", "rho"]

cases = []
for k, (method, n, cp) in enumerate([("linear", 12, "c"), ("cubic", 14, "p"), ("cubicspline", 16, "c"), ("pchip", 12, "p"),
                                     ("linear", 20, "mixed"), ("krogh", 10, "c")]):
    df = MG.frame(n, 300 + k)
    df["strike"] = 25000.0 + 250.0 * (k - 2)
    df["interest_rate"] = 0.01 + 0.002 * k
    if cp == "mixed":                                   # option type changes along the series; one null strike / rate cell
        df["callput"] = np.array((["c"] * 7 + ["p"] * (n - 7)), dtype=object)
        df.loc[3, "strike"] = np.nan; df.loc[9, "interest_rate"] = np.nan
    else:
        df["callput"] = np.array([cp] * n, dtype=object)
    df.loc[4:5, "iv"] = np.nan; df.loc[8, "underlying_price"] = np.nan; df.loc[1, "time_to_maturity"] = np.nan
    cases.append((f"gf{k}_{method}_{cp}", df, method))

arrays, names, methods = {}, [], []
for name, df, method in cases:
    out = IVInterpolator(method, 10).interpolate_symbol(df.copy())
    assert out is not None, name
    g = {c: np.empty(len(out)) for c in GREEKS}
    for i, (_, row) in enumerate(out.iterrows()):
        typ = "call" if str(row["callput"]).lower().startswith("c") else "put"
        r = BlackScholesGreeks.calculate_greeks(row["underlying_price"], row["strike"], row["time_to_maturity"],
                                                row["interest_rate"], row["iv"], typ)
        for c in GREEKS:
            g[c][i] = r[c]
    names.append(name); methods.append(method)
    for c in df.columns:
        a, tag = MG.enc(df[c])
        arrays[f"{name}/in_tag/{c}"] = np.array(tag)
        for kk, v in a.items():
            arrays[f"{name}/in/{c}/{kk}"] = v
    arrays[f"{name}/in_columns"] = np.array(list(df.columns))
    arrays[f"{name}/rows"] = np.array(len(out))
    arrays[f"{name}/date"] = out["date"].astype("datetime64[ns]").astype("int64").to_numpy()
    for c in ("iv", "underlying_price", "time_to_maturity"):
        arrays[f"{name}/out/{c}"] = out[c].to_numpy(np.float64)
    for c in GREEKS:
        arrays[f"{name}/greeks/{c}"] = g[c]
arrays["names"] = np.array(names); arrays["methods"] = np.array(methods)
np.savez_compressed(os.path.join(HERE, "greeks_frames.npz"), **arrays)
print("greeks-frame golden cases:", len(names), [int(arrays[f"{n}/rows"]) for n in names])
