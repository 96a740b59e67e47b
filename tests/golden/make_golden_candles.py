#!/usr/bin/env python3
"""Golden vectors for candle aggregation from the REAL reference (src/candle_reconstruction/core.py), data only.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_candles.py"""
import os, sys, logging
sys.dont_write_bytecode = True
import numpy as np, pandas as pd
sys.path.insert(0, "/root/reference/src")
from candle_reconstruction.core import CandleReconstructor   # the real reference
logging.getLogger("candle_reconstruction.core").setLevel(logging.CRITICAL)

r = np.random.default_rng(20230320)
out = {}
cases = []
def case(name, ts, freq, nan_frac=0.0):
    n = len(ts)
    o = 100 + np.cumsum(r.normal(0, 1, n)); h = o + r.uniform(0, 2, n); l = o - r.uniform(0, 2, n); c = o + r.normal(0, .5, n)
    v = r.uniform(0, 50, n).round(2)
    for a in (o, h, l, c, v):
        a[r.random(n) < nan_frac] = np.nan
    df = pd.DataFrame({"symbol": "btc-20mar23-25000-c", "timestamp": pd.to_datetime(ts), "open": o, "high": h, "low": l, "close": c, "volume": v})
    res = CandleReconstructor(freq).reconstruct_symbol_candles(df)
    out[f"{name}/ts"] = np.asarray(ts).astype("datetime64[ns]").astype(np.int64)
    for k, a in (("o", o), ("h", h), ("l", l), ("c", c), ("v", v)):
        out[f"{name}/{k}"] = a
    out[f"{name}/none"] = np.array(res is None)
    if res is not None:
        out[f"{name}/out_ts"] = res["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64)
        for k in ("open", "high", "low", "close", "volume"):
            out[f"{name}/out_{k}"] = res[k].to_numpy(np.float64)
    cases.append((name, freq))
base = pd.Timestamp("2023-03-01 09:00:00")
mins = lambda idx: (base + pd.to_timedelta(np.asarray(idx), unit="min")).to_numpy()
# KAT of SURVEY 8f: 12 one-minute candles from 09:00 -> two 5-min candles, third group dropped
ts = mins(np.arange(12)); n = 12
df = pd.DataFrame({"symbol": "s", "timestamp": pd.to_datetime(ts), "open": 100.0 + np.arange(n), "high": 102.0 + np.arange(n),
                   "low": 99.0 + np.arange(n), "close": 101.0 + np.arange(n), "volume": np.arange(n, dtype=float)})
res = CandleReconstructor("5min").reconstruct_symbol_candles(df)
out["kat/ts"] = ts.astype("datetime64[ns]").astype(np.int64)
for k, col in (("o", "open"), ("h", "high"), ("l", "low"), ("c", "close"), ("v", "volume")):
    out[f"kat/{k}"] = df[col].to_numpy()
out["kat/none"] = np.array(False); out["kat/out_ts"] = res["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64)
for k in ("open", "high", "low", "close", "volume"):
    out[f"kat/out_{k}"] = res[k].to_numpy(np.float64)
cases.append(("kat", "5min"))
case("full_day_5", mins(np.arange(1440)), "5min")
case("full_day_15", mins(np.arange(1440) + 7), "15min")
case("gaps_5", mins(np.sort(r.choice(2000, 1400, replace=False))), "5min")
case("nans_5", mins(np.arange(600)), "5min", nan_frac=0.1)
case("shuffled_3", mins(r.permutation(300)), "3m")
case("too_few", mins(np.arange(3)), "5min")
case("offgrid_seconds", (base + pd.to_timedelta(np.arange(400) * 60 + 17, unit="s")).to_numpy(), "5min")
out["names"] = np.array([c[0] for c in cases]); out["freqs"] = np.array([c[1] for c in cases])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "candles.npz"), **out)
print("candle golden cases:", len(cases))
