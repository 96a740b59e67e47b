#!/usr/bin/env python3
"""Golden vectors for the IV -> OHLCV bridge from the REAL reference (src/data_bridge/ohlcv_converter.py), data only.
The reference draws from the unseeded global NumPy generator; each case seeds it first (np.random.seed) so that the
outputs are reproducible.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bridge.py"""
import logging
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import pandas as pd

sys.path.insert(0, "/root/reference/src")
from data_bridge.ohlcv_converter import InterpolatedToOHLCVConverter   # the real reference

logging.getLogger("data_bridge.ohlcv_converter").setLevel(logging.CRITICAL)


class NS:
    pass


def converter(strategy):
    cfg = NS(); cfg.data_bridge = NS()
    cfg.data_bridge.conversion_strategy = strategy
    cfg.data_bridge.spread_method = "adaptive"
    cfg.data_bridge.enable_quality_checks = True
    cfg.data_bridge.spread_parameters = {"base_spread_percent": 0.002, "volatility_factor": 1.5}
    return InterpolatedToOHLCVConverter(None, cfg)


r = np.random.default_rng(20230320)
out = {}
names = []


def case(name, strategy, seed, price, volume, with_volume=True, price_col="mark_price"):
    n = len(price)
    df = pd.DataFrame({"symbol": ["BTC-29MAR24-25000-C"] * n,
                       "timestamp": pd.date_range("2024-01-01", periods=n, freq="1min"), price_col: price})
    if with_volume:
        df["volume"] = volume
    c = converter(strategy)
    np.random.seed(seed)
    res = c._generate_ohlcv_from_interpolated(df)
    out[f"{name}/price"] = np.asarray(price, np.float64)
    out[f"{name}/volume"] = np.asarray(volume, np.float64) if with_volume else np.zeros(0)
    out[f"{name}/has_volume"] = np.array(with_volume)
    out[f"{name}/seed"] = np.array(seed)
    out[f"{name}/strategy"] = np.array(strategy)
    out[f"{name}/none"] = np.array(res is None)
    if res is not None:
        out[f"{name}/ts"] = res["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64)
        for k in ("open", "high", "low", "close", "volume", "source_price"):
            out[f"{name}/out_{k}"] = res[k].to_numpy(np.float64)
        out[f"{name}/method_label"] = np.array(str(res["conversion_method"].iloc[0]))
        out[f"{name}/columns"] = np.array(list(res.columns))
        out[f"{name}/index"] = res.index.to_numpy()
        q = c._validate_ohlcv_quality(res)
        out[f"{name}/quality_valid"] = np.array(bool(q["valid"]))
        out[f"{name}/quality_reason"] = np.array(q["reason"])
    names.append(name)


def walk(n, s0=25000.0, vol=0.0008):
    return s0 * np.exp(np.cumsum(r.normal(0, vol, n)))


for strat in ("spread_simulation", "price_as_midpoint", "trend_following", "simple_spread", "no_such_strategy"):
    n = 400
    p = walk(n)
    v = r.uniform(0.01, 80.0, n).round(3)
    case(f"{strat}/clean", strat, 11, p, v)
    p2 = walk(n, 1800.0)
    p2[r.random(n) < 0.1] = np.nan
    p2[r.random(n) < 0.03] = 0.0
    p2[r.random(n) < 0.02] = -5.0
    v2 = r.uniform(0.01, 80.0, n).round(3)
    v2[r.random(n) < 0.25] = np.nan
    v2[r.random(n) < 0.10] = 0.0
    case(f"{strat}/holes", strat, 12, p2, v2)
    case(f"{strat}/novolume", strat, 13, walk(150, 0.37), None, with_volume=False)
    # dyadic prices: exact decimal ties in round(x, 4) (x * 1e4 = k + 0.5 exactly)
    pd_ = 100.0 + np.arange(64) / 32.0
    case(f"{strat}/dyadic", strat, 14, pd_, np.full(64, 0.03125 * 3))
    case(f"{strat}/tiny", strat, 15, walk(40, 1.2e-3, 0.01), r.uniform(1e-7, 1e-5, 40))
    case(f"{strat}/flat", strat, 16, np.full(50, 123.456), np.full(50, 2.5))
case("spread_simulation/all_invalid", "spread_simulation", 17, np.full(10, np.nan), np.ones(10))
# the reference's known quality-check verdicts on hand-made frames
c = converter("simple_spread")
for nm, fr in (("ok", dict(open=[1.0], high=[1.1], low=[0.9], close=[1.05], source_price=[1.0])),
               ("hl", dict(open=[1.0], high=[0.8], low=[0.9], close=[1.0], source_price=[1.0])),
               ("ho", dict(open=[1.2], high=[1.1], low=[0.9], close=[1.0], source_price=[1.0])),
               ("lo", dict(open=[1.0], high=[1.3], low=[1.05], close=[1.1], source_price=[1.0])),
               ("wide", dict(open=[1.0], high=[1.2], low=[0.9], close=[1.0], source_price=[1.0])),
               ("neg", dict(open=[-1.0], high=[1.0], low=[-1.0], close=[1.0], source_price=[100.0]))):
    q = c._validate_ohlcv_quality(pd.DataFrame(fr))
    for k, a in fr.items():
        out[f"quality/{nm}/{k}"] = np.asarray(a, np.float64)
    out[f"quality/{nm}/valid"] = np.array(bool(q["valid"])); out[f"quality/{nm}/reason"] = np.array(q["reason"])
out["quality_names"] = np.array(["ok", "hl", "ho", "lo", "wide", "neg"])
# price-column selection (ohlcv_converter.py:189-207)
sel = []
for nm, cols in (("u", dict(underlying_price=[1.0] * 10, mark_price=[2.0] * 10)),
                 ("m", dict(underlying_price=[np.nan] * 5 + [1.0] * 5, mark_price=[2.0] * 10)),
                 ("i", dict(index_price=[3.0] * 10)),
                 ("fallback", dict(underlying_price=[np.nan] * 9 + [1.0], mark_price=[np.nan] * 10)),
                 ("edge80", dict(underlying_price=[np.nan] * 2 + [1.0] * 8, mark_price=[2.0] * 10))):
    sel.append((nm, c._select_price_column(pd.DataFrame(cols))))
out["select_names"] = np.array([s[0] for s in sel]); out["select_cols"] = np.array([s[1] for s in sel])
out["names"] = np.array(names)
# raw generator words for the device MT19937
for seed in (0, 1, 11, 20230320, 4294967295):
    out[f"mt/{seed}"] = np.frombuffer(np.random.RandomState(seed).bytes(4 * 2000), dtype="<u4").copy()
out["mt_seeds"] = np.array([0, 1, 11, 20230320, 4294967295], np.int64)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bridge.npz"), **out)
print("bridge golden cases:", len(names))
