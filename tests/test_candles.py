"""Candle aggregation: oracle vs the real reference's golden outputs (CPU), HIP kernel vs both (GPU, bit-exact)."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import candles_oracle as CO  # noqa: E402
from golden_io import GOLDEN  # noqa: E402

g = np.load(os.path.join(GOLDEN, "candles.npz"))
CASES = [(str(n), str(f)) for n, f in zip(g["names"], g["freqs"])]
KEYS = ("open", "high", "low", "close", "volume")


def _minutes(f):
    return int(f[:-3]) if f.endswith("min") else int(f[:-1])


@pytest.mark.parametrize("name,freq", CASES)
def test_oracle_against_reference_golden(name, freq):
    r = CO.aggregate(*[g[f"{name}/{k}"] for k in ("ts", "o", "h", "l", "c", "v")], _minutes(freq))
    if bool(g[f"{name}/none"]):
        assert r is None
        return
    assert np.array_equal(r["timestamp"], g[f"{name}/out_ts"])
    for k in KEYS:
        assert np.array_equal(r[k], g[f"{name}/out_{k}"], equal_nan=True), k     # bit-exact incl. Kahan volume


def test_known_answer_survey_8f():
    r = CO.aggregate(g["kat/ts"], g["kat/o"], g["kat/h"], g["kat/l"], g["kat/c"], g["kat/v"], 5)
    assert [tuple(r[k][i] for k in KEYS) for i in range(2)] == [(100, 106, 99, 105, 10), (105, 111, 104, 110, 35)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,freq", CASES)
def test_hip_candles_against_reference_golden(name, freq):
    from iv_interpolation_amd.candles import CandleReconstructor
    df = pd.DataFrame({"symbol": "btc-20mar23-25000-c", "timestamp": pd.to_datetime(g[f"{name}/ts"]),
                       **{c: g[f"{name}/{k}"] for c, k in zip(KEYS, "ohlcv")}})
    out = CandleReconstructor(freq).reconstruct_symbol_candles(df)
    if bool(g[f"{name}/none"]):
        assert out is None
        return
    assert list(out.columns) == ["symbol", "timestamp", "open", "high", "low", "close", "volume", "frequency", "source_candles", "created_at"]
    assert np.array_equal(out["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64), g[f"{name}/out_ts"])
    for k in KEYS:
        assert np.array_equal(out[k].to_numpy(), g[f"{name}/out_{k}"], equal_nan=True), k


@pytest.mark.gpu
def test_hip_candles_batch_vs_oracle():
    from iv_interpolation_amd.candles import CandleReconstructor
    r = np.random.default_rng(3)
    frames = []
    for s in range(50):
        n = int(r.integers(3, 3000))
        ts = pd.Timestamp("2023-03-01") + pd.to_timedelta(np.sort(r.choice(4000, n, replace=False)), unit="min")
        o = r.uniform(90, 110, n); frames.append(pd.DataFrame({"symbol": f"s{s}", "timestamp": ts, "open": o, "high": o + 1, "low": o - 1,
                                                                "close": o + r.normal(0, .3, n), "volume": r.uniform(0, 9, n)}))
    outs = CandleReconstructor("5min").reconstruct_batch(frames)
    for f, o in zip(frames, outs):
        ref = CO.aggregate(f["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64), f["open"], f["high"], f["low"], f["close"], f["volume"], 5)
        if ref is None:
            assert o is None
            continue
        assert np.array_equal(o["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64), ref["timestamp"])
        for k in KEYS:
            assert np.array_equal(o[k].to_numpy(), ref[k], equal_nan=True)
