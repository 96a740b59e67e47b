"""IV -> OHLCV bridge.
CPU: the oracle (oracle/bridge_oracle.py) against the real reference's seeded outputs (tests/golden/bridge.npz), the
host mirror of the converter class over the oracle backend.  GPU: the HIP path through the C ABI against both."""
import os
import sys

import numpy as np
import pandas as pd
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import bridge_oracle as BO  # noqa: E402
from golden_io import GOLDEN  # noqa: E402

G = np.load(os.path.join(GOLDEN, "bridge.npz"))
NAMES = [str(n) for n in G["names"]]
COLS = ("open", "high", "low", "close", "volume", "source_price")


def golden_case(name):
    price = G[f"{name}/price"]
    volume = G[f"{name}/volume"] if bool(G[f"{name}/has_volume"]) else None
    return price, volume, BO.strategy_code(str(G[f"{name}/strategy"])), int(G[f"{name}/seed"])


def exp_draws(price, volume):
    """rows whose volume comes from an exponential draw (log: 1-ulp library differences are allowed there)"""
    ok = ~(np.isnan(price) | (price <= 0))
    if volume is None:
        return np.ones(int(ok.sum()), bool)
    v = volume[ok]
    return np.isnan(v) | (v <= 0)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_reference(name):
    price, volume, strat, seed = golden_case(name)
    valid, d = BO.candles(price, volume, strat, seed=seed)
    if bool(G[f"{name}/none"]):
        assert not valid.any()
        return
    for k in COLS:
        assert np.array_equal(d[k], G[f"{name}/out_{k}"]), (name, k)          # same library, same order: bit-exact
    ok, reason = BO.validate_quality(d["open"], d["high"], d["low"], d["close"], d["source_price"])
    assert ok == bool(G[f"{name}/quality_valid"]) and reason == str(G[f"{name}/quality_reason"])


def test_quality_verdicts_and_stream_words():
    for nm in G["quality_names"]:
        a = {k: G[f"quality/{nm}/{k}"] for k in ("open", "high", "low", "close", "source_price")}
        ok, reason = BO.validate_quality(a["open"], a["high"], a["low"], a["close"], a["source_price"])
        assert ok == bool(G[f"quality/{nm}/valid"]) and reason == str(G[f"quality/{nm}/reason"]), nm
    for seed in G["mt_seeds"]:
        assert np.array_equal(BO.mt19937_words(int(seed), 2000), G[f"mt/{int(seed)}"])


def test_batch_stream_continues_across_symbols():
    r = np.random.default_rng(3)
    price = 100 * np.exp(np.cumsum(r.normal(0, 1e-3, 300))); volume = r.uniform(0, 5, 300); volume[::7] = np.nan
    off = np.array([0, 100, 100, 180, 300])
    for strat in range(5):
        valid, out = BO.candles_batch(price, volume, off, strat, seed=5)
        rs = np.random.RandomState(5)
        for s in range(4):
            a, b = off[s], off[s + 1]
            v, d = BO.candles(price[a:b], volume[a:b], strat, rs=rs)
            assert np.array_equal(out[3, a:b][v], d["close"])


# ---------------------------------------------------------------- host mirror of the converter class (CPU, oracle backend)
def _config(strategy):
    import config as cfgmod
    cfg = cfgmod.get_config()
    cfg.data_bridge.conversion_strategy = strategy
    return cfg


def _frame(name):
    price, volume, _, _ = golden_case(name)
    n = len(price)
    df = pd.DataFrame({"symbol": ["BTC-29MAR24-25000-C"] * n, "timestamp": pd.date_range("2024-01-01", periods=n, freq="1min"),
                       "mark_price": price})
    if volume is not None:
        df["volume"] = volume
    return df


def _check_frame_against_golden(res, name, exact=True, atol=0.0):
    if bool(G[f"{name}/none"]):
        assert res is None
        return
    assert list(res.columns) == [str(c) for c in G[f"{name}/columns"]]
    assert np.array_equal(res.index.to_numpy(), G[f"{name}/index"])
    assert np.array_equal(res["timestamp"].to_numpy().astype("datetime64[ns]").astype(np.int64), G[f"{name}/ts"])
    assert (res["conversion_method"] == str(G[f"{name}/method_label"])).all() and res["is_synthetic"].all()
    price, volume, _, _ = golden_case(name)
    drawn = exp_draws(price, volume)
    for k in COLS:
        got, exp = res[k].to_numpy(np.float64), G[f"{name}/out_{k}"]
        if exact and k != "volume":
            assert np.array_equal(got, exp), (name, k, np.abs(got - exp).max())
        elif exact:
            assert np.array_equal(got[~drawn], exp[~drawn]), (name, k)
            assert np.allclose(got[drawn], exp[drawn], rtol=0, atol=1.000001e-6), (name, k)   # device log vs libm
        else:
            assert np.allclose(got, exp, rtol=0, atol=atol), (name, k, np.abs(got - exp).max())


@pytest.mark.parametrize("name", NAMES)
def test_converter_class_over_oracle_backend(name):
    from oracle_backend import OracleBridgeBackend
    from iv_interpolation_amd.bridge import InterpolatedToOHLCVConverter
    c = InterpolatedToOHLCVConverter(None, _config(str(G[f"{name}/strategy"])), backend=OracleBridgeBackend(),
                                     seed=int(G[f"{name}/seed"]))
    res = c._generate_ohlcv_from_interpolated(_frame(name))
    _check_frame_against_golden(res, name)
    if res is not None:
        q = c._validate_ohlcv_quality(res)
        assert q["valid"] == bool(G[f"{name}/quality_valid"]) and q["reason"] == str(G[f"{name}/quality_reason"])


def test_converter_import_path_selection_quality_and_stream_state():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "src"))
    from data_bridge.ohlcv_converter import InterpolatedToOHLCVConverter          # the reference's import path
    from oracle_backend import OracleBridgeBackend
    c = InterpolatedToOHLCVConverter(None, _config("spread_simulation"), backend=OracleBridgeBackend(), seed=3)
    sel = {"u": dict(underlying_price=[1.0] * 10, mark_price=[2.0] * 10),
           "m": dict(underlying_price=[np.nan] * 5 + [1.0] * 5, mark_price=[2.0] * 10),
           "i": dict(index_price=[3.0] * 10),
           "fallback": dict(underlying_price=[np.nan] * 9 + [1.0], mark_price=[np.nan] * 10),
           "edge80": dict(underlying_price=[np.nan] * 2 + [1.0] * 8, mark_price=[2.0] * 10)}
    for nm, col in zip(G["select_names"], G["select_cols"]):
        assert c._select_price_column(pd.DataFrame(sel[str(nm)])) == str(col)
    with pytest.raises(ValueError):
        c._select_price_column(pd.DataFrame({"iv": [1.0]}))
    for nm in G["quality_names"]:
        fr = pd.DataFrame({k: G[f"quality/{nm}/{k}"] for k in ("open", "high", "low", "close", "source_price")})
        q = c._validate_ohlcv_quality(fr)
        assert q["valid"] == bool(G[f"quality/{nm}/valid"]) and q["reason"] == str(G[f"quality/{nm}/reason"]), nm
    # the stream continues from call to call exactly like the global generator of the reference
    for strat in ("spread_simulation", "trend_following"):
        c = InterpolatedToOHLCVConverter(None, _config(strat), backend=OracleBridgeBackend(), seed=21)
        f1, f2 = _frame("spread_simulation/holes"), _frame("trend_following/novolume")
        a1 = c._generate_ohlcv_from_interpolated(f1)
        a2 = c._generate_ohlcv_from_interpolated(f2)
        c2 = InterpolatedToOHLCVConverter(None, _config(strat), backend=OracleBridgeBackend(), seed=21)
        b1, b2 = c2.convert_frames([f1, f2])
        pd.testing.assert_frame_equal(a1, b1); pd.testing.assert_frame_equal(a2, b2)
        rs = np.random.RandomState(21)
        _, d1 = BO.candles(f1["mark_price"].to_numpy(), f1["volume"].to_numpy(), BO.strategy_code(strat), rs=rs)
        _, d2 = BO.candles(f2["mark_price"].to_numpy(), None, BO.strategy_code(strat), rs=rs)
        assert np.array_equal(a1["close"].to_numpy(), d1["close"]) and np.array_equal(a2["close"].to_numpy(), d2["close"])
    # frames the reference turns into None
    assert c._generate_ohlcv_from_interpolated(pd.DataFrame()) is None
    assert c._generate_ohlcv_from_interpolated(pd.DataFrame({"symbol": ["s"], "timestamp": ["2024-01-01"], "iv": [0.5]})) is None


def test_convert_batch_and_symbol_listing_on_the_frame_store(tmp_path):
    """ohlcv_converter.py:31-55 and :463-493 (reference main.py:400 calls convert_batch): frame-store backed, one
    generator stream over the symbols in order, the reference's tally dictionary."""
    from oracle_backend import OracleBridgeBackend
    from iv_interpolation_amd.bridge import InterpolatedToOHLCVConverter
    from iv_interpolation_amd.frame_store import FrameStore, synthetic_symbol
    store = FrameStore(str(tmp_path))
    for i, (sym, batch) in enumerate((("aaa", 7), ("bbb", 7), ("ccc", 8))):
        f = synthetic_symbol(sym, 6, seed=i); f["is_interpolated"] = False
        store.write_output(sym, f, batch)
    store.write_output("empty", synthetic_symbol("empty", 6, seed=9).iloc[:0], 7)
    c = InterpolatedToOHLCVConverter(store, _config("simple_spread"), backend=OracleBridgeBackend(), seed=5)
    assert c.get_interpolated_symbols() == ["aaa", "bbb", "ccc", "empty"]
    assert c.get_interpolated_symbols(batch_id=7) == ["aaa", "bbb"]
    tally = c.convert_batch(batch_id=7)
    assert tally == {"total": 2, "success": 2, "errors": 0, "skipped": 0}
    assert store.symbols("minute_candles") == ["aaa", "bbb"] and len(store.read_table("minute_candles", "aaa")) == 6
    assert c.convert_batch(symbols=["empty", "nope"]) == {"total": 2, "success": 0, "errors": 0, "skipped": 2}
    assert c.convert_batch(symbols=[]) == {"total": 0, "success": 0, "errors": 0, "skipped": 0}


# ---------------------------------------------------------------- GPU
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_gpu_mt19937_stream_equals_numpy():
    from iv_interpolation_amd import engine
    for seed in G["mt_seeds"]:
        w = engine.mt19937_words(int(seed), 2000).cpu().numpy().view(np.uint32)
        assert np.array_equal(w, G[f"mt/{int(seed)}"]), int(seed)
    n = 624 * 7 + 5                                   # several regenerations, ragged tail
    w = engine.mt19937_words(12345, n).cpu().numpy().view(np.uint32)
    assert np.array_equal(w, BO.mt19937_words(12345, n))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_converter_equals_reference(name):
    from iv_interpolation_amd.bridge import InterpolatedToOHLCVConverter
    strat = str(G[f"{name}/strategy"])
    c = InterpolatedToOHLCVConverter(None, _config(strat), seed=int(G[f"{name}/seed"]))
    res = c._generate_ohlcv_from_interpolated(_frame(name))
    if strat == "trend_following":        # log / sqrt of the polar method: 1-ulp library differences, rounded to 1e-4
        _check_frame_against_golden(res, name, exact=False, atol=2.0e-4)
    else:
        _check_frame_against_golden(res, name)


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", [0, 1, 2, 3, 4])     # 4 = the pipeline's inline variant: parity unpinned (oracle only)
def test_gpu_bridge_batch_vs_oracle(strategy):
    """300 symbols x ragged lengths (empty ones included), holes in price and volume, one stream; then a second call
    continuing the same stream."""
    from iv_interpolation_amd import engine
    r = np.random.default_rng(100 + strategy)
    S = 300
    lens = r.integers(0, 400, S); lens[::17] = 0; lens[5] = 1500
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = int(off[-1])
    price = 2000 * np.exp(np.cumsum(r.normal(0, 5e-4, n))); price[r.random(n) < 0.05] = np.nan; price[r.random(n) < 0.01] = -1.0
    volume = r.uniform(0, 30, n).round(4); volume[r.random(n) < 0.2] = np.nan; volume[r.random(n) < 0.1] = 0.0
    seed = 77
    words = engine.mt19937_words(seed, 2 * engine.bridge_words_bound(n, strategy))
    out, valid, tail = engine.bridge_candles(_dev(price), _dev(volume), _dev(off), strategy, words)
    tail_h = tail.cpu().numpy()
    assert tail_h[3] == 0
    rvalid, rout = BO.candles_batch(price, volume, off, strategy, seed)
    got = out.cpu().numpy()
    assert np.array_equal(valid.cpu().numpy().astype(bool), rvalid)
    assert np.array_equal(np.isnan(got), np.isnan(rout))
    tol = 2.0e-4 if strategy == 2 else 0.0
    for j, k in enumerate(COLS):
        if k == "volume":
            assert np.allclose(got[j], rout[j], rtol=0, atol=1.000001e-6, equal_nan=True)
        elif tol:
            assert np.allclose(got[j], rout[j], rtol=0, atol=tol, equal_nan=True), (k, np.nanmax(np.abs(got[j] - rout[j])))
        else:
            assert np.array_equal(got[j], rout[j], equal_nan=True), (k, np.nanmax(np.abs(got[j] - rout[j])))
    # second call: continue the stream where the first stopped (position + cached deviate)
    used = int(tail_h[0])
    t2 = tail.clone(); t2[0] = 0; t2[3] = 0
    out2, valid2, tail2 = engine.bridge_candles(_dev(price[:500]), None, _dev(np.array([0, 200, 500])), strategy, words[used:], t2)
    rs = np.random.RandomState(seed); rs.bytes(4 * used)
    st = rs.get_state(); rs.set_state((st[0], st[1], st[2], int(tail_h[1]), float(np.int64(tail_h[2]).view(np.float64))))
    ref2 = []
    for a, b in ((0, 200), (200, 500)):
        v, d = BO.candles(price[a:b], None, strategy, rs=rs)
        ref2.append(d["close"])
    got2 = out2.cpu().numpy()[3]
    got2 = got2[~np.isnan(got2)]
    assert np.allclose(got2, np.concatenate(ref2), rtol=0, atol=tol)


@pytest.mark.gpu
def test_gpu_bridge_reports_short_word_buffer_and_empty_input():
    from iv_interpolation_amd import engine
    price = np.full(1000, 100.0); off = np.array([0, 1000])
    words = engine.mt19937_words(1, 100)
    for strategy in (0, 2):
        _, _, tail = engine.bridge_candles(_dev(price), None, _dev(off), strategy, words)
        assert int(tail.cpu()[3]) == 1
    out, valid, tail = engine.bridge_candles(_dev(np.zeros(0)), None, _dev(np.array([0])), 0, words)
    assert out.shape == (6, 0) and int(tail.cpu()[0]) == 0
