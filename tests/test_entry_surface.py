"""CPU: the kept entry surface (config.py / complete_pipeline.py / main.py) runs end to end against the
DB-less frame store.  The device calls are answered by the oracle backend (tests only)."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle_backend import OracleBackend, OracleBridgeBackend, OracleCandleBackend


def test_config_surface():
    import config
    c = config.get_config()
    assert c.interpolation.method == "linear" and c.interpolation.min_data_points == 10      # reference config.py:42,44
    assert c.processing.max_workers == 32 and c.processing.symbols_per_batch == 100
    assert set(c.database.to_dict()) == {"host", "database", "user", "password", "port"}
    os.environ["ENVIRONMENT"] = "development"
    try:
        assert config.get_config().processing.max_workers == 4 and config.get_config().debug
    finally:
        os.environ.pop("ENVIRONMENT")


def test_complete_pipeline_task1(tmp_path):
    import complete_pipeline
    rc = complete_pipeline.main(["--task", "interpolation", "--synthetic", "4", "--data-dir", str(tmp_path)],
                                backend=OracleBackend())
    assert rc == 0
    store = complete_pipeline.FrameStore(str(tmp_path))
    syms = store.symbols("interpolated_trading_tickers")
    assert len(syms) == 4
    out = store.read_output(syms[0])
    src = store.read_symbol(syms[0])
    assert len(out) == (len(src) - 1) * 60 + 1                       # hourly -> 1-minute lattice
    # config.interpolation.preserve_greeks defaults to True (reference config.py:46): the rows carry delta..rho
    assert list(out.columns)[-7:] == ["is_interpolated", "delta", "gamma", "theta", "vega", "rho", "batch_id"]
    assert not out["is_interpolated"].any() and out[["delta", "gamma", "theta", "vega", "rho"]].notna().all().all()
    assert ((out["delta"] > 0) & (out["delta"] < 1) & (out["gamma"] > 0) & (out["vega"] > 0)).all()      # calls
    # knots are reproduced exactly, interior rows are np.interp of the neighbours
    on = out[out["date"].isin(src["date"])]
    assert np.array_equal(on["iv"].to_numpy(), src["iv"].to_numpy())
    mid = out.iloc[30]
    assert abs(mid["iv"] - 0.5 * (src["iv"][0] + src["iv"][1])) < 1e-15
    # second run: nothing pending (idempotent by anti-join)
    assert complete_pipeline.main(["--task", "interpolation", "--data-dir", str(tmp_path)], backend=OracleBackend()) == 1


def test_main_entry(tmp_path):
    import complete_pipeline
    import main as entry
    complete_pipeline.main(["--setup-only", "--synthetic", "2", "--data-dir", str(tmp_path)], backend=OracleBackend())
    assert entry.main(["--test", "--data-dir", str(tmp_path)], backend=OracleBackend()) == 0
    assert entry.main(["--task", "candles", "--data-dir", str(tmp_path)], backend=OracleBackend(),
                      candle_backend=OracleCandleBackend()) == 1                     # no minute candles yet


def test_full_pipeline_bridge_and_candles(tmp_path):
    """Task 1 -> bridge -> Task 2 through both entry scripts on the frame store (device calls answered by the oracles)."""
    _full_pipeline(tmp_path, dict(backend=OracleBackend(), bridge_backend=OracleBridgeBackend(),
                                  candle_backend=OracleCandleBackend(), seed=99))


@pytest.mark.gpu
def test_full_pipeline_on_gpu(tmp_path):
    """The same three stages on the MI355X engine (no injected backends), checked against the oracles."""
    _full_pipeline(tmp_path, dict(seed=99))


def _full_pipeline(tmp_path, kw):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import bridge_oracle as BO
    import complete_pipeline
    rc = complete_pipeline.main(["--task", "all", "--synthetic", "3", "--data-dir", str(tmp_path)], **kw)
    assert rc == 0
    store = complete_pipeline.FrameStore(str(tmp_path))
    syms = store.symbols("interpolated_trading_tickers")
    assert store.symbols("minute_candles") == syms and store.symbols("reconstructed_candles") == syms
    # bridge: one candle per interpolated row, the inline builder of the reference pipeline on ONE stream in symbol order
    rs = np.random.RandomState(99)
    for s in syms:
        itp = store.read_output(s).sort_values("date")
        mc = store.read_table("minute_candles", s)
        base = BO.pipeline_base_price(itp["underlying_price"], itp["mark_price"], itp["index_price"])
        v, d = BO.candles(base, itp["volume"].to_numpy(np.float64), BO.PIPELINE_INLINE, rs=rs)
        assert len(mc) == int(v.sum()) == len(itp)
        assert list(mc.columns) == ["symbol", "timestamp", "open", "high", "low", "close", "volume", "source_price", "is_synthetic"]
        for k in ("open", "high", "low", "close", "volume", "source_price"):
            # CSV round trip (read_csv fast parser); a drawn volume goes through the device's log on the GPU
            assert np.allclose(mc[k].to_numpy(), d[k], rtol=1e-14, atol=1.1e-6 if k == "volume" else 0), k
        assert (mc["high"] >= mc[["open", "close"]].max(axis=1)).all() and (mc["low"] <= mc[["open", "close"]].min(axis=1)).all()
        # Task 2: complete 5-minute groups only
        rc5 = store.read_table("reconstructed_candles", s)
        assert list(rc5.columns) == ["timestamp", "open", "high", "low", "close", "volume", "symbol", "frequency",
                                     "source_candles", "batch_id"]
        g = mc.assign(b=mc["timestamp"].dt.floor("5min")).groupby("b")
        full = g.size()[g.size() >= 5].index
        assert np.array_equal(rc5["timestamp"].to_numpy(), full.to_numpy())
        assert np.allclose(rc5["high"].to_numpy(), g["high"].max()[full].to_numpy())
        assert np.allclose(rc5["open"].to_numpy(), g["open"].first()[full].to_numpy())
        assert np.allclose(rc5["volume"].to_numpy(), g["volume"].sum()[full].to_numpy())
    st = complete_pipeline.CompleteOptimizedPipeline(complete_pipeline.get_config(), data_dir=str(tmp_path)).get_pipeline_status()
    assert st["bridge_symbols"] == 3 and st["task2_symbols"] == 3
    # main.py dispatch
    import main as entry
    assert entry.main(["--task", "bridge", "--data-dir", str(tmp_path)], **kw) == 1       # anti-join: nothing left to convert
    assert entry.main(["--task", "pipeline", "--test", "--data-dir", str(tmp_path)], **kw) == 0
