"""CPU: the kept entry surface (config.py / complete_pipeline.py / main.py) runs end to end against the
DB-less frame store.  The device calls are answered by the oracle backend (tests only)."""
import os

import numpy as np
import pandas as pd

from oracle_backend import OracleBackend


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
    assert list(out.columns)[-2:] == ["is_interpolated", "batch_id"] and not out["is_interpolated"].any()
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
    assert entry.main(["--task", "candles", "--data-dir", str(tmp_path)], backend=OracleBackend()) == 1
