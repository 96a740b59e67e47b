"""DB-less source/sink with the table contract of the reference's PostgreSQL layer.

The reference reads ``trading_tickers`` one symbol at a time (``SELECT <13 columns> ... WHERE symbol=%s
ORDER BY date``, batch_processor.py:77-87) and INSERTs the interpolated rows plus ``batch_id`` into
``interpolated_trading_tickers`` (batch_processor.py:154-173); the downstream stages only ever see that table.
This store keeps the same two "tables" as directories of per-symbol CSV files so the entry scripts run
without PostgreSQL/psycopg2 (absent in the build image).  Storage is out of the hot path's scope."""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import pandas as pd

SOURCE_TABLE = "trading_tickers"
OUTPUT_TABLE = "interpolated_trading_tickers"
SOURCE_COLUMNS = ["symbol", "date", "iv", "underlying_price", "time_to_maturity", "strike", "callput",
                  "interest_rate", "mark_price", "index_price", "volume", "quote_volume", "record_time"]


class FrameStore:
    def __init__(self, root: str):
        self.root = root
        for t in (SOURCE_TABLE, OUTPUT_TABLE):
            os.makedirs(os.path.join(root, t), exist_ok=True)

    def _path(self, table: str, symbol: str) -> str:
        return os.path.join(self.root, table, symbol.replace("/", "_") + ".csv")

    def symbols(self, table: str = SOURCE_TABLE) -> List[str]:
        d = os.path.join(self.root, table)
        if not os.path.isdir(d):
            return []
        return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".csv"))

    def pending_symbols(self) -> List[str]:
        """Symbols without output yet (the anti-join of complete_pipeline.py:280-296)."""
        done = set(self.symbols(OUTPUT_TABLE))
        return [s for s in self.symbols(SOURCE_TABLE) if s not in done]

    def read_symbol(self, symbol: str) -> pd.DataFrame:
        df = pd.read_csv(self._path(SOURCE_TABLE, symbol))
        df["date"] = pd.to_datetime(df["date"])
        return df.sort_values("date").reset_index(drop=True)

    def write_source(self, symbol: str, df: pd.DataFrame) -> None:
        df.to_csv(self._path(SOURCE_TABLE, symbol), index=False)

    def write_output(self, symbol: str, df: pd.DataFrame, batch_id: int) -> int:
        out = df.copy()
        out["batch_id"] = batch_id                      # batch_processor.py:105, complete_pipeline.py:323
        out.to_csv(self._path(OUTPUT_TABLE, symbol), index=False)
        return len(out)

    def read_output(self, symbol: str) -> Optional[pd.DataFrame]:
        p = self._path(OUTPUT_TABLE, symbol)
        return pd.read_csv(p, parse_dates=["date"]) if os.path.exists(p) else None

    # downstream tables (minute_candles, reconstructed candles): same per-symbol CSV convention
    def write_table(self, table: str, symbol: str, df: pd.DataFrame) -> int:
        os.makedirs(os.path.join(self.root, table), exist_ok=True)
        df.to_csv(self._path(table, symbol), index=False)
        return len(df)

    def read_table(self, table: str, symbol: str) -> Optional[pd.DataFrame]:
        p = self._path(table, symbol)
        if not os.path.exists(p):
            return None
        df = pd.read_csv(p)
        if "timestamp" in df.columns:
            df["timestamp"] = pd.to_datetime(df["timestamp"])
        return df


def synthetic_symbol(symbol: str, n_hours: int = 48, seed: int = 0, start: str = "2023-03-01") -> pd.DataFrame:
    """An hourly option-quote series shaped like the reference's source rows (13 columns)."""
    r = np.random.default_rng(seed)
    dates = pd.date_range(start, periods=n_hours, freq="1h")
    spot = 25000 + np.cumsum(r.normal(0, 40, n_hours))
    return pd.DataFrame({
        "symbol": symbol, "date": dates,
        "iv": 0.6 + 0.05 * np.sin(np.arange(n_hours) / 3.0) + r.normal(0, 0.01, n_hours),
        "underlying_price": spot, "time_to_maturity": 0.05 - np.arange(n_hours) / (24 * 365.0),
        "strike": 25000.0, "callput": "c", "interest_rate": 0.01,
        "mark_price": 0.03 + r.normal(0, 0.001, n_hours), "index_price": spot + r.normal(0, 5, n_hours),
        "volume": r.uniform(0, 10, n_hours).round(3), "quote_volume": r.uniform(0, 1e5, n_hours).round(1),
        "record_time": (dates.asi8 // 10**6).astype(np.float64)})[SOURCE_COLUMNS]
