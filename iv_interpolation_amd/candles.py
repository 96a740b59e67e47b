"""Drop-in for the reference's ``CandleReconstructor`` (src/candle_reconstruction/core.py:11-106) on the MI355X
engine: same constructor, same ``reconstruct_symbol_candles(minute_data) -> Optional[DataFrame]`` contract and output
columns; ``reconstruct_batch`` aggregates many symbols in one launch."""
import logging
from datetime import datetime
from typing import List, Optional, Sequence

import numpy as np
import pandas as pd

from . import engine

logger = logging.getLogger("candle_reconstruction.core")
COLUMN_ORDER = ["symbol", "timestamp", "open", "high", "low", "close", "volume", "frequency", "source_candles", "created_at"]


class HipCandleBackend:
    """Packed host columns -> device -> sparse per-row result of ivs_candle_aggregate_f64 (see engine.candle_aggregate)."""

    def candle_aggregate(self, ts, cols, off, minutes):
        torch = engine.require_device()
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
        out = engine.candle_aggregate(dev(ts), *[dev(c) for c in cols], dev(off), minutes)
        return [t.cpu().numpy() for t in out]


class CandleReconstructor:
    def __init__(self, target_frequency: str = "5min", backend=None):
        self.target_frequency = target_frequency
        self.frequency_minutes = self._parse_frequency(target_frequency)
        self._backend = backend          # None -> HipCandleBackend on first use

    def _parse_frequency(self, freq: str) -> int:
        if freq.endswith("min"):
            return int(freq[:-3])
        if freq.endswith("m"):
            return int(freq[:-1])
        raise ValueError(f"Unsupported frequency: {freq}")

    def reconstruct_symbol_candles(self, minute_data: pd.DataFrame) -> Optional[pd.DataFrame]:
        return self.reconstruct_batch([minute_data])[0]

    def reconstruct_batch(self, frames: Sequence[pd.DataFrame]) -> List[Optional[pd.DataFrame]]:
        res: List[Optional[pd.DataFrame]] = [None] * len(frames)
        live, packed = [], []
        for i, f in enumerate(frames):
            if f.empty:
                logger.warning("No minute data provided"); continue
            missing = [c for c in ("timestamp", "open", "high", "low", "close", "volume") if c not in f.columns]
            if missing:
                logger.error(f"Missing required columns: {missing}"); continue
            try:
                d = f.sort_values("timestamp").copy()                       # core.py:59
                d["timestamp"] = pd.to_datetime(d["timestamp"])
                if len(d) < self.frequency_minutes:                          # core.py:63-66
                    logger.warning(f"Insufficient data: {len(d)} rows, need at least {self.frequency_minutes}"); continue
                live.append(i); packed.append(d)
            except Exception as e:
                logger.error(f"Candle reconstruction failed: {e}")
        if not live:
            return res
        off = np.concatenate([[0], np.cumsum([len(d) for d in packed])]).astype(np.int64)
        ts = np.concatenate([pd.DatetimeIndex(d["timestamp"]).as_unit("ns").asi8 for d in packed])
        cols = [np.concatenate([d[c].to_numpy(np.float64) for d in packed]) for c in ("open", "high", "low", "close", "volume")]
        be = self._backend or HipCandleBackend()
        out = be.candle_aggregate(ts, cols, off, self.frequency_minutes)
        cnt = out[6]
        for k, i in enumerate(live):
            a, b = off[k], off[k + 1]
            keep = np.flatnonzero(cnt[a:b] >= self.frequency_minutes) + a     # complete groups only (core.py:86-88)
            d = packed[k]
            tsv = pd.DatetimeIndex(out[0][keep])
            if getattr(d["timestamp"].dt, "tz", None) is not None:
                tsv = tsv.tz_localize("UTC").tz_convert(d["timestamp"].dt.tz)
            sym = d["symbol"].iloc[0] if "symbol" in d.columns else None
            agg = pd.DataFrame({"symbol": sym, "timestamp": tsv, "open": out[1][keep], "high": out[2][keep],
                                "low": out[3][keep], "close": out[4][keep], "volume": out[5][keep]})
            agg["frequency"] = self.target_frequency
            agg["source_candles"] = self.frequency_minutes
            agg["created_at"] = datetime.now()
            res[i] = agg[COLUMN_ORDER]
        return res
