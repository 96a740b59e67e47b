"""Drop-in for the reference's ``CandleReconstructor`` (src/candle_reconstruction/core.py:11-106) on the MI355X
engine: same constructor, same ``reconstruct_symbol_candles(minute_data) -> Optional[DataFrame]`` contract and output
columns; ``reconstruct_batch`` aggregates many symbols in one launch."""
import logging
import time
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
        # the reference aggregates 'symbol': 'first' too: a frame without that column fails inside agg() -> None; so do
        # OHLCV columns that are not numeric (both inside its try block, core.py:50-106)
        ok_live, ok_packed, num = [], [], []
        for i, d in zip(live, packed):
            try:
                if "symbol" not in d.columns:
                    raise KeyError("Column(s) ['symbol'] do not exist")
                num.append([d[c].to_numpy(np.float64) for c in ("open", "high", "low", "close", "volume")])
                ok_live.append(i); ok_packed.append(d)
            except Exception as e:
                logger.error(f"Candle reconstruction failed: {e}")
        live, packed = ok_live, ok_packed
        if not live:
            return res
        off = np.concatenate([[0], np.cumsum([len(d) for d in packed])]).astype(np.int64)
        ts = np.concatenate([pd.DatetimeIndex(d["timestamp"]).as_unit("ns").asi8 for d in packed])
        cols = [np.concatenate([n[c] for n in num]) for c in range(5)]
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
            symcol = d["symbol"]
            if symcol.notna().all() and symcol.nunique() == 1:
                sym = symcol.iloc[0]                                          # the usual case: one symbol per frame
            else:                                                             # 'first' = first non-null symbol of each group
                grp = d["timestamp"].dt.floor(f"{self.frequency_minutes}min")
                first = symcol.groupby(grp.to_numpy()).first()
                sym = first.reindex(pd.DatetimeIndex(out[0][keep]) if tsv.tz is None else tsv).to_numpy()
            agg = pd.DataFrame({"symbol": sym, "timestamp": tsv, "open": out[1][keep], "high": out[2][keep],
                                "low": out[3][keep], "close": out[4][keep], "volume": out[5][keep]})
            agg["frequency"] = self.target_frequency
            agg["source_candles"] = self.frequency_minutes
            agg["created_at"] = datetime.now()
            res[i] = agg[COLUMN_ORDER]
        return res

    # ------------------------------------------------------------------ host-side checks (core.py:108-171)
    def validate_candle_data(self, df: pd.DataFrame) -> bool:
        """OHLC ordering, non-negative volume, no nulls in open/high/low/close (core.py:108-144): True / False, never raises."""
        if df.empty:
            return False
        try:
            o, h, lo, c = (df[k] for k in ("open", "high", "low", "close"))
            bad = (h < lo) | (h < o) | (h < c) | (lo > o) | (lo > c)
            if bad.any():
                logger.warning(f"Found {bad.sum()} candles with invalid OHLC relationships")
                return False
            neg = df["volume"] < 0
            if neg.any():
                logger.warning(f"Found {neg.sum()} candles with negative volume")
                return False
            for col in ("open", "high", "low", "close"):
                if df[col].isnull().any():
                    logger.warning(f"Found null values in {col}")
                    return False
            logger.debug("Candle data validation passed")
            return True
        except Exception as e:
            logger.error(f"Validation failed: {e}")
            return False

    def get_reconstruction_stats(self, original_data: pd.DataFrame, reconstructed_data: pd.DataFrame) -> dict:
        """The nine statistics of core.py:146-171 ({} when either frame is empty or anything fails)."""
        if original_data.empty or reconstructed_data.empty:
            return {}
        try:
            span_o = original_data["timestamp"].max() - original_data["timestamp"].min()
            span_r = reconstructed_data["timestamp"].max() - reconstructed_data["timestamp"].min()
            vol_o, vol_r = original_data["volume"].sum(), reconstructed_data["volume"].sum()
            n_o, n_r = len(original_data), len(reconstructed_data)
            return {
                "original_candles": n_o,
                "reconstructed_candles": n_r,
                "compression_ratio": n_o / n_r if n_r > 0 else 0,
                "original_timespan": span_o,
                "reconstructed_timespan": span_r,
                "coverage_ratio": span_r / span_o if span_o.total_seconds() > 0 else 0,
                "total_volume_original": vol_o,
                "total_volume_reconstructed": vol_r,
                "volume_preservation": abs(1 - vol_r / vol_o) if vol_o > 0 else 1,
            }
        except Exception as e:
            logger.error(f"Failed to calculate stats: {e}")
            return {}


class MultiSymbolCandleReconstructor:
    """core.py:174-300 over the frame store (``db_manager`` = iv_interpolation_amd.frame_store.FrameStore) instead of
    PostgreSQL: same constructor arguments, same per-symbol result dictionaries.  ``process_symbols`` is the batch
    extension (all symbols aggregated in ONE device launch)."""

    MINUTE_TABLE = "minute_candles"
    OUTPUT_TABLE = "reconstructed_candles"

    def __init__(self, db_manager, config, backend=None):
        self.db_manager = db_manager
        self.config = config
        self.reconstructor = CandleReconstructor(config.candle_reconstruction.target_frequency, backend=backend)

    def get_symbols_with_minute_data(self, start_date: str = None, end_date: str = None) -> List[str]:
        try:
            symbols = list(self.db_manager.symbols(self.MINUTE_TABLE))
            if start_date and end_date:
                lo, hi = pd.Timestamp(start_date), pd.Timestamp(end_date)
                kept = []
                for sym in symbols:
                    df = self.db_manager.read_table(self.MINUTE_TABLE, sym)
                    if df is not None and ((df["timestamp"] >= lo) & (df["timestamp"] <= hi)).any():
                        kept.append(sym)
                symbols = kept
            logger.info(f"Found {len(symbols)} symbols with minute candle data")
            return sorted(symbols)
        except Exception as e:
            logger.error(f"Failed to retrieve symbols: {e}")
            return []

    def _finish(self, symbol: str, minute_data, reconstructed, t0: float) -> dict:
        if reconstructed is None:
            return {"symbol": symbol, "status": "skipped", "reason": "Reconstruction failed"}
        if not self.reconstructor.validate_candle_data(reconstructed):
            return {"symbol": symbol, "status": "error", "error": "Invalid reconstructed candle data"}
        if not self._save_reconstructed_candles(reconstructed):
            return {"symbol": symbol, "status": "error", "error": "Failed to save reconstructed candles"}
        return {"symbol": symbol, "status": "success", "input_candles": len(minute_data), "output_candles": len(reconstructed),
                "processing_time": time.time() - t0,
                "stats": self.reconstructor.get_reconstruction_stats(minute_data, reconstructed)}

    def _load(self, symbol: str):
        df = self.db_manager.read_table(self.MINUTE_TABLE, symbol)
        if df is None or df.empty:
            return None, {"symbol": symbol, "status": "skipped", "reason": "No minute data found"}
        df = df[[c for c in ("symbol", "timestamp", "open", "high", "low", "close", "volume") if c in df.columns]]
        df = df.sort_values("timestamp")
        if not self.reconstructor.validate_candle_data(df):
            return None, {"symbol": symbol, "status": "error", "error": "Invalid input candle data"}
        return df, None

    def process_symbol(self, symbol: str) -> dict:
        t0 = time.time()
        try:
            df, verdict = self._load(symbol)
            if verdict:
                return verdict
            return self._finish(symbol, df, self.reconstructor.reconstruct_symbol_candles(df), t0)
        except Exception as e:
            logger.error(f"Error processing {symbol}: {e}")
            return {"symbol": symbol, "status": "error", "error": str(e), "processing_time": time.time() - t0}

    def process_symbols(self, symbols: Sequence[str]) -> List[dict]:
        """Batch extension: what [process_symbol(s) for s in symbols] returns, with one device launch for all of them."""
        t0 = time.time()
        out: List[Optional[dict]] = [None] * len(symbols)
        frames, slots = [], []
        for i, sym in enumerate(symbols):
            try:
                df, verdict = self._load(sym)
                if verdict:
                    out[i] = verdict
                else:
                    frames.append(df); slots.append(i)
            except Exception as e:
                out[i] = {"symbol": sym, "status": "error", "error": str(e), "processing_time": time.time() - t0}
        for i, df, rec in zip(slots, frames, self.reconstructor.reconstruct_batch(frames) if frames else []):
            try:
                out[i] = self._finish(symbols[i], df, rec, t0)
            except Exception as e:
                out[i] = {"symbol": symbols[i], "status": "error", "error": str(e), "processing_time": time.time() - t0}
        return out

    def _save_reconstructed_candles(self, candle_data: pd.DataFrame) -> bool:
        """core.py:272-300: one row per (symbol, timestamp, frequency), later writes replace earlier ones."""
        try:
            if candle_data.empty:
                return True
            for sym, part in candle_data.groupby("symbol", sort=False):
                old = self.db_manager.read_table(self.OUTPUT_TABLE, str(sym))
                if old is not None and len(old):
                    part = pd.concat([old, part], ignore_index=True)
                    part["timestamp"] = pd.to_datetime(part["timestamp"])
                    part = part.drop_duplicates(subset=["symbol", "timestamp", "frequency"], keep="last")
                self.db_manager.write_table(self.OUTPUT_TABLE, str(sym), part.sort_values("timestamp"))
            return True
        except Exception as e:
            logger.error(f"Failed to save reconstructed candles: {e}")
            return False
