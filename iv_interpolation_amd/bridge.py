"""Drop-in for the reference's ``InterpolatedToOHLCVConverter`` (src/data_bridge/ohlcv_converter.py:16-30, 138-418) on
the MI355X engine: same constructor, same ``_generate_ohlcv_from_interpolated(df) -> Optional[DataFrame]`` contract,
output columns, price-column rule, quality verdicts and clean-up.

The reference draws its candles from NumPy's process-global, unseeded legacy generator.  Here the generator is an
explicit MT19937 stream on the device: with ``seed=s`` the converter produces, call after call, exactly what the
reference produces after ``np.random.seed(s)`` (bit for bit, except that a volume drawn from the exponential
distribution goes through the device's ``log``); with ``seed=None`` a seed is taken from the OS, which is the
reference's unseeded behaviour.  ``convert_frames`` converts many symbols in one launch (one stream, symbols in order).
"""
import logging
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import pandas as pd

logger = logging.getLogger("data_bridge.ohlcv_converter")

STRATEGIES = {"spread_simulation": 0, "price_as_midpoint": 1, "trend_following": 2}     # anything else -> simple_spread (3)
STRATEGY_LABEL = {0: "spread_simulation", 1: "price_midpoint", 2: "trend_following", 3: "simple_spread"}
OUT_COLUMNS = ["symbol", "timestamp", "open", "high", "low", "close", "volume", "source_price", "conversion_method",
               "is_synthetic"]
PRICE_PRIORITY = ["underlying_price", "mark_price", "index_price"]


class HipBridgeBackend:
    """Owns the device-side generator: the words of the stream are produced once and extended by doubling."""

    def __init__(self):
        self._seed = None
        self._words = None

    def _ensure(self, seed: int, n_words: int):
        from . import engine
        if self._seed != seed or self._words is None or self._words.numel() < n_words:
            have = 0 if (self._seed != seed or self._words is None) else self._words.numel()
            self._words = engine.mt19937_words(seed, max(n_words, 2 * have))
            self._seed = seed
        return self._words

    def bridge_candles(self, price, volume, row_off, strategy, seed, pos, tail, base_spread_pct, vol_factor):
        from . import engine
        torch = engine.require_device()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
        price_d = d(np.asarray(price, np.float64))
        vol_d = None if volume is None else d(np.asarray(volume, np.float64))
        off_d = d(np.asarray(row_off, np.int64))
        need = engine.bridge_words_bound(len(price), strategy)
        while True:
            words = self._ensure(seed, pos + need)
            t = torch.tensor([0, int(tail[0]), int(tail[1]), 0], dtype=torch.int64).cuda()
            out, valid, t = engine.bridge_candles(price_d, vol_d, off_d, strategy, words[pos:], t, base_spread_pct, vol_factor)
            th = t.cpu().numpy()
            if th[3] == 0:
                break
            need *= 2                                   # rejection sampling needed more words than estimated
        return out.cpu().numpy(), valid.cpu().numpy().astype(bool), pos + int(th[0]), (int(th[1]), int(th[2]))


class InterpolatedToOHLCVConverter:
    def __init__(self, db_manager, config, backend=None, seed: Optional[int] = None):
        self.db_manager = db_manager
        self.config = config
        self.conversion_strategy = config.data_bridge.conversion_strategy
        self.spread_method = config.data_bridge.spread_method
        self.quality_checks = config.data_bridge.enable_quality_checks
        self._backend = backend
        self.seed = int.from_bytes(os.urandom(4), "little") if seed is None else int(seed)
        self._pos = 0                                   # words of the stream consumed so far
        self._tail = (0, 0)                             # the legacy generator's cached normal deviate (has_gauss, bits)

    # ------------------------------------------------------------------ conversion
    def _generate_ohlcv_from_interpolated(self, df: pd.DataFrame) -> Optional[pd.DataFrame]:
        return self.convert_frames([df])[0]

    def convert_frames(self, frames: Sequence[pd.DataFrame]) -> List[Optional[pd.DataFrame]]:
        """Element i is what the reference's ``_generate_ohlcv_from_interpolated(frames[i])`` returns when the frames
        are converted one after the other (ohlcv_converter.py:138-187)."""
        res: List[Optional[pd.DataFrame]] = [None] * len(frames)
        live, prices, volumes = [], [], []
        for i, df in enumerate(frames):
            if df.empty:
                continue
            try:
                col = self._select_price_column(df)
                pd.to_datetime(df["timestamp"])          # the reference converts row by row and fails the frame on error
                p = pd.to_numeric(df[col], errors="raise").to_numpy(np.float64, na_value=np.nan)
                if "volume" in df.columns:
                    v = pd.to_numeric(df["volume"], errors="raise").to_numpy(np.float64, na_value=np.nan)
                else:
                    v = np.zeros(len(df))                # row.get('volume', 0) -> 0 -> drawn (ohlcv_converter.py:362-367)
                _ = df["symbol"]
            except Exception as e:
                logger.error(f"OHLCV generation failed: {e}")
                continue
            live.append(i); prices.append(p); volumes.append(v)
        if not live:
            return res
        off = np.concatenate([[0], np.cumsum([len(p) for p in prices])]).astype(np.int64)
        strategy = STRATEGIES.get(self.conversion_strategy, 3)
        sp = self.config.data_bridge.spread_parameters or {}
        be = self._backend or HipBridgeBackend()
        self._backend = be
        out, valid, self._pos, self._tail = be.bridge_candles(
            np.concatenate(prices), np.concatenate(volumes), off, strategy, self.seed, self._pos, self._tail,
            float(sp.get("base_spread_percent", 0.002)), float(sp.get("volatility_factor", 1.5)))
        for k, i in enumerate(live):
            a, b = int(off[k]), int(off[k + 1])
            keep = np.flatnonzero(valid[a:b])
            if len(keep) == 0:
                continue                                  # no candle -> None (ohlcv_converter.py:174-175)
            df = frames[i]
            o = out[:, a:b][:, keep]
            cand = pd.DataFrame({
                "symbol": df["symbol"].to_numpy()[keep],
                "timestamp": pd.to_datetime(df["timestamp"]).to_numpy()[keep],
                "open": o[0], "high": o[1], "low": o[2], "close": o[3], "volume": o[4], "source_price": o[5],
                "conversion_method": STRATEGY_LABEL[strategy], "is_synthetic": True})
            res[i] = self._cleanup_ohlcv_data(cand[OUT_COLUMNS])
            logger.debug(f"Generated {len(res[i])} OHLCV candles from {len(df)} interpolated points")
        return res

    def _select_price_column(self, df: pd.DataFrame) -> str:
        """ohlcv_converter.py:189-207: first column of the priority list with more than 80 % data, else the first present."""
        for col in PRICE_PRIORITY:
            if col in df.columns and df[col].notna().sum() > len(df) * 0.8:
                return col
        for col in PRICE_PRIORITY:
            if col in df.columns:
                logger.warning(f"Using {col} as fallback price source (limited data)")
                return col
        raise ValueError("No suitable price column found")

    def _validate_ohlcv_quality(self, ohlcv_df: pd.DataFrame) -> Dict:
        """ohlcv_converter.py:371-402, same order of checks and the same reasons."""
        try:
            o, h, lo, c = (ohlcv_df[k].to_numpy(np.float64) for k in ("open", "high", "low", "close"))
            if (h < lo).any():
                return {"valid": False, "reason": "High < Low found"}
            if ((h < o) | (h < c)).any():
                return {"valid": False, "reason": "High < Open/Close found"}
            if ((lo > o) | (lo > c)).any():
                return {"valid": False, "reason": "Low > Open/Close found"}
            if (((h - lo) / ohlcv_df["source_price"].to_numpy(np.float64)) > 0.1).any():
                return {"valid": False, "reason": "Unrealistic spreads detected"}
            if (np.stack([o, h, lo, c]) <= 0).any():
                return {"valid": False, "reason": "Negative or zero prices found"}
            return {"valid": True, "reason": "All quality checks passed"}
        except Exception as e:
            return {"valid": False, "reason": f"Validation error: {str(e)}"}

    def _cleanup_ohlcv_data(self, ohlcv_df: pd.DataFrame) -> pd.DataFrame:
        """ohlcv_converter.py:404-418."""
        ohlcv_df = ohlcv_df.dropna(subset=["open", "high", "low", "close"])
        for col in ["open", "high", "low", "close", "volume"]:
            ohlcv_df[col] = pd.to_numeric(ohlcv_df[col], errors="coerce")
        return ohlcv_df.sort_values("timestamp")

    # ------------------------------------------------------------------ store-facing wrappers
    def get_interpolated_symbols(self, batch_id: Optional[int] = None) -> List[str]:
        """ohlcv_converter.py:31-55 against the frame store: symbols with interpolated rows (of that batch), sorted."""
        try:
            symbols = list(self.db_manager.symbols("interpolated_trading_tickers"))
            if batch_id:
                kept = []
                for sym in symbols:
                    df = self.db_manager.read_output(sym)
                    if df is not None and "batch_id" in df.columns and (df["batch_id"] == batch_id).any():
                        kept.append(sym)
                symbols = kept
            logger.info(f"Found {len(symbols)} symbols with interpolated data")
            return sorted(symbols)
        except Exception as e:
            logger.error(f"Failed to retrieve interpolated symbols: {e}")
            return []

    def convert_batch(self, symbols: Optional[List[str]] = None, batch_id: Optional[int] = None) -> Dict:
        """ohlcv_converter.py:463-493: the same tally dictionary; symbols are converted in order on ONE generator stream."""
        if symbols is None:
            symbols = self.get_interpolated_symbols(batch_id)
        if not symbols:
            logger.warning("No symbols found for OHLCV conversion")
            return {"total": 0, "success": 0, "errors": 0, "skipped": 0}
        logger.info(f"Converting {len(symbols)} symbols to OHLCV format")
        results = {"total": len(symbols), "success": 0, "errors": 0, "skipped": 0}
        for symbol in symbols:
            r = self.convert_symbol_to_ohlcv(symbol, batch_id)
            if r["status"] == "success":
                results["success"] += 1
                logger.info(f"{symbol}: {r['input_points']} -> {r['output_candles']} candles")
            elif r["status"] == "skipped":
                results["skipped"] += 1
                logger.warning(f"{symbol}: {r['reason']}")
            else:
                results["errors"] += 1
                logger.error(f"{symbol}: {r['error']}")
        logger.info(f"OHLCV conversion complete: {results['success']} success, {results['errors']} errors, {results['skipped']} skipped")
        return results

    def convert_symbol_to_ohlcv(self, symbol: str, batch_id: Optional[int] = None) -> Dict:
        """ohlcv_converter.py:57-136 against a frame store (``db_manager`` = iv_interpolation_amd.frame_store.FrameStore)
        instead of PostgreSQL: same result dictionaries."""
        t0 = time.time()
        try:
            df = self.db_manager.read_output(symbol)
            if df is None or df.empty:
                return {"symbol": symbol, "status": "skipped", "reason": "No interpolated data found"}
            if batch_id and "batch_id" in df.columns:
                df = df[df["batch_id"] == batch_id]
            df = df.rename(columns={"date": "timestamp"}).sort_values("timestamp")
            ohlcv = self._generate_ohlcv_from_interpolated(df)
            if ohlcv is None or ohlcv.empty:
                return {"symbol": symbol, "status": "error", "error": "OHLCV conversion failed"}
            if self.quality_checks:
                q = self._validate_ohlcv_quality(ohlcv)
                if not q["valid"]:
                    return {"symbol": symbol, "status": "error", "error": f"Quality check failed: {q['reason']}"}
            self.db_manager.write_table("minute_candles", symbol, ohlcv)
            return {"symbol": symbol, "status": "success", "input_points": len(df), "output_candles": len(ohlcv),
                    "processing_time": time.time() - t0, "conversion_method": self.conversion_strategy}
        except Exception as e:
            logger.error(f"Error converting {symbol}: {e}")
            return {"symbol": symbol, "status": "error", "error": str(e), "processing_time": time.time() - t0}
