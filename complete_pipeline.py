#!/usr/bin/env python3
"""Entry surface kept from the reference's ``complete_pipeline.py`` (reference :838-921): same class name,
same ``--task/--test/--symbols/--estimate/--setup-only`` flags, same result dictionaries.  Task 1 (IV
interpolation) runs on the MI355X engine -- many symbols per device launch through
``IVInterpolator.interpolate_batch`` instead of the reference's one-symbol-at-a-time loop
(:243-262) -- against a DB-less frame store (``--data-dir``).  The data bridge (reference :367-510, the pipeline's
own inline candle builder) and the 5-minute candle stage (:567-710) run on the engine too (SURVEY.md section 8f
ranks 3 and 4): all symbols of a stage in one device launch, tables ``minute_candles`` and ``reconstructed_candles``
in the same store."""
import argparse
import os
import sys
import time
from typing import List, Optional

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(ROOT, "src"))          # reference complete_pipeline.py:30

from config import get_config                          # noqa: E402
from interpolation.core import IVInterpolator          # noqa: E402  (reference :34)

import numpy as np                                      # noqa: E402
import pandas as pd                                     # noqa: E402

from iv_interpolation_amd.frame_store import FrameStore, synthetic_symbol   # noqa: E402

MINUTE_TABLE = "minute_candles"
CANDLE_TABLE = "reconstructed_candles"
PIPELINE_INLINE = 4                                     # engine strategy code of reference complete_pipeline.py:473-510


class CompleteOptimizedPipeline:
    def __init__(self, config, data_dir: Optional[str] = None, backend=None, bridge_backend=None, candle_backend=None,
                 seed: Optional[int] = None):
        self.config = config
        self.store = FrameStore(data_dir or config.data_dir)
        self._bridge_backend = bridge_backend            # None -> HipBridgeBackend / HipCandleBackend on first use
        self._candle_backend = candle_backend
        # the reference draws from the unseeded global NumPy generator; an explicit seed makes a run reproducible
        self.seed = int.from_bytes(os.urandom(4), "little") if seed is None else int(seed)
        self._rng_pos, self._rng_tail = 0, (0, 0)
        # the reference builds IVInterpolator() with defaults here (:49); config values are honoured instead
        # ... including `preserve_greeks` (reference config.py:46, read by nothing there): the interpolated rows then carry
        # the delta..rho columns the reference's schema reserves for them (src/database/schema.py:36-40)
        self.iv_interpolator = IVInterpolator(method=config.interpolation.method,
                                              min_points=config.interpolation.min_data_points, backend=backend,
                                              preserve_greeks=bool(getattr(config.interpolation, "preserve_greeks", False)))
        self.interrupted = False

    def setup_database_tables(self) -> dict:
        return {"success": True, "tables": ["trading_tickers", "interpolated_trading_tickers"], "root": self.store.root}

    def get_pipeline_status(self) -> dict:
        src = self.store.symbols()
        done = self.store.symbols("interpolated_trading_tickers")
        return {"source_symbols": len(src), "task1_symbols": len(done),
                "bridge_symbols": len(self.store.symbols(MINUTE_TABLE)), "task2_symbols": len(self.store.symbols(CANDLE_TABLE)),
                "source_rows": sum(len(self.store.read_symbol(s)) for s in src)}

    def run_task1_interpolation(self, symbols: List[str] = None, batch_id: int = None) -> dict:
        print("TASK 1: IV INTERPOLATION (MI355X engine)")
        print("-" * 40)
        if symbols is None:
            symbols = self.store.pending_symbols()
        if not symbols:
            return {"success": False, "error": "No symbols found for Task 1"}
        if batch_id is None:
            batch_id = int(time.time())
        start = time.time()
        ok = err = total_in = total_out = 0
        step = max(1, int(self.config.processing.symbols_per_batch))
        for a in range(0, len(symbols), step):
            if self.interrupted:
                break
            chunk = symbols[a:a + step]
            frames = [self.store.read_symbol(s) for s in chunk]
            results = self.iv_interpolator.interpolate_batch(frames)          # one device round trip per chunk
            for sym, src, out in zip(chunk, frames, results):
                if out is None:                                               # reference: status 'skipped'
                    err += 1
                    print(f"  {sym}: skipped (no valid data after interpolation)")
                    continue
                out = out.copy()
                out["is_interpolated"] = out["symbol"].isna()                 # recomputed as complete_pipeline.py:324
                n = self.store.write_output(sym, out, batch_id)
                ok += 1; total_in += len(src); total_out += n
                print(f"  {sym}: {len(src)} -> {n} rows")
        duration = time.time() - start
        print(f"\nTASK 1 COMPLETE: {duration:.1f}s, success {ok}, errors {err}, rows {total_in:,} -> {total_out:,}")
        return {"success": ok > 0, "batch_id": batch_id, "symbols_processed": ok, "total_input": total_in,
                "total_output": total_out, "duration": duration}

    def run_data_bridge(self, symbols: List[str] = None, batch_id: int = None) -> dict:
        """Reference :367-510.  One candle per interpolated row; base price = ``underlying or mark or index`` per row
        (:479, first TRUTHY value); all symbols in one launch, one generator stream in symbol order."""
        print("\nDATA BRIDGE: IV -> OHLCV CONVERSION (MI355X engine)")
        print("-" * 40)
        if symbols is None:                                                   # anti-join of :419-435
            done = set(self.store.symbols(MINUTE_TABLE))
            symbols = [s for s in self.store.symbols("interpolated_trading_tickers") if s not in done]
        if not symbols:
            return {"success": False, "error": "No symbols found for data bridge"}
        start = time.time()
        ok = err = total_in = total_out = 0
        frames, live = [], []
        for sym in symbols:
            df = self.store.read_output(sym)
            if df is None or df.empty:
                err += 1
                print(f"  {sym}: skipped (No interpolated data)")
                continue
            frames.append(df.sort_values("date").reset_index(drop=True)); live.append(sym)
        if frames:
            def col(df, name):
                return pd.to_numeric(df[name], errors="coerce").to_numpy(np.float64) if name in df.columns else np.full(len(df), np.nan)
            base, vol = [], []
            for df in frames:
                u, m, i = col(df, "underlying_price"), col(df, "mark_price"), col(df, "index_price")
                base.append(np.where(u != 0, u, np.where(m != 0, m, i)))
                vol.append(col(df, "volume"))
            off = np.concatenate([[0], np.cumsum([len(f) for f in frames])]).astype(np.int64)
            if self._bridge_backend is None:
                from iv_interpolation_amd.bridge import HipBridgeBackend
                self._bridge_backend = HipBridgeBackend()
            out, valid, self._rng_pos, self._rng_tail = self._bridge_backend.bridge_candles(
                np.concatenate(base), np.concatenate(vol), off, PIPELINE_INLINE, self.seed, self._rng_pos, self._rng_tail,
                0.002, 1.5)
            for k, (sym, df) in enumerate(zip(live, frames)):
                a, b = int(off[k]), int(off[k + 1])
                keep = np.flatnonzero(valid[a:b])
                if len(keep) == 0:
                    err += 1
                    print(f"  {sym}: OHLCV generation failed")
                    continue
                o = out[:, a:b][:, keep]
                cand = pd.DataFrame({"symbol": df["symbol"].to_numpy()[keep], "timestamp": df["date"].to_numpy()[keep],
                                     "open": o[0], "high": o[1], "low": o[2], "close": o[3], "volume": o[4],
                                     "source_price": o[5], "is_synthetic": True})
                self.store.write_table(MINUTE_TABLE, sym, cand)
                ok += 1; total_in += len(df); total_out += len(cand)
                print(f"  {sym}: {len(df)} -> {len(cand)} candles")
        duration = time.time() - start
        print(f"\nDATA BRIDGE COMPLETE: {duration:.1f}s, success {ok}, errors {err}, OHLCV candles {total_out:,}")
        return {"success": ok > 0, "symbols_processed": ok, "total_input": total_in, "total_output": total_out,
                "duration": duration}

    def run_task2_candle_reconstruction(self, symbols: List[str] = None, batch_id: int = None) -> dict:
        """Reference :567-710: 1-minute -> 5-minute candles (first / max / min / last / sum, groups of fewer than five
        rows dropped), all symbols in one launch."""
        from iv_interpolation_amd.candles import CandleReconstructor
        print("\nTASK 2: CANDLE RECONSTRUCTION (MI355X engine)")
        print("-" * 40)
        if symbols is None:                                                   # anti-join of :623-639
            done = set(self.store.symbols(CANDLE_TABLE))
            symbols = [s for s in self.store.symbols(MINUTE_TABLE) if s not in done]
        if not symbols:
            return {"success": False, "error": "No symbols found for Task 2"}
        if batch_id is None:
            batch_id = int(time.time())
        start = time.time()
        ok = err = total_in = total_out = 0
        frames, live = [], []
        for sym in symbols:
            df = self.store.read_table(MINUTE_TABLE, sym)
            if df is None or df.empty:
                err += 1; print(f"  {sym}: skipped (No minute candles found)"); continue
            if len(df) < 5:
                err += 1; print(f"  {sym}: skipped (Insufficient data for 5-min candles)"); continue
            frames.append(df[["symbol", "timestamp", "open", "high", "low", "close", "volume"]]); live.append(sym)
        if frames:
            rec = CandleReconstructor("5min", backend=self._candle_backend)
            for sym, src, res in zip(live, frames, rec.reconstruct_batch(frames)):
                if res is None or res.empty:
                    err += 1; print(f"  {sym}: Reconstruction failed"); continue
                res = res[["timestamp", "open", "high", "low", "close", "volume", "symbol"]].copy()     # :689-704 column order
                res["frequency"] = "5min"; res["source_candles"] = 5; res["batch_id"] = batch_id          # :663-666
                self.store.write_table(CANDLE_TABLE, sym, res)
                ok += 1; total_in += len(src); total_out += len(res)
                print(f"  {sym}: {len(src)} -> {len(res)} candles ({len(src) / len(res):.1f}:1)")
        duration = time.time() - start
        print(f"\nTASK 2 COMPLETE: {duration:.1f}s, success {ok}, errors {err}, 5-min candles {total_out:,}")
        return {"success": ok > 0, "batch_id": batch_id, "symbols_processed": ok, "total_input": total_in,
                "total_output": total_out, "duration": duration}

    def run_complete_pipeline(self, test_mode: bool = False, symbol_limit: int = None) -> dict:
        """Reference :740-831: Task 1 -> bridge -> Task 2 over the same symbol list, stopping at the first failed stage."""
        symbols = self.store.symbols()
        if test_mode:
            symbols = symbols[:3]
        elif symbol_limit:
            symbols = symbols[:symbol_limit]
        if not symbols:
            return {"success": False, "error": "No symbols found"}
        batch_id = int(time.time())
        t0 = time.time()
        results = {}
        try:
            results["task1"] = self.run_task1_interpolation(symbols, batch_id)
            if not results["task1"]["success"]:
                return {"success": False, "error": "Task 1 failed", "results": results}
            results["bridge"] = self.run_data_bridge(symbols, batch_id)
            if not results["bridge"]["success"]:
                return {"success": False, "error": "Data bridge failed", "results": results}
            results["task2"] = self.run_task2_candle_reconstruction(symbols, batch_id)
            if not results["task2"]["success"]:
                return {"success": False, "error": "Task 2 failed", "results": results}
            return {"success": True, "batch_id": batch_id, "duration": time.time() - t0, "results": results,
                    "final_status": self.get_pipeline_status()}
        except Exception as e:
            return {"success": False, "error": str(e), "results": results}

    def cleanup(self):
        pass


def main(argv=None, backend=None, bridge_backend=None, candle_backend=None, seed=None):
    parser = argparse.ArgumentParser(description="Complete Optimized Pipeline (MI355X engine)")
    parser.add_argument("--task", choices=["interpolation", "bridge", "candles", "all"], default="all")
    parser.add_argument("--test", action="store_true", help="Test mode with 3 symbols")
    parser.add_argument("--symbols", type=int, help="Limit number of symbols")
    parser.add_argument("--estimate", action="store_true", help="Show estimates only")
    parser.add_argument("--setup-only", action="store_true", help="Setup tables only")
    parser.add_argument("--data-dir", help="frame store directory (default: config.data_dir / $IVS_DATA_DIR)")
    parser.add_argument("--synthetic", type=int, metavar="N", help="first fill the store with N synthetic hourly symbols")
    args = parser.parse_args(argv)
    try:
        config = get_config()
        pipeline = CompleteOptimizedPipeline(config, data_dir=args.data_dir, backend=backend, bridge_backend=bridge_backend,
                                             candle_backend=candle_backend, seed=seed)
        if args.synthetic:
            for i in range(args.synthetic):
                sym = f"btc-20mar23-{20000 + 500 * i}-c"
                pipeline.store.write_source(sym, synthetic_symbol(sym, seed=i))
        if args.setup_only:
            pipeline.setup_database_tables()
            print("Setup complete")
            return 0
        if args.estimate:
            st = pipeline.get_pipeline_status()
            print(f"Source data: {st['source_rows']:,} rows, {st['source_symbols']:,} symbols")
            print(f"Task 1 progress: {st['task1_symbols']}/{st['source_symbols']} symbols")
            return 0
        if args.task == "all":
            result = pipeline.run_complete_pipeline(test_mode=args.test, symbol_limit=args.symbols)
        else:
            symbols = None
            if args.symbols or args.test:
                symbols = pipeline.store.symbols()[:3 if args.test else args.symbols]
            if args.task == "interpolation":
                result = pipeline.run_task1_interpolation(symbols)
            elif args.task == "bridge":
                result = pipeline.run_data_bridge(symbols)
            else:
                result = pipeline.run_task2_candle_reconstruction(symbols)
        return 0 if result["success"] else 1
    except KeyboardInterrupt:
        print("\nPipeline interrupted by user")
        return 130


if __name__ == "__main__":
    sys.exit(main())
