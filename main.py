#!/usr/bin/env python3
"""Entry surface kept from the reference's ``main.py`` (reference :459-602): ``--task`` /
``--test`` / ``--env`` / ``--validate-only`` flags and the reference's task dispatch (interpolation, bridge, candles,
both, pipeline) over ``complete_pipeline.CompleteOptimizedPipeline`` on the MI355X engine and the DB-less frame store."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(ROOT, "src"))          # reference main.py:21


def main(argv=None, backend=None, bridge_backend=None, candle_backend=None, seed=None):
    parser = argparse.ArgumentParser(description="Dual-Task Pipeline: IV Interpolation (MI355X engine)")
    parser.add_argument("--task", choices=["interpolation", "bridge", "candles", "both", "pipeline"], default="interpolation")
    parser.add_argument("--test", action="store_true", help="Run in test mode (3 symbols)")
    parser.add_argument("--auto-bridge", action="store_true")
    parser.add_argument("--bridge-batch-id", type=int)
    parser.add_argument("--generate-sample-candles", action="store_true")
    parser.add_argument("--validate-only", action="store_true", help="Only validate the setup (engine + frame store)")
    parser.add_argument("--env", choices=["development", "testing", "production"], default="production")
    parser.add_argument("--data-dir")
    args = parser.parse_args(argv)
    os.environ["ENVIRONMENT"] = args.env
    from config import get_config
    from complete_pipeline import CompleteOptimizedPipeline
    config = get_config()
    print(f"ENVIRONMENT: {config.environment}")
    if args.validate_only:
        from iv_interpolation_amd import _lib
        lib = _lib.load()
        print(f"libivs ABI {lib.ivs_version()}, HIP devices visible: {lib.ivs_device_count()}")
        return 0 if lib.ivs_device_count() > 0 else 1
    if args.generate_sample_candles:
        print("--generate-sample-candles writes to PostgreSQL in the reference (main.py:377-456); not available with the frame store.")
        return 1
    pipe = CompleteOptimizedPipeline(config, data_dir=args.data_dir, backend=backend, bridge_backend=bridge_backend,
                                     candle_backend=candle_backend, seed=seed)
    symbols = pipe.store.symbols()[:3] if args.test else None

    def task1():
        return pipe.run_task1_interpolation(symbols)["success"]

    def bridge():
        return pipe.run_data_bridge(symbols, args.bridge_batch_id)["success"]

    def task2():
        return pipe.run_task2_candle_reconstruction(symbols)["success"]

    if args.task == "interpolation":
        ok = task1()
    elif args.task == "bridge":
        ok = bridge()
    elif args.task == "candles":
        ok = task2()
    elif args.task == "both" and not args.auto_bridge:          # reference :551-559: Task 1, then Task 2 on whatever candles exist
        ok = task1() and task2()
    else:                                                       # 'pipeline', or 'both' with --auto-bridge (:540-550, :561-575)
        ok = task1() and bridge() and task2()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
