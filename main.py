#!/usr/bin/env python3
"""Entry surface kept from the reference's ``main.py`` (reference :459-602): ``--task`` /
``--test`` / ``--env`` / ``--validate-only`` flags.  Task ``interpolation`` (the default) runs on the MI355X
engine through ``complete_pipeline.CompleteOptimizedPipeline``; the other tasks belong to the reference's
unchanged downstream stages and are reported as out of scope."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.append(os.path.join(ROOT, "src"))          # reference main.py:21


def main(argv=None, backend=None):
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
    if args.task != "interpolation" or args.generate_sample_candles:
        print("Only --task interpolation is implemented here; bridge/candles are the reference's unchanged stages.")
        return 1
    pipe = CompleteOptimizedPipeline(config, data_dir=args.data_dir, backend=backend)
    symbols = pipe.store.symbols()[:3] if args.test else None
    return 0 if pipe.run_task1_interpolation(symbols)["success"] else 1


if __name__ == "__main__":
    sys.exit(main())
