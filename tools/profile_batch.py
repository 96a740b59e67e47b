"""cProfile of IVInterpolator.interpolate_batch on the GPU box: where the host time of the columnar batch path goes.
    python tools/profile_batch.py [symbols] [method]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iv_interpolation_amd import IVInterpolator
from iv_interpolation_amd.frame_store import synthetic_symbol
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
method = sys.argv[2] if len(sys.argv) > 2 else "linear"
frames = [synthetic_symbol(f"s{i}", 64, seed=i) for i in range(N)]
iv = IVInterpolator(method)
iv.interpolate_batch(frames[:8])
for _ in range(2):
    t0 = time.perf_counter(); out = iv.interpolate_batch(frames); dt = time.perf_counter() - t0
    print("%d symbols: %.3f s -> %.0f symbols/s" % (N, dt, N / dt)); del out
import gc
gc.disable()
for _ in range(2):
    t0 = time.perf_counter(); out = iv.interpolate_batch(frames); dt = time.perf_counter() - t0
    print("gc disabled, %d symbols: %.3f s -> %.0f symbols/s" % (N, dt, N / dt)); del out
gc.enable()
pr = cProfile.Profile(); pr.enable(); out = iv.interpolate_batch(frames); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
