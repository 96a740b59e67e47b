"""cProfile of IVInterpolator.interpolate_frame on the GPU box: where the host time of the long-frame path goes.
    python tools/profile_frame_host.py [symbols] [method]"""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd
from iv_interpolation_amd import IVInterpolator
from iv_interpolation_amd.frame_store import synthetic_symbol
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
method = sys.argv[2] if len(sys.argv) > 2 else "linear"
long = pd.concat([synthetic_symbol(f"s{i:05d}", 64, seed=i) for i in range(N)], ignore_index=True)
iv = IVInterpolator(method)
iv.interpolate_frame(long.iloc[:64 * 64])
for _ in range(3):
    t0 = time.perf_counter(); out = iv.interpolate_frame(long); dt = time.perf_counter() - t0
    print("%d symbols: %.3f s -> %.0f symbols/s, %d rows" % (N, dt, N / dt, len(out))); del out
gc.disable()
for _ in range(2):
    t0 = time.perf_counter(); out = iv.interpolate_frame(long); dt = time.perf_counter() - t0
    print("gc disabled: %.3f s -> %.0f symbols/s" % (dt, N / dt)); del out
gc.enable()
pr = cProfile.Profile(); pr.enable(); out = iv.interpolate_frame(long); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
