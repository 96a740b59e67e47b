#!/usr/bin/env python3
"""cProfile of IVInterpolator.interpolate_frame on 2048 symbols (where does the end-to-end time go?)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pandas as pd
from iv_interpolation_amd import IVInterpolator
from iv_interpolation_amd.frame_store import synthetic_symbol
n = 64
big = [synthetic_symbol(f"s{i:05d}", n, seed=i) for i in range(2048)]
long = pd.concat(big, ignore_index=True)
iv = IVInterpolator("linear")
iv.interpolate_frame(long.iloc[: 64 * n])
t0 = time.perf_counter(); iv.interpolate_frame(long); print("wall", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); iv.interpolate_frame(long); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
