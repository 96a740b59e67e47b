"""cProfile of IVInterpolator.interpolate_symbol (one symbol per call, the reference's own call pattern) on the GPU box."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iv_interpolation_amd import IVInterpolator
from iv_interpolation_amd.frame_store import synthetic_symbol
method = sys.argv[1] if len(sys.argv) > 1 else "linear"
frames = [synthetic_symbol(f"s{i}", 64, seed=i) for i in range(200)]
iv = IVInterpolator(method)
for f in frames[:5]: iv.interpolate_symbol(f)
t0 = time.perf_counter(); out = [iv.interpolate_symbol(f) for f in frames]; dt = time.perf_counter() - t0
print("%d calls: %.3f s -> %.0f symbols/s (%.0f us per call)" % (len(frames), dt, len(frames) / dt, dt / len(frames) * 1e6))
pr = cProfile.Profile(); pr.enable(); out = [iv.interpolate_symbol(f) for f in frames]; pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
