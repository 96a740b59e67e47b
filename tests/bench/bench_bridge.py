#!/usr/bin/env python3
"""Device-side rate of the IV -> OHLCV bridge: S symbols x m one-minute rows, stream generation and candle kernels
timed separately with HIP events; the CPU oracle (the reference's per-row Python loop restated) on a small sample.
    python tests/bench/bench_bridge.py [--symbols 4096] [--rows 3781] [--strategy spread_simulation]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
from iv_interpolation_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--symbols", type=int, default=4096); ap.add_argument("--rows", type=int, default=3781)
ap.add_argument("--strategy", default="spread_simulation", choices=list(engine.BRIDGE_STRATEGIES))
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
S, m = a.symbols, a.rows
code = engine.BRIDGE_STRATEGIES[a.strategy]
r = np.random.default_rng(1)
n = S * m
price = torch.from_numpy(25000 * np.exp(np.cumsum(r.normal(0, 2e-4, n)))).cuda()
volume = torch.from_numpy(r.uniform(0.0, 40.0, n)).cuda()
volume[torch.rand(n, device="cuda") < 0.2] = float("nan")
off = torch.arange(0, n + 1, m, dtype=torch.int64, device="cuda")
nw = engine.bridge_words_bound(n, code)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2], out


t_words, words = timed(lambda: engine.mt19937_words(7, nw))
t_candles, res = timed(lambda: engine.bridge_candles(price, volume, off, code, words))
assert int(res[2].cpu()[3]) == 0
import bridge_oracle as BO
ns = min(n, 20000)
ph, vh = price[:ns].cpu().numpy(), volume[:ns].cpu().numpy()
t0 = time.perf_counter(); BO.candles(ph, vh, code, seed=7); t_cpu = time.perf_counter() - t0
print(json.dumps({"strategy": a.strategy, "symbols": S, "rows_per_symbol": m, "rows": n, "stream_words": nw,
                  "mt19937_ms": t_words, "mt19937_Mwords_per_s": nw / t_words / 1e3,
                  "candles_ms": t_candles, "candles_Mrows_per_s": n / t_candles / 1e3,
                  "total_Mrows_per_s": n / (t_words + t_candles) / 1e3,
                  "cpu_oracle_rows_per_s": ns / t_cpu, "cpu_sample_rows": ns}))
