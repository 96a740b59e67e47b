#!/usr/bin/env python3
"""Throughput of the variable-shape kernels vs the distribution of strike counts (ragged CSR batches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import engine, synth
B = 500_000
Kq, Tq = synth.query_grids(64, 16)
Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
for method in ("cubic", "linear"):
    for lo, hi in ((32, 32), (48, 48), (64, 64), (8, 64), (33, 64), (8, 32), (96, 96), (128, 128), (65, 128), (8, 128)):
        d = synth.torch_ragged_batch(B, 16, lo, hi)
        out = torch.empty((B, 16, 64), dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda")
        run = lambda: engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, method, k_off=d["k_off"], nK_max=d["nK_max"],
                                           n_maturities=16, out=out, status=st)
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        nbytes = 8 * (17 * d["K"].numel() + 16) + 8 * B * 1024
        print(f"{method:7s} n in [{lo:3d},{hi:3d}]: {B / ms / 1e3:7.1f} M surfaces/s  {nbytes / ms / 1e6:7.0f} GB/s  [{engine.last_kernel()}]", flush=True)
        del d, out, st
