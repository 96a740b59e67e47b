#!/usr/bin/env python3
"""Dense 64x16 kernel: throughput by input layout (shared vs per-surface maturities / query grids / strikes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
from iv_interpolation_amd import engine, synth
ap = argparse.ArgumentParser(); ap.add_argument("--methods", default="cubic,pchip,linear"); ap.add_argument("--one-pass", action="store_true")
ap.add_argument("--only", default="", help="substring filter on the case names")
a = ap.parse_args()
B = 1_000_000
d = synth.torch_batch(B, 64, 16)
Kq, Tq = synth.query_grids(64, 16); Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
out = torch.empty((B, 16, 64), dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda")
Tb = d["T"][None, :].repeat(B, 1).contiguous(); Tqb = Tq[None, :].repeat(B, 1).contiguous(); Kqb = Kq[None, :].repeat(B, 1).contiguous()
cases = {"shared T, Tq, Kq (benchmark layout)": (d["K"], d["T"], Kq, Tq),
         "per-surface T and Tq": (d["K"], Tb, Kq, Tqb),
         "per-surface Kq": (d["K"], d["T"], Kqb, Tq),
         "per-surface T, Tq and Kq": (d["K"], Tb, Kqb, Tqb),
         "shared strikes (one K row)": (d["K"][0].contiguous(), d["T"], Kq, Tq),
         "strike rows in runs of 8 (snapshots of one chain)": (d["K"][(torch.arange(B, device="cuda") // 8) * 8].contiguous(), d["T"], Kq, Tq)}
for method in a.methods.split(","):
    for name, (K, T, kq, tq) in cases.items():
        if a.only and a.only not in name:
            continue
        run = lambda: engine.surface_batch(K, T, d["sigma"], kq, tq, method, out=out, status=st, one_pass=a.one_pass)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{method:7s} {name:40s} {B / ms / 1e3:7.1f} M surfaces/s  [{engine.last_kernel()}]", flush=True)
