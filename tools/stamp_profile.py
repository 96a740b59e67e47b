#!/usr/bin/env python3
"""Phase shares of the dense kernel from the stamped diagnostic build (ivs_debug_stamps).
Shares only -- the stamps serialise phases, so this run's wall time is not a timing claim."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import _lib, engine, synth

ap = argparse.ArgumentParser(); ap.add_argument("--method", default="cubic"); ap.add_argument("--batch", type=int, default=200000)
ap.add_argument("--mk", type=int, default=64); ap.add_argument("--mt", type=int, default=16)
a = ap.parse_args()
lib = _lib.load(); engine.require_device()
d = synth.torch_batch(a.batch, 64, 16)
Kq, Tq = synth.query_grids(a.mk, a.mt)
Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, a.method)      # warm
buf = torch.zeros(8 * 8 * 512, dtype=torch.int64, device="cuda")
lib.ivs_debug_stamps(buf.data_ptr(), buf.numel())
engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, a.method)
torch.cuda.synchronize()
grid = lib.ivs_debug_last_grid()
lib.ivs_debug_stamps(None, 0)
s = buf[: grid * 8].view(grid, 8).double()
n = s[:, 7].sum()
names = ["stage+prefetch-issue", "k-phase", "strike sweeps", "strike eval", "maturity solve", "maturity eval+store"]
per = s[:, :6].sum(0) / n
tot = float(per.sum())
print(json.dumps({"method": a.method, "workgroups": int(grid), "kernel": engine.last_kernel(),
                  "cycles_per_surface_per_wave": {k: round(float(v), 1) for k, v in zip(names, per)},
                  "share": {k: round(float(v) / tot, 3) for k, v in zip(names, per)}, "total_cycles": round(tot, 1)}))
