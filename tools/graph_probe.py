"""hipGraph capture / replay of one surface call, stage by stage (prints flush immediately).  usage: graph_probe.py uniform|ragged"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import engine, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
def say(*a): print(*a, flush=True)
B = 20000
Kq, Tq = synth.query_grids(64, 16); Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
if kind == "uniform":
    d = synth.torch_batch(B, 64, 16, seed=1); kw = {}
else:
    d = synth.torch_ragged_batch(B, 16, 8, 128, seed=2); kw = dict(k_off=d["k_off"], nK_max=128, n_maturities=16)
eager, _ = engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, "cubic", **kw)
torch.cuda.synchronize(); say("eager ok", engine.last_kernel())
out = torch.empty((B, 16, 64), dtype=torch.float64, device="cuda"); st = torch.empty((B,), dtype=torch.int32, device="cuda")
ws = engine.surface_workspace(B, kind == "ragged")
engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, "cubic", out=out, status=st, workspace=ws, **kw)
torch.cuda.synchronize(); say("eager with caller buffers ok", bool(torch.equal(out, eager)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, "cubic", out=out, status=st, workspace=ws, **kw)
say("captured")
out.fill_(float("nan")); torch.cuda.synchronize(); say("filled")
g.replay(); say("replay enqueued")
torch.cuda.synchronize(); say("replay done", bool(torch.equal(out, eager)))
