"""Per-launch time of the first launches after data generation (does the kernel need many warm-up launches?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import engine, synth
B = 1_000_000
d = synth.torch_batch(B, 64, 16)
Kq, Tq = synth.query_grids(64, 16); Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
out = torch.empty((B, 16, 64), dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, "cubic", out=out, status=st); e1.record()
        torch.cuda.synchronize(); ts.append(round(e0.elapsed_time(e1), 3))
    print("back-to-back sync'd launches:", ts)
    time.sleep(2.0)
ts = []
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
for i in range(12):
    evs[i][0].record(); engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, "cubic", out=out, status=st); evs[i][1].record()
torch.cuda.synchronize()
print("after 2 s idle, queued launches:   ", [round(a.elapsed_time(b), 3) for a, b in evs])
