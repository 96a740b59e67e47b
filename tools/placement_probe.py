"""Which property of an output ALLOCATION moves the headline kernel?  (run on the GPU box, normally under rocprofv3 --pmc)

DESIGN 5: the same surface kernel on the same inputs runs up to 8 % faster or slower depending on which allocation it
writes to.  This probe allocates `--bufs` output tensors (all alive at once), runs the config-3 call `--launches` times on
each in turn (the first `--warm` of them untimed) and writes the launch sequence with the HIP-event times to a JSON file.
Under `rocprofv3 --pmc ... --kernel-trace` every dispatch of surface_pass_kernel gets its counters; tools/placement_summarize.py
joins the two by dispatch order and correlates each counter with the per-buffer time.

    python tools/placement_probe.py --json gpurun_out/placement/plain.json
    rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum ... --kernel-trace --output-format csv -d DIR -- python3 tools/placement_probe.py --json DIR/seq.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iv_interpolation_amd import engine, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1_000_000)
ap.add_argument("--method", default="cubic")
ap.add_argument("--bufs", type=int, default=10)
ap.add_argument("--launches", type=int, default=8)
ap.add_argument("--warm", type=int, default=3)
ap.add_argument("--json", required=True)
a = ap.parse_args()

B = a.batch
d = synth.torch_batch(B, 64, 16, seed=synth.BASE_SEED)
Kq_h, Tq_h = synth.query_grids(64, 16)
Kq = torch.from_numpy(Kq_h).cuda(); Tq = torch.from_numpy(Tq_h).cuda()
status = torch.empty((B,), dtype=torch.int32, device="cuda")
ws = engine.surface_workspace(B, False)
outs = [torch.empty((B, 16, 64), dtype=torch.float64, device="cuda") for _ in range(a.bufs)]


def step(o):
    engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, a.method, out=o, status=status, workspace=ws)


import time  # noqa: E402
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:            # clocks up; these launches are counted in `spin`
    pass
spin = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    step(outs[0]); torch.cuda.synchronize(); spin += 1
seq = []
for rnd in range(2):                             # two rounds: is the ranking stable within the process?
    for i, o in enumerate(outs):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
        for s_, e_ in ev:
            s_.record(); step(o); e_.record()
        torch.cuda.synchronize()
        ms = [s_.elapsed_time(e_) for s_, e_ in ev]
        seq.append({"round": rnd, "buf": i, "ptr": o.data_ptr(), "ms": ms, "median_ms": sorted(ms[a.warm:])[(a.launches - a.warm) // 2]})
res = {"batch": B, "method": a.method, "kernel": engine.last_kernel(), "spin_launches": spin, "launches": a.launches, "warm": a.warm,
       "sigma_ptr": d["sigma"].data_ptr(), "K_ptr": d["K"].data_ptr(), "seq": seq}
os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
with open(a.json, "w") as f:
    json.dump(res, f)
for e in seq:
    print("round %d buf %2d at %#x: %.4f ms" % (e["round"], e["buf"], e["ptr"], e["median_ms"]))
