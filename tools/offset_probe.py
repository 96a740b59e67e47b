"""Does the RELATIVE placement of the quote and output buffers move the headline kernel?  (run on the GPU box)
One process, one set of inputs; the output (or the quote) tensor is a view into a larger buffer at a byte offset.
    python tools/offset_probe.py [--batch 1000000] [--method cubic]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import engine, synth

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=1_000_000); ap.add_argument("--method", default="cubic")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
B = a.batch
d = synth.torch_batch(B, 64, 16, seed=synth.BASE_SEED)
Kq_h, Tq_h = synth.query_grids(64, 16)
Kq = torch.from_numpy(Kq_h).cuda(); Tq = torch.from_numpy(Tq_h).cuda()
status = torch.empty((B,), dtype=torch.int32, device="cuda")
ws = engine.surface_workspace(B, False)
PAD = 33 << 30                                              # bytes
obuf = torch.empty(B * 1024 + PAD // 8, dtype=torch.float64, device="cuda")
sbuf = torch.empty(B * 1024 + (64 << 20), dtype=torch.float64, device="cuda")
offs = [m << 20 for m in (0, 1024, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 9216, 10240, 12288, 14336, 16384, 20480, 24576, 28672, 32768)]


def view(buf, off):
    return buf[off // 8: off // 8 + B * 1024].view(B, 16, 64)


def run(sig, out):
    def step():
        engine.surface_batch(d["K"], d["T"], sig, Kq, Tq, a.method, out=out, status=status, workspace=ws)
    for _ in range(3):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for s, e in ev:
        s.record(); step(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in ev)[2]


print("data_ptr sigma %#x  obuf %#x  sbuf %#x" % (d["sigma"].data_ptr(), obuf.data_ptr(), sbuf.data_ptr()))
t_spin = __import__("time").perf_counter()
while __import__("time").perf_counter() - t_spin < 0.3:
    run(d["sigma"], view(obuf, 0))
print("-- output view at byte offset (quotes fixed)")
for r in range(a.rounds):
    print("round", r, " ".join("%d:%.3f" % (o >> 20, run(d["sigma"], view(obuf, o))) for o in offs))
print("-- quote view at byte offset (output fixed at 0)")
for o in [0, 16 << 20, 32 << 20]:
    view(sbuf, o).copy_(d["sigma"])
    print(o >> 20, "MB %.3f ms" % run(view(sbuf, o), view(obuf, 0)))
print("-- fresh output tensors (new allocations)")
keep = []
for i in range(6):
    o = torch.empty((B, 16, 64), dtype=torch.float64, device="cuda"); keep.append(o)
    print("%#x %.3f ms" % (o.data_ptr(), run(d["sigma"], o)))
