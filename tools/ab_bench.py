#!/usr/bin/env python3
"""Interleaved A/B timing of two builds of libivs.so in ONE process on ONE device (cdna guide rule 24).
    python tools/ab_bench.py libA.so libB.so [--method cubic] [--rounds 8]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import _lib, synth

ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--method", default="cubic")
ap.add_argument("--rounds", type=int, default=8); ap.add_argument("--groups", default="", help="comma list: IVS_MAP_GROUPS seen by each lib at its first call"); ap.add_argument("--batch", type=int, default=1_000_000)
ap.add_argument("--nk", type=int, default=64); ap.add_argument("--mk", type=int, default=64); ap.add_argument("--mt", type=int, default=16)
a = ap.parse_args()
libs = []
for pth in a.libs:
    lib = C.CDLL(os.path.abspath(pth))
    res, args = _lib.SIGNATURES["ivs_surface_batch_f64"]
    lib.ivs_surface_batch_f64.restype = res; lib.ivs_surface_batch_f64.argtypes = args
    libs.append(lib)
d = synth.torch_batch(a.batch, a.nk, 16)
Kq, Tq = synth.query_grids(a.mk, a.mt); Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
out = torch.empty((a.batch, a.mt, a.mk), dtype=torch.float64, device="cuda"); st = torch.empty(a.batch, dtype=torch.int32, device="cuda")
code = _lib.METHOD_CODES[a.method]
def run(lib):
    rc = lib.ivs_surface_batch_f64(d["K"].data_ptr(), None, a.nk, a.nk, d["T"].data_ptr(), 0, 16, d["sigma"].data_ptr(), a.batch,
                                   Kq.data_ptr(), 0, a.mk, Tq.data_ptr(), 0, a.mt, out.data_ptr(), st.data_ptr(), code, 0,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
times = [[] for _ in libs]
groups = a.groups.split(",") if a.groups else []
for i, lib in enumerate(libs):
    if groups:
        os.environ["IVS_MAP_GROUPS"] = groups[i]      # read once per library instance, at its first call
    run(lib); run(lib)
torch.cuda.synchronize()
for r in range(a.rounds):
    for i, lib in enumerate(libs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(lib); run(lib); run(lib); e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 3)
for pth, t in zip(a.libs, times):
    t = sorted(t)
    print(f"{pth}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms  -> {a.batch / t[len(t)//2] / 1e3:.1f} M surfaces/s (median)")
