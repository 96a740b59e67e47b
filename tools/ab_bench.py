#!/usr/bin/env python3
"""Interleaved A/B timing of builds of libivs.so in ONE process on ONE device (cdna guide rule 24).
    python tools/ab_bench.py libA.so libB.so [--method cubic] [--rounds 8] [--ragged] [--groups 8,16]
Libraries of ABI 1 (no workspace argument) and ABI 2 can be mixed: the call is made per ivs_version()."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import _lib, synth

ap = argparse.ArgumentParser(); ap.add_argument("libs", nargs="+"); ap.add_argument("--method", default="cubic")
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--groups", default="", help="comma list: IVS_FLAG_MAP_GROUPS override per library (ABI 2)")
ap.add_argument("--batch", type=int, default=1_000_000)
ap.add_argument("--nk", type=int, default=64); ap.add_argument("--mk", type=int, default=64); ap.add_argument("--mt", type=int, default=16)
ap.add_argument("--ragged", action="store_true", help="config 5: strike counts lo..hi per surface")
ap.add_argument("--lo", type=int, default=8); ap.add_argument("--hi", type=int, default=128)
ap.add_argument("--outs", type=int, default=1, help="time every library on this many separately allocated output buffers (placement sensitivity)")
ap.add_argument("--nan-frac", type=float, default=0.0, help="this share of the quotes missing (NaN), as bench.py --nan-frac")
ap.add_argument("--check", action="store_true", help="compare the outputs of the libraries (max abs diff vs the first)")
a = ap.parse_args()
_p, _i64, _i32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
V1 = [_p, _p, _i64, _i32, _p, _i64, _i32, _p, _i64, _p, _i64, _i32, _p, _i64, _i32, _p, _p, _i32, _i32, _p]
V2 = V1[:-1] + [_p, _sz, _p]
libs = []
for pth in a.libs:
    lib = C.CDLL(os.path.abspath(pth))
    lib.ivs_version.restype = C.c_int
    v = lib.ivs_version()
    lib.ivs_surface_batch_f64.restype = C.c_int
    lib.ivs_surface_batch_f64.argtypes = V2 if v >= 2 else V1
    if v >= 2:
        lib.ivs_surface_workspace_bytes.restype = _sz; lib.ivs_surface_workspace_bytes.argtypes = [_i64, _i32]
    libs.append((lib, v))
if a.ragged:
    d = synth.torch_ragged_batch(a.batch, 16, a.lo, a.hi)
    koff, nk, kstr = d["k_off"].data_ptr(), d["nK_max"], 0
else:
    d = synth.torch_batch(a.batch, a.nk, 16)
    koff, nk, kstr = None, a.nk, a.nk
if a.nan_frac > 0:
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    d["sigma"][torch.rand(d["sigma"].shape, generator=g, device="cuda") < a.nan_frac] = float("nan")
Kq, Tq = synth.query_grids(a.mk, a.mt); Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
out = torch.empty((a.batch, a.mt, a.mk), dtype=torch.float64, device="cuda"); st = torch.empty(a.batch, dtype=torch.int32, device="cuda")
code = _lib.METHOD_CODES[a.method]
groups = [int(g) for g in a.groups.split(",")] if a.groups else []
ws = {}
def run(i):
    lib, v = libs[i]
    args = [d["K"].data_ptr(), koff, kstr, nk, d["T"].data_ptr(), 0, 16, d["sigma"].data_ptr(), a.batch,
            Kq.data_ptr(), 0, a.mk, Tq.data_ptr(), 0, a.mt, out.data_ptr(), st.data_ptr(), code]
    s = torch.cuda.current_stream().cuda_stream
    if v >= 2:
        if i not in ws:
            ws[i] = torch.empty(lib.ivs_surface_workspace_bytes(a.batch, 1 if a.ragged else 0), dtype=torch.uint8, device="cuda")
        flags = ((groups[i] & 0xff) << 8) if i < len(groups) else 0
        rc = lib.ivs_surface_batch_f64(*args, flags, ws[i].data_ptr(), ws[i].numel(), s)
    else:
        rc = lib.ivs_surface_batch_f64(*args, 0, s)
    assert rc == 0, rc
times = [[] for _ in libs]
ref = None
for i in range(len(libs)):
    run(i); run(i)
    if a.check:
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        else:
            dif = (out - ref).abs(); dif[torch.isnan(out) & torch.isnan(ref)] = 0
            print(f"{a.libs[i]}: max |diff| vs {a.libs[0]} = {float(dif.nan_to_num(nan=float('inf')).max()):.3e}")
torch.cuda.synchronize()
for r in range(a.rounds):
    for i in range(len(libs)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(i); run(i); run(i); e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 3)
if a.outs > 1:
    outs = [out] + [torch.empty_like(out) for _ in range(a.outs - 1)]
    for oi, o in enumerate(outs):
        out = o
        row = []
        for i in range(len(libs)):
            run(i); tt = []
            for r in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(i); run(i); run(i); e1.record(); torch.cuda.synchronize()
                tt.append(e0.elapsed_time(e1) / 3)
            row.append(sorted(tt)[1])
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        flat = o.view(-1)
        flat.zero_(); torch.cuda.synchronize()
        e0.record(); flat.zero_(); flat.zero_(); e1.record(); sm = flat.sum(); sm2 = flat.sum(); e2.record(); torch.cuda.synchronize()
        gb = flat.numel() * 8 / 1e9
        print(f"out buffer {oi} at {o.data_ptr():#x} (sigma at {d['sigma'].data_ptr():#x}): " + "  ".join(f"{x:.3f}" for x in row) +
              f" ms | plain write {2 * gb / (e0.elapsed_time(e1) * 1e-3):.0f} GB/s, plain read {2 * gb / (e1.elapsed_time(e2) * 1e-3):.0f} GB/s")
for pth, t in zip(a.libs, times):
    t = sorted(t)
    print(f"{pth}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f} ms  -> {a.batch / t[len(t)//2] / 1e3:.1f} M surfaces/s (median)")
