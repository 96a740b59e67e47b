"""Spread of the workgroups' finishing times in the row-pass kernel (diagnostic build ab/libpass_ends.so; GPU box).
    make -C tools ../ab/libpass_ends.so && python tools/ends_probe.py
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iv_interpolation_amd import synth
lib = C.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "ab/libpass_ends.so"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
d = synth.torch_batch(B, 64, 16)
Kq, Tq = synth.query_grids(64, 16); Kq = torch.from_numpy(Kq).cuda(); Tq = torch.from_numpy(Tq).cuda()
out = torch.empty((B, 16, 64), dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda")
lib.ivs_surface_workspace_bytes.restype = C.c_size_t; lib.ivs_surface_workspace_bytes.argtypes = [C.c_int64, C.c_int32]
ws = torch.empty(lib.ivs_surface_workspace_bytes(B, 0), dtype=torch.uint8, device="cuda")
p, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
lib.ivs_surface_batch_f64.argtypes = [p, p, i64, i32, p, i64, i32, p, i64, p, i64, i32, p, i64, i32, p, p, i32, i32, p, sz, p]
lib.ivs_diag_ends.argtypes = [p, C.c_int]
def run():
    rc = lib.ivs_surface_batch_f64(d["K"].data_ptr(), None, 64, 64, d["T"].data_ptr(), 0, 16, d["sigma"].data_ptr(), B, Kq.data_ptr(), 0, 64,
                                   Tq.data_ptr(), 0, 16, out.data_ptr(), st.data_ptr(), 1, 0, ws.data_ptr(), ws.numel(),
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0
for _ in range(20): run()
torch.cuda.synchronize()
assert lib.ivs_diag_ends(None, 0) == 1
for _ in range(5): run()
torch.cuda.synchronize()
n = 3072
h = np.zeros((n, 2), dtype=np.uint64)
assert lib.ivs_diag_ends(h.ctypes.data, n) == 0
t0 = h[:, 0].astype(np.int64); t1 = h[:, 1].astype(np.int64)
z = t0.min()
s = (t0 - z) * 1e-5; e = (t1 - z) * 1e-5                       # 100 MHz -> ms
print("kernel span %.3f ms; starts: max %.3f ms; ends: min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f ms" %
      (e.max(), s.max(), e.min(), np.percentile(e, 10), np.median(e), np.percentile(e, 90), e.max()))
dur = e - s
print("busy time per workgroup: min %.3f median %.3f max %.3f ms; idle share at the tail = %.1f %%" %
      (dur.min(), np.median(dur), dur.max(), 100 * (1 - dur.sum() / (n * e.max()))))
wg = np.arange(n)
for g in range(8):
    m = wg % 8 == g
    print("group/XCD %d: mean end %.3f  max end %.3f  mean busy %.3f ms" % (g, e[m].mean(), e[m].max(), dur[m].mean()))
