#!/usr/bin/env python3
"""Join tools/placement_probe.py's launch sequence with the rocprofv3 counters of the same process (one directory per PMC
pass: seq.json + **/*counter_collection.csv) and print, per counter, its per-buffer averages next to the per-buffer time and
the correlation between the two.   python3 tools/placement_summarize.py gpurun_out/placement"""
import csv
import glob
import json
import os
import sys

import numpy as np

root = sys.argv[1]
report = {}
for seqf in sorted(glob.glob(os.path.join(root, "*", "seq.json"))):
    d = os.path.dirname(seqf)
    s = json.load(open(seqf))
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "surface_pass_kernel" in r.get("Kernel_Name", "")]
    if not rows:
        continue
    by_disp = {}
    for r in rows:
        by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [by_disp[k] for k in sorted(by_disp)]
    L, W = s["launches"], s["warm"]
    disp = disp[s["spin_launches"]:]
    if len(disp) != L * len(s["seq"]):
        print(f"{d}: {len(disp)} dispatches after the spin-up, expected {L * len(s['seq'])}; skipped")
        continue
    ms = np.array([e["median_ms"] for e in s["seq"]])
    out = {"ms": [round(float(x), 4) for x in ms], "buf": [e["buf"] for e in s["seq"]], "round": [e["round"] for e in s["seq"]]}
    for c in sorted(disp[0]):
        v = np.array([np.mean([disp[i * L + j][c] for j in range(W, L)]) for i in range(len(s["seq"]))])
        cc = float(np.corrcoef(ms, v)[0, 1]) if v.std() > 0 and ms.std() > 0 else 0.0
        out[c] = {"per_buffer": [float(f"{x:.6g}") for x in v], "corr_with_ms": round(cc, 3),
                  "fastest": float(f"{v[ms.argmin()]:.6g}"), "slowest": float(f"{v[ms.argmax()]:.6g}")}
    report[os.path.basename(d)] = out
    print(os.path.basename(d), "ms:", out["ms"])
    for c in sorted(disp[0]):
        print(f"   {c:44s} corr {out[c]['corr_with_ms']:+.3f}  fastest {out[c]['fastest']:.6g}  slowest {out[c]['slowest']:.6g}")
json.dump(report, open(os.path.join(root, "summary.json"), "w"), indent=1)
