#!/usr/bin/env python3
"""Average PMC counters per dispatch of the ivs:: kernels found under a rocprofv3 output dir."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "ivs::" not in k: continue
        name = k.split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
