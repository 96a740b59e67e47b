#!/bin/bash
# does the allocator mode change the per-allocation spread of the headline kernel?  (plain runs of tools/placement_probe.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/alloc_modes; mkdir -p $O; cd $R
for mode in default expandable nocache; do
  case $mode in
    default) unset PYTORCH_HIP_ALLOC_CONF PYTORCH_CUDA_ALLOC_CONF PYTORCH_NO_CUDA_MEMORY_CACHING;;
    expandable) export PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True; unset PYTORCH_NO_CUDA_MEMORY_CACHING;;
    nocache) unset PYTORCH_HIP_ALLOC_CONF PYTORCH_CUDA_ALLOC_CONF; export PYTORCH_NO_CUDA_MEMORY_CACHING=1;;
  esac
  for rep in 1 2; do
    timeout -k 10 200 python3 tools/placement_probe.py --bufs 8 --json $O/${mode}_$rep.json > $O/${mode}_$rep.txt 2>&1 || { tail -5 $O/${mode}_$rep.txt; continue; }
    python3 - <<PY
import json
d=json.load(open("$O/${mode}_$rep.json"))
ms=[e["median_ms"] for e in d["seq"] if e["round"]==1]
print("$mode run $rep: per-buffer median ms min %.4f max %.4f mean %.4f  first %.4f  -> spread %.1f %%" % (min(ms), max(ms), sum(ms)/len(ms), ms[0], (max(ms)/min(ms)-1)*100))
PY
  done
done
