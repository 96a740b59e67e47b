#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
for b in 125000 1000000; do for l in eager graph; do
  timeout -k 10 300 python3 bench.py --batch $b --launch $l --no-other-configs --no-cpu-baseline --check 0 > $O/bench_cfg3_b${b}_$l.json 2>>$O/bench_err.txt
  timeout -k 10 300 python3 bench.py --workload cfg5 --batch $b --launch $l --no-other-configs --no-cpu-baseline --check 0 > $O/bench_cfg5_b${b}_$l.json 2>>$O/bench_err.txt
done; done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/bench_cfg*_b*_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],4), "ms", "frac", round(d["roofline"]["frac"],3), d["config"]["launch"][:20])
    except Exception as e: print(f, e)
PY
tail -3 $O/bench_err.txt
