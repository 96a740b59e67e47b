#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
IVS_ERRLOG=$O/errlog_masked.txt timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "nan or masked or fuzz or random_case or golden" > $O/gputests_job9.txt 2>&1; tail -4 $O/gputests_job9.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted" $O/gputests_job9.txt && exit 1
for m in cubic cubicspline linear pchip; do
  timeout -k 10 200 python3 bench.py --nan-frac 0.1 --method $m --no-other-configs --no-cpu-baseline --check 0 --steps 10 > $O/bench_cfg3_${m}_nan10.json 2>>$O/bench_err.txt
done
timeout -k 10 200 python3 bench.py --nan-frac 0.02 --method cubic --no-other-configs --no-cpu-baseline --check 0 --steps 10 > $O/bench_cfg3_cubic_nan02.json 2>>$O/bench_err.txt
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/bench_cfg3_*_nan*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],3), "ms frac", round(d["roofline"]["frac"],3), d["config"]["kernel"])
    except Exception as e: print(f, e)
PY
sort -g -r $O/errlog_masked.txt | grep -i "nan\|masked" | head -5
