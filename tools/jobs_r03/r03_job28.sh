#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "missing_quotes or config2 or nan" > $O/gputests_job28.txt 2>&1; tail -2 $O/gputests_job28.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job28.txt && { tail -30 $O/gputests_job28.txt; exit 1; }
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tq -- python3 $R/bench.py --steps 10 --warmup 2 --batch 125000 --no-cpu-baseline --no-other-configs --check 0 > $O/bench_tq.json 2> $O/trace_tq.err
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace_tq/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ivs::" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg ns", r["AverageNs"])
PY
rm -rf $O/trace_tq
