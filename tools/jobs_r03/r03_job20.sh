#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "missing_quotes or nan" > $O/gputests_job20.txt 2>&1; tail -2 $O/gputests_job20.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job20.txt && { tail -30 $O/gputests_job20.txt; exit 1; }
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tq -- python3 $R/bench.py --steps 10 --warmup 2 --batch 125000 --no-cpu-baseline --no-other-configs --check 0 > $O/bench_tq.json 2> $O/trace_tq.err
grep "ivs::" "$(find $O/trace_tq -name '*kernel_stats.csv' | head -1)" | cut -d, -f1-4 | cut -c1-150
rm -rf $O/trace_tq
cd $R
for i in 1 2; do python3 bench.py --batch 125000 --no-other-configs --no-cpu-baseline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('125k: %.1f M/s %.4f ms'%(d['value']/1e6, d['roofline']['kernel_ms_avg']))"; done
