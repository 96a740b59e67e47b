#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "ragged or var or config5 or runtime_maturity or fuzz or random_case or one_pass or config1 or guarded or capturable" > $O/gputests_job10.txt 2>&1; tail -3 $O/gputests_job10.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted" $O/gputests_job10.txt && exit 1
for m in cubic linear pchip akima quadratic; do
  timeout -k 10 200 python3 bench.py --workload cfg5 --method $m --no-other-configs --no-cpu-baseline --check 0 --steps 10 > $O/bench_cfg5_${m}.json 2>>$O/bench_err.txt
  python3 -c "import json;d=json.load(open('$O/bench_cfg5_${m}.json'));print('cfg5 $m', round(d['value']/1e6,1), 'M/s frac', round(d['roofline']['frac'],3), d['config']['kernel'])"
done
