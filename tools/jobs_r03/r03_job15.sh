#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/nanfew; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for f in 0.0005 0.005 0.02; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$f -- python3 $R/bench.py --steps 5 --warmup 2 --nan-frac $f --no-cpu-baseline --no-other-configs --check 0 > $O/bench_nan_$f.json 2> $O/trace_$f.err || { tail -5 $O/trace_$f.err; exit 1; }
cp "$(find $O/trace_$f -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_nan_$f.csv
echo "== nan-frac $f"; python3 -c "import json;d=json.load(open('$O/bench_nan_$f.json'));print(round(d['value']/1e6,1),'M/s', d['roofline']['kernel_ms_avg'],'ms')"
grep "ivs::" $O/kernel_stats_nan_$f.csv | cut -d, -f1-4 | cut -c1-150
rm -rf $O/trace_$f
done
