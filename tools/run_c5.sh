set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
for spec in "cubic cfg5 5" "linear cfg5 5" "pchip cfg5 5" "akima cfg5 5"; do
  set -- $spec
  python bench.py --steps $3 --warmup 2 --method $1 --workload $2 --no-cpu-baseline > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err || { tail -5 $O/bench_$2_$1.err; exit 1; }
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_cfg5_cubic2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg5_cubic2 -- python3 $R/bench.py --steps 5 --warmup 2 --method cubic --workload cfg5 --no-cpu-baseline --check 0 > $O/bench_cfg5_cubic_under_rocprof.json 2> $O/trace_cfg5.err || { tail -5 $O/trace_cfg5.err; exit 1; }
f=$(find $O/trace_cfg5_cubic2 -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_cfg5_cubic.csv
cd $R
rm -rf $R/gpurun_out/pmc_cfg5
bash tools/pmc_run.sh cfg5 --steps 3 --warmup 1 --workload cfg5 > $O/pmc_cfg5.txt 2>&1 || { tail -5 $O/pmc_cfg5.txt; exit 1; }
python tools/ragged_probe.py > $O/ragged_probe.txt 2>&1
for f in $O/bench_cfg5_*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']/1e6,1), round(d['roofline']['kernel_ms_avg'],3), round(d['roofline']['achieved']), round(d['roofline']['frac'],3))"; done
grep "ivs::" $O/kernel_stats_cfg5_cubic.csv | cut -d, -f1-4 | cut -c1-120
