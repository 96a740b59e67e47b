#!/bin/bash
# usage: tools/final_profile.sh [part1|part2|all]   (two gpurun calls of <= 20 min: part1 = tests + bench lines, part2 = traces + PMC)
# Round-end measurement set on the GPU box: full GPU test suite, bench lines for every workload, rocprofv3
# kernel-trace summaries of the same commands, PMC passes.  Everything lands under gpurun_out/final/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
PART=${1:-all}
if [ "$PART" != "part2" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
# the entry scripts from a fresh process (no torch imported by the caller): three stages on a scratch frame store
rm -rf /tmp/ivs_fs && python complete_pipeline.py --task all --synthetic 3 --data-dir /tmp/ivs_fs > $O/cli_pipeline.log 2>&1 || { tail -20 $O/cli_pipeline.log; exit 1; }
tail -3 $O/cli_pipeline.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
for spec in "cubic cfg3 20" "linear cfg3 20" "pchip cfg3 20" "akima cfg3 20" "cubicspline cfg3 20" "cubic cfg4 10" "linear cfg4 10" "cubic cfg5 5" "linear cfg5 5" "pchip cfg5 5" "akima cfg5 5"; do
  set -- $spec
  python bench.py --steps $3 --warmup 2 --method $1 --workload $2 --no-cpu-baseline --no-other-configs > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err || { tail -5 $O/bench_$2_$1.err; exit 1; }
done
python bench.py --steps 5 --warmup 1 --force-generic --no-cpu-baseline --no-other-configs > $O/bench_cfg3_cubic_generic.json 2> $O/bench_generic.err || exit 1
for m in cubic linear pchip akima; do      # 10 % of the quotes missing: row-pass / dense kernel tags, compaction kernel redoes
  python bench.py --steps 5 --warmup 2 --method $m --nan-frac 0.1 --no-cpu-baseline --no-other-configs > $O/bench_cfg3_${m}_nan10.json 2> $O/bench_nan10_$m.err || { tail -5 $O/bench_nan10_$m.err; exit 1; }
done
fi
if [ "$PART" != "part1" ]; then
cd /tmp && export TMPDIR=/tmp
for spec in "cubic cfg3 20" "linear cfg3 20" "pchip cfg3 20" "cubic cfg4 10" "cubic cfg5 5"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$2_$1 -- python3 $R/bench.py --steps $3 --warmup 2 --method $1 --workload $2 --no-cpu-baseline --no-other-configs --check 0 > $O/bench_$2_$1_under_rocprof.json 2> $O/trace_$2_$1.err || { tail -5 $O/trace_$2_$1.err; exit 1; }
  f=$(find $O/trace_$2_$1 -name '*kernel_stats.csv' | head -1)
  cp "$f" $O/kernel_stats_$2_$1.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3_cubic_nan10 -- python3 $R/bench.py --steps 5 --warmup 2 --nan-frac 0.1 --no-cpu-baseline --no-other-configs --check 0 > $O/bench_cfg3_cubic_nan10_under_rocprof.json 2> $O/trace_nan10.err || { tail -5 $O/trace_nan10.err; exit 1; }
cp "$(find $O/trace_cfg3_cubic_nan10 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_cfg3_cubic_nan10.csv
cd $R
bash tools/pmc_run.sh cubic --steps 5 --warmup 1 --no-other-configs > $O/pmc_cubic.txt 2>&1 || { tail -5 $O/pmc_cubic.txt; exit 1; }
bash tools/pmc_run.sh cfg5 --steps 3 --warmup 1 --workload cfg5 > $O/pmc_cfg5.txt 2>&1 || { tail -5 $O/pmc_cfg5.txt; exit 1; }
bash tools/pmc_run.sh nan10 --steps 3 --warmup 1 --nan-frac 0.1 --no-other-configs > $O/pmc_nan10.txt 2>&1 || { tail -5 $O/pmc_nan10.txt; exit 1; }
python tests/bench/bench_symbols.py --method cubic > $O/bench_symbols_cubic.json 2>> $O/bench_symbols.err || { tail -5 $O/bench_symbols.err; exit 1; }
for s in spread_simulation price_as_midpoint simple_spread pipeline_inline trend_following; do python tests/bench/bench_bridge.py --strategy $s > $O/bench_bridge_$s.json 2>> $O/bench_bridge.err || { tail -5 $O/bench_bridge.err; exit 1; }; done
python tests/bench/bench_symbols.py > $O/bench_symbols_linear.json 2> $O/bench_symbols.err || { tail -5 $O/bench_symbols.err; exit 1; }
python tools/ragged_probe.py > $O/ragged_probe.txt 2>&1 || exit 1
fi
python - <<'PY'
import json,glob,os
O=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/final")
for f in sorted(glob.glob(O+"/bench_*.json")):
    try: d=json.loads(open(f).read())
    except Exception as e: print(f, "unreadable", e); continue
    if "roofline" not in d: continue
    print(os.path.basename(f), "%.1fM surf/s"%(d["value"]/1e6), "%.3f ms"%d["roofline"]["kernel_ms_avg"], "%.0f GB/s frac %.3f"%(d["roofline"]["achieved"], d["roofline"]["frac"]), d["roofline"].get("kernel"))
PY
