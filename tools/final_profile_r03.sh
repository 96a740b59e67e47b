#!/bin/bash
# usage: tools/final_profile_r03.sh [part1|part2|part3|part4]   (gpurun calls of <= 20 min each)
# Round-3 measurement set on the GPU box -> gpurun_out/final/ (copied to profiles/r03/final/).
#   part1: GPU tests, CLI pipeline, bench lines (first-allocation numbers: --placement-tries 1 is the default now)
#   part2: rocprofv3 kernel traces of the same commands (rocprof avg must agree with the bench line), symbols / bridge benches
#   part3: PMC passes (separate --pmc runs) for cfg3 cubic, cfg5 cubic, cfg3 cubic with 10 % missing quotes, the fused frame pass
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
PART=${1:-part1}
if [ "$PART" = "part1" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
rm -rf /tmp/ivs_fs && python complete_pipeline.py --task all --synthetic 3 --data-dir /tmp/ivs_fs > $O/cli_pipeline.log 2>&1 || { tail -20 $O/cli_pipeline.log; exit 1; }
tail -3 $O/cli_pipeline.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
for spec in "cubic cfg3 20" "linear cfg3 20" "pchip cfg3 20" "akima cfg3 20" "cubicspline cfg3 20" "quadratic cfg3 20" "nearest cfg3 20" "cubic cfg4 10" "linear cfg4 10" "cubic cfg5 5" "linear cfg5 5" "pchip cfg5 5" "akima cfg5 5"; do
  set -- $spec
  python bench.py --steps $3 --warmup 2 --method $1 --workload $2 --no-cpu-baseline --no-other-configs > $O/bench_$2_$1.json 2> $O/bench_$2_$1.err || { tail -5 $O/bench_$2_$1.err; exit 1; }
done
python bench.py --steps 5 --warmup 1 --force-generic --no-cpu-baseline --no-other-configs > $O/bench_cfg3_cubic_generic.json 2> $O/bench_generic.err || exit 1
for m in cubic cubicspline linear pchip akima quadratic nearest; do      # 10 % of the quotes missing: 'missing quotes first'
  python bench.py --steps 5 --warmup 2 --method $m --nan-frac 0.1 --no-cpu-baseline --no-other-configs > $O/bench_cfg3_${m}_nan10.json 2> $O/bench_nan10_$m.err || { tail -5 $O/bench_nan10_$m.err; exit 1; }
done
python bench.py --steps 5 --warmup 2 --method cubic --nan-frac 0.0005 --no-cpu-baseline --no-other-configs > $O/bench_cfg3_cubic_nan_few.json 2> $O/bench_nan_few.err || exit 1
for m in cubic linear; do      # 0.5 % missing: the sparse rule of the probe
  python bench.py --steps 5 --warmup 2 --method $m --nan-frac 0.005 --no-cpu-baseline --no-other-configs > $O/bench_cfg3_${m}_nan05.json 2>> $O/bench_nan_few.err || exit 1
done
for b in 125000 250000 500000; do      # the cfg3 / cfg5 shards of the 8-, 4- and 2-GPU split
  python bench.py --batch $b --no-other-configs --no-cpu-baseline > $O/bench_cfg3_cubic_b$b.json 2>> $O/bench_small.err || exit 1
  python bench.py --workload cfg5 --batch $b --no-other-configs --no-cpu-baseline > $O/bench_cfg5_cubic_b$b.json 2>> $O/bench_small.err || exit 1
done
python bench.py --placement-tries 8 --no-other-configs --no-cpu-baseline > $O/bench_cfg3_cubic_placed.json 2> $O/bench_placed.err || exit 1
fi
if [ "$PART" = "part2" ]; then
cd /tmp && export TMPDIR=/tmp
for spec in "cubic cfg3 20" "linear cfg3 20" "pchip cfg3 20" "cubic cfg4 10" "cubic cfg5 5"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$2_$1 -- python3 $R/bench.py --steps $3 --warmup 2 --method $1 --workload $2 --no-cpu-baseline --no-other-configs --check 0 > $O/bench_$2_$1_under_rocprof.json 2> $O/trace_$2_$1.err || { tail -5 $O/trace_$2_$1.err; exit 1; }
  cp "$(find $O/trace_$2_$1 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_$2_$1.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3_cubic_nan10 -- python3 $R/bench.py --steps 5 --warmup 2 --nan-frac 0.1 --no-cpu-baseline --no-other-configs --check 0 > $O/bench_cfg3_cubic_nan10_under_rocprof.json 2> $O/trace_nan10.err || { tail -5 $O/trace_nan10.err; exit 1; }
cp "$(find $O/trace_cfg3_cubic_nan10 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_cfg3_cubic_nan10.csv
for m in linear cubic; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_symbols_$m -- python3 $R/tests/bench/bench_symbols.py --method $m --e2e 8 > /dev/null 2> $O/trace_symbols_$m.err || { tail -5 $O/trace_symbols_$m.err; exit 1; }
  cp "$(find $O/trace_symbols_$m -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_symbols_$m.csv
  python3 $R/tests/bench/bench_symbols.py --method $m > $O/bench_symbols_$m.json 2>> $O/bench_symbols.err || { tail -5 $O/bench_symbols.err; exit 1; }
done
cd $R
for s in spread_simulation price_as_midpoint simple_spread pipeline_inline trend_following; do python tests/bench/bench_bridge.py --strategy $s > $O/bench_bridge_$s.json 2>> $O/bench_bridge.err || { tail -5 $O/bench_bridge.err; exit 1; }; done
python tools/ragged_probe.py > $O/ragged_probe.txt 2>&1 || exit 1
python tools/layout_probe.py > $O/layout_probe.txt 2>&1 || exit 1
rm -rf $O/trace_*/
fi
if [ "$PART" = "part4" ]; then      # the frame pass only (after a change to ivs_frame.hpp): traces, bench, counters
cd /tmp && export TMPDIR=/tmp
for m in linear cubic; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_symbols_$m -- python3 $R/tests/bench/bench_symbols.py --method $m --e2e 8 > /dev/null 2> $O/trace_symbols_$m.err || { tail -5 $O/trace_symbols_$m.err; exit 1; }
  cp "$(find $O/trace_symbols_$m -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_symbols_$m.csv
  python3 $R/tests/bench/bench_symbols.py --method $m > $O/bench_symbols_$m.json 2>> $O/bench_symbols.err || { tail -5 $O/bench_symbols.err; exit 1; }
done
cd $R; rm -rf $O/trace_*/
PMC_PROG="tests/bench/bench_symbols.py --device-only" bash tools/pmc_run.sh symbols --method linear > $O/pmc_symbols.txt 2>&1 || { tail -5 $O/pmc_symbols.txt; exit 1; }
cp $R/gpurun_out/pmc_symbols/summary.json $O/pmc_symbols.json
fi
if [ "$PART" = "part3" ]; then
bash tools/pmc_run.sh cubic --steps 5 --warmup 1 --no-other-configs > $O/pmc_cubic.txt 2>&1 || { tail -5 $O/pmc_cubic.txt; exit 1; }
bash tools/pmc_run.sh cfg5 --steps 3 --warmup 1 --workload cfg5 > $O/pmc_cfg5.txt 2>&1 || { tail -5 $O/pmc_cfg5.txt; exit 1; }
bash tools/pmc_run.sh nan10 --steps 3 --warmup 1 --nan-frac 0.1 --no-other-configs > $O/pmc_nan10.txt 2>&1 || { tail -5 $O/pmc_nan10.txt; exit 1; }
PMC_PROG="tests/bench/bench_symbols.py --device-only" bash tools/pmc_run.sh symbols --method linear > $O/pmc_symbols.txt 2>&1 || { tail -5 $O/pmc_symbols.txt; exit 1; }
for t in cubic cfg5 nan10 symbols; do cp $R/gpurun_out/pmc_$t/summary.json $O/pmc_$t.json; done
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
