#!/bin/bash
# One GPU-box cycle: smoke, surface parity tests, phase stamps, benches.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not symbol_cases" > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
python tools/stamp_profile.py --method cubic > gpurun_out/stamp_cubic.json && python tools/stamp_profile.py --method linear > gpurun_out/stamp_linear.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_dense_cubic.json 2> gpurun_out/bench_dense_cubic.err
python bench.py --steps 20 --warmup 3 --method linear --no-cpu-baseline > gpurun_out/bench_dense_linear.json 2>gpurun_out/bench_dense_linear.err
python bench.py --steps 10 --warmup 2 --workload cfg4 --no-cpu-baseline > gpurun_out/bench_dense_cfg4.json 2>gpurun_out/bench_dense_cfg4.err
python bench.py --steps 10 --warmup 2 --workload cfg4 --method linear --no-cpu-baseline > gpurun_out/bench_dense_cfg4lin.json 2>gpurun_out/bench_dense_cfg4lin.err
python bench.py --steps 5 --warmup 1 --workload cfg5 --no-cpu-baseline > gpurun_out/bench_dense_cfg5.json 2>gpurun_out/bench_dense_cfg5.err
python - <<'PY'
import json,glob
for f in ("gpurun_out/stamp_cubic.json","gpurun_out/stamp_linear.json"):
    d=json.load(open(f)); print(d["method"], d["total_cycles"], d["cycles_per_surface_per_wave"])
for f in sorted(glob.glob("gpurun_out/bench_dense_*.json")):
    d=json.loads(open(f).read()); print(d["config"]["workload"][:4], d["config"]["method"], "%.1fM surf/s"%(d["value"]/1e6), "%.2f ms"%d["roofline"]["kernel_ms_avg"], "%.0f GB/s frac %.3f"%(d["roofline"]["achieved"], d["roofline"]["frac"]), d["parity_check"])
PY
