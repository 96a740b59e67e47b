#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "per_surface or fuzz or random_case or maturit or golden or symbol_batch or capturable" > $O/gputests_job6.txt 2>&1; tail -3 $O/gputests_job6.txt
grep -q "MEMORY_APERTURE\|Memory access fault" $O/gputests_job6.txt && exit 1
echo "== per-surface maturities: row-pass (new) vs one-pass kernels" > $O/layout_tsh.txt
timeout -k 10 200 python3 tools/layout_probe.py --methods cubic,cubicspline,linear --only "per-surface T" >> $O/layout_tsh.txt 2>&1
timeout -k 10 200 python3 tools/layout_probe.py --methods cubic,cubicspline --only "per-surface T" --one-pass >> $O/layout_tsh.txt 2>&1
timeout -k 10 200 python3 tools/layout_probe.py --methods cubic --only "benchmark" >> $O/layout_tsh.txt 2>&1
grep -v amdgpu.ids $O/layout_tsh.txt
python3 tests/bench/bench_symbols.py --method linear --e2e 2048 > $O/bench_symbols_linear_e2e.json 2>/dev/null
python3 - <<PY
import json; d=json.load(open("$O/bench_symbols_linear_e2e.json"))
print("e2e batch", d["end_to_end_batch"], "frame", d["end_to_end_frame"]["symbols_per_s"], "single", d["end_to_end_single"])
PY
