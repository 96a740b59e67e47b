#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "nearest or missing_quotes or nan or masked or fuzz" > $O/gputests_job21.txt 2>&1; tail -2 $O/gputests_job21.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job21.txt && { tail -30 $O/gputests_job21.txt; exit 1; }
L="tools/abx/libivs_r3d.so iv_interpolation_amd/libivs.so"
for f in 0.1 0.005; do
echo "== nearest, share of quotes missing $f"; timeout -k 10 200 python3 tools/ab_bench.py $L --method nearest --nan-frac $f --check --rounds 5 2>&1 | grep -v amdgpu.ids
done
