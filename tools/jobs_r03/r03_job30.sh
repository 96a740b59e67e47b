#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "quadratic or akima or cubic or missing_quotes or masked or nan or fuzz" > $O/gputests_job30.txt 2>&1; tail -2 $O/gputests_job30.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job30.txt && { tail -40 $O/gputests_job30.txt; exit 1; }
L="tools/abx/libivs_r3k.so iv_interpolation_amd/libivs.so"
for m in cubic cubicspline akima quadratic; do
echo "== $m, 10 % of the quotes missing"; timeout -k 10 200 python3 tools/ab_bench.py $L --method $m --nan-frac 0.1 --check --rounds 5 2>&1 | grep -v amdgpu.ids
done
