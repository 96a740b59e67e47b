#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -q -x -k "quadratic or missing_quotes or masked" > $O/gputests_job26.txt 2>&1; tail -2 $O/gputests_job26.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job26.txt && { tail -40 $O/gputests_job26.txt; exit 1; }
L="tools/abx/libivs_r3i.so iv_interpolation_amd/libivs.so"
echo "== quadratic, 10 % missing"; timeout -k 10 200 python3 tools/ab_bench.py $L --method quadratic --nan-frac 0.1 --check --rounds 5 2>&1 | grep -v amdgpu.ids
