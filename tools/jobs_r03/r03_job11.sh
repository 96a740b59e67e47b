#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "nan or masked or missing or quadratic or symbol or golden or fuzz" > $O/gputests_job11.txt 2>&1; tail -3 $O/gputests_job11.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job11.txt && exit 1
for m in cubic quadratic; do
  echo "== $m, 10 % of the quotes missing"
  timeout -k 10 200 python3 tools/ab_bench.py tools/abx/libivs_r3a.so tools/abx/libivs_r3b.so --method $m --nan-frac 0.1 --check --rounds 6 2>&1 | grep -v amdgpu.ids
done
echo "== quadratic dense cfg3"
timeout -k 10 200 python3 tools/ab_bench.py tools/abx/libivs_r3a.so tools/abx/libivs_r3b.so --method quadratic --check --rounds 6 2>&1 | grep -v amdgpu.ids
