#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "missing_quotes or nan or masked" > $O/gputests_job16.txt 2>&1; tail -2 $O/gputests_job16.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job16.txt && { tail -30 $O/gputests_job16.txt; exit 1; }
for m in cubic linear; do
for f in 0.0005 0.001 0.002 0.005 0.02; do
  echo "== $m, share of quotes missing $f"
  timeout -k 10 200 python3 tools/ab_bench.py tools/abx/libivs_r3c.so iv_interpolation_amd/libivs.so --method $m --nan-frac $f --check --rounds 5 2>&1 | grep -v amdgpu.ids
done; done
