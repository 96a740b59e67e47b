#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
L="tools/abx/libivs_r3j.so tools/abx/libivs_mpabl1.so tools/abx/libivs_mpabl2.so"
for m in cubic pchip akima quadratic; do
echo "== $m, 10 % of the quotes missing: full / no slope solve / no strike evaluation"; timeout -k 10 200 python3 tools/ab_bench.py $L --method $m --nan-frac 0.1 --rounds 4 2>&1 | grep -v amdgpu.ids
done
