#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
L="tools/abx/libivs_r3h.so tools/abx/libivs_maxilp.so tools/abx/libivs_memclause.so tools/abx/libivs_bias0.so tools/abx/libivs_bias100.so"
echo "== cfg3 cubic"; timeout -k 10 300 python3 tools/ab_bench.py $L --method cubic --check --rounds 8 2>&1 | grep -v amdgpu.ids
echo "== cfg3 linear"; timeout -k 10 300 python3 tools/ab_bench.py $L --method linear --rounds 6 2>&1 | grep -v amdgpu.ids
echo "== cfg5 cubic"; timeout -k 10 300 python3 tools/ab_bench.py $L --method cubic --ragged --rounds 6 2>&1 | grep -v amdgpu.ids
echo "== cfg3 cubic 10 % missing"; timeout -k 10 300 python3 tools/ab_bench.py $L --method cubic --nan-frac 0.1 --rounds 4 2>&1 | grep -v amdgpu.ids
echo "== cfg4 cubic (400k)"; timeout -k 10 300 python3 tools/ab_bench.py $L --method cubic --mk 256 --mt 64 --batch 400000 --rounds 4 2>&1 | grep -v amdgpu.ids
