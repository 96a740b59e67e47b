#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "akima or fuzz or random_case or golden" > $O/gputests_job17.txt 2>&1; tail -2 $O/gputests_job17.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job17.txt && { tail -30 $O/gputests_job17.txt; exit 1; }
L="tools/abx/libivs_r3c.so iv_interpolation_amd/libivs.so"
echo "== cfg3 akima"; timeout -k 10 200 python3 tools/ab_bench.py $L --method akima --check --rounds 8 2>&1 | grep -v amdgpu.ids
echo "== cfg5 akima (ragged 8..128)"; timeout -k 10 200 python3 tools/ab_bench.py $L --method akima --ragged --check --rounds 8 2>&1 | grep -v amdgpu.ids
echo "== akima ragged 8..64"; timeout -k 10 200 python3 tools/ab_bench.py $L --method akima --ragged --lo 8 --hi 64 --check --rounds 6 2>&1 | grep -v amdgpu.ids
echo "== akima ragged 65..128"; timeout -k 10 200 python3 tools/ab_bench.py $L --method akima --ragged --lo 65 --hi 128 --batch 500000 --check --rounds 6 2>&1 | grep -v amdgpu.ids
echo "== akima, 10 % of the quotes missing"; timeout -k 10 200 python3 tools/ab_bench.py $L --method akima --nan-frac 0.1 --check --rounds 5 2>&1 | grep -v amdgpu.ids
echo "== cfg4 akima (256x64 out)"; timeout -k 10 200 python3 tools/ab_bench.py $L --method akima --mk 256 --mt 64 --batch 400000 --check --rounds 4 2>&1 | grep -v amdgpu.ids
