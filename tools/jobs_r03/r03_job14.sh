#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "symbol or frame or batch or fused" > $O/gputests_job14.txt 2>&1; tail -2 $O/gputests_job14.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job14.txt && exit 1
L="tools/abx/libivs_r3c.so tools/abx/libivs_stag20.so tools/abx/libivs_stag40.so tools/abx/libivs_chunk2.so tools/abx/libivs_chunk1.so"
for b in 125000 250000 1000000; do
  echo "== cfg3 cubic, batch $b"
  timeout -k 10 200 python3 tools/ab_bench.py $L --batch $b --check --rounds 10 2>&1 | grep -v amdgpu.ids
done
echo "== cfg5 cubic, batch 125000"
timeout -k 10 200 python3 tools/ab_bench.py $L --batch 125000 --ragged --rounds 10 2>&1 | grep -v amdgpu.ids
