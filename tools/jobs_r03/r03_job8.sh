#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; X=$R/tools/abx; mkdir -p $O; cd $R
echo "== guided tail of the work queues: cfg3 cubic" > $O/ab_tail.txt
for b in 125000 250000 1000000; do
  echo "-- batch $b" >> $O/ab_tail.txt
  timeout -k 10 200 python3 tools/ab_bench.py $X/libpass_old.so $X/libpass_tail.so --batch $b --rounds 12 --check >> $O/ab_tail.txt 2>&1
done
echo "-- ragged 8..128, batch 125000 / 1000000" >> $O/ab_tail.txt
timeout -k 10 200 python3 tools/ab_bench.py $X/libpass_old.so $X/libpass_tail.so --ragged --batch 125000 --rounds 12 --check >> $O/ab_tail.txt 2>&1
timeout -k 10 200 python3 tools/ab_bench.py $X/libpass_old.so $X/libpass_tail.so --ragged --batch 1000000 --rounds 8 >> $O/ab_tail.txt 2>&1
grep -v amdgpu.ids $O/ab_tail.txt
