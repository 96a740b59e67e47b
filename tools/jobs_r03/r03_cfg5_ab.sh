#!/bin/bash
# round 3: config-5 size classes, old (8 B/lane row loads) vs wide (16 B/lane) staging and the phase ablations of the
# run-time-shape row-pass kernels (a1: no K-phase / sweeps, a2: no gathers, a6: neither = streaming skeleton)
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/tools/abx
O=$R/gpurun_out/ab_cfg5.txt
mkdir -p $R/gpurun_out
cd $R
for cls in "65 128" "8 64" "8 32" "33 64" "8 128"; do
  set -- $cls
  echo "== ragged $1..$2, 500k surfaces, cubic" >> $O
  timeout -k 10 300 python3 tools/ab_bench.py $X/libpass_old.so $X/libpass_wide.so $X/libpass_a1.so $X/libpass_a2.so $X/libpass_a6.so --ragged --lo $1 --hi $2 --batch 500000 --rounds 6 >> $O 2>&1 || exit 1
done
echo "== check wide vs old (cubic, linear), ragged 8..128" >> $O
timeout -k 10 300 python3 tools/ab_bench.py $X/libpass_old.so $X/libpass_wide.so --ragged --lo 8 --hi 128 --batch 200000 --rounds 2 --check >> $O 2>&1
tail -40 $O
