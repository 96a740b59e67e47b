#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
cp iv_interpolation_amd/libivs.so /tmp/libivs_keep.so
for v in keep cap160 cap176 cap192 cap208 cap224 keep cap192; do
  [ $v = keep ] && cp /tmp/libivs_keep.so iv_interpolation_amd/libivs.so || cp tools/abx/libivs_$v.so iv_interpolation_amd/libivs.so
  for m in linear cubic; do
    timeout -k 10 120 python3 tests/bench/bench_symbols.py --method $m --device-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['device_frame_fused']; print('$v $m fused %.4f ms frac %.3f'%(f['ms'], f['frac_of_8TBps']))"
  done
done
cp /tmp/libivs_keep.so iv_interpolation_amd/libivs.so
