#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "symbol or frame or batch or fused" > $O/gputests_job12.txt 2>&1; tail -3 $O/gputests_job12.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job12.txt && exit 1
cp iv_interpolation_amd/libivs.so /tmp/libivs_keep.so
for v in keep cap192 cap128 keep cap128; do
  [ $v = keep ] && cp /tmp/libivs_keep.so iv_interpolation_amd/libivs.so || cp tools/abx/libivs_$v.so iv_interpolation_amd/libivs.so
  for m in linear cubic; do
    timeout -k 10 120 python3 tests/bench/bench_symbols.py --method $m --device-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['device_frame_fused']; print('$v $m fused %.4f ms frac %.3f'%(f['ms'], f['frac_of_8TBps']))"
  done
done
cp /tmp/libivs_keep.so iv_interpolation_amd/libivs.so
timeout -k 10 300 python3 tests/bench/bench_symbols.py --method linear > $O/bench_symbols_linear_job12.json 2>$O/bench_symbols_job12.err
python3 -c "import json; d=json.load(open('$O/bench_symbols_linear_job12.json')); print(d['end_to_end_batch']); print(d['end_to_end_frame'])"
