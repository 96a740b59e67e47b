#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "symbol or frame or batch or fused or greek" > $O/gputests_job19.txt 2>&1; tail -2 $O/gputests_job19.txt
grep -q "MEMORY_APERTURE\|Memory access fault\|Aborted\|failed" $O/gputests_job19.txt && { tail -30 $O/gputests_job19.txt; exit 1; }
timeout -k 10 300 python -m pytest tests/test_greeks.py -q -x -m gpu > $O/gputests_job19b.txt 2>&1; tail -2 $O/gputests_job19b.txt
for m in linear cubic linear cubic; do
  timeout -k 10 120 python3 tests/bench/bench_symbols.py --method $m --device-only 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); f=d['device_frame_fused']; print('$m fused %.4f ms %.0f GB/s frac %.3f'%(f['ms'], f['GBps'], f['frac_of_8TBps']))"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_frame_w -- python3 $R/tests/bench/bench_symbols.py --device-only --method linear > /dev/null 2> $O/pmc_frame_w.err
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/pmc_frame_w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "frame_fused" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, len(v), sum(v)/len(v), "KB per launch")
PY
