#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "fused" > $O/gputests_fused.txt 2>&1; tail -3 $O/gputests_fused.txt
cd /tmp && export TMPDIR=/tmp
for m in linear cubic; do
  python3 $R/tests/bench/bench_symbols.py --method $m --e2e 64 > $O/bench_symbols_$m.json 2>$O/bench_symbols_$m.err
  python3 - <<PY
import json; d=json.load(open("$O/bench_symbols_$m.json"))
for k in ("device","device_frame_fused","device_frame_separate_calls"): print("$m", k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items() if a in ("ms","GBps","frac_of_8TBps","rows_per_s")})
print("$m e2e frame", d["end_to_end_frame"]["symbols_per_s"], "batch", d["end_to_end_batch"]["symbols_per_s"])
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_symbols_$m -- python3 $R/tests/bench/bench_symbols.py --method $m --e2e 8 > /dev/null 2>$O/prof_symbols_$m.log
  f=$(find $O/prof_symbols_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_symbols_$m.csv && grep "ivs::" $f | cut -c1-150
done
