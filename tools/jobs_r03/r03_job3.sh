#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/tools/abx
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== cfg3 cubic: 8 B/lane row stores vs paired 16 B/lane stores (row-pass kernel)" > $O/ab_put_pair.txt
timeout -k 10 300 python3 tools/ab_bench.py $X/libpass_old.so $X/libpass_pair.so --rounds 10 --check >> $O/ab_put_pair.txt 2>&1
echo "== cfg4 cubic 256x64 out (one-pass kernel), 400k surfaces" >> $O/ab_put_pair.txt
timeout -k 10 300 python3 tools/ab_bench.py $X/libabl0.so $X/libabl0_pair.so --rounds 6 --mk 256 --mt 64 --batch 400000 --check >> $O/ab_put_pair.txt 2>&1
echo "== cfg3 linear (row-pass lerp instantiation is cubic-only in the diag build: one-pass kernel)" >> $O/ab_put_pair.txt
timeout -k 10 300 python3 tools/ab_bench.py $X/libabl0.so $X/libabl0_pair.so --rounds 6 --method linear --check >> $O/ab_put_pair.txt 2>&1
grep -v amdgpu.ids $O/ab_put_pair.txt
for b in 125000 250000 500000; do
  timeout -k 10 300 python3 bench.py --batch $b --no-other-configs --no-cpu-baseline --check 0 > $O/bench_cfg3_b$b.json 2>$O/bench_err.txt
  timeout -k 10 300 python3 bench.py --workload cfg5 --batch $b --no-other-configs --no-cpu-baseline --check 0 > $O/bench_cfg5_b$b.json 2>>$O/bench_err.txt
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/bench_cfg*_b*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]/1e6,1), "M/s", round(d["ms_per_step"],4), "ms", "frac", round(d["roofline"]["frac"],3))
    except Exception as e: print(f, e)
PY
cd /tmp && export TMPDIR=/tmp
for m in linear cubic; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_symbols_$m -- python3 $R/tests/bench/bench_symbols.py --method $m --e2e 64 > $O/bench_symbols_$m.json 2>$O/prof_symbols_$m.log
  f=$(find $O/prof_symbols_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_symbols_$m.csv && head -12 $f
done
