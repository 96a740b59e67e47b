#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...>   -- runs on the GPU box; separate --pmc passes, csv into gpurun_out/pmc_<tag>/
# PMC_PROG="tests/bench/bench_symbols.py --device-only" tools/pmc_run.sh symbols --method linear   -- another program than bench.py
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$n -- python3 ${PMC_PROG:+$R/}${PMC_PROG:-$R/bench.py --no-cpu-baseline --check 0} "$@" > $R/gpurun_out/pmc_$TAG/$n.log 2>&1 || { echo "pass $n failed"; tail -5 $R/gpurun_out/pmc_$TAG/$n.log; }
done
python3 $R/tools/pmc_summarize.py $R/gpurun_out/pmc_$TAG
