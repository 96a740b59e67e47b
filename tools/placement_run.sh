#!/bin/bash
# usage: tools/placement_run.sh   -- on the GPU box: plain run + separate --pmc passes of tools/placement_probe.py into gpurun_out/placement/
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/placement
mkdir -p $O/plain
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/placement_probe.py --json $O/plain/seq.json > $O/plain/log.txt 2>&1
tail -22 $O/plain/log.txt
i=0
for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  mkdir -p $O/pass$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pass$i -- python3 $R/tools/placement_probe.py --json $O/pass$i/seq.json > $O/pass$i/log.txt 2>&1 || { echo "pass $i ($set) failed"; tail -5 $O/pass$i/log.txt; }
done
python3 $R/tools/placement_summarize.py $O | tee $O/summary.txt
