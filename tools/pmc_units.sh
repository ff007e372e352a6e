#!/bin/bash
# Which unit of the memory pipeline is busy / stalled during k_frontier_step: TA, TCP, TCC and SQ-side VMEM counters (separate passes).
# Usage (on the GPU box): bash tools/pmc_units.sh <out file under gpurun_out> [bench args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; : > $out; cd /tmp
sets=("GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
      "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"
      "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"
      "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
      "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_NORMAL_WRITEBACK_sum")
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmc_u
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_u -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-host --steps 1 --warmup 0 "$@" > /tmp/pmc_u.log 2>&1
  f=$(find /tmp/pmc_u -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_aggregate.py $f | grep -E "k_frontier_step" >> $out; else echo "pass $i ($set) failed: $(grep -iE "error|invalid|not" /tmp/pmc_u.log | head -2)" >> $out; fi
done
cat $out
