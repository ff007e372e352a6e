#!/bin/bash
# Which unit of the memory pipeline is busy / stalled during k_frontier_step: TLB, TCP, TCC counters in separate short passes
# (a set the hardware cannot collect together makes rocprofv3 abort: every pass has its own timeout).
# Usage (on the GPU box): bash tools/pmc_units.sh <out file under gpurun_out> [bench args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; shift; : > $out; cd /tmp
sets=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"
      "TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum"
      "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum"
      "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_IB_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum")
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmc_u
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_u -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-host --target off --steps 1 --warmup 0 "$@" > /tmp/pmc_u.log 2>&1
  f=$(find /tmp/pmc_u -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_aggregate.py $f | grep -E "k_frontier_step" >> $out; else echo "pass $i ($set) failed: $(grep -iE "error code|invalid" /tmp/pmc_u.log | head -1)" >> $out; fi
done
cat $out
