#!/bin/bash
# HBM-traffic and SQ counter passes over one merge step of bench.py (config 2), aggregated per kernel on the GPU box.
# Separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass); no tracing options combined.
# Usage (on the GPU box): bash tools/pmc_passes.sh <out-subdir-of-gpurun_out> [sq]      (PMC_BENCH_ARGS="--workload genome --coverage 300": another workload than config 2)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; mkdir -p $out; : > $out/pmc.txt
sets=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum")
if [ "$2" = "sq" ]; then
  sets+=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
fi
cd /tmp
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 900 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python3 $R/bench.py $PMC_BENCH_ARGS --no-cpu-baseline --no-verify --no-host --target off --steps 1 --warmup 0 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_aggregate.py $f >> $out/pmc.txt; else echo "pass $i ($set) failed" >> $out/pmc.txt; tail -5 /tmp/pmc_$i.log >> $out/pmc.txt; fi
done
