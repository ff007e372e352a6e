export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_g
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|SQ_WAVES|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_VMEM_RD|SQ_INSTS_VMEM_WR|SQ_INSTS_SMEM|SQ_LDS_BANK_CONFLICT|SQ_LDS_IDX_ACTIVE|SQ_ACTIVE_INST_VMEM|SQ_ACTIVE_INST_SCA|SQ_WAIT_INST_LDS|SQ_INST_CYCLES_VMEM|TCC_HIT_sum|TCC_MISS_sum|TCP_TCC_READ_REQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCP_PENDING_STALL_CYCLES_sum|TCP_TA_TCP_STATE_READ_sum)\b" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc_g/avail.txt
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 900 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$i -- python3 $R/bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 0 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_aggregate.py $f >> $R/gpurun_out/pmc_g/pmc.txt; else echo "pass $i failed" >> $R/gpurun_out/pmc_g/pmc.txt; tail -5 /tmp/pmc_$i.log >> $R/gpurun_out/pmc_g/pmc.txt; fi
done
cat $R/gpurun_out/pmc_g/avail.txt; echo; grep -E "k_frontier_step|k_build_recs|k_tile_build|k_enc_emit|k_interleave " $R/gpurun_out/pmc_g/pmc.txt
