#!/bin/bash
# HBM-traffic counters of the dominant search kernel (k_frontier_step) at ANY size: separate rocprofv3 --pmc passes restricted to that
# kernel (--kernel-include-regex: the input build and the other kernels run uninstrumented), aggregated over the launches of the one
# timed search (the builder's merges launch the same kernel earlier: the last `launches_per_step` rows are the search's).
# Usage (on the GPU box): bash tools/pmc_step_kernel.sh <out-subdir-of-gpurun_out> <reads per set> [extra bench args]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$1; reads=$2; shift; shift
mkdir -p $out; : > $out/pmc.txt
sets=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum")
cd /tmp
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); rm -rf /tmp/pmcs_$i
  timeout 1500 rocprofv3 --pmc $set --kernel-include-regex "k_frontier_step" --output-format csv -d /tmp/pmcs_$i -- python3 $R/bench.py --reads $reads --no-cpu-baseline --no-verify --no-host --target off --keep-pool --steps 1 --warmup 1 "$@" > /tmp/pmcs_$i.json 2> /tmp/pmcs_$i.log
  f=$(find /tmp/pmcs_$i -name "*counter_collection.csv" | head -1)
  n=$(python3 -c "import json,sys; print(int(round(json.loads([l for l in open('/tmp/pmcs_$i.json').read().splitlines() if l.startswith('{\"metric')][-1])['roofline']['launches_per_step'])))" 2>/dev/null)
  if [ -n "$f" ] && [ -n "$n" ]; then python3 $R/tools/pmc_aggregate.py $f --last $n --kernel k_frontier_step >> $out/pmc.txt; else echo "pass $i ($set) failed" >> $out/pmc.txt; tail -5 /tmp/pmcs_$i.log >> $out/pmc.txt; fi
  cp /tmp/pmcs_$i.json $out/bench_pass$i.json 2>/dev/null; grep -v "bench\] input" /tmp/pmcs_$i.log | tail -40 > $out/bench_pass$i.log
done
