#!/bin/bash
# rocprofv3 runs of bench.py (config 2) on the GPU box; summaries land in gpurun_out/<tag>_*.
# Usage: bash tools/profile_round.sh <tag>
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; cd /tmp
rm -rf /tmp/prof_k /tmp/prof_c
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-host --target off --steps 5 --warmup 1 > $R/gpurun_out/${tag}_bench_kernel_trace.log 2>&1
f=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/summarize_kernel_trace.py $f 5 > $R/gpurun_out/${tag}_config2_summary.md
s=$(find /tmp/prof_k -name "*kernel_stats.csv" | head -1)
grep -E "Name|bwtm::" $s | head -40 > $R/gpurun_out/${tag}_config2_rocprofv3_kernel_stats_bwtm.csv
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_c -- python3 $R/bench.py --no-cpu-baseline --no-verify --target off --steps 1 --warmup 0 --host-steps 1 > $R/gpurun_out/${tag}_bench_copy_trace.log 2>&1
k=$(find /tmp/prof_c -name "*kernel_trace.csv" | head -1); c=$(find /tmp/prof_c -name "*memory_copy_trace.csv" | head -1)
head -3 $c > $R/gpurun_out/${tag}_memory_copy_trace_head.csv
python3 - $k $c > $R/gpurun_out/${tag}_trace_inventory.txt <<'PY'
import csv, sys, collections
names = collections.Counter(); dur = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"][:60]; names[n] += 1; dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, c in names.most_common(40):
    print("kernel %-62s x %6d  %10.2f ms" % (n, c, dur[n] / 1e6))
dirs = collections.Counter(); ddur = collections.Counter()
for r in csv.DictReader(open(sys.argv[2])):
    d = r["Direction"]; dirs[d] += 1; ddur[d] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for d, c in dirs.items():
    print("copy %-40s x %6d  %10.2f ms" % (d, c, ddur[d] / 1e6))
PY
python3 $R/tools/copy_overlap.py $k $c > $R/gpurun_out/${tag}_host_to_host_overlap.md 2> $R/gpurun_out/${tag}_overlap_err.log
tail -2 $R/gpurun_out/${tag}_bench_kernel_trace.log | head -c 3000; cat $R/gpurun_out/${tag}_config2_summary.md $R/gpurun_out/${tag}_host_to_host_overlap.md; cat $R/gpurun_out/${tag}_overlap_err.log | tail -5
