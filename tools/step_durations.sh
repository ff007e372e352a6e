#!/bin/bash
# Per-launch durations of k_frontier_step over the last search of a bench run (rocprofv3 kernel trace).
# Usage: bash tools/step_durations.sh <tag> <bench args...>
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; shift; cd /tmp; rm -rf /tmp/prof_s
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-host --target off --steps 1 --warmup 0 "$@" > /tmp/prof_s.log 2>&1
k=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python3 - $k > $R/gpurun_out/${tag}_step_durations.txt <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])) if "bwtm::" in r["Kernel_Name"])
inits = [k for k, r in enumerate(rows) if "k_frontier_init" in r[2]]
steps = [r for r in rows[inits[-1]:] if "k_frontier_step" in r[2]]
print("launches", len(steps))
print(" ".join("%.2f" % ((e - s) / 1e6) for s, e, _ in steps))
PY
cat $R/gpurun_out/${tag}_step_durations.txt
