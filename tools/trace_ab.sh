#!/bin/bash
# rocprofv3 kernel traces of bench.py (config 2) under different builds / knob settings on one box: per-kernel table, wall time between the
# first and the last kernel of a step, and the gap table (profiles/summarize_kernel_trace.py).
# Usage: bash tools/trace_ab.sh <tag> <label>[@<variant>][:knob=value,...] ...        (variant = bwt-merge_amd/_variants/<variant>.so)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; shift
for spec in "$@"; do
  label=${spec%%[@:]*}; rest=${spec#$label}
  variant=""; tunes=""
  if [ "${rest:0:1}" = "@" ]; then rest=${rest:1}; variant=${rest%%:*}; rest=${rest#$variant}; fi
  if [ "${rest:0:1}" = ":" ]; then tunes=${rest:1}; fi
  if [ -n "$variant" ]; then export BWTM_LIB=$R/bwt-merge_amd/_variants/$variant.so; else unset BWTM_LIB; fi
  targs=""; for kv in ${tunes//,/ }; do targs="$targs --tune $kv"; done
  cd /tmp; rm -rf /tmp/prof_$label
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-host --target off --steps 5 --warmup 1 $targs > $R/gpurun_out/${tag}_${label}_bench.log 2>&1
  f=$(find /tmp/prof_$label -name "*kernel_trace.csv" | head -1)
  python3 $R/profiles/summarize_kernel_trace.py $f 5 > $R/gpurun_out/${tag}_${label}_summary.md
  grep -E "timed region|all gaps|all bwtm kernels|k_frontier_scan|k_scan_reduce|k_frontier_step" $R/gpurun_out/${tag}_${label}_summary.md | sed "s/^/$label: /"
done
