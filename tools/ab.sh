#!/bin/bash
# Same-box comparison of library builds and knob settings on the default bench shape (config 2).  Every argument is
#   <label>[@<variant>][:<knob>=<value>[,<knob>=<value>...]]      variant = bwt-merge_amd/_variants/<variant>.so, default = the library in the tree
# Prints per run: ms per merge, the step kernel, the tail (everything else), the per-step scan launches and the largest tail kernels.
# Usage: bash tools/ab.sh [-r rounds] [-s steps] spec1 spec2 ...
rounds=2; steps=6
while [ "${1:0:1}" = "-" ]; do case $1 in -r) rounds=$2; shift 2;; -s) steps=$2; shift 2;; *) break;; esac; done
for round in $(seq $rounds); do
  for spec in "$@"; do
    label=${spec%%[@:]*}; rest=${spec#$label}
    variant=""; tunes=""
    if [ "${rest:0:1}" = "@" ]; then rest=${rest:1}; variant=${rest%%:*}; rest=${rest#$variant}; fi
    if [ "${rest:0:1}" = ":" ]; then tunes=${rest:1}; fi
    if [ -n "$variant" ]; then export BWTM_LIB=$PWD/bwt-merge_amd/_variants/$variant.so; else unset BWTM_LIB; fi
    targs=""; for kv in ${tunes//,/ }; do targs="$targs --tune $kv"; done
    timeout 300 python bench.py --steps $steps --warmup 2 --no-host --no-cpu-baseline --no-verify --target off $targs 2>/dev/null | tail -1 > gpurun_out/tmp_ab_$label.json
    python3 - $label <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tmp_ab_%s.json" % sys.argv[1]).read()); k = d["kernel_ms_per_step"]
tail = sum(v for n, v in k.items() if n != "frontier_step")
keys = ("build_recs", "block_len", "interleave", "enc_emit", "enc_size", "tile_build", "range_step", "scan_apply", "scan_reduce", "frontier_scan")
print("%-10s %7.2f ms  kernels %6.2f  tail %6.2f  step %6.2f  %s" % (sys.argv[1], d["ms_per_step"], tail + k.get("frontier_step", 0), tail, k.get("frontier_step", 0),
      " ".join("%s %.2f" % (n, k.get(n, 0)) for n in keys)), flush=True)
PY
  done
done
