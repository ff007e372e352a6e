#!/bin/bash
# Same-box comparison of several builds of the library (bwt-merge_amd/_variants/<name>.so, tools/build_variant.sh): two rounds over all of them.
# Timing only: variants may compute wrong results (--no-verify).  Usage: bash tools/ab_variants.sh name1 name2 ... [-- bench args]
names=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done; [ "$1" = "--" ] && shift
for round in 1 2; do
  for v in "${names[@]}"; do
    BWTM_LIB=$PWD/bwt-merge_amd/_variants/$v.so timeout 120 python bench.py --steps 6 --warmup 2 --no-host --no-cpu-baseline --no-verify --target off "$@" 2>/dev/null | tail -1 > gpurun_out/tmp_var_$v.json
    python3 - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tmp_var_%s.json" % sys.argv[1]).read()); k = d["kernel_ms_per_step"]
print(sys.argv[1], d["ms_per_step"], {x: k.get(x) for x in ("frontier_step", "build_recs", "enc_emit", "interleave")}, d["roofline"]["launches_per_step"])
PY
  done
done
