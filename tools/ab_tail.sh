#!/bin/bash
# Same-box comparison of builds of the library on the TAIL kernels (everything but the search): rounds over bwt-merge_amd/_variants/<name>.so
# ("cur" = the library in the tree).  Prints ms per merge, the tail's sum and its kernels.  Usage: bash tools/ab_tail.sh [-r rounds] name1 name2 ...
rounds=2; if [ "$1" = "-r" ]; then rounds=$2; shift; shift; fi
for round in $(seq $rounds); do
  for v in "$@"; do
    if [ $v = cur ]; then unset BWTM_LIB; else export BWTM_LIB=$PWD/bwt-merge_amd/_variants/$v.so; fi
    timeout 300 python bench.py --steps 6 --warmup 2 --no-host --no-cpu-baseline --no-verify --target off 2>/dev/null | tail -1 > gpurun_out/tmp_tail_$v.json
    python3 - $v <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tmp_tail_%s.json" % sys.argv[1]).read()); k = d["kernel_ms_per_step"]
tail = sum(v for n, v in k.items() if n != "frontier_step")
keys = ("build_recs", "block_len", "interleave", "enc_emit", "enc_size", "enc_lasthead", "block_cum", "tile_build")
print("%-8s %7.2f ms  tail %6.2f  step %6.2f  %s" % (sys.argv[1], d["ms_per_step"], tail, k.get("frontier_step", 0), " ".join("%s %.2f" % (n, k.get(n, 0)) for n in keys)), flush=True)
PY
  done
done
