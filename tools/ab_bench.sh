#!/bin/bash
# A/B on ONE box (boxes of the pool differ by 8 % on the search kernel): alternates bench.py between bwt-merge_amd/_variants/base.so
# (tools/build_variant.sh base "" on the commit to compare with) and the library in the tree.  Usage: bash tools/ab_bench.sh [kernel ...]
keys=${@:-frontier_step}
for v in base cur base cur base cur; do
  if [ $v = base ]; then export BWTM_LIB=$PWD/bwt-merge_amd/_variants/base.so; else unset BWTM_LIB; fi
  python bench.py --steps 10 --warmup 2 --no-host --no-cpu-baseline --no-verify --target off 2>/dev/null | tail -1 > gpurun_out/tmp_ab_$v.json
  python3 - $v $keys <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tmp_ab_%s.json" % sys.argv[1]).read()); k = d["kernel_ms_per_step"]
print(sys.argv[1], d["ms_per_step"], {x: k.get(x) for x in sys.argv[2:]})
PY
done
