#!/bin/bash
# What a SCALE session on a multi-GPU node runs to compare the multi-GPU designs: the C++ tool on the
# same two inputs with 1, 2, 4, 8 GPUs: with partitioned records (product default: bwt_merge -g ... -P), with sequence blocks (product: -B) and with the
# sliced frontier search (experimental build: bwt_merge_experimental -g ... -S).  Every run prints the phases of mergeMultiGPU() on stderr (upload / search /
# exchange / interleave + encode / download, exchanged bytes per GPU; built with -DVERBOSE_STATUS_INFO like the reference); this script
# collects those lines into gpurun_out/scale_cli.txt.  The inputs are two synthetic read sets written as native files by the tool itself.
# Usage: bash tools/scale_cli.sh [reads per set = 50000000] [max gpus = 8]
set -e
reads=${1:-50000000}; maxg=${2:-8}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; H=$R/bwt-merge_amd/csrc/host; W=${TMPDIR:-/tmp}/scale_cli; mkdir -p $W $R/gpurun_out
make -C $H -s; make -C $H -s experimental
cd $R
python3 - $reads $W <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, _pkg
pkg = _pkg.load(); pkg.init(0)
from bwt_merge_amd import synth
n, w = int(sys.argv[1]), sys.argv[2]
chars = np.frombuffer(b"$ACGTN", dtype=np.uint8)
for k, seed in enumerate((1001, 1002)):
    ix = synth.build_index(pkg, seed, n, 100, device=torch.device("cuda", 0))
    chars[ix.extract(0, ix.bases)].tofile(os.path.join(w, "in%d.plain" % k))
    ix.free()
PY
for k in 0 1; do $H/bwt_convert -i plain_default -o native $W/in$k.plain $W/in$k.native > /dev/null; rm -f $W/in$k.plain; done
: > $R/gpurun_out/scale_cli.txt
ngpu=$(python3 -c "import torch; print(torch.cuda.device_count())")
# BWTM_SCALE_SAME_DEVICE=1: the "GPUs" are contexts of GPU 0 (a one-GPU box: the times say nothing about scaling, the runs check every mode at size)
for g in 1 2 4 8; do
  if [ $g -gt $maxg ]; then break; fi
  if [ -z "$BWTM_SCALE_SAME_DEVICE" ] && [ $g -gt $ngpu ]; then break; fi
  if [ -n "$BWTM_SCALE_SAME_DEVICE" ]; then list=$(printf '0,%.0s' $(seq $g)); list=${list%,}; else list=$(seq -s, 0 $((g-1))); fi
  for mode in blocks sliced partitioned; do
    exe=$H/bwt_merge; extra="-B"
    if [ $g -lt 2 ]; then extra=""; if [ $mode != blocks ]; then continue; fi; fi
    if [ $mode = sliced ]; then exe=$H/bwt_merge_experimental; extra="-S"; fi
    if [ $mode = partitioned ]; then extra="-P"; fi
    for round in 1 2; do                                   # the first run of a device list pays ncclCommInitAll and the pools' first allocations
      echo "== $g GPU(s), $mode, run $round" | tee -a $R/gpurun_out/scale_cli.txt
      $exe -g $list $extra -i native $W/in0.native $W/in1.native $W/out.native 2>&1 | grep -E "mergeMultiGPU|BWTs merged|Total time|rror" | tee -a $R/gpurun_out/scale_cli.txt || true
      echo "   result: $(md5sum < $W/out.native | cut -c1-32)" | tee -a $R/gpurun_out/scale_cli.txt      # every mode and device count must write the same file
    done
  done
done
rm -f $W/out.native
