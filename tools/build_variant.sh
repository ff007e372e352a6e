#!/bin/bash
# Builds bwt-merge_amd/_variants/<name>.so from the current sources with one sed edit applied to the kernel
# headers (csrc/kernels/*.hip.h) (A/B measurements: `BWTM_LIB=<path> python bench.py ...`).  The sources are restored.
# Usage: build_variant.sh <name> <sed expression | ""> [extra hipcc flags, e.g. -DBWTM_DIAGNOSTICS for the timing-only kernels
#        and bwtm_tune keys that tools/walk_experiments.py uses]
set -e
name=$1; expr=$2; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/bwt-merge_amd/csrc
mkdir -p $root/bwt-merge_amd/_variants
rm -rf /tmp/bwtm_kernels.orig && cp -r $src/kernels /tmp/bwtm_kernels.orig
trap 'cp /tmp/bwtm_kernels.orig/*.hip.h $src/kernels/' EXIT
if [ -n "$expr" ]; then sed -i "$expr" $src/kernels/*.hip.h; fi
(cd $src && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread "$@" -o $root/bwt-merge_amd/_variants/$name.so bwtm_api.hip)
