#!/bin/bash
# Builds bwt-merge_amd/_variants/<name>.so from the current sources with one sed edit applied to
# bwtm_kernels.hip.h (A/B measurements: `BWTM_LIB=<path> python bench.py ...`).  The sources are restored.
set -e
name=$1; expr=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/bwt-merge_amd/csrc
mkdir -p $root/bwt-merge_amd/_variants
cp $src/bwtm_kernels.hip.h /tmp/bwtm_kernels.hip.h.orig
trap 'cp /tmp/bwtm_kernels.hip.h.orig $src/bwtm_kernels.hip.h' EXIT
sed -i "$expr" $src/bwtm_kernels.hip.h
(cd $src && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o $root/bwt-merge_amd/_variants/$name.so bwtm_api.hip)
