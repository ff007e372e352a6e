#!/bin/bash
# Per-kernel ISA metadata (VGPRs, SGPRs, scratch, occupancy, LDS) of the library's device code, compiled here (no GPU needed).
# Usage: bash tools/isa_meta.sh [kernel-name-substring ...]      (the ISA text stays in /tmp/isa/bwtm_api.s)
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only $EXTRA -o /tmp/isa/bwtm_api.s $root/bwt-merge_amd/csrc/bwtm_api.hip 2>/dev/null
python3 - "$@" <<'PY'
import re, sys
cur, meta = None, {}
for line in open("/tmp/isa/bwtm_api.s"):
    m = re.match(r"^(_ZN4bwtm\w+):", line)
    if m: cur = m.group(1); meta[cur] = {}; continue
    m = re.match(r"; (ScratchSize|NumVgprs|NumSgprs|Occupancy|LDSByteSize): (\d+)", line)
    if m and cur: meta[cur].setdefault(m.group(1), int(m.group(2)))
for k, v in meta.items():
    if not sys.argv[1:] or any(a in k for a in sys.argv[1:]):
        print("%-90s %s" % (k[:90], v))
PY
