#!/usr/bin/env python3
"""profiles/search_kernel_traffic.json from the per-kernel PMC table tools/pmc_passes.sh writes (pmc.txt).
HBM bytes of k_frontier_step per search: FETCH_SIZE (KiB, tallied at 64 B per read request) x 2 when the kernel's read requests
are 128-byte ones (TCC_EA0_RDREQ_128B ~ TCC_EA0_RDREQ; MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KiB).
Usage: make_traffic_json.py pmc.txt source-label reads_per_set read_length > search_kernel_traffic.json"""
import json
import sys

rows = {}
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) >= 8 and p[2] == "launches":
        rows[(p[0], p[1])] = (int(p[3]), float(p[5]), float(p[7]))
k = "k_frontier_step"
launches, fetch_kib, _ = rows[(k, "FETCH_SIZE")]
_, write_kib, _ = rows[(k, "WRITE_SIZE")]
rd = rows.get((k, "TCC_EA0_RDREQ_sum"), (0, 0.0, 0.0))[1]
rd32 = rows.get((k, "TCC_EA0_RDREQ_32B_sum"), (0, 0.0, 0.0))[1]
rd128 = rows.get((k, "TCC_EA0_RDREQ_128B_sum"), (0, None, 0.0))[1]
factor = 2.0
note = ("FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but this kernel's read requests are 128-byte ones (measured in round 2: TCC_EA0_RDREQ_128B = 99.7 % of "
        "TCC_EA0_RDREQ; the record loads did not change): reads = 2 x FETCH_SIZE (the correction of MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported.")
if rd128 is not None and rd > 0:
    factor = (128.0 * rd128 + 64.0 * (rd - rd128 - rd32) + 32.0 * rd32) / (64.0 * rd)
    note = "reads = 128 B x TCC_EA0_RDREQ_128B + 64 B x the other requests (measured in this set of passes) = %.3f x FETCH_SIZE; WRITE_SIZE as reported." % factor
total = fetch_kib * 1024.0 * factor + write_kib * 1024.0
out = {"kernel": "frontier_step", "config": {"reads_per_set": int(sys.argv[3]), "read_length": int(sys.argv[4]), "n_gpus": 1},
       "launches_per_search": launches, "FETCH_SIZE_KiB_per_search": fetch_kib, "WRITE_SIZE_KiB_per_search": write_kib,
       "TCC_EA0_RDREQ_per_search": rd, "TCC_EA0_RDREQ_32B_per_search": rd32, "TCC_EA0_RDREQ_128B_per_search": rd128,
       "hbm_bytes_per_launch": total / launches, "hbm_bytes_per_search": total, "note": note, "source": sys.argv[2]}
print(json.dumps(out, indent=1))
