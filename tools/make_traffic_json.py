#!/usr/bin/env python3
"""Adds / replaces one entry of profiles/search_kernel_traffic.json from the per-kernel PMC table tools/pmc_passes.sh writes (pmc.txt).
HBM bytes of k_frontier_step per search: FETCH_SIZE (KiB, tallied at 64 B per read request) corrected by the measured share of 128-byte
requests (TCC_EA0_RDREQ_128B; MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (KiB).
Every entry carries what bench.py checks before it uses the bytes: a hash of the search kernels' sources, the knobs of the run, the
launches and the LF steps of the profiled search.
Usage: make_traffic_json.py pmc.txt source-label reads_per_set read_length lf_steps_per_search [KEY=VALUE knobs ...]
       (rewrites profiles/search_kernel_traffic.json in place; lf_steps_per_search = roofline.lf_steps_of_this_kernel_per_step of the
       bench line of the same configuration, 0 = unknown)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import search_code_hash, SEARCH_KERNEL_SOURCES

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
note = ("FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but this kernel's read requests are 128-byte ones: reads = 2 x FETCH_SIZE "
        "(the correction of MI355X_MICROARCH.md, HBM section; TCC_EA0_RDREQ_128B not collected in this set); WRITE_SIZE as reported.")
if rd128 is not None and rd > 0:
    factor = (128.0 * rd128 + 64.0 * (rd - rd128 - rd32) + 32.0 * rd32) / (64.0 * rd)
    note = "reads = 128 B x TCC_EA0_RDREQ_128B + 64 B x the other requests (measured in this set of passes) = %.3f x FETCH_SIZE; WRITE_SIZE as reported." % factor
total = fetch_kib * 1024.0 * factor + write_kib * 1024.0
lf_steps = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
entry = {"kernel": "frontier_step", "config": {"reads_per_set": int(sys.argv[3]), "read_length": int(sys.argv[4]), "n_gpus": 1},
         "code_hash": search_code_hash(), "code_hash_of": list(SEARCH_KERNEL_SOURCES), "tune": sorted(sys.argv[6:]),
         "launches_per_search": launches, "FETCH_SIZE_KiB_per_search": fetch_kib, "WRITE_SIZE_KiB_per_search": write_kib,
         "TCC_EA0_RDREQ_per_search": rd, "TCC_EA0_RDREQ_32B_per_search": rd32, "TCC_EA0_RDREQ_128B_per_search": rd128,
         "hbm_bytes_per_launch": total / launches, "hbm_bytes_per_search": total, "note": note, "source": sys.argv[2]}
if lf_steps > 0:
    entry["lf_steps_per_search"] = lf_steps
path = os.path.join(ROOT, "profiles", "search_kernel_traffic.json")
try:
    stored = json.load(open(path))
    entries = stored.get("entries", [])
except (OSError, ValueError):
    entries = []
entries = [e for e in entries if e.get("config") != entry["config"] or e.get("kernel") != entry["kernel"] or e.get("tune", []) != entry["tune"]]
entries.append(entry)
entries.sort(key=lambda e: (e["config"]["reads_per_set"], e["config"]["read_length"]))
json.dump({"what": "HBM bytes of the dominant search kernel per configuration, from separate rocprofv3 --pmc passes (tools/pmc_passes.sh); "
                   "bench.py uses an entry only when code hash, knobs, launches and LF steps match the run",
           "entries": entries}, open(path, "w"), indent=1)
print("entry for %s: %.3f GB per launch over %d launches" % (entry["config"], total / launches / 1e9, launches))
