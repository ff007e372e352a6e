#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection.csv of `bench.py --steps 1 --warmup 1` per bwtm kernel and
counter over the launches of the LAST merge step (the timed one): for every kernel the last `n` launches, where n
is the number of launches of that kernel in one step.  Usage: pmc_aggregate.py counter_collection.csv
       pmc_aggregate.py counter_collection.csv --last N --kernel NAME   (only the last N launches of one kernel: passes restricted to it)"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bwtm::" in r["Kernel_Name"]]
if "--last" in sys.argv:
    last_n = int(sys.argv[sys.argv.index("--last") + 1]); only = sys.argv[sys.argv.index("--kernel") + 1]
    per = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("bwtm::")[1].split("(")[0].split("<")[0]
        if name == only:
            per[r["Counter_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for counter, v in sorted(per.items()):
        sel = sorted(v)[-last_n:]
        print("%-26s %-26s launches %4d  sum %.6g  kernel_ms %.3f" % (only, counter, len(sel), sum(x for _, x, _ in sel), sum(d for _, _, d in sel) / 1e6))
    sys.exit(0)
per = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("bwtm::")[1].split("(")[0].split("<")[0]
    per[(name, r["Counter_Name"])].append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
# the last merge step starts with the k_block_len launches of its two inputs (the output's block starts come
# from the encoder) -> use the start of the second-from-last k_block_len launch
starts = sorted(set(t for (n, c), v in per.items() if n == "k_block_len" for t, _, _ in v))
t0 = starts[-2] if len(starts) >= 2 else 0
for (name, counter), v in sorted(per.items()):
    sel = [(val, dur) for t, val, dur in v if t >= t0]
    if sel:
        print("%-26s %-26s launches %4d  sum %.6g  kernel_ms %.3f" % (name, counter, len(sel), sum(x for x, _ in sel), sum(d for _, d in sel) / 1e6))
