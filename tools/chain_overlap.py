#!/usr/bin/env python3
"""Copy / kernel overlap of the LAST round of bench.py's pipelined host chain (--chain N) from a rocprofv3
--kernel-trace --memory-copy-trace run.  Usage: chain_overlap.py kernel_trace.csv memory_copy_trace.csv merges > summary.md
The last round = the last `merges` searches (each starts with a k_range_init / k_frontier_init launch)."""
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from copy_overlap import intersect, total, union


def main():
    kpath, cpath, merges = sys.argv[1], sys.argv[2], int(sys.argv[3])
    all_kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kpath))]
    kernels = [k for k in all_kernels if "bwtm::" in k[2]]
    copies = [(s, e, "DEVICE_TO_HOST") for s, e, n in all_kernels if "copyBuffer" in n or "CopyBuffer" in n]
    for r in csv.DictReader(open(cpath)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")))
    big = [c for c in copies if c[1] - c[0] >= 100000]
    h2d = sorted(c for c in big if "HOST_TO_DEVICE" in c[2].upper())
    d2h = sorted(c for c in big if "DEVICE_TO_HOST" in c[2].upper())
    starts = sorted(s for s, e, n in kernels if "k_range_init" in n)
    if len(starts) < merges:
        starts = sorted(s for s, e, n in kernels if "k_frontier_init" in n)
    mine = starts[-merges:]
    t_first = mine[0]
    # the round's own uploads: H2D chunks before its first search, walking back while the gaps stay below 20 ms
    ups = [c for c in h2d if c[1] <= t_first]
    first_up = t_first
    for c in reversed(ups):
        if first_up - c[1] > 2e7:
            break
        first_up = c[0]
    t0 = first_up
    t1 = max([c[1] for c in d2h if c[0] >= t_first] + [k[1] for k in kernels if k[0] >= t_first])
    ks = union([(s, e) for s, e, _ in kernels if e > t0 and s < t1])
    us = union([(c[0], c[1]) for c in h2d if c[1] > t0 and c[0] < t1])
    ds = union([(c[0], c[1]) for c in d2h if c[1] > t0 and c[0] < t1])
    after = union([(c[0], c[1]) for c in h2d if c[0] >= t_first and c[0] < t1])       # the announced increments: H2D after the first search began
    ms = lambda x: x / 1e6
    print("### the last round of the pipelined chain (%d merges)\n" % merges)
    print("| quantity | value |\n|---|---|")
    print("| window (first H2D chunk .. last D2H chunk / kernel) | %.1f ms |" % ms(t1 - t0))
    print("| H2D link busy | %.1f ms (%d chunks) |" % (ms(total(us)), len(us)))
    print("| ... of which after the first search began (the announced increments) | %.1f ms |" % ms(total(after)))
    print("| ... of which under a running kernel | %.1f ms |" % ms(intersect(after, ks)))
    print("| D2H link busy | %.1f ms |" % ms(total(ds)))
    print("| kernels busy | %.1f ms |" % ms(total(ks)))
    both = union([list(x) for x in us] + [list(x) for x in ds] + [list(x) for x in ks])
    print("| neither copying nor computing | %.1f ms |" % ms((t1 - t0) - total(both)))
    for k, s in enumerate(mine):
        pending = [c for c in h2d if c[0] >= s and (k + 1 == len(mine) or c[0] < mine[k + 1])]
        print("| search %d starts at %.1f ms; H2D chunks that start during it: %d (%.1f ms) |  |" % (k + 1, ms(s - t0), len(pending), ms(sum(c[1] - c[0] for c in pending))))


if __name__ == "__main__":
    main()
