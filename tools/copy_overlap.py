#!/usr/bin/env python3
"""Copy / kernel overlap of the host-to-host merge from a rocprofv3 --kernel-trace --memory-copy-trace run of bench.py.
Usage: copy_overlap.py kernel_trace.csv memory_copy_trace.csv > summary.md
Looks at the LAST bwtm_merge_host call with samples (bench.py runs the data-only calls after it): its window starts at the
first large host-to-device copy after the previous download and ends at its last device-to-host copy."""
import csv
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    acc = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            acc += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return acc


def main():
    kpath, cpath = sys.argv[1], sys.argv[2]
    kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kpath)) if "bwtm::" in r["Kernel_Name"]]
    copies = []
    for r in csv.DictReader(open(cpath)):
        direction = r.get("Direction", r.get("direction", ""))
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        # this rocprofv3 release reports no size: estimate it from the duration at the link's ~57 GB/s
        size = int(r.get("Size", 0) or 0) or int((e0 - s0) * 57.0)
        copies.append((s0, e0, direction, size))
    big = [c for c in copies if c[1] - c[0] >= 200000]          # >= 0.2 ms: the chunked transfers, not the small result fetches
    h2d = [c for c in big if "HOST_TO_DEVICE" in c[2].upper() or c[2].upper().startswith("H2D")]
    d2h = [c for c in big if "DEVICE_TO_HOST" in c[2].upper() or c[2].upper().startswith("D2H")]
    # merges = groups of H2D chunks separated by D2H activity
    events = sorted([(c[0], "u", c) for c in h2d] + [(c[0], "d", c) for c in d2h])
    groups, cur = [], None
    for t, kind, c in events:
        if kind == "u":
            if cur is None or cur["d"]:
                cur = {"u": [], "d": []}; groups.append(cur)
            cur["u"].append(c)
        elif cur is not None:
            cur["d"].append(c)
    groups = [g for g in groups if g["u"] and g["d"]]
    with_samples = [g for g in groups if sum(c[1] - c[0] for c in g["d"]) > 1.3 * sum(c[1] - c[0] for c in g["u"])]
    g = (with_samples or groups)[-1]
    t0, t1 = g["u"][0][0], max(c[1] for c in g["d"])
    ks = union([(s, e) for s, e, _ in kernels if e > t0 and s < t1])
    us = union([(c[0], c[1]) for c in g["u"]]); ds = union([(c[0], c[1]) for c in g["d"]])
    up_bytes = sum(c[3] for c in g["u"]); down_bytes = sum(c[3] for c in g["d"])
    ms = 1e-6
    print("| quantity | value |")
    print("|---|---|")
    print("| window of one bwtm_merge_host call (first H2D chunk .. last D2H chunk) | %.1f ms |" % ((t1 - t0) * ms))
    print("| H2D: %d chunks | link busy %.1f ms |" % (len(g["u"]), total(us) * ms))
    print("| D2H: %d chunks | link busy %.1f ms |" % (len(g["d"]), total(ds) * ms))
    print("| kernels busy inside the window | %.1f ms |" % (total(ks) * ms))
    print("| kernels running while an H2D copy is in flight | %.1f ms |" % (intersect(ks, us) * ms))
    print("| kernels running while a D2H copy is in flight | %.1f ms |" % (intersect(ks, ds) * ms))
    print("| neither copying nor computing | %.1f ms |" % ((t1 - t0 - total(union(ks + us + ds))) * ms))
    first_search = min((s for s, e, n in kernels if s >= t0 and "k_frontier" in n), default=None)
    if first_search:
        print("| last H2D chunk done -> first search kernel | %.1f ms |" % ((first_search - max(c[1] for c in g["u"])) * ms))


if __name__ == "__main__":
    main()
