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


def main():  # noqa: C901
    kpath, cpath = sys.argv[1], sys.argv[2]
    all_kernels = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kpath))]
    kernels = [k for k in all_kernels if "bwtm::" in k[2]]
    copies = []
    # device-to-host transfers into page-locked memory may be executed by the runtime's copy kernels instead of the DMA engines:
    # they show up in the kernel trace, not in the memory-copy trace
    for s0, e0, n in all_kernels:
        if "copyBuffer" in n or "CopyBuffer" in n:
            copies.append((s0, e0, "BLIT_KERNEL_DEVICE_TO_HOST", int((e0 - s0) * 57.0)))
    for r in csv.DictReader(open(cpath)):
        direction = r.get("Direction", r.get("direction", ""))
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        # this rocprofv3 release reports no size: estimate it from the duration at the link's ~57 GB/s
        size = int(r.get("Size", 0) or 0) or int((e0 - s0) * 57.0)
        copies.append((s0, e0, direction, size))
    big = [c for c in copies if c[1] - c[0] >= 200000]          # >= 0.2 ms: the chunked transfers, not the small result fetches
    h2d = [c for c in big if "HOST_TO_DEVICE" in c[2].upper() or c[2].upper().startswith("H2D")]
    d2h = [c for c in big if "DEVICE_TO_HOST" in c[2].upper() or c[2].upper().startswith("D2H")]
    inits = sorted(s0 for s0, e0, n in kernels if "k_frontier_init" in n)
    for label, which in (("the last bwtm_merge_host call that downloads the samples (third search from the end: bench.py runs two data-only calls after it)", -3),
                         ("the last call (data only)", -1)):
        if len(inits) < -which:
            continue
        t_search = inits[which]
        t_next = (inits[which + 1] if which + 1 < 0 else float("inf"))
        # its upload: the H2D chunks before the search, walking back while the gaps stay below 20 ms
        ups = sorted([c for c in h2d if c[1] <= t_search], key=lambda c: c[0])
        mine = []
        for c in reversed(ups):
            if mine and mine[-1][0] - c[1] > 2e7:
                break
            mine.append(c)
        mine.reverse()
        # its download: the D2H chunks after the search and before the next merge's upload
        next_up = min((c[0] for c in h2d if c[0] > t_search), default=float("inf"))
        downs = [c for c in d2h if c[0] >= t_search and c[0] < min(next_up, t_next)]
        if not mine or not downs:
            continue
        t0, t1 = mine[0][0], max(c[1] for c in downs)
        ks = union([(s0, e0) for s0, e0, _ in kernels if e0 > t0 and s0 < t1])
        us = union([(c[0], c[1]) for c in mine]); ds = union([(c[0], c[1]) for c in downs])
        ms = 1e-6
        print("### " + label)
        print()
        print("| quantity | value |")
        print("|---|---|")
        print("| window (first H2D chunk .. last D2H chunk) | %.1f ms |" % ((t1 - t0) * ms))
        print("| H2D: %d chunks | link busy %.1f ms |" % (len(mine), total(us) * ms))
        print("| D2H: %d chunks | link busy %.1f ms |" % (len(downs), total(ds) * ms))
        print("| kernels busy inside the window | %.1f ms |" % (total(ks) * ms))
        print("| kernels running while an H2D copy is in flight | %.1f ms |" % (intersect(ks, us) * ms))
        print("| kernels running while a D2H copy is in flight | %.1f ms |" % (intersect(ks, ds) * ms))
        print("| neither copying nor computing | %.1f ms |" % ((t1 - t0 - total(union(ks + us + ds))) * ms))
        print("| last H2D chunk done -> first kernel of the search | %.2f ms |" % ((t_search - mine[-1][1]) * ms))
        first_down = min(c[0] for c in downs)
        emits = [e0 for s0, e0, n in kernels if "k_enc_emit" in n and s0 >= t_search and s0 < t1]
        if emits:
            print("| first D2H chunk starts %.2f ms after the first k_enc_emit range ends; last k_enc_emit ends %.1f ms before the last D2H chunk does | |" %
                  ((first_down - min(emits)) * ms, (t1 - max(emits)) * ms))
        print()


if __name__ == "__main__":
    main()
