#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of `python3 bench.py ...` per bwtm kernel,
restricted to the TIMED steps of the bench (the input tooling launches the same kernels).

The timed region is recognised structurally: every merge step ends with the encoder
(k_enc_emit) followed by the sample builder's last entry (k_block_cum until round 4, k_block_cum32 since round 5); the last `steps` emits are
the timed steps -- not counting the `extra` untimed steps bench.py runs after them (since round 3: one step with every kernel
bracketed by events, for the per-kernel table).  Usage: summarize_kernel_trace.py kernel_trace.csv steps [extra = 1] > summary.md
"""
import csv
import re
import sys
from collections import defaultdict


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    extra = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "bwtm::" not in name:
            continue
        short = re.sub(r"<.*", "", name.split("bwtm::")[1].split("(")[0])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r))
    rows.sort()
    emits = [k for k, r in enumerate(rows) if r[2] == "k_enc_emit"]
    if extra > 0:
        emits = emits[:-extra]
    first = emits[-(steps + 1)] if len(emits) > steps else -1
    # the step before the timed ones ends with its own k_block_cum
    closers = ("k_block_cum", "k_block_cum32")
    t_begin = next(r[1] for r in rows[first:] if r[2] in closers) if first >= 0 else 0
    last_emit = emits[-1]
    t_end = next(r[1] for r in rows[last_emit:] if r[2] in closers)
    sel = [r for r in rows if r[0] >= t_begin and r[1] <= t_end]
    agg = defaultdict(list)
    for s, e, short, r in sel:
        agg[short].append(e - s)
    total = sum(sum(v) for v in agg.values())
    print("| kernel | launches in %d timed steps | avg per launch (ms) | total per step (ms) | share |" % steps)
    print("|---|---|---|---|---|")
    for short, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("| %s | %d | %.3f | %.3f | %.1f %% |" % (short, len(v), sum(v) / len(v) / 1e6, sum(v) / steps / 1e6, 100.0 * sum(v) / total))
    print("| all bwtm kernels | | | %.3f | 100 %% |" % (total / steps / 1e6))
    print()
    print("timed region: %.3f ms wall per step between first and last kernel" % ((t_end - t_begin) / steps / 1e6))
    # idle time of the device between consecutive kernels of the timed steps, by the kernel that follows the gap
    gaps = defaultdict(list)
    for prev, cur in zip(sel, sel[1:]):
        gaps[cur[2]].append(max(0, cur[0] - prev[1]))
    print()
    print("| gap before kernel | count | avg (us) | total per step (ms) |")
    print("|---|---|---|---|")
    for short, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print("| %s | %d | %.1f | %.3f |" % (short, len(v), sum(v) / len(v) / 1e3, sum(v) / steps / 1e6))
    print("| all gaps | | | %.3f |" % (sum(sum(v) for v in gaps.values()) / steps / 1e6))
    print()
    top = max(agg.items(), key=lambda kv: sum(kv[1]))[0]
    r = next(r for r in sel if r[2] == top)[3]
    print(top + " launch geometry: grid %s x workgroup %s, VGPR %s, SGPR %s, LDS %s B" %
          (r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")))


if __name__ == "__main__":
    main()
