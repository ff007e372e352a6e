#!/usr/bin/env python3
"""What a GPU of a G-GPU machine would do in the partitioned-records search, measured on ONE GPU: G contexts
stand in for G GPUs, each holds its windows of the two indexes' records and advances the elements of its position range; the contexts
run one after the other, so the time of part g IS the time GPU g would need for its share (the exchange itself -- peer reads over
xGMI -- is a device-local read here and is not what this measures).

Compared with: the whole search on one GPU without the node phase (range_ratio = 0: the prototype advances elements from the roots on),
and the sequence-sharded search a GPU runs today (1 / G of the sequences over ALL records).

Needs the experimental library:  BWTM_LIB=$PWD/bwt-merge_amd/libbwtm_experimental.so python tools/partitioned_scale.py [--reads N] [--parts 2,4,8]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--readlen", type=int, default=100)
    ap.add_argument("--parts", default="2,4,8")
    ap.add_argument("--k", type=int, default=5, help="cuts are chosen among the 5^k k-mers")
    ap.add_argument("--emit-budget", type=int, default=2 << 30)
    ap.add_argument("--node-ratio", type=int, default=8, help="the first levels on trie nodes while a level has at most sequences / this many nodes (0: elements from the roots on)")
    ap.add_argument("--slack", type=float, default=1.25, help="with the node phase: a GPU's frontier buffers hold slack x sequences / G elements")
    args = ap.parse_args()
    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd import synth
    from bwt_merge_amd import experimental as X
    assert X.loaded(), "start with BWTM_LIB=<libbwtm_experimental.so>"
    torch.cuda.set_device(0)
    pkg.init(0)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    host, meta = [], []
    for seed in (1001, 1002):
        ix = synth.build_index(pkg, seed, args.reads, args.readlen, device=dev)
        ix.encode()
        hb = pkg.HostBuffer(ix.nbytes)
        ix.download_into(hb.array)
        meta.append((ix.sequences, ix.bases, ix.nbytes))
        ix.free(); host.append(hb)
    torch.cuda.empty_cache(); pkg.trim()
    print("inputs: 2 x %d reads of %d bp (%.2f + %.2f Gbase) in %.0f s" % (args.reads, args.readlen, meta[0][1] / 1e9, meta[1][1] / 1e9, time.time() - t0), flush=True)
    pkg.tune("emit_budget", args.emit_budget)

    def upload():
        return (pkg.Index.upload(host[0].array[: meta[0][2]], meta[0][0], meta[0][1]), pkg.Index.upload(host[1].array[: meta[1][2]], meta[1][0], meta[1][1]))

    def profiled(fn):
        pkg.profile_only(None); pkg.profile_reset(); pkg.profile_enable(True)
        pkg.synchronize(); t = time.perf_counter()
        out = fn()
        pkg.synchronize(); wall = (time.perf_counter() - t) * 1e3
        prof = pkg.profile_read(); pkg.profile_enable(False)
        return out, wall, prof

    def ms(prof, *names):
        return sum(prof.get(n, (0.0, 0))[0] for n in names)

    # ---- the whole search on one GPU: as shipped (node phase), and element-only like the prototype
    A, B = upload()
    m_b = meta[1][0]
    rows = []
    for label, ratio in (("one GPU, as shipped (node phase)", -1), ("one GPU, elements from the roots on", 0)):
        pkg.tune("range_ratio", ratio)
        ra = pkg.RankArray(A, B)
        ra.search(A, B, 0, m_b - 1); pkg.synchronize(); ra.free()               # warm-up (pool)
        ra = pkg.RankArray(A, B)
        _, wall, prof = profiled(lambda: ra.search(A, B, 0, m_b - 1))
        rows.append((label, wall, ms(prof, "frontier_step"), prof.get("frontier_step", (0, 0))[1], sum(v[0] for v in prof.values())))
        if ratio == 0:
            whole_bits = ra                                                     # kept: the parts' union must equal it
        else:
            ra.free()
    print("\n| search of input2's %d sequences | wall (ms) | k_frontier_step (ms) | launches | all kernels (ms) |\n|---|---|---|---|---|" % m_b)
    for r in rows:
        print("| %s | %.1f | %.1f | %d | %.1f |" % r)
    # ---- what one GPU of G does today: 1 / G of the sequences over all records
    print("\n| sequence shard of 1 / G (what a GPU of G runs today), as shipped | wall (ms) | k_frontier_step (ms) | x of the whole search / G | all kernels (ms) |\n|---|---|---|---|---|")
    pkg.tune("range_ratio", -1)
    parts_list = [int(x) for x in args.parts.split(",")]
    for G in parts_list:
        last = (m_b + G - 1) // G - 1
        ra = pkg.RankArray(A, B)
        _, wall, prof = profiled(lambda: ra.search(A, B, 0, last))
        ra.free()
        print("| G = %d | %.1f | %.1f | %.2f | %.1f |" % (G, wall, ms(prof, "frontier_step"), ms(prof, "frontier_step") / (rows[0][2] / G), sum(v[0] for v in prof.values())), flush=True)
    whole_bytes = X.index_record_bytes(A) + X.index_record_bytes(B)

    def node_capacity(G):
        if args.node_ratio == 0 or G == 1:
            return None
        return int(args.slack * min(5 * (m_b // args.node_ratio), m_b) / G) + 65536

    def capacity(G):
        return None if args.node_ratio == 0 or G == 1 else int(args.slack * m_b / G) + 65536

    # ---- partitioned records
    print("\n| partitioned records | part | window (MB) | elements advanced (share) | k_frontier_step (ms) | compact (ms) | gather (ms) | cut search (ms) | scans, tables, tiles (ms) | all kernels (ms) |\n|---|---|---|---|---|---|---|---|---|---|")
    summary = []
    for G in parts_list:
        I, R = X.partition_cuts(A, B, G, args.k)
        ctxs = [pkg.Context(0) for _ in range(G)]
        ras = []
        windows = []
        for g in range(G):
            ctxs[g].make_current()
            a2, b2 = upload()
            wa, wb = X.index_window(a2, I[g], I[g + 1]), X.index_window(b2, R[g], R[g + 1])
            a2.free(); b2.free(); pkg.trim()
            windows.append((wa, wb)); ras.append(pkg.RankArray(wa, wb))
            pkg.profile_only(None); pkg.profile_reset(); pkg.profile_enable(True)
        t = time.perf_counter()
        steps, levels, largest, work = X.search_partitioned(pkg, windows, ras, m_b, R, lambda g: ctxs[g].make_current(), capacity=capacity(G), node_ratio=args.node_ratio, node_capacity=node_capacity(G))
        wall = (time.perf_counter() - t) * 1e3
        per, alls = [], []
        for g in range(G):
            ctxs[g].make_current()
            prof = pkg.profile_read(); pkg.profile_enable(False)
            step_ms = ms(prof, "frontier_step")
            per.append(step_ms)
            all_ms = sum(v[0] for v in prof.values())
            alls.append(all_ms)
            print("| G = %d | %d | %.0f | %d (%.3f) | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f |" % (G, g, (X.index_record_bytes(windows[g][0]) + X.index_record_bytes(windows[g][1])) / 1e6,
                  work[g], work[g] / max(1, sum(work)), step_ms, ms(prof, "compact_outputs"), ms(prof, "frontier_gather"), ms(prof, "cut_counts"),
                  all_ms - step_ms - ms(prof, "compact_outputs", "frontier_gather", "cut_counts"), all_ms), flush=True)
        # parity: the union of the parts' bits is the whole search's bitvector (the parts' sets are disjoint: n_b bits in all)
        pkg.make_default_current()
        acc = pkg.RankArray(A, B)
        for g in range(G):
            acc.or_from(ras[g])
        ones, outside = acc.subset_check(whole_bits)
        same = (ones == meta[1][1] and outside == 0)
        acc.free()
        summary.append((G, "%d + %d node levels" % (steps, levels), max(per), sum(per), (rows[0][2], rows[1][2]), largest, same, max(alls)))
        for g in range(G):
            ctxs[g].make_current()
            ras[g].free(); windows[g][0].free(); windows[g][1].free(); pkg.trim()
        pkg.make_default_current()
        for c in ctxs:
            c.destroy()
    print("\n| G | LF steps | slowest part's k_frontier_step (ms) | all parts together (ms) | the whole search on one GPU: as shipped / elements only (ms) | speed-up of the step kernel = whole as shipped / slowest part | largest frontier a part held | union of the parts' bits == the whole search's bitvector | slowest part, all its kernels (ms) |\n|---|---|---|---|---|---|---|---|---|")
    for G, steps, slow, total, whole, largest, same, wall in summary:
        print("| %d | %s | %.1f | %.1f | %.1f / %.1f | %.2f | %d | %s | %.1f |" % (G, steps, slow, total, whole[0], whole[1], whole[0] / slow, largest, same, wall))
    print("\nrecords of both indexes: %.0f MB" % (whole_bytes / 1e6))
    whole_bits.free(); A.free(); B.free()


if __name__ == "__main__":
    main()
