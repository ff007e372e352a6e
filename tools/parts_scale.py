#!/usr/bin/env python3
"""The merge over PARTITIONED records (include/bwtm.h: bwtm_group_*, bwtm_part_*) at full size on ONE GPU: G threads of this process, one
library context each, stand in for G GPUs.  Every part transcodes its windows from its own share of the native bytes, searches in lock step
with the others -- its step kernel reads its input straight out of the other parts' output buffers --, and finalizes, interleaves and encodes
its own range of the output; the parts' bytes, laid end to end, must be the product merge's native stream.

With BWTM_GROUP_SERIAL=1 (set here) every compute section of a part runs ALONE on the device, so a part's kernel milliseconds are what its
own GPU would need; they are reported by phase and, for the search, by kernel, next to the single-GPU merge.

    python tools/parts_scale.py [--reads N] [--parts 2,4,8] [--workload iid|genome]
"""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--reads-a", type=int, default=0, help="reads of input1 when it differs from input2 (BASELINE config 4's shape: a large input1, a small input2)")
    ap.add_argument("--readlen", type=int, default=100)
    ap.add_argument("--parts", default="2,4,8")
    ap.add_argument("--kmer", type=int, default=0)
    ap.add_argument("--emit-budget", type=int, default=2 << 30)
    ap.add_argument("--workload", choices=("iid", "genome"), default="iid")
    ap.add_argument("--coverage", type=int, default=30)
    ap.add_argument("--concurrent", action="store_true", help="let the parts' kernels overlap on the one GPU (wall time of the whole merge instead of per-part kernel times)")
    args = ap.parse_args()
    if not args.concurrent:
        os.environ["BWTM_GROUP_SERIAL"] = "1"
    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd import partitioned, synth
    torch.cuda.set_device(0)
    pkg.init(0)
    dev = torch.device("cuda", 0)
    t0 = time.time()
    hosts = []
    for seed, nreads in ((1001, args.reads_a or args.reads), (1002, args.reads)):
        wargs = ({"coverage": args.coverage, "error_percent": 1} if args.workload == "genome" else {})
        ix = synth.build_index(pkg, seed, nreads, args.readlen, device=dev, workload=args.workload, **wargs)
        ix.encode()
        data = pkg.HostBuffer(ix.nbytes)
        ix.download_into(data.array)
        be, cum = ix.samples()
        hosts.append(types.SimpleNamespace(buf=data, data=data.array, cum=cum, bases=ix.bases, sequences=ix.sequences))
        ix.free()
    torch.cuda.empty_cache(); pkg.trim()
    a, b = hosts
    print("inputs: %d + %d reads of %d bp (%.2f + %.2f Gbase, %.2f + %.2f GB native) in %.0f s" %
          (args.reads_a or args.reads, args.reads, args.readlen, a.bases / 1e9, b.bases / 1e9, a.data.size / 1e9, b.data.size / 1e9, time.time() - t0), flush=True)
    pkg.tune("emit_budget", args.emit_budget)

    # the product merge on the one GPU: the reference bytes, and its kernels by phase
    A = pkg.Index.upload(a.data, a.sequences, a.bases); B = pkg.Index.upload(b.data, b.sequences, b.bases)
    M = pkg.merge(A, B); M.free()                                    # warm-up (pool)
    A.free(); B.free()
    pkg.trim()
    pkg.profile_only(None); pkg.profile_reset(); pkg.profile_enable(True)
    A2 = pkg.Index.upload(a.data, a.sequences, a.bases); B2 = pkg.Index.upload(b.data, b.sequences, b.bases)
    M = pkg.merge(A2, B2)
    pkg.synchronize()
    prof = pkg.profile_read(); pkg.profile_enable(False)
    A2.free(); B2.free()
    ref = np.empty(M.nbytes, dtype=np.uint8)
    M.download_into(ref)
    M.free()
    pkg.trim()                                                       # the default context's pool gives its blocks back: the parts' contexts need the memory
    transcode = sum(prof.get(n, (0, 0))[0] for n in ("block_len", "build_recs", "build_sup"))
    tail_names = ("interleave", "interleave_base", "interleave_sup", "enc_emit", "enc_size", "enc_lasthead", "fold_top", "fold_group", "fold_seg", "chunk_popc", "block_cum")
    tail = sum(v[0] for n, v in prof.items() if n in tail_names)
    total = sum(v[0] for v in prof.values())
    print("\n| the product merge on one GPU, kernels (ms) | transcode | search (everything between) | of it k_frontier_step | finalize + interleave + encode | all |\n|---|---|---|---|---|---|")
    print("| | %.1f | %.1f | %.1f | %.1f | %.1f |" % (transcode, total - transcode - tail, prof.get("frontier_step", (0, 0))[0], tail, total), flush=True)
    ha = pkg.host_index(a.data, a.cum, a.sequences, a.bases); hb = pkg.host_index(b.data, b.cum, b.sequences, b.bases)
    rows = []
    for G in [int(x) for x in args.parts.split(",")]:
        tc = time.perf_counter()
        cuts = pkg.partition_cuts_host(ha, hb, G, args.kmer)
        cuts_ms = (time.perf_counter() - tc) * 1e3

        def collect(g, s):
            got = s.data()
            return bool(np.array_equal(got, ref[s.byte_offset: s.byte_offset + got.size])), got.size

        tw = time.perf_counter()
        out = partitioned.merge_parts(pkg, ha, hb, G, cuts=cuts, profile=not args.concurrent, collect=collect)
        wall_ms = (time.perf_counter() - tw) * 1e3
        same = all(c[0] for c in out["collected"]) and sum(c[1] for c in out["collected"]) == ref.size
        st = out["stats"]
        print("\n| G = %d (cuts on the host: %.1f ms) | part | records held (MB) | bitvector held (MB) | transcode from its byte share (ms) | search: %d node levels + %d element steps (ms) | of it: k_frontier_step | pull_tables + scans + cut_search | node levels | tile builds | finalize + interleave + encode of its range (ms) | all (ms) | elements advanced (share) | pulled from the parts' buffers (MB) | output bytes |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|" %
              (G, cuts_ms, st[0]["node_levels"], st[0]["steps"]))
        alls = []
        for g in range(G):
            ph = out["phases"][g]
            tr = sum(ph.get("transcode", {}).values()); se = ph.get("search", {}); fi = sum(ph.get("finish", {}).values())
            search = sum(se.values())
            routing = sum(se.get(n, 0) for n in ("pull_tables", "frontier_scan", "scan_reduce", "cut_search", "frontier_prep", "scan_apply"))
            nodes = sum(se.get(n, 0) for n in ("range_step", "range_emit", "range_children", "cut_counts", "nodes_gather", "range_expand", "range_expand_pieces", "range_init", "frontier_init"))
            tiles = sum(se.get(n, 0) for n in ("bound_seg_min", "bound_suffix_min", "tile_build"))
            allg = tr + search + fi; alls.append(allg)
            print("| | %d | %.0f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %d (%.3f) | %.0f | %d |" %
                  (g, st[g]["record_bytes"] / 1e6, st[g]["bitvector_bytes"] / 1e6, tr, search, se.get("frontier_step", 0), routing, nodes, tiles, fi, allg,
                   st[g]["elements"], st[g]["elements"] / max(1, sum(x["elements"] for x in st)), st[g]["pulled_bytes"] / 1e6, out["collected"][g][1]), flush=True)
        rows.append((G, max(alls) if alls else 0, sum(alls), total, same, wall_ms, max(x["largest"] for x in st)))
        out["release"]()
        pkg.trim()
    print("\n| G | slowest part, all its kernels (ms) | all parts together (ms) | the product merge on one GPU (ms) | all parts / product | speed-up = product / slowest part | largest frontier a part held | wall of the whole merge on the one GPU (ms) | the parts' bytes == the product's stream |\n|---|---|---|---|---|---|---|---|---|")
    for G, slow, alltog, whole, same, wall, largest in rows:
        print("| %d | %.1f | %.1f | %.1f | %.2f | %.2f | %d | %.0f | %s |" % (G, slow, alltog, whole, alltog / whole, whole / max(slow, 1e-9), largest, wall, same))


if __name__ == "__main__":
    main()
