#!/usr/bin/env python3
"""The WHOLE merge over partitioned records at full size, on one GPU: G contexts stand in for G GPUs (experimental.merge_partitioned).  Every
part transcodes its windows from its own share of the native bytes, searches with the elements routed by position, and finalizes, interleaves
and encodes its own range of the output; the parts' bytes, laid end to end, must be the product merge's native stream.  Reports every part's
kernel milliseconds by phase (the contexts run one after the other: a part's time is what its GPU would need) next to the product merge on
the one GPU.

Needs the experimental library:  BWTM_LIB=$PWD/bwt-merge_amd/libbwtm_experimental.so python tools/partitioned_merge_scale.py [--reads N] [--parts 2,4,8]
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
    ap.add_argument("--readlen", type=int, default=100)
    ap.add_argument("--parts", default="2,4,8")
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--emit-budget", type=int, default=2 << 30)
    ap.add_argument("--slack", type=float, default=1.25)
    ap.add_argument("--workload", choices=("iid", "genome"), default="iid", help="genome: reads of a shared random genome (bench.py's secondary workload): a compressible BWT of long runs")
    ap.add_argument("--coverage", type=int, default=30)
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
    hosts = []
    for seed in (1001, 1002):
        wargs = ({"coverage": args.coverage, "error_percent": 1} if args.workload == "genome" else {})
        ix = synth.build_index(pkg, seed, args.reads, args.readlen, device=dev, workload=args.workload, **wargs)
        ix.encode()
        data = np.empty(ix.nbytes, dtype=np.uint8)
        ix.download_into(data)
        be, cum = ix.samples()
        hosts.append(types.SimpleNamespace(data=data, samples=(be, cum), bases=ix.bases, sequences=ix.sequences))
        ix.free()
    torch.cuda.empty_cache(); pkg.trim()
    a, b = hosts
    print("inputs: 2 x %d reads of %d bp (%.2f + %.2f Gbase, %.2f + %.2f GB native) in %.0f s" %
          (args.reads, args.readlen, a.bases / 1e9, b.bases / 1e9, a.data.size / 1e9, b.data.size / 1e9, time.time() - t0), flush=True)
    pkg.tune("emit_budget", args.emit_budget)

    # the product merge on the one GPU: the reference bytes, and its kernels by phase
    A = pkg.Index.upload(a.data, a.sequences, a.bases); B = pkg.Index.upload(b.data, b.sequences, b.bases)
    M = pkg.merge(A, B); M.free()                                    # warm-up (pool)
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
    tail = sum(v[0] for n, v in prof.items() if n in ("interleave", "interleave_base", "interleave_sup", "enc_emit", "enc_size", "enc_lasthead", "fold_top", "fold_group", "fold_seg",
                                                      "chunk_popc", "block_cum"))
    total = sum(v[0] for v in prof.values())
    print("\n| the product merge on one GPU, kernels (ms) | transcode | search (everything between) | finalize + interleave + encode | all |\n|---|---|---|---|---|")
    print("| | %.1f | %.1f | %.1f | %.1f |" % (transcode, total - transcode - tail, tail, total), flush=True)
    m_b = b.sequences
    rows = []
    for G in [int(x) for x in args.parts.split(",")]:
        cuts = X.partition_cuts(A, B, G, args.k)                      # from whole indexes here; the host's own index in a deployment
        limit = m_b // 8
        out = X.merge_partitioned(pkg, a, b, G, cuts, from_bytes=True, node_ratio=8, capacity=int(args.slack * m_b / G) + 65536,
                                  node_capacity=int(args.slack * min(5 * limit, m_b) / G) + 65536, profile=True, keep_data=False)
        # the parts' bytes, end to end, are the product's stream
        same = (out["offsets"][-1] == ref.size)
        for g, s in enumerate(out["slices"]):
            if not same:
                break
            out_g = s.data()
            same = same and np.array_equal(out_g, ref[out["offsets"][g]: out["offsets"][g] + out_g.size])
        print("\n| G = %d | part | records held (MB) | bitvector held (MB) | transcode from its byte share (ms) | search: node levels + %d element steps (ms) | finalize + interleave + encode of its range (ms) | all (ms) | output bytes |\n|---|---|---|---|---|---|---|---|---|" % (G, out["steps"]))
        alls = []
        for g in range(G):
            ph = out["phases"][g]
            allg = sum(ph.values()); alls.append(allg)
            print("| | %d | %.0f | %.1f | %.1f | %.1f | %.1f | %.1f | %d |" % (g, out["held"][g] / 1e6, out["ra_bytes"][g] / 1e6, ph.get("transcode", 0), ph.get("search", 0),
                  ph.get("finalize_interleave_encode", 0), allg, out["nbytes"][g]), flush=True)
        rows.append((G, max(alls), sum(alls), total, same))
        out["release"]()
        pkg.trim()
    print("\n| G | slowest part, all its kernels (ms) | all parts together (ms) | the product merge on one GPU (ms) | speed-up = product / slowest part | the parts' bytes == the product's stream |\n|---|---|---|---|---|---|")
    for G, slow, alltog, whole, same in rows:
        print("| %d | %.1f | %.1f | %.1f | %.2f | %s |" % (G, slow, alltog, whole, whole / slow, same))
    A.free(); B.free()


if __name__ == "__main__":
    main()
