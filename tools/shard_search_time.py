#!/usr/bin/env python3
"""What one rank of a G-GPU run spends in the search: times bwtm_search on the first 1/G of input2's sequences at
config-2 size (the indexes are replicated, so this is exactly a rank's work), for both forms of the search.
Usage: python tools/shard_search_time.py [reads_per_set]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg
pkg = _pkg.load()
from bwt_merge_amd import synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
pkg.init(0)
dev = torch.device("cuda", 0)
A = synth.build_index(pkg, 1001, reads, 100, device=dev)
B = synth.build_index(pkg, 1002, reads, 100, device=dev)
torch.cuda.empty_cache(); pkg.trim()
for G in (1, 2, 4, 8):
    last = B.sequences // G - 1
    for algo, name in ((2, "frontier"), (1, "walk")):
        pkg.tune("search_algo", algo)
        best = 1e9
        for _ in range(2):
            ra = pkg.RankArray(A, B)
            pkg.synchronize(); t0 = time.perf_counter()
            ra.search(A, B, 0, last)
            pkg.synchronize(); best = min(best, time.perf_counter() - t0)
            ra.free()
        print("G = %d: shard of %9d sequences, %-8s %7.1f ms" % (G, last + 1, name, best * 1e3), flush=True)
pkg.tune("search_algo", 0)
