#!/usr/bin/env python3
"""NEEDS A DIAGNOSTICS BUILD: bash tools/build_variant.sh diag "" -DBWTM_DIAGNOSTICS; BWTM_LIB=bwt-merge_amd/_variants/diag.so python tools/walk_experiments.py
(the timing-only kernel variants and their bwtm_tune keys are not part of the product library).
Diagnostic: prices the parts of the search kernel on config-2-sized inputs (GPU only).
Usage: python tools/walk_experiments.py [reads_per_set]"""
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
t0 = time.time()
A = synth.build_index(pkg, 1001, reads, 100, device=dev)
B = synth.build_index(pkg, 1002, reads, 100, device=dev)
torch.cuda.empty_cache(); pkg.trim()
print("inputs in %.0f s" % (time.time() - t0), flush=True)
pkg.profile_enable(True)

def run(label, emit, blocks):
    pkg.tune("walk_emit", emit); pkg.tune("walk_blocks", blocks)
    pkg.profile_reset()
    for _ in range(2):
        ra = pkg.RankArray(A, B)
        ra.search(A, B, 0, B.sequences - 1)
        pkg.synchronize()
        ra.free()
    prof = pkg.profile_read()
    for k, (ms, n) in prof.items():
        if k.startswith("lf_walk"):
            print("%-34s %-16s %9.2f ms  %6.2f Gsteps/s" % (label, k, ms / n, B.bases / (ms / n) / 1e6), flush=True)

mode = sys.argv[2] if len(sys.argv) > 2 else "sweep"
def run_product(label, algo=1):
    pkg.tune("walk_emit", 0); pkg.tune("walk_blocks", 0); pkg.tune("walk_kernel", 0); pkg.tune("emit_path", 0); pkg.tune("search_algo", algo)
    import time as _t
    pkg.profile_reset()
    pkg.synchronize(); _t0 = _t.perf_counter()
    for _ in range(2):
        ra = pkg.RankArray(A, B)
        ra.search(A, B, 0, B.sequences - 1)
        pkg.synchronize()
        ra.free()
    prof = pkg.profile_read()
    tot = 0
    for k, (ms, n) in prof.items():
        print("%-34s %-16s %9.2f ms per search" % (label, k, ms / 2), flush=True); tot += ms / 2
    print("%-34s %-16s %9.2f ms  %6.2f Gsteps/s   (wall %.1f ms per search)" % (label, "TOTAL", tot, B.bases / tot / 1e6, (_t.perf_counter() - _t0) * 500), flush=True)

run_product("per-chain walk + partitioned emit", 1)
if mode == "frontier":
    run_product("frontier search (product)", 2)
    pkg.tune("search_algo", 2)
    for mode_emit in (1,):
        pkg.tune("walk_emit", mode_emit)
        pkg.profile_reset()
        for _ in range(2):
            ra = pkg.RankArray(A, B); ra.search(A, B, 0, B.sequences - 1); pkg.synchronize(); ra.free()
        for k, (ms, n) in pkg.profile_read().items():
            if k.startswith("frontier_step"):
                print("frontier timing-only variant       %-28s %9.2f ms per search" % (k, ms / 2), flush=True)
    pkg.tune("walk_emit", 0); pkg.tune("search_algo", 0)
    sys.exit(0)
if mode == "ablate":
    pkg.tune("emit_path", 1); pkg.tune("walk_kernel", 0)
    for abl, name in ((0, "full"), (8, "synthetic chain, all loads"), (1, "no sup loads"), (2, "no A record"), (3, "no sup, no A record"), (7, "no loads at all")):
        pkg.tune("walk_ablate", abl)
        run("quad no-emit ablation: %s" % name, 1, 2048)
    pkg.tune("walk_ablate", 0); pkg.tune("emit_path", 0)
    sys.exit(0)
pkg.tune("emit_path", 1); pkg.tune("search_algo", 1)
for kernel, kname in ((0, "quad"), (1, "lane")):
    pkg.tune("walk_kernel", kernel)
    if mode == "sweep" and kernel == 0:
        for blocks in (1024, 4096, 8192):
            run("%s atomicOr, %d blocks" % (kname, blocks), 0, blocks)
    run("%s atomicOr, 2048 blocks" % kname, 0, 2048)
    run("%s no emit, 2048 blocks" % kname, 1, 2048)
    run("%s 8-byte store emit, 2048 blocks" % kname, 2, 2048)
pkg.tune("walk_emit", 0); pkg.tune("walk_blocks", 0); pkg.tune("walk_kernel", 0); pkg.tune("emit_path", 0)
