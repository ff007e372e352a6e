#!/usr/bin/env python3
"""Two (or N) PROCESSES that build synthetic inputs at the same time on ONE GPU -- the situation in which round 6 once saw a rank array with 116 of
1.06e8 bits missing inside bwtm_builder_add.  Every process builds `--rounds` indexes of `--reads` reads and checks each
against the generator (extracted reads); the library's own consistency check (set bits == positions of the increment) raises on the failure seen.

    python tools/two_builders.py [--procs 2] [--reads 10000000] [--rounds 3]          (BWTM_POOL_VMM=0 in the environment: without the mapped-memory pool)
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(rank, reads, rounds):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd import synth
    torch.cuda.set_device(0); pkg.init(0)
    dev = torch.device("cuda", 0)
    bad = 0
    for k in range(rounds):
        try:
            ix = synth.build_index(pkg, 1001 + 10 * rank + k, reads, 100, device=dev)
            ids = np.sort(np.random.default_rng(k).integers(0, ix.sequences, 2000))
            got = synth.extract_sequences_matrix(ix, ids, max_len=102)
            ref = synth.reads_matrix("iid", 1001 + 10 * rank + k, ids, 100, ix.sequences, 102)
            ok = bool(np.array_equal(got, ref))
            ix.free()
            print("proc %d round %d: %s" % (rank, k, "ok" if ok else "WRONG READS"), flush=True)
            bad += (0 if ok else 1)
        except pkg.BwtmError as e:
            print("proc %d round %d: ERROR %s" % (rank, k, e), flush=True)
            bad += 1
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--child", type=int, default=-1)
    args = ap.parse_args()
    if args.child >= 0:
        sys.exit(1 if child(args.child, args.reads, args.rounds) else 0)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), "--reads", str(args.reads), "--rounds", str(args.rounds)]) for r in range(args.procs)]
    rcs = [p.wait() for p in procs]
    print("pool: %s; processes: %d; failures: %d" % ("hipMalloc only" if os.environ.get("BWTM_POOL_VMM") == "0" else "mapped blocks", args.procs, sum(1 for r in rcs if r != 0)))


if __name__ == "__main__":
    main()
