#!/usr/bin/env python3
"""The whole merge over partitioned records on N GPUs of one node, one process per GPU over RCCL (experimental_dist.merge_partitioned_dist;
DESIGN.md section 6.3): every rank transcodes its windows from its own share of the native bytes, searches with the per-step all-to-all,
and finalizes / interleaves / encodes its own range of the output.  Prints ONE JSON line from rank 0: ms per merge (barrier to barrier, the
maximum over the ranks) and merged Gbases/s with the inputs in page-locked host memory -- the windows' upload is inside the timed region,
because that is where a rank's share of the input first meets its GPU.

  N = 1:   python tools/partitioned_dist_run.py [--reads N]
  N > 1:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/partitioned_dist_run.py
Never run on more than one GPU so far (no such machine was available): N = 1 runs every collective with itself
(tests/experimental/test_gpu_partitioned_dist.py), the exchange logic for N = 2, 3 is covered over gloo on the CPU (tests/test_partition_dist_host.py).
--verify: rank 0 also runs the product merge and checks every rank's slice against its bytes of that stream.
"""
import argparse
import hashlib
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BWTM_LIB", os.path.join(ROOT, "bwt-merge_amd", "libbwtm_experimental.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--readlen", type=int, default=100)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--verify", action="store_true")
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd import synth
    from bwt_merge_amd import experimental as X
    from bwt_merge_amd.experimental_dist import merge_partitioned_dist
    assert X.loaded(), "BWTM_LIB must name libbwtm_experimental.so"
    torch.cuda.set_device(local)
    pkg.init(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    hosts = []
    for seed in (1001, 1002):                                        # every rank builds the same inputs and keeps them on the HOST only
        ix = synth.build_index(pkg, seed, args.reads, args.readlen, device=dev)
        ix.encode()
        data = pkg.HostBuffer(ix.nbytes)
        ix.download_into(data.array)
        be, cum = ix.samples()
        hosts.append(types.SimpleNamespace(data=data.array[: ix.nbytes], samples=(be, cum), bases=ix.bases, sequences=ix.sequences, keep=data))
        ix.free()
    torch.cuda.empty_cache(); pkg.trim()
    a, b = hosts
    cuts = [None]
    if rank == 0:                                                    # the cuts from whole indexes on rank 0 (a deployment asks the host's index)
        A = pkg.Index.upload(a.data, a.sequences, a.bases); B = pkg.Index.upload(b.data, b.sequences, b.bases)
        cuts[0] = X.partition_cuts(A, B, world, args.k)
        A.free(); B.free(); pkg.trim()
    dist.broadcast_object_list(cuts, src=0)
    cuts = cuts[0]

    times = {}

    def one(keep=False):
        S, handles, steps = merge_partitioned_dist(pkg, a, b, cuts, rank, world, dist, torch, dev, times=times)
        pkg.synchronize()
        if keep:
            return S, handles, steps
        S.free()
        for h in handles:
            h.free()
        return None, None, steps

    for _ in range(args.warmup):
        one()
    times.clear()
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = one(keep=(k == args.steps - 1))
    torch.cuda.synchronize(); dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    S, handles, steps = last
    mine = S.data()
    info = [None] * world
    dist.all_gather_object(info, (int(S.byte_first), int(mine.size), hashlib.sha256(mine.tobytes()).hexdigest(), X.index_record_bytes(handles[1]) + X.index_record_bytes(handles[2]),
                                  X.ra_bytes(handles[0])))
    verified = None
    if args.verify and rank == 0:
        S.free()
        for h in handles:
            h.free()
        pkg.trim()
        A = pkg.Index.upload(a.data, a.sequences, a.bases); B = pkg.Index.upload(b.data, b.sequences, b.bases)
        M = pkg.merge(A, B)
        ref = np.empty(M.nbytes, dtype=np.uint8)
        M.download_into(ref)
        verified = (sum(x[1] for x in info) == ref.size) and all(hashlib.sha256(ref[off: off + n].tobytes()).hexdigest() == h for off, n, h, _, _ in info)
        M.free(); A.free(); B.free()
    if rank == 0:
        ms = float(elapsed.item()) * 1e3 / args.steps
        print(json.dumps({"metric": "merged Gbases/sec (input1+input2), partitioned records, native inputs in host memory", "value": round((a.bases + b.bases) / ms / 1e6, 4),
                          "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "lf_steps": steps[0], "node_levels": steps[1],
                          "phases_ms_rank0": {k: round(v / times["merges"], 1) for k, v in times.items() if k != "merges"},
                          "note": "a functional prototype driven step by step from Python: every LF step costs host synchronisations, an all-gather, five to ten all-to-alls and a barrier, and the exported buffers are hipMalloc'ed per merge; the kernels' share is in profiles/r05_partitioned_merge_config2.txt",
                          "records_bytes_per_gpu": [x[3] for x in info], "bitvector_bytes_per_gpu": [x[4] for x in info], "output_bytes_per_gpu": [x[1] for x in info],
                          "verified_against_the_product_merge": verified, "library": os.environ["BWTM_LIB"]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
