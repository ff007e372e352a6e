#!/usr/bin/env python3
"""bench.py -- merged Gbases/s of the rank-array / interleave path on N MI355X GPUs.

One "step" = one whole merge job over synthetic inputs that are already resident in HBM in
the reference's native byte format: transcode of both inputs to the device rank structure,
LF-walk search, rank-array finalize, interleave, canonical run encoder, sample build
(the work between the timers of merge(), bwt_merge.cpp:287-299).  At N > 1 the sequences of
input2 are sharded over the ranks and the rank-array bitvectors are combined with one RCCL
all-reduce (sum == or: set bits are disjoint); the rest is replicated ("strong" scaling:
the job is fixed as N grows).

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  roofline      HBM roofline of the dominant kernel (k_frontier_step; k_lf_walk_binned for small inputs), duration
                measured with HIP events on the library's stream
  cpu_baseline  the CPU oracle (port of the reference algorithm) timed on this host's cores
                on a bounded sample of the same workload (N == 1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
SEARCH_BYTES_PER_BASE = 160                # one 64-byte block + 8 + 8 bytes on each side


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per input set (config 2: 5e7 x 100 bp = 5.05 Gbase)")
    ap.add_argument("--readlen", type=int, default=100)
    ap.add_argument("--leaf-reads", type=int, default=1 << 19)
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads per set for the CPU baseline sample (0 = auto)")
    ap.add_argument("--workload", choices=("iid", "genome"), default="iid",
                    help="iid = the headline distribution; genome = reads from a shared random genome, 30x coverage, 1 %% substitutions (SURVEY 8(d), secondary)")
    ap.add_argument("--coverage", type=int, default=30, help="genome workload: coverage (a smaller genome = a more repetitive BWT)")
    ap.add_argument("--error-percent", type=int, default=1, help="genome workload: substitution rate in percent")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (process group, caller-owned bitvector, all-reduce) even with one rank: "
                         "a smoke test of that path on a 1-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd import synth

    torch.cuda.set_device(local_rank)
    pkg.init(local_rank)
    dist = None
    sharded = world > 1 or args.force_dist
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    def barrier():
        pkg.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---------------------------------------------------------------- inputs (untimed)
    t_gen = time.time()
    wargs = ({"coverage": args.coverage, "error_percent": args.error_percent} if args.workload == "genome" else {})
    sets = []
    for k, seed in enumerate((1001, 1002)):
        def progress(done, total, k=k):
            if rank == 0 and (done == total or (done // args.leaf_reads) % 16 == 0):
                log("input%d: %d / %d reads (%.0f s)" % (k + 1, done, total, time.time() - t_gen))
        ix = synth.build_index(pkg, seed, args.reads, args.readlen, leaf_reads=args.leaf_reads, device=dev, progress=progress,
                               workload=args.workload, **wargs)
        ix.encode()
        sets.append(ix)
    torch.cuda.empty_cache()
    pkg.trim()
    A0, B0 = sets
    n_a, n_b = A0.bases, B0.bases
    ptr_a, bytes_a = A0.device_data()
    ptr_b, bytes_b = B0.device_data()
    if rank == 0:
        log("inputs ready in %.0f s: %d + %d bases, %.3f + %.3f GB native (%.3f bytes/base)" %
            (time.time() - t_gen, n_a, n_b, bytes_a / 1e9, bytes_b / 1e9, (bytes_a + bytes_b) / (n_a + n_b)))

    # shard of input2's sequences for this rank (getBounds, utils.cpp:169-187)
    from bwt_merge_amd.dist import shard_range
    seq_first, seq_last = shard_range(B0.sequences, rank, world)

    # ---------------------------------------------------------------- one step
    def step(keep=False):
        # BWT::load of both inputs: the resident native bytes are read in place (no second copy in HBM)
        A = pkg.Index.from_device(ptr_a, bytes_a, A0.sequences, n_a, borrow=True)
        B = pkg.Index.from_device(ptr_b, bytes_b, B0.sequences, n_b, borrow=True)
        if not sharded:
            M = pkg.merge(A, B)
        else:
            nbytes = pkg.ra_buffer_bytes(A, B)
            buf = torch.zeros(nbytes // 8, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ra = pkg.RankArray(A, B, buf.data_ptr(), nbytes)
            if seq_first <= seq_last:
                ra.search(A, B, seq_first, seq_last)
            pkg.synchronize()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)      # RCCL over xGMI; disjoint bits: sum == or
            torch.cuda.synchronize()
            ra.finalize()
            M = pkg.interleave(A, B, ra)
            M.encode()
            ra.free()
            del buf
        pkg.synchronize()
        A.free(); B.free()
        if keep:
            return M
        M.free()
        return None

    for _ in range(args.warmup):
        step()
    pkg.profile_enable(True)
    pkg.profile_reset()
    barrier()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = step(keep=(k == args.steps - 1))
    barrier()
    elapsed = time.perf_counter() - t0
    prof = pkg.profile_read()
    pkg.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sec_per_step = elapsed / max(1, args.steps)
    value = (n_a + n_b) / 1e9 / sec_per_step

    # ---------------------------------------------------------------- roofline of the dominant kernel
    # The search kernel: k_frontier_step (one launch per LF step of the level-synchronous search) or,
    # when the library falls back to it, k_lf_walk_binned (one launch per search).  SURVEY.md 8(d):
    # 160 algorithmic bytes per LF step (one 64-byte block + 8 + 8 bytes on each side).
    dom = "frontier_step" if "frontier_step" in prof else "lf_walk"
    dom_ms, dom_launches = prof.get(dom, (0.0, 0))
    searches = max(1, args.steps)
    units_per_search = (seq_last - seq_first + 1) / max(1, B0.sequences) * n_b if seq_first <= seq_last else 0
    launches_per_search = dom_launches / searches if dom_launches else 0
    avg_launch_s = (dom_ms / 1e3 / dom_launches) if dom_launches else float("nan")
    units_per_launch = units_per_search / launches_per_search if launches_per_search else 0
    achieved = SEARCH_BYTES_PER_BASE * units_per_launch / avg_launch_s / 1e9 if dom_launches else 0.0
    traffic, traffic_source = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "search_kernel_traffic.json")) as f:
            t = json.load(f)
        if (t["kernel"] == dom and t["config"]["reads_per_set"] == args.reads and t["config"]["read_length"] == args.readlen and world == 1):
            traffic, traffic_source = t["hbm_bytes_per_launch"], t["source"]
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": "k_" + dom + ("_binned" if dom == "lf_walk" else ""), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launches_per_step": round(launches_per_search, 2), "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                "kernel_ms_per_step": round(dom_ms / searches, 3),
                "algorithmic_bytes_per_launch": SEARCH_BYTES_PER_BASE * units_per_launch}
    kernel_ms = {name: round(ms / max(1, args.steps), 3) for name, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0])}
    # whole-job algorithmic bytes W = 176 n_B + |A| + |B| + 2 |Out|
    out_bytes = last.nbytes if last is not None else 0
    W = 176 * n_b + bytes_a + bytes_b + 2 * out_bytes
    job = {"algorithmic_bytes": W, "achieved_GBs": round(W / sec_per_step / 1e9, 1), "frac": round(W / sec_per_step / 1e9 / HBM_PEAK_GBS, 4)}

    # ---------------------------------------------------------------- verification at full size
    verified = None
    if rank == 0 and last is not None and not args.no_verify:
        verified = bool(np.array_equal(last.C, A0.C + B0.C) and last.bases == n_a + n_b and last.sequences == A0.sequences + B0.sequences)
        rng = np.random.default_rng(12345)
        ids = np.sort(rng.integers(0, last.sequences, 48))
        got = synth.extract_sequences(pkg, last, ids, max_len=args.readlen + 8)
        for j, seq in zip(ids, got):
            seed, idx = (1001, int(j)) if j < A0.sequences else (1002, int(j - A0.sequences))
            ref = synth.make_reads(args.workload, seed, idx, 1, args.readlen, args.reads, **wargs)[0].tolist()
            verified = verified and (seq == ref)
        # the emitted native stream must decode back to the merged index (header check in upload)
        p, nb = last.device_data()
        R = pkg.Index.from_device(p, nb, last.sequences, last.bases)
        w0 = int(rng.integers(0, max(1, last.bases - 4096)))
        verified = verified and bool(np.array_equal(R.extract(w0, min(4096, last.bases)), last.extract(w0, min(4096, last.bases))))
        R.free()
        log("full-size verification: %s" % verified)
    if last is not None:
        last.free()

    # ---------------------------------------------------------------- CPU baseline (rank 0, N == 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pkg, synth, torch, np, dev, args)

    if rank == 0:
        out = {
            "metric": "merged Gbases/sec (input1+input2), bit-exact native BWT",
            "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "two %.3g Gbase synthetic %d bp read sets (%s), native format, inputs resident in HBM" %
                       (n_a / 1e9, args.readlen, "sigma=6" if args.workload == "iid" else "reads from a shared random genome, 30x coverage, 1% substitutions"),
                       "reads_per_set": args.reads, "read_length": args.readlen,
                       "bases": [n_a, n_b], "native_bytes": [bytes_a, bytes_b, out_bytes],
                       "parallelism": "sequence blocks of input2 sharded over %d GPU(s)%s" %
                       (world, ", RCCL all-reduce of the rank-array bitvector" if world > 1 else "")},
            "roofline": roofline, "job_roofline": job, "kernel_ms_per_step": kernel_ms,
            "cpu_baseline": cpu, "verified": verified,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(pkg, synth, torch, np, dev, args):
    """Times the CPU oracle (port of the reference algorithm, all host cores, reference default
    buffer sizes) on a bounded sample of the same workload and checks the GPU result on it."""
    from oracle import oracle as orc
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    n = args.cpu_sample_reads or int(min(1 << 21, max(1 << 16, cores * (1 << 14))))
    n = min(n, args.reads)
    t0 = time.time()
    fm = []
    for seed in (1001, 1002):
        sym = synth.leaf_bwt(synth.make_reads(args.workload, seed, 0, n, args.readlen, n, device=dev)).cpu().numpy()
        fm.append(orc.FMI.from_symbols(sym))
    a, b = fm
    A = pkg.Index.upload(a.data, a.sequences, a.bases)
    B = pkg.Index.upload(b.data, b.sequences, b.bases)
    M = pkg.merge(A, B)
    gpu_bytes = M.data()
    log("cpu baseline sample: 2 x %d reads prepared in %.1f s; running the oracle on %d threads" % (n, time.time() - t0, cores))
    t0 = time.perf_counter()
    m, secs = orc.merge(a, b, threads=cores)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(gpu_bytes, m.data))
    merged = 2 * n * (args.readlen + 1)
    log("cpu baseline: %.2f s (search %.2f s, interleave %.2f s), parity with GPU on the sample: %s" % (dt, secs[0], secs[1], ok))
    return {"value": round(merged / 1e9 / dt, 6), "unit": "Gbases/s", "cores": cores, "kind": "port",
            "sample": "two sets of %d synthetic %d bp reads (%.3g Gbase merged), oracle merge with %d threads, reference default buffers" %
                      (n, args.readlen, merged / 1e9, cores),
            "seconds": round(dt, 3), "gpu_parity_on_sample": ok}


if __name__ == "__main__":
    main()
