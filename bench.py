#!/usr/bin/env python3
"""bench.py -- merged Gbases/s of the rank-array / interleave path on N MI355X GPUs.

One "step" = one whole merge job over synthetic inputs that are already resident in HBM in
the reference's native byte format: transcode of both inputs to the device rank structure,
LF-walk search, rank-array finalize, interleave, canonical run encoder, sample build
(the work between the timers of merge(), bwt_merge.cpp:287-299).  At N > 1 the sequences of
input2 are sharded over the ranks and the rank-array bitvectors are combined with one RCCL
reduce-scatter by output range (sum == or: set bits are disjoint); every rank then interleaves and
encodes only its own range of the output (bwtm_slice_*), so the result is left sharded by byte range.

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  value         the HBM-resident rate (inputs and result stay on the device)
  host_to_host  the rate SURVEY.md 8(d) defines: page-locked host inputs -> page-locked host result incl. samples
                (bwtm_merge_host: pipelined H2D / device work / D2H), with a PCIe calibration next to it
  roofline      HBM roofline of the dominant kernel (k_frontier_step; k_lf_walk_binned for small inputs), duration
                measured with HIP events on the library's stream
  cpu_baseline  the CPU oracle (port of the reference algorithm) timed on this host's cores
                on a bounded sample of the same workload (N == 1, rank 0 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_START = time.time()
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_SPEC_GBS = 63.0                       # PCIe Gen5 x16 per direction
SEARCH_BYTES_PER_BASE = 160                # one 64-byte block + 8 + 8 bytes on each side


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per input set (config 2: 5e7 x 100 bp = 5.05 Gbase)")
    ap.add_argument("--reads-a", type=int, default=0, help="reads of input1 when it differs from input2 (BASELINE config 4: asymmetric insert)")
    ap.add_argument("--readlen", type=int, default=100)
    ap.add_argument("--leaf-reads", type=int, default=1 << 19)
    ap.add_argument("--torch-leaves", action="store_true", help="build the inputs' leaves with tensor ops instead of the library's builder (cross-check)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads per set for the CPU baseline's large sample (0 = auto)")
    ap.add_argument("--workload", choices=("iid", "genome", "mixed"), default="iid",
                    help="iid = the headline distribution; genome = reads from a shared random genome, 30x coverage, 1 %% substitutions "
                         "(SURVEY 8(d), secondary); mixed = iid reads of 100 and 150 bp, half of the bases each (BASELINE config 5)")
    ap.add_argument("--coverage", type=int, default=30, help="genome workload: coverage (a smaller genome = a more repetitive BWT)")
    ap.add_argument("--error-percent", type=int, default=1, help="genome workload: substitution rate in percent")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config1", action="store_true", help="skip the one-thread CPU run of BASELINE config 1 inside the CPU baseline")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel launch of the timed region with HIP events (default: only the dominant kernel; "
                                                               "the per-kernel table then comes from one extra untimed step)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the host-to-host measurement")
    ap.add_argument("--host-steps", type=int, default=3)
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="bwtm_tune knobs for experiments (product knobs never change results)")
    ap.add_argument("--verify-reads", type=int, default=10000, help="reads extracted from the merged index and compared with the generator")
    ap.add_argument("--chain", type=int, default=2,
                    help="number of input sets; more than 2 = chained merge in command-line order (BASELINE config 5: bwt_merge in1 in2 in3 in4 out), "
                         "intermediate results stay on the device as rank structures, only the last merge encodes")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (process group, caller-owned bitvector, reduce-scatter by output range, output slices) even with one rank: "
                         "a smoke test of that path on a 1-GPU box")
    ap.add_argument("--search", choices=("auto", "blocks", "partitioned"), default="auto",
                    help="several ranks: `partitioned` = the merge over partitioned records (one part per rank: windows from byte shares, the frontier's "
                         "elements read from the peers' buffers, no collective library; DESIGN.md section 6.3), `blocks` = sequence blocks of input2 per rank, "
                         "replicated records, RCCL reduce-scatter of the bitvector; auto = partitioned for two ranks and more, blocks for --force-dist on one")
    ap.add_argument("--same-device", action="store_true",
                    help="all ranks use GPU 0 (a one-GPU box): the ranks are still processes of their own that map each other's buffers through HIP IPC; "
                         "the process group is gloo (RCCL refuses two ranks on one device); --search partitioned only")
    ap.add_argument("--target", choices=("auto", "on", "off"), default="auto",
                    help="after the configured workload, measure the north star's target size (two sets of --target-reads reads on ONE GPU) and add it to "
                         "the line as `target`; auto = when this is the default single-GPU config-2 run and the device has the memory for it")
    ap.add_argument("--target-reads", type=int, default=500_000_000, help="reads per set of the target measurement (5e8 x 100 bp = 50.5 Gbase)")
    ap.add_argument("--target-steps", type=int, default=2)
    ap.add_argument("--target-timeout", type=int, default=1500, help="seconds after which the target measurement's child process is stopped (the configured "
                                                                     "workload's line is printed either way)")
    ap.add_argument("--as-target", action="store_true", help=argparse.SUPPRESS)        # set by the parent for its target-size child: fewer repeats of every leg
    ap.add_argument("--keep-pool", action="store_true",
                    help="do not return the library's pooled memory to the driver after the inputs are built (under rocprofv3 --pmc memory released "
                         "with hipMemRelease does not come back: at 2 x 50 Gbase the merge then finds 118 GB less than it should)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts the N ranks itself (it has not touched the GPU) and
        # relays rank 0's JSON line.
        sys.exit(launch_ranks(args.gpus, args.same_device))

    # stdout carries the ONE JSON line and nothing else: native libraries (RCCL prints a version banner through C stdio) are sent
    # to stderr by pointing file descriptor 1 there; the line itself goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import types
    import numpy as np
    import torch
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd import synth

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    pkg.init(local_rank)
    dist = None
    sharded = world > 1 or args.force_dist
    args.partitioned = sharded and (args.search == "partitioned" or (args.search == "auto" and world > 1))
    if args.same_device and world > 1 and not args.partitioned:
        raise SystemExit("--same-device needs --search partitioned (RCCL refuses two ranks on one device)")
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.same_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    args.coll_dev = (torch.device("cpu") if args.same_device else dev)          # where the bench's own small collectives (timing, flags) live

    for kv in args.tune:
        key, _, value = kv.partition("=")
        pkg.tune(key, int(value))

    env = types.SimpleNamespace(pkg=pkg, synth=synth, np=np, torch=torch, dist=dist, dev=dev, rank=rank, world=world, sharded=sharded)
    args.is_target = bool(args.as_target)
    out = measure(env, args)

    # ---------------------------------------------------------------- the north star's target size on ONE GPU
    # "2 x 50 Gbase synthetic 100 bp reads at 1 MI355X" (BASELINE.json north_star): measured in the same run, after the configured
    # workload has released everything, so that the driver's line carries it.  Not a second bench line: `value` stays config 2's.
    default_shape = (args.reads == 50_000_000 and not args.reads_a and args.readlen == 100 and args.workload == "iid" and args.chain == 2
                     and not args.torch_leaves and not args.tune)
    want_target = (args.target == "on" or (args.target == "auto" and default_shape)) and world == 1 and not args.force_dist
    if want_target and rank == 0:
        # The configured workload's line is final at this point: it goes to stderr (and to a side file) BEFORE the target measurement starts,
        # so that no failure of that measurement -- which runs in a child process with its own timeout -- can lose it.
        log("line before the target measurement: " + json.dumps(out))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_line_before_target.json"), "w") as f:
                f.write(json.dumps(out) + "\n")
        except OSError:
            pass
        free_b, total_b = torch.cuda.mem_get_info()
        need = 230e9 * (args.target_reads / 5e8)
        if total_b < need:
            out["target"] = {"skipped": "the device has %.0f GB of memory, the target size needs %.0f GB" % (total_b / 1e9, need / 1e9)}
        else:
            out["target"] = target_record(args, out)
    if rank == 0:
        out["process_seconds"] = round(time.time() - T_START, 1)       # this process from its start to this line (the driver's clock sees launch + this)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        # the numbers a reader of a truncated log needs, as the LAST line on stderr (the JSON line above is ~16 KB)
        tg = out.get("target") or {}
        summary = {"value": out.get("value"), "ms_per_step": out.get("ms_per_step"), "host_to_host_value": out.get("host_to_host_value"), "n_gpus": out.get("n_gpus"),
                   "search": (out.get("config") or {}).get("search"), "roofline_frac": (out.get("roofline") or {}).get("frac"), "verified": out.get("verified"),
                   "target_value": tg.get("value"), "target_ms_per_step": tg.get("ms_per_step"), "target_host_to_host_value": tg.get("host_to_host_value"), "target_verified": tg.get("verified")}
        sys.stderr.flush()
        os.write(2, ("[bench] summary " + json.dumps(summary) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def target_record(args, base):
    """The target-size measurement as a record of the line: resident value, host to host (full + compact), verification, roofline of
    the dominant kernel, and the ratio to the CPU baseline of the same run.  Measured by a CHILD process (this one has released its device
    memory; it keeps only its HIP context): a hang, an abort or an out-of-memory kill there costs the record, not the line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--reads", str(args.target_reads), "--readlen", str(args.readlen), "--leaf-reads", str(args.leaf_reads),
           "--steps", str(max(1, args.target_steps)), "--warmup", "1", "--host-steps", "1", "--verify-reads", str(args.verify_reads),
           "--no-cpu-baseline", "--target", "off", "--as-target"]
    if args.no_verify:
        cmd.append("--no-verify")
    if args.no_host:
        cmd.append("--no-host")
    t0 = time.time()
    try:
        run = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=args.target_timeout)         # stderr (the progress log) passes through
        lines = [l for l in run.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        if run.returncode != 0 or not lines:
            return {"failed": "the child process ended with code %d and %d JSON lines" % (run.returncode, len(lines)), "seconds": round(time.time() - t0, 1)}
        full = json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {"failed": "stopped after %d s (--target-timeout)" % args.target_timeout, "seconds": round(time.time() - t0, 1)}
    except (OSError, ValueError) as e:
        return {"failed": repr(e), "seconds": round(time.time() - t0, 1)}
    rec = {k: full.get(k) for k in ("value", "value_basis", "host_to_host_value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline", "job_roofline",
                                    "kernel_ms_per_step", "host_to_host", "peak_device_bytes", "verified", "verification")}
    rec["process"] = "child"
    rec["seconds"] = round(time.time() - t0, 1)
    cpu = base.get("cpu_baseline")
    if cpu and cpu.get("value"):
        h2h = full.get("host_to_host") or {}
        rec["vs_cpu_baseline"] = {"cpu_value": cpu["value"], "cpu_cores": cpu["cores"], "resident": round(full["value"] / cpu["value"], 1),
                                  "host_to_host": (round(h2h["value"] / cpu["value"], 1) if h2h.get("value") else None),
                                  "host_to_host_compact": (round(h2h["compact_samples"]["value"] / cpu["value"], 1) if h2h.get("compact_samples") else None),
                                  "north_star": ">= 30 x the reference CPU Gbases/s at this size on one GPU"}
    return rec


def measure(env, args):
    """One workload: inputs, timed steps, roofline, verification, host to host, CPU baseline.  Returns the line's dict on rank 0."""
    pkg, synth, np, torch, dist, dev, rank, world, sharded = env.pkg, env.synth, env.np, env.torch, env.dist, env.dev, env.rank, env.world, env.sharded
    is_target = bool(getattr(args, "is_target", False))

    def barrier():
        pkg.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---------------------------------------------------------------- inputs (untimed)
    # Built on the device by the merge tree, encoded to the native format, parked in page-locked host memory (the
    # host-to-host leg starts from there) and copied back into plain device buffers for the HBM-resident leg, so that the
    # library holds nothing of the inputs when a step starts.
    t_gen = time.time()
    wargs = ({"coverage": args.coverage, "error_percent": args.error_percent} if args.workload == "genome" else {})
    nsets = max(2, args.chain)
    reads_per_set = [args.reads_a or args.reads] + [args.reads] * (nsets - 1)
    seeds = [1001 + k for k in range(nsets)]
    host_in, dev_in, meta = [], [], []

    def build_inputs():
        for k, seed in enumerate(seeds):
            def progress(done, total, k=k):
                if rank == 0 and (done == total or (done // args.leaf_reads) % 64 == 0):
                    log("input%d: %d / %d reads (%.0f s)" % (k + 1, done, total, time.time() - t_gen))
            ix = synth.build_index(pkg, seed, reads_per_set[k], args.readlen, leaf_reads=args.leaf_reads, device=dev, progress=progress,
                                   workload=args.workload, native=not args.torch_leaves, **wargs)
            ix.encode()
            hb = pkg.HostBuffer(ix.nbytes)
            ix.download_into(hb.array)
            meta.append({"sequences": ix.sequences, "bases": ix.bases, "nbytes": ix.nbytes, "C": ix.C})
            if args.partitioned:
                meta[-1]["cum"] = ix.samples()[1]                  # the samples of BWT::build: what the reference's loaded FMI holds next to the bytes
            ix.free()
            host_in.append(hb)
        torch.cuda.empty_cache()
        if args.same_device:
            pkg.trim()

    if args.same_device and world > 1:
        for r in range(world):                                  # ranks that share one GPU build their inputs one after the other
            if r == rank:
                build_inputs()
            dist.barrier()
    else:
        build_inputs()
    torch.cuda.empty_cache()
    if not args.keep_pool:
        pkg.trim()
    part_env = None
    if args.partitioned:
        # One part per rank.  The group lives as long as the process; the HBM-resident inputs of a part are its byte SHARES (the 64-byte blocks
        # that cover its windows): no rank holds a whole input on its device.  The cuts are computed again inside every timed step.
        from bwt_merge_amd import partitioned as bwtm_parts
        if nsets != 2:
            raise SystemExit("--search partitioned measures single merges (--chain 2)")
        gname = None
        if world > 1:
            # one shared-memory name for all ranks, whoever launched them (bench.py itself, torch.distributed.run): rank 0 makes it up and the
            # process group carries it to the others (a name derived from the environment could meet a leftover of an earlier run)
            obj = ["/bwtm-bench-%d-%x" % (os.getpid(), int(time.time() * 1e6) & 0xFFFFFFFFFF) if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0, device=(args.coll_dev if args.same_device else dev))
            gname = obj[0]
        group = pkg.Group(gname, rank, world)
        hidx = [pkg.host_index(host_in[k].array, meta[k]["cum"], meta[k]["sequences"], meta[k]["bases"]) for k in range(2)]
        cuts0 = pkg.partition_cuts_host(hidx[0], hidx[1], world)
        probe = pkg.Part(group, hidx[0], hidx[1], cuts0[0], cuts0[1])
        share_meta = [probe.share(k, hidx[k]) for k in range(2)]
        probe.free()
        for k, (off, count, fp, before) in enumerate(share_meta):
            t = torch.empty(count + 16, dtype=torch.uint8, device=dev)
            t[:count].copy_(torch.from_numpy(host_in[k].array[off: off + count]))
            t[count:].zero_()
            dev_in.append(t)
        part_env = types.SimpleNamespace(mod=bwtm_parts, group=group, hidx=hidx, cuts0=cuts0, share_meta=share_meta, stats=[], cuts_ms=[], transcode_ms=[])
    else:
        for hb in host_in:
            t = torch.empty(hb.nbytes + 16, dtype=torch.uint8, device=dev)       # 16 readable bytes after the stream (borrowed form)
            t[: hb.nbytes].copy_(torch.from_numpy(hb.array))
            t[hb.nbytes:].zero_()
            dev_in.append(t)
    torch.cuda.synchronize()
    n_a, n_b = meta[0]["bases"], meta[1]["bases"]
    m_a, m_b = meta[0]["sequences"], meta[1]["sequences"]
    bytes_a, bytes_b = meta[0]["nbytes"], meta[1]["nbytes"]
    # bases that pass through merges in one step: (input1 + input2) for a single merge; the running total + the increment for every merge of a chain
    merged_bases, acc = 0, n_a
    for k in range(1, nsets):
        acc += meta[k]["bases"]; merged_bases += acc
    searched_bases = sum(mt["bases"] for mt in meta[1:])
    if rank == 0:
        free_b, total_b = torch.cuda.mem_get_info()
        log("device memory before the first step: %.1f GB free of %.1f; torch holds %.1f GB (reserved), the inputs take %.1f GB, the library's pool %.1f GB" %
            (free_b / 1e9, total_b / 1e9, torch.cuda.memory_reserved() / 1e9, sum(t.numel() for t in dev_in) / 1e9, pkg.pool_stats().get("held_bytes", 0) / 1e9))
        log("inputs ready in %.0f s: %s bases, %s GB native (%.3f bytes/base)" %
            (time.time() - t_gen, " + ".join(str(mt["bases"]) for mt in meta), " + ".join("%.3f" % (mt["nbytes"] / 1e9) for mt in meta),
             sum(mt["nbytes"] for mt in meta) / sum(mt["bases"] for mt in meta)))

    # shard of input2's sequences for this rank (getBounds, utils.cpp:169-187)
    from bwt_merge_amd.dist import shard_range, merge_sharded
    seq_first, seq_last = shard_range(m_b, rank, world)

    def load_input(k):
        # BWT::load of an input: the resident native bytes are read in place (no second copy in HBM)
        return pkg.Index.from_device(dev_in[k].data_ptr(), meta[k]["nbytes"], meta[k]["sequences"], meta[k]["bases"], borrow=True)

    def load_inputs():
        return load_input(0), load_input(1)

    shard_times = {}                                 # phases of the sharded merges of this rank (N > 1 or --force-dist)

    # ---------------------------------------------------------------- one step
    def step_partitioned(keep=False):
        # the whole merge as this rank's part: cuts (host rank queries on the inputs' samples), windows transcoded from the resident byte shares,
        # the search in lock step with the other ranks, the second half
        tc = time.perf_counter()
        cuts = pkg.partition_cuts_host(part_env.hidx[0], part_env.hidx[1], world)
        part_env.cuts_ms.append((time.perf_counter() - tc) * 1e3)
        if cuts != part_env.cuts0:
            raise RuntimeError("the cuts changed between two steps")
        shares = []
        for k in range(2):
            off, count, fp, before = part_env.share_meta[k]
            shares.append((dev_in[k].data_ptr(), count, fp, before))
        S, st = part_env.mod.merge_part(part_env.group, part_env.hidx[0], part_env.hidx[1], cuts=cuts, shares=shares)
        pkg.synchronize()
        part_env.stats.append(st)
        if keep:
            return S
        S.free()
        return None

    def step(keep=False):
        if args.partitioned:
            return step_partitioned(keep)
        A, B = load_inputs()
        if not sharded:
            for k in range(2, nsets):            # a chain: the running result is a device rank structure, the next increment is loaded
                A = synth.merge_indexes(pkg, A, B)
                B = load_input(k)
            M = pkg.merge_consume(A, B)          # "merges a and b, destroying them": their records are released after the interleave
        else:
            M = merge_sharded(pkg, A, B, rank, world, dist, torch, dev, times=shard_times)      # this rank's slice of the result
            A.free(); B.free()
        pkg.synchronize()
        if keep:
            return M
        M.free()
        return None

    if args.partitioned and world > 1:
        # The first partitioned merge of the run.  It has never met two physical GPUs: when it fails on ANY rank -- the
        # parts' buffers cannot be mapped between the devices, a part runs out of room -- every rank gets an error from the same collective step,
        # and the run goes on with sequence blocks instead of dying without a line; `config.search` and `partitioned_fallback` say so.
        failed, why = 0, None
        try:
            step()
        except pkg.BwtmError as e:
            failed, why = 1, str(e)
        flag = torch.tensor([failed], dtype=torch.int64, device=args.coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) != 0:
            if args.same_device:
                raise SystemExit("the partitioned merge failed (%s) and --same-device has no other path" % why)
            log("rank %d: the merge over partitioned records failed (%s); this run goes on with sequence blocks" % (rank, why or "on another rank"))
            args.partitioned_fallback = why or "failed on another rank"
            args.partitioned = False
            try:
                part_env.group.free()
            except Exception:                                     # noqa: BLE001
                pass
            part_env = None
            dev_in.clear(); torch.cuda.empty_cache(); pkg.trim()
            for hb in host_in:
                t = torch.empty(hb.nbytes + 16, dtype=torch.uint8, device=dev)
                t[: hb.nbytes].copy_(torch.from_numpy(hb.array))
                t[hb.nbytes:].zero_()
                dev_in.append(t)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    pkg.device_bytes_peak(reset=True)
    shard_times.clear()
    if part_env is not None:
        part_env.stats.clear(); part_env.cuts_ms.clear()
    # Inside the timed region only the dominant kernel's launches are bracketed by HIP events (two event records per launch cost
    # microseconds each on the stream, and a search is ~330 launches); the per-kernel table comes from one more, untimed step.
    pkg.profile_only(None if args.profile_all else "frontier_step,lf_walk")
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
    shard_phases = None
    if sharded and shard_times.get("merges"):
        # rank 0's view, per merge of the timed region; the maximum over ranks of the whole step is `ms_per_step`
        shard_phases = {k: round(shard_times[k] / shard_times["merges"], 2) for k in ("ms_search", "ms_exchange", "ms_interleave_encode")}
        shard_phases["exchange_bytes_per_gpu"] = int(shard_times["exchange_bytes_per_gpu"])
    part_phases = None
    timed_stats = list(part_env.stats) if part_env is not None else []
    if timed_stats:
        n = len(timed_stats)
        part_phases = {"ms_cuts_on_the_host": round(sum(part_env.cuts_ms) / max(1, len(part_env.cuts_ms)), 3),
                       "ms_search": round(sum(x["ms_search"] for x in timed_stats) / n, 2), "ms_search_waiting_for_peers": round(sum(x["ms_search_wait"] for x in timed_stats) / n, 2),
                       "ms_finish": round(sum(x["ms_finish"] for x in timed_stats) / n, 2),
                       "lf_steps": int(timed_stats[-1]["steps"]), "node_levels": int(timed_stats[-1]["node_levels"]),
                       "elements_advanced": int(timed_stats[-1]["elements"]), "largest_frontier": int(timed_stats[-1]["largest"]),
                       "bytes_read_from_the_parts_buffers_per_merge": int(timed_stats[-1]["pulled_bytes"]), "boundary_bytes_per_pair": int(timed_stats[-1]["boundary_bytes"]),
                       "record_bytes_held": int(timed_stats[-1]["record_bytes"]), "bitvector_bytes_held": int(timed_stats[-1]["bitvector_bytes"])}
    prof_all, prof_all_steps = prof, max(1, args.steps)
    if not args.profile_all:
        if last is not None and (n_a + n_b) > 4e10:
            last.free(); last = None                                   # no room for a second result next to the first at this size
        pkg.profile_only(None)
        pkg.profile_reset()
        extra = step(keep=(last is None))
        if last is None:
            last = extra
        prof_all, prof_all_steps = pkg.profile_read(), 1
        barrier()
    pkg.profile_enable(False)
    peak_device = pkg.device_bytes_peak() + sum(t.numel() for t in dev_in)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    sec_per_step = elapsed / max(1, args.steps)
    value = merged_bases / 1e9 / sec_per_step

    # ---------------------------------------------------------------- roofline of the dominant kernel
    # The search kernel: k_frontier_step (one launch per LF step of the level-synchronous search) or,
    # when the library falls back to it, k_lf_walk_binned (one launch per search).  SURVEY.md 8(d):
    # 160 algorithmic bytes per LF step (one 64-byte block + 8 + 8 bytes on each side).
    dom = "frontier_step" if "frontier_step" in prof else "lf_walk"
    dom_ms, dom_launches = prof.get(dom, (0.0, 0))
    searches = max(1, args.steps)
    units_per_search = ((seq_last - seq_first + 1) / max(1, m_b) * n_b if seq_first <= seq_last else 0) if nsets == 2 else searched_bases
    # LF steps taken on trie NODES (k_range_step, one launch per level) are not this kernel's: every sequence of the shard is alive on
    # each of those first levels (they are far fewer than the shortest read is long), so they account for levels x sequences steps
    node_levels = (prof_all.get("range_step", (0.0, 0))[1] / prof_all_steps / max(1, nsets - 1)) if dom == "frontier_step" else 0
    if nsets == 2 and dom == "frontier_step":
        units_per_search = max(0.0, units_per_search - node_levels * (seq_last - seq_first + 1))
    if timed_stats:
        units_per_search = float(timed_stats[-1]["elements"])           # what THIS rank's step kernels advanced (its share of every step)
        node_levels = float(timed_stats[-1]["node_levels"])
    launches_per_search = dom_launches / searches if dom_launches else 0
    avg_launch_s = (dom_ms / 1e3 / dom_launches) if dom_launches else float("nan")
    units_per_launch = units_per_search / launches_per_search if launches_per_search else 0
    achieved = SEARCH_BYTES_PER_BASE * units_per_launch / avg_launch_s / 1e9 if dom_launches else 0.0
    # HBM bytes of the dominant kernel from the stored PMC passes -- used only when they were collected on THIS code (hash of the kernel's
    # sources), with the same knobs, for the same workload, and the run reproduces the launches and LF steps of the profiled search.
    traffic, traffic_source, traffic_gbs, traffic_why = None, None, None, None
    entry, traffic_why = stored_traffic(dom, args, world, nsets, launches_per_search, units_per_search)
    if entry is not None:
        # per launch over ALL launches of a search (it ends with a few empty ones: the host learns the frontier size with a delay),
        # like avg_launch_ms; traffic_GBs = bytes of a search / kernel time of a search
        traffic, traffic_source = entry["hbm_bytes_per_launch"], entry["source"]
        traffic_gbs = entry["hbm_bytes_per_search"] / (dom_ms / searches / 1e3) / 1e9
    # What this design must move per launch at the least (the "design floor"): 22 bytes of coordinates and emit per element (10 in,
    # 10 out, 2 emit; 18 without the high bytes) + the DISTINCT 64-byte records of both indexes that hold an element.  Along the
    # sorted frontier N elements spread over R records touch R (1 - exp(-N / R)) of them (0.72 R at config 2).  HBM traffic above
    # this floor is waste (untouched neighbours in 128-byte lines, re-reads); the contract's algorithmic count (160 B per LF step,
    # every step charged its own two blocks) is ABOVE it because neighbours share records.
    import math
    floor_bytes = None
    if dom == "frontier_step" and units_per_launch > 0:
        wide = (n_a >= (1 << 32) or n_b >= (1 << 32))
        def distinct(n_pos):
            recs = n_pos / 128.0 + 1
            return recs * (1.0 - math.exp(-units_per_launch / recs))
        floor_bytes = units_per_launch * (22 if wide else 18) + 64.0 * (distinct(n_a if nsets == 2 else merged_bases / max(1, nsets - 1)) + distinct(n_b))
    # `frac` = HBM bytes really moved per second / peak: from the PMC counters when they were collected for exactly this
    # configuration and code (profiles/search_kernel_traffic.json), otherwise from the design floor (a lower bound of the traffic).
    # `algorithmic_frac` is the contract's figure (SURVEY 8(d)'s 160 B per LF step / measured duration): it exceeds the real
    # utilisation -- and can pass 1.0 -- because the sorted frontier shares records between neighbouring elements.
    if traffic_gbs:
        frac, basis = traffic_gbs / HBM_PEAK_GBS, ("HBM bytes of a stored PMC profile of this code and configuration (rocprofv3 passes: FETCH_SIZE, WRITE_SIZE with the gfx950 "
                                                   "corrections; code hash, knobs, launches and LF steps checked) / duration measured live in this run")
    elif floor_bytes:
        frac, basis = floor_bytes / avg_launch_s / 1e9 / HBM_PEAK_GBS, "design floor bytes (no usable PMC profile: %s) / duration measured in this run" % traffic_why
    else:
        frac, basis = achieved / HBM_PEAK_GBS, "algorithmic bytes (SURVEY 8(d): 160 B per LF step; the per-chain walk fetches exactly these) / duration measured in this run"
    roofline = {"bound": "hbm", "kernel": "k_" + dom + ("_binned" if dom == "lf_walk" else ""),
                "achieved": round((traffic_gbs if traffic_gbs else frac * HBM_PEAK_GBS), 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(frac, 4), "frac_basis": basis,
                "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_GBs": round(achieved, 1), "algorithmic_frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_note": "160 B per LF step x LF steps of a launch / the launch's duration: every step is charged its own two 64-byte blocks "
                                    "(the reference's access pattern); neighbouring elements of the sorted frontier share records, so this exceeds the bytes moved",
                "algorithmic_bytes_per_launch": SEARCH_BYTES_PER_BASE * units_per_launch,
                "design_floor_bytes_per_launch": (round(floor_bytes) if floor_bytes else None),
                "traffic_over_design_floor": (round(traffic / floor_bytes, 3) if traffic and floor_bytes else None),
                "traffic_over_algorithmic": (round(traffic / (SEARCH_BYTES_PER_BASE * units_per_launch), 3) if traffic and units_per_launch else None),
                "traffic_profile_check": ("matched" if traffic else traffic_why),
                "copy_ceiling_GBs": 6290.0, "frac_of_copy_ceiling": round(frac * HBM_PEAK_GBS / 6290.0, 4),
                "node_levels": round(node_levels, 2), "lf_steps_of_this_kernel_per_step": int(units_per_search),
                "launches_per_step": round(launches_per_search, 2), "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                "kernel_ms_per_step": round(dom_ms / searches, 3)}
    kernel_ms = {name: round(ms / prof_all_steps, 3) for name, (ms, n) in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
    # whole-job algorithmic bytes W = 176 n_B + |A| + |B| + 2 |Out|
    out_bytes = last.total_nbytes if last is not None else 0
    W = 176 * searched_bases + sum(mt["nbytes"] for mt in meta) + 2 * out_bytes
    job = {"algorithmic_bytes": W, "algorithmic_GBs": round(W / sec_per_step / 1e9, 1), "ratio_of_survey_W_rate_to_peak": round(W / sec_per_step / 1e9 / HBM_PEAK_GBS, 4),
           "note": "SURVEY 8(d)'s W charges every LF step its own two 64-byte blocks; the sorted frontier shares records between neighbouring elements, so the "
                   "bytes really moved are ~0.73 W for the search (roofline.traffic_over_algorithmic) and this ratio can pass 1.0: it is the contract's "
                   "algorithmic figure, not a bandwidth measurement and not an upper bound for this algorithm -- roofline.frac is the measured one"}

    # ---------------------------------------------------------------- verification at full size
    verified, checks = None, {}
    if rank == 0 and last is not None and not args.no_verify and not sharded:
        verified, checks = verify_full_size(pkg, synth, np, last, meta, args, wargs, load_inputs)
        log("full-size verification: %s %s" % (verified, checks))
    if sharded and last is not None and not args.no_verify:
        # every rank: its slice of the sharded result == the same byte range of the result it computes alone
        if args.partitioned:
            # (from the host copies: a part's resident inputs are its byte shares; ranks that share one GPU take turns)
            def load_full():
                return tuple(pkg.Index.upload(host_in[k].array, meta[k]["sequences"], meta[k]["bases"]) for k in range(2))
            ok = True
            for r in range(world if args.same_device else 1):
                if not args.same_device or r == rank:
                    ok = verify_slice(pkg, np, last, load_full)
                    pkg.trim()
                if args.same_device:
                    dist.barrier()
        else:
            ok = verify_slice(pkg, np, last, load_inputs)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=args.coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item() == 1)
        checks = {"every_slice_equals_the_single_gpu_result": verified, "slices": world}
        if rank == 0:
            log("sharded verification: %s" % verified)
    chain_tail = None
    if last is not None and nsets > 2 and not sharded and rank == 0 and not args.no_host and not args.no_verify:
        # the tail of the device-resident chain's stream (it depends on everything before it): what the host chain must reproduce
        ptr, nb = last.device_data()
        ref = np.zeros(min(nb, 1 << 26), dtype=np.uint8)
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        if hip.hipMemcpy(ref.ctypes.data, ptr + (nb - ref.size), ref.size, 2) == 0:
            chain_tail = (nb, ref)
    if last is not None:
        last.free()

    # ---------------------------------------------------------------- host to host (SURVEY 8(d)'s T), rank 0 at N == 1
    host = None
    dev_in.clear()                                   # the device copies of the inputs are not needed any more
    from bwt_merge_amd import dist as bwtm_dist
    bwtm_dist.release_buffers()                      # the cached bitvector of the sharded steps
    torch.cuda.empty_cache()
    if rank == 0 and not sharded and not args.no_host and nsets == 2:
        host = host_to_host(pkg, np, torch, dev, host_in, meta, args)
    if sharded and not args.no_host and nsets == 2 and args.partitioned:
        host = host_to_host_partitioned(pkg, np, torch, args, host_in, meta, rank, world, dist, part_env)      # collective
    elif sharded and not args.no_host and nsets == 2:
        host = host_to_host_sharded(pkg, np, torch, dev, host_in, meta, args, rank, world, dist)       # collective: every rank takes part
    if rank == 0 and not sharded and not args.no_host and nsets > 2:
        host = host_chain(pkg, np, torch, host_in, meta, args, chain_tail)

    # ---------------------------------------------------------------- CPU baseline (rank 0, N == 1)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.empty_cache(); pkg.trim()
        cpu = cpu_baseline(pkg, synth, torch, np, dev, args)

    rccl_ranks = None
    if dist is not None:
        one_t = torch.ones(1, dtype=torch.int64, device=args.coll_dev)
        dist.all_reduce(one_t, op=dist.ReduceOp.SUM)                       # counted by the collective itself, not read from the environment
        rccl_ranks = int(one_t.item())
    if part_env is not None:
        part_env.group.barrier()
        part_env.group.free()
    for hb in host_in:
        hb.free()
    host_in.clear()
    torch.cuda.empty_cache(); pkg.trim()
    if rank != 0:
        return None
    wname = {"iid": "sigma=6", "genome": "reads from a shared random genome, %dx coverage, %d%% substitutions" % (args.coverage, args.error_percent),
             "mixed": "sigma=6, 100 / 150 bp mixed"}[args.workload]
    return {
        "metric": "merged Gbases/sec (input1+input2), bit-exact native BWT; value = inputs and result resident in HBM, host_to_host_value = SURVEY 8(d)'s T (page-locked host to page-locked host)",
        "value": round(value, 4), "value_basis": "hbm_resident", "host_to_host_value": (host or {}).get("value"), "unit": "Gbases/s", "n_gpus": world,
        "rccl_ranks": (None if args.same_device else rccl_ranks), "process_group": (None if dist is None else ("gloo" if args.same_device else "nccl")), "ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(sec_per_step * 1e3, 2), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%s Gbase synthetic %d bp read sets (%s)%s, native format, inputs resident in HBM" %
                   (" + ".join("%.3g" % (mt["bases"] / 1e9) for mt in meta), args.readlen, wname,
                    ", chained merge in command-line order (value = bases through all merges / time)" if nsets > 2 else ""),
                   "reads_per_set": args.reads, "reads_input1": reads_per_set[0], "read_length": args.readlen,
                   "bases": [mt["bases"] for mt in meta], "native_bytes": [mt["nbytes"] for mt in meta] + [out_bytes],
                   "parallelism": (("partitioned records: one part per rank (%d), windows transcoded from the rank's byte shares, the frontier's elements read from the "
                                    "peers' output buffers inside the step kernel (%s), no collective library; result sharded by output range" %
                                    (world, "HIP IPC between the ranks' processes, all on GPU 0" if args.same_device else "peer-mapped memory over xGMI")) if args.partitioned else
                                   "sequence blocks of input2 sharded over %d GPU(s)%s" %
                                   (world, ", RCCL reduce-scatter of the rank-array bitvector by output range, result sharded by output range" if sharded else "")),
                   "search": ("partitioned" if args.partitioned else ("blocks" if sharded else "single")), "same_device": bool(args.same_device),
                   "partitioned_fallback": getattr(args, "partitioned_fallback", None)},
        "sharded_phases_rank0": shard_phases, "partitioned_phases_rank0": part_phases,
        "roofline": roofline, "job_roofline": job, "kernel_ms_per_step": kernel_ms,
        "host_to_host": host, "peak_device_bytes": peak_device,
        "cpu_baseline": cpu, "verified": verified, "verification": checks,
    }


SEARCH_KERNEL_SOURCES = ("bwt-merge_amd/csrc/kernels/search_frontier.hip.h", "bwt-merge_amd/csrc/kernels/search_walk.hip.h",
                         "bwt-merge_amd/csrc/kernels/common.hip.h", "bwt-merge_amd/csrc/bwtm_device.h",
                         "bwt-merge_amd/csrc/api/search.hip.h")            # launch geometry and knobs of the search change its traffic too


def search_code_hash():
    """sha256 over the sources the search kernels are compiled from: a stored PMC profile describes exactly this code."""
    import hashlib
    h = hashlib.sha256()
    for rel in SEARCH_KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


# Knobs that cannot change what k_frontier_step reads or writes per launch: the epoch length only decides how often the dense emits are
# turned into tiles (another kernel); a PMC pass may have needed a smaller emit buffer than the run it describes (the profiler's own
# device memory left too little for the default budget at 2 x 50 Gbase).
TRAFFIC_NEUTRAL_KNOBS = ("emit_budget", "frontier_epoch")


def effective_tune(args):
    """The knobs this run changed (--tune arguments plus the BWTM_TUNE environment of the library), without the traffic-neutral ones."""
    knobs = sorted(kv for kv in args.tune if kv.partition("=")[0] not in TRAFFIC_NEUTRAL_KNOBS)
    if os.environ.get("BWTM_TUNE"):
        knobs.append("env:" + os.environ["BWTM_TUNE"])
    return knobs


def stored_traffic(dom, args, world, nsets, launches_per_search, units_per_search):
    """(entry, None) of profiles/search_kernel_traffic.json that describes this run, or (None, why not)."""
    try:
        with open(os.path.join(ROOT, "profiles", "search_kernel_traffic.json")) as f:
            stored = json.load(f)
    except (OSError, ValueError) as e:
        return None, "profiles/search_kernel_traffic.json unreadable (%s)" % e.__class__.__name__
    if args.workload != "iid" or world != 1 or nsets != 2 or args.reads_a:
        return None, "no PMC passes for this workload"
    why = "no PMC passes for %d reads of %d bp" % (args.reads, args.readlen)
    try:
        code = search_code_hash()
    except OSError:
        return None, "kernel sources not readable"
    for e in stored.get("entries", []):
        try:
            if e["kernel"] != dom or e["config"]["reads_per_set"] != args.reads or e["config"]["read_length"] != args.readlen:
                continue
            if e.get("code_hash") != code:
                why = "the kernel sources changed since the PMC passes (code hash differs)"
            elif [kv for kv in e.get("tune", []) if kv.partition("=")[0] not in TRAFFIC_NEUTRAL_KNOBS] != effective_tune(args):
                why = "the PMC passes ran with other knobs (%s)" % (e.get("tune") or "defaults")
            elif abs(e["launches_per_search"] - launches_per_search) >= 0.5:
                why = "%.1f launches per search, the PMC passes saw %d" % (launches_per_search, e["launches_per_search"])
            elif abs(e.get("lf_steps_per_search", units_per_search) - units_per_search) > 1e-6 * max(1.0, units_per_search):
                why = "the PMC passes covered another number of LF steps"
            else:
                return e, None
        except (KeyError, TypeError):
            why = "malformed entry"
    return None, why


def launch_ranks(world, same_device=False):
    """One fresh child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same command line),
    started before this process has made any GPU call -- a process that has initialised the GPU must never be replaced or
    forked on this pool.  Rank 0's stdout (the JSON line) is passed through; the exit code is non-zero when any rank fails."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if same_device else r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    rc = 0
    failed = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc, failed = code, p
                for q in procs:                      # a rank died: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.05)
    if rc != 0:
        log("a rank exited with code %d" % rc)
    return rc if rc >= 0 else 1


def verify_full_size(pkg, synth, np, last, meta, args, wargs, load_inputs):
    """Size-independent checks of the full-size result: C additivity and header, `verify_reads` reads extracted from the merged
    index by LF walk == the generator's, the emitted stream decodes back to the same index, and -- two implementations of the
    search against each other -- the rank array of the level-synchronous frontier search == that of the per-chain walk."""
    checks = {}
    C = sum(mt["C"] for mt in meta)
    checks["header_and_C"] = bool(np.array_equal(last.C, C) and last.bases == sum(mt["bases"] for mt in meta)
                                  and last.sequences == sum(mt["sequences"] for mt in meta))
    rng = np.random.default_rng(12345)
    ids = np.sort(rng.integers(0, last.sequences, args.verify_reads))
    maxlen = (150 if args.workload == "mixed" else args.readlen)
    got = synth.extract_sequences_matrix(last, ids, max_len=maxlen + 2)
    ok = True
    first_seq = 0
    for which, mt in enumerate(meta):                    # sequences of the merged collection are the inputs' sequences in command-line order
        sel = (ids >= first_seq) & (ids < first_seq + mt["sequences"])
        ref = synth.reads_matrix(args.workload, 1001 + which, ids[sel] - first_seq, args.readlen, mt["sequences"], maxlen + 2, **wargs)
        ok = ok and bool(np.array_equal(got[sel], ref))
        first_seq += mt["sequences"]
    checks["extracted_reads"] = ok
    checks["extracted_reads_count"] = int(ids.size)
    # the emitted native stream must decode back to the merged index (header check in upload)
    p, nb = last.device_data()
    R = pkg.Index.from_device(p, nb, last.sequences, last.bases, borrow=True)
    w0 = int(rng.integers(0, max(1, last.bases - (1 << 20))))
    cnt = min(1 << 20, last.bases)
    checks["stream_decodes_back"] = bool(np.array_equal(R.extract(w0, cnt), last.extract(w0, cnt)))
    R.free()
    if meta[1]["bases"] <= 2e10 and len(meta) == 2:
        A, B = load_inputs()
        bits = []
        for algo in (2, 1):
            pkg.tune("search_algo", algo)
            ra = pkg.RankArray(A, B)
            ra.search(A, B, 0, meta[1]["sequences"] - 1)
            ra.finalize()
            bits.append(ra.bits())
            ra.free()
        pkg.tune("search_algo", 0)
        A.free(); B.free()
        checks["frontier_equals_walk"] = bool(np.array_equal(bits[0], bits[1]))
        del bits
    elif len(meta) == 2:
        # Too large for two full bitvectors on the host: the level-synchronous search of the WHOLE collection against the per-chain
        # walk of a 1 % block of its sequences, compared on the device -- every bit the walk sets must be set by the frontier
        # search, and the walk must set exactly one bit per position of its sequences.
        last.free()                                            # the result (126 GB at 2 x 50 Gbase) makes room for a second pair of record arrays
        pkg.trim()
        A, B = load_inputs()
        m_b = meta[1]["sequences"]
        s0 = int(rng.integers(0, max(1, m_b - m_b // 100)))
        s1 = min(m_b - 1, s0 + max(1, m_b // 100) - 1)
        pkg.tune("search_algo", 2)
        whole = pkg.RankArray(A, B); whole.search(A, B, 0, m_b - 1)
        pkg.tune("search_algo", 1)
        part = pkg.RankArray(A, B); part.search(A, B, s0, s1)
        pkg.tune("search_algo", 0)
        ones, outside = part.subset_check(whole)
        part.free(); whole.free(); A.free(); B.free()
        per_seq = meta[1]["bases"] // m_b
        checks["walk_of_a_shard_inside_frontier"] = bool(outside == 0 and (ones == (s1 - s0 + 1) * per_seq or args.workload != "iid"))
        checks["walk_shard_sequences"] = int(s1 - s0 + 1)
    return all(v for k, v in checks.items() if isinstance(v, bool)), checks


def verify_slice(pkg, np, slice_, load_inputs):
    """The bytes of this rank's output slice against the same range of the stream the rank computes on its own."""
    A, B = load_inputs()
    full = pkg.merge_consume(A, B)
    ok = (full.nbytes == slice_.total_nbytes)
    first, count = slice_.byte_first, slice_.nbytes
    if ok and count > 0:
        ptr, _ = full.device_data()
        ref = np.zeros(count, dtype=np.uint8)
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        ok = (hip.hipMemcpy(ref.ctypes.data, ptr + first, count, 2) == 0) and bool(np.array_equal(ref, slice_.data()))
    full.free()
    return ok


def host_to_host(pkg, np, torch, dev, host_in, meta, args):
    """T of SURVEY.md 8(d): page-locked native inputs -> page-locked native result + samples, through bwtm_merge_host."""
    a = (host_in[0].array, meta[0]["sequences"], meta[0]["bases"])
    b = (host_in[1].array, meta[1]["sequences"], meta[1]["bases"])
    torch.cuda.empty_cache(); pkg.trim()
    buffers = {}
    # Warmup: page-locked output buffers, the library's device pool -- and the link: after the seconds of idle time that pinning
    # 14 GB of output buffers takes, the first calls move data at 45 - 70 % of the steady PCIe rate, sometimes on a plateau of
    # several calls (measured: 1218, 813, 799, 808, 569, then 512 +- 1 ms), so calls are repeated until the upload phase runs at
    # 75 % of the link's specified rate or better twice in a row (at most 12 calls).
    in_bytes_w = meta[0]["nbytes"] + meta[1]["nbytes"]
    upload_ok_ms = in_bytes_w / (0.75 * PCIE_SPEC_GBS * 1e9) * 1e3
    warm, good = [], 0
    while len(warm) < 12 and good < 2:
        t0 = time.perf_counter()
        res = pkg.merge_host(a, b, samples=True, buffers=buffers)
        warm.append(time.perf_counter() - t0)
        good = good + 1 if res.times["ms_upload"] <= upload_ok_ms else 0
    log("host to host warmup calls: %s ms" % [round(t * 1e3, 1) for t in warm])
    out_bytes, blocks = res.out.nbytes, res.out.blocks
    times, best = [], None
    for _ in range(max(1, args.host_steps)):
        t0 = time.perf_counter()
        res = pkg.merge_host(a, b, samples=True, buffers=buffers)
        dt = time.perf_counter() - t0
        times.append(dt)
        log("host to host call %d: %.1f ms wall; %s" % (len(times), dt * 1e3, {k: round(v, 1) for k, v in res.times.items()}))
        if best is None or dt <= min(times):
            best = dict(res.times)
    t_data, t_compact, compact_phases, width = [], [], None, None
    quick = bool(getattr(args, "is_target", False))                      # the target-size record: one call less of each kind (seconds each)
    for _ in range(1 if quick else 2):
        t0 = time.perf_counter()
        r2 = pkg.merge_host(a, b, samples=False, buffers=buffers)
        t_data.append(time.perf_counter() - t0)
    for k in range(2 if quick else 3):                                   # the first call allocates the buffers of the compact form
        t0 = time.perf_counter()
        r3 = pkg.merge_host(a, b, samples=2, buffers=buffers)
        if k > 0:
            t_compact.append(time.perf_counter() - t0)
        compact_phases, width = dict(r3.times), r3.out.sample_width
        log("host to host, compact samples, call %d: %s" % (k + 1, {kk: round(v, 1) for kk, v in r3.times.items()}))
    # PCIe calibration: plain page-locked copies of the same buffers
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n = host_in[0].nbytes
    scratch = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    cal = {}
    for name, dst, src, kind in (("h2d_GBs", scratch.data_ptr(), host_in[0].ptr, 1), ("d2h_GBs", buffers[0].ptr, scratch.data_ptr(), 2)):
        best_t = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            hip.hipMemcpy(dst, src, min(n, buffers[0].nbytes), kind)
            best_t = min(best_t, time.perf_counter() - t0)
        cal[name] = round(min(n, buffers[0].nbytes) / best_t / 1e9, 1)
    del scratch
    merged = meta[0]["bases"] + meta[1]["bases"]
    sec = sum(times) / len(times)
    in_bytes = meta[0]["nbytes"] + meta[1]["nbytes"]
    sample_bytes = 8 * blocks + 48 * (blocks + 1)
    floor_ms = (in_bytes / (cal["h2d_GBs"] * 1e9) + (out_bytes + sample_bytes) / (cal["d2h_GBs"] * 1e9)) * 1e3
    host = {"value": round(merged / 1e9 / sec, 4), "unit": "Gbases/s", "ms_per_step": round(sec * 1e3, 2), "steps": len(times),
            "ms_each": [round(t * 1e3, 1) for t in times], "warmup_ms_each": [round(t * 1e3, 1) for t in warm],
            "includes": "H2D of both native inputs, transcode, search, interleave, encode, D2H of the native result and of its samples "
                        "(block_end + 6 cumulative arrays, 56 bytes per 64-byte block)",
            "phases_ms": {k: round(v, 2) for k, v in best.items()},
            "data_only": {"value": round(merged / 1e9 / min(t_data), 4), "ms_per_step": round(min(t_data) * 1e3, 2),
                          "note": "without the D2H of the samples"},
            "compact_samples": {"value": round(merged / 1e9 / (sum(t_compact) / len(t_compact)), 4), "ms_per_step": round(sum(t_compact) / len(t_compact) * 1e3, 2),
                                "sample_width": width, "phases_ms": {k: round(v, 2) for k, v in compact_phases.items()},
                                "note": "samples as %d-byte fields + anchors (%d bytes per block instead of 56): what the C++ facade downloads" % (width, 6 * width + 1)},
            "bytes": {"h2d": in_bytes, "d2h_data": out_bytes, "d2h_samples": sample_bytes},
            "pcie": dict(cal, spec_GBs=PCIE_SPEC_GBS, transfer_floor_ms=round(floor_ms, 2))}
    log("host to host: %.1f ms per merge (%.2f Gbases/s); phases %s; pcie %s" % (sec * 1e3, host["value"], host["phases_ms"], host["pcie"]))
    for bfr in buffers.values():
        bfr.free()
    return host


def host_to_host_sharded(pkg, np, torch, dev, host_in, meta, args, rank, world, dist):
    """SURVEY 8(d)'s T on N GPUs: page-locked inputs -> this rank's byte range of the native result in page-locked memory.  Every
    native byte crosses PCIe once (each rank uploads 1 / N of both inputs, the parts are all-gathered over xGMI), the sequences
    of input2 are sharded, the rank arrays are combined by one reduce-scatter by output range and every rank encodes and downloads its output slice.
    Data only (the samples of a slice need its successor's first block start; the C++ host downloads them too)."""
    from bwt_merge_amd.dist import merge_sharded, upload_sharded
    torch.cuda.empty_cache(); pkg.trim()
    out_buf, times, phases, from_host = None, [], None, 0
    for it in range(2 + max(1, args.host_steps)):
        pkg.synchronize(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        A, la = upload_sharded(pkg, host_in[0].array, meta[0]["sequences"], meta[0]["bases"], rank, world, dist, torch, dev)
        B, lb = upload_sharded(pkg, host_in[1].array, meta[1]["sequences"], meta[1]["bases"], rank, world, dist, torch, dev)
        t1 = time.perf_counter()
        S = merge_sharded(pkg, A, B, rank, world, dist, torch, dev)
        A.free(); B.free()
        t2 = time.perf_counter()
        if out_buf is None or out_buf.nbytes < S.nbytes:
            if out_buf is not None:
                out_buf.free()
            out_buf = pkg.HostBuffer(S.nbytes + (1 << 20))
        S.data_into(out_buf.array)
        total = S.total_nbytes
        S.free()
        pkg.synchronize(); torch.cuda.synchronize(); dist.barrier()
        t3 = time.perf_counter()
        t = torch.tensor([t3 - t0, t1 - t0, t2 - t1, t3 - t2], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if it >= 2:
            times.append(float(t[0].item()))
            phases = {"ms_upload_allgather_transcode": round(float(t[1].item()) * 1e3, 2), "ms_search_exchange_encode": round(float(t[2].item()) * 1e3, 2),
                      "ms_download": round(float(t[3].item()) * 1e3, 2)}
        from_host = la + lb
        if rank == 0:
            log("host to host on %d GPUs, round %d: %.1f ms" % (world, it, float(t[0].item()) * 1e3))
    if out_buf is not None:
        out_buf.free()
    merged = meta[0]["bases"] + meta[1]["bases"]
    sec = sum(times) / len(times)
    return {"value": round(merged / 1e9 / sec, 4), "unit": "Gbases/s", "ms_per_step": round(sec * 1e3, 2), "ms_each": [round(x * 1e3, 1) for x in times],
            "phases_ms": phases, "includes": "sharded H2D of both native inputs (1 / N per link) + all-gather, transcode, sharded search, reduce-scatter, "
            "interleave + encode of the rank's output slice, D2H of the slice; without the samples",
            "bytes": {"h2d_this_rank": int(from_host), "h2d_all_inputs": meta[0]["nbytes"] + meta[1]["nbytes"], "d2h_all_ranks": int(total)}}


def host_to_host_partitioned(pkg, np, torch, args, host_in, meta, rank, world, dist, part_env):
    """SURVEY 8(d)'s T over partitioned records: page-locked inputs -> this rank's byte range of the native result in page-locked memory.  Every
    rank uploads only the 64-byte blocks that cover its windows (its PCIe link carries 1 / N of the inputs and two margins), transcodes them, searches
    in lock step with the others, encodes its range of the output and downloads it.  Data only, like host_to_host_sharded."""
    torch.cuda.empty_cache(); pkg.trim()
    out_buf, times, phases, from_host, total = None, [], None, 0, 0
    for it in range(2 + max(1, args.host_steps)):
        pkg.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        cuts = pkg.partition_cuts_host(part_env.hidx[0], part_env.hidx[1], world)
        S, st = part_env.mod.merge_part(part_env.group, part_env.hidx[0], part_env.hidx[1], cuts=cuts)
        t1 = time.perf_counter()
        if out_buf is None or out_buf.nbytes < S.nbytes:
            if out_buf is not None:
                out_buf.free()
            out_buf = pkg.HostBuffer(S.nbytes + (1 << 20))
        S.data_into(out_buf.array)
        total = S.total_nbytes
        S.free()
        pkg.synchronize(); dist.barrier()
        t2 = time.perf_counter()
        t = torch.tensor([t2 - t0, t1 - t0, t2 - t1], dtype=torch.float64, device=args.coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if it >= 2:
            times.append(float(t[0].item()))
            phases = {"ms_cuts_upload_transcode_search_encode": round(float(t[1].item()) * 1e3, 2), "ms_download": round(float(t[2].item()) * 1e3, 2)}
        from_host = sum(part_env.share_meta[k][1] for k in range(2))
        if rank == 0:
            log("host to host over partitioned records on %d ranks, round %d: %.1f ms" % (world, it, float(t[0].item()) * 1e3))
    if out_buf is not None:
        out_buf.free()
    merged = meta[0]["bases"] + meta[1]["bases"]
    sec = sum(times) / len(times)
    return {"value": round(merged / 1e9 / sec, 4), "unit": "Gbases/s", "ms_per_step": round(sec * 1e3, 2), "ms_each": [round(x * 1e3, 1) for x in times],
            "phases_ms": phases, "includes": "cuts on the host, H2D of the rank's byte shares of both inputs, transcode of its windows, the partitioned search, "
            "interleave + encode of the rank's output range, D2H of the slice; without the samples",
            "bytes": {"h2d_this_rank": int(from_host), "h2d_all_inputs": meta[0]["nbytes"] + meta[1]["nbytes"], "d2h_all_ranks": int(total)}}


def host_chain(pkg, np, torch, host_in, meta, args, chain_tail):
    """A chained merge from page-locked host inputs to a page-locked host result (bwt_merge in1 in2 ... out, bwt_merge.cpp:167-173):
    intermediate results stay on the device; the last merge downloads data + compact samples.  Measured twice: every merge uploading
    its own increment when it starts (`sequential`), and every merge announcing the NEXT increment so that its bytes travel under
    the search (`pipelined`, bwtm_merge_host_pipelined)."""
    torch.cuda.empty_cache(); pkg.trim()
    inp = [(host_in[k].array, meta[k]["sequences"], meta[k]["bases"]) for k in range(len(host_in))]
    n = len(inp)
    buffers = {}

    def run(pipelined):
        t0 = time.perf_counter()
        phases = []
        r, pend = pkg.merge_host_pipelined(a=inp[0], b=inp[1], next=(inp[2] if pipelined else None), samples=pkg.RESULT_ON_DEVICE, keep=True, buffers=buffers)
        phases.append(dict(r.times))
        kept, r.keep = r.keep, None
        for k in range(2, n):
            final = (k == n - 1)
            nxt = (inp[k + 1] if pipelined and not final else None)
            if pipelined:
                r, pend = pkg.merge_host_pipelined(chained=kept, pending=pend, next=nxt, samples=(2 if final else pkg.RESULT_ON_DEVICE), keep=not final, buffers=buffers)
            else:
                r, _ = pkg.merge_host_pipelined(chained=kept, b=inp[k], samples=(2 if final else pkg.RESULT_ON_DEVICE), keep=not final, buffers=buffers)
            phases.append(dict(r.times))
            if not final:
                kept, r.keep = r.keep, None
        return time.perf_counter() - t0, phases, r

    out = {}
    check = None
    for name, pipelined in (("sequential", False), ("pipelined", True)):
        times = []
        for it in range(2 + max(1, args.host_steps)):                  # two warm-up rounds (page-locked output buffers, the pool, the link)
            dt, phases, r = run(pipelined)
            if it >= 2:
                times.append(dt)
            log("host chain (%s) round %d: %.1f ms; upload / search per merge: %s" %
                (name, it, dt * 1e3, [(round(ph["ms_upload"], 1), round(ph["ms_search"], 1)) for ph in phases]))
        if chain_tail is not None:
            nb, ref = chain_tail
            ok = bool(r.out.nbytes == nb and np.array_equal(r.data[nb - ref.size:], ref))
            check = ok if check is None else (check and ok)
        sec = sum(times) / len(times)
        out[name] = {"ms": round(sec * 1e3, 2), "ms_each": [round(t * 1e3, 1) for t in times],
                     "phases_ms_per_merge": [{k: round(v, 1) for k, v in ph.items()} for ph in phases]}
    merged_bases, acc = 0, meta[0]["bases"]
    for k in range(1, n):
        acc += meta[k]["bases"]; merged_bases += acc
    out["value"] = round(merged_bases / 1e9 / (out["pipelined"]["ms"] / 1e3), 4)
    out["unit"] = "Gbases/s (bases through all merges / time, page-locked host inputs -> page-locked host result with compact samples)"
    out["saved_ms"] = round(out["sequential"]["ms"] - out["pipelined"]["ms"], 2)
    out["equals_device_chain"] = check
    out["bytes"] = {"h2d": sum(mt["nbytes"] for mt in meta), "d2h_data": r.out.nbytes}
    for bfr in buffers.values():
        bfr.free()
    return out


def cpu_quota():
    """CPUs' worth of time the container may use per scheduling period (cgroup v2 cpu.max or v1 cfs quota); None = unlimited or unknown.
    The GPU boxes of this pool show 256 logical CPUs to sched_getaffinity and give 16 (profiles/r05_cpu_quota_probe.txt): threads beyond
    the quota only take turns."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            names = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        return "%s (%d logical CPUs)" % (names[0], len(names)) if names else "unknown"
    except OSError:
        return "unknown"


def cpu_baseline(pkg, synth, torch, np, dev, args):
    """Times the CPU oracle (port of the reference algorithm, reference default buffer sizes) on bounded samples of the same workload:
    a sweep over thread counts on a small sample, then the best thread count on a large one (the reported value), and BASELINE
    config 1 on one thread.  The GPU result on the large sample is compared with the oracle's byte for byte."""
    from oracle import oracle as orc
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = cpu_quota()
    usable = max(1, min(cores, int(quota + 0.5))) if quota else cores     # threads beyond the container's CPU quota only take turns

    def sample(n):
        return [orc.FMI.from_symbols(synth.leaf_symbols(args.workload, seed, 0, n, args.readlen, n, dev).cpu().numpy()) for seed in (1001, 1002)]

    # (1) thread sweep: the reference's thread / merge-buffer hierarchy stops scaling long before 256 threads
    # (profiles/r01g_cpu_baseline_threads.log), so "all cores" is not the best the CPU path can do
    t0 = time.time()
    n_sweep = min(1 << 19, args.reads)
    candidates = sorted(set(t for t in (max(1, usable // 2), usable, 2 * usable, 4 * usable) if t <= cores)) or [cores]
    a_s, b_s = sample(n_sweep)
    bases_s = a_s.bases + b_s.bases
    def accounting(tm, search_seconds):
        """Where the threads of the search phase were (oracle/bwtm_oracle.cpp, SearchStats): shares of threads x search wall time."""
        th = max(1.0, tm["threads"])
        total = th * max(search_seconds, 1e-9)
        busy = {k: tm[k] for k in ("dfs", "sort_encode", "merge_thread", "lock_wait", "merge_global", "write")}
        shares = {k: round(v / total, 4) for k, v in busy.items()}
        shares["flush_one_thread"] = round(tm["flush"] * th / total, 4)          # every other thread is idle during the flush
        shares["idle_after_last_block"] = round(max(0.0, th * (search_seconds - tm["flush"]) - tm["threads_wall"]) / total, 4)
        return {"threads": int(th), "sequence_blocks": int(tm["sequence_blocks"]), "shares_of_thread_seconds": shares,
                "flush_seconds": round(tm["flush"], 3)}

    sweep = []
    for th in candidates:
        t1 = time.perf_counter()
        _, secs, tm = orc.merge_timed(a_s.clone(), b_s.clone(), threads=th)
        dt = time.perf_counter() - t1
        sweep.append({"threads": th, "seconds": round(dt, 3), "search_seconds": round(secs[0], 3), "value": round(bases_s / 1e9 / dt, 6),
                      "accounting": accounting(tm, secs[0])})
        log("cpu baseline sweep, %d threads: %.2f s (%.4f Gbases/s) on 2 x %d reads; %s" % (th, dt, sweep[-1]["value"], n_sweep, sweep[-1]["accounting"]["shares_of_thread_seconds"]))
    best = max(sweep, key=lambda s: s["value"])["threads"]
    # the reference's default is 4 sequence blocks per thread (fmi.h:52); more blocks = smaller thread buffers and more two-way merges
    blocks_sweep = []
    for per_thread in (16, 64):
        t1 = time.perf_counter()
        _, secs, tm = orc.merge_timed(a_s.clone(), b_s.clone(), threads=best, sequence_blocks=per_thread * best)
        dt = time.perf_counter() - t1
        blocks_sweep.append({"threads": best, "sequence_blocks": per_thread * best, "seconds": round(dt, 3), "search_seconds": round(secs[0], 3),
                             "value": round(bases_s / 1e9 / dt, 6), "accounting": accounting(tm, secs[0])})
        log("cpu baseline, %d threads x %d blocks per thread: %.2f s (%.4f Gbases/s)" % (best, per_thread, dt, blocks_sweep[-1]["value"]))
    del a_s, b_s

    # (2) the reported value: the best thread count on the large sample
    n = args.cpu_sample_reads or (1 << 22 if usable >= 16 else 1 << 18)
    n = min(n, args.reads)
    a, b = sample(n)
    A = pkg.Index.upload(a.data, a.sequences, a.bases)
    B = pkg.Index.upload(b.data, b.sequences, b.bases)
    M = pkg.merge(A, B)
    gpu_bytes = M.data()
    M.free(); A.free(); B.free()
    log("cpu baseline sample: 2 x %d reads prepared (%.1f s since the start of the baseline); running the oracle on %d threads" % (n, time.time() - t0, best))
    t1 = time.perf_counter()
    merged = a.bases + b.bases
    n_b_large = b.bases
    m, secs, tm_large = orc.merge_timed(a, b, threads=best)
    dt = time.perf_counter() - t1
    ok = bool(np.array_equal(gpu_bytes, m.data))
    del m, gpu_bytes
    log("cpu baseline: %.2f s (search %.2f s, interleave %.2f s), parity with GPU on the sample: %s" % (dt, secs[0], secs[1], ok))
    # (3) BASELINE config 1 (two sets of 10^5 reads) on ONE thread: the reference's `bwt_merge -t 1` plumbing case
    one = None
    efficiency = None
    if not args.no_config1:
        n1 = min(100000, args.reads)
        fm1 = sample(n1)
        bases1 = fm1[0].bases + fm1[1].bases
        n_b_one = fm1[1].bases
        t1 = time.perf_counter()
        _, secs1 = orc.merge(fm1[0], fm1[1], threads=1)
        dt1 = time.perf_counter() - t1
        # search phase only: rank-array values per second and thread on the large sample / the same on one thread
        if secs1[0] > 0 and secs[0] > 0:
            efficiency = round((n_b_large / secs[0]) / (min(best, usable) * (n_b_one / secs1[0])), 4)     # per CPU the container really gives
        one = {"value": round(bases1 / 1e9 / dt1, 6), "unit": "Gbases/s", "cores": 1, "seconds": round(dt1, 3), "search_seconds": round(secs1[0], 3),
               "sample": "BASELINE config 1: two sets of %d synthetic reads (%.3g Gbase merged), oracle merge with 1 thread (bwt_merge -t 1), "
                         "timer around the merging constructor as in bwt_merge.cpp:290-297" % (n1, bases1 / 1e9)}
        log("cpu baseline, config 1 on one thread: %.2f s (%.4f Gbases/s)" % (dt1, one["value"]))
    return {"value": round(merged / 1e9 / dt, 6), "unit": "Gbases/s", "cores": best, "cores_available": cores, "cpu_quota": quota,
            "cores_note": "threads of the reported run = the best of the sweep; the container shows %d logical CPUs%s" %
                          (cores, (" but its cgroup quota is %.1f CPUs' worth of time: more threads than that only take turns, which is why the sweep "
                                   "stops scaling there (profiles/r05_cpu_quota_probe.txt, r05_cpu_baseline_study.txt)" % quota) if quota else ""),
            "cpu_model": cpu_model(),
            "thread_sweep": {"sample": "two sets of %d reads (%.3g Gbase merged)" % (n_sweep, bases_s / 1e9), "runs": sweep, "best_threads": best,
                             "blocks_per_thread_sweep": blocks_sweep},
            "search_parallel_efficiency": efficiency,
            "search_parallel_efficiency_basis": "rank-array values per second of the search phase on the large sample / (min(threads, CPU quota) x the same on one "
                                                "thread, config 1)",
            "search_accounting": accounting(tm_large, secs[0]),
            "search_accounting_note": "shares of (threads x search wall time) by what a thread was doing; the reference's thread / merge-buffer hierarchy "
                                      "(fmi.cpp:139-257) is restated as it is: the two-way merges of ever larger buffers by single threads and the final flush "
                                      "bound the search phase once the trie walk is spread over many threads",
            "config1_one_thread": one, "kind": "port",
            "sample": "two sets of %d synthetic reads of the same workload (%.3g Gbase merged), oracle merge with %d threads (the best of the sweep %s), "
                      "reference default buffers" % (n, merged / 1e9, best, candidates),
            "seconds": round(dt, 3), "search_seconds": round(secs[0], 3), "gpu_parity_on_sample": ok,
            "context": "the paper reports 8.3 - 9.4 Mbp/s INSERTED on 32 cores for real read sets (paper.tex:266), i.e. the same order of magnitude "
                       "as this port on synthetic reads; baseline only"}


if __name__ == "__main__":
    main()
