"""Where the CPU baseline's search phase spends its threads: the oracle's merge on one sample with 1 ... all threads,
the reference's default of 4 sequence blocks per thread and 16 / 64 per thread, every run with the per-thread time accounting of
oracle/bwtm_oracle.cpp (SearchStats).  Run on the GPU box (its host CPU is the baseline's CPU): python tools/cpu_baseline_study.py [log2 reads]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import _pkg
pkg = _pkg.load()
from bwt_merge_amd import synth
from oracle import oracle as orc

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
fm = []
for seed in (1001, 1002):
    sym = synth.leaf_symbols("iid", seed, 0, n, 100, n, dev).cpu().numpy()
    fm.append(orc.FMI.from_symbols(sym))
a, b = fm
cores = len(os.sched_getaffinity(0))
print("sample: two sets of %d reads of 100 bp (%.3g Gbase merged, %d rank-array values); %d logical CPUs" % (n, (a.bases + b.bases) / 1e9, b.bases, cores))
print("%7s %7s | %8s %8s %8s | %6s %6s %6s %6s %6s %6s %6s %6s | %9s %6s" % ("threads", "blocks", "total s", "search s", "interl s", "dfs", "sort", "mrg_th", "lockw", "mrg_gl", "write", "flush", "idle", "Mvalues/s", "eff"))
base_rate = None
for th in [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= cores]:
    for per in ((4,) if th == 1 else (4, 16, 64)):
        t0 = time.perf_counter()
        m, secs, tm = orc.merge_timed(a.clone(), b.clone(), threads=th, sequence_blocks=per * th)
        dt = time.perf_counter() - t0
        total = th * secs[0]
        idle = max(0.0, th * (secs[0] - tm["flush"]) - tm["threads_wall"])
        rate = b.bases / secs[0] / 1e6
        if base_rate is None:
            base_rate = rate
        print("%7d %7d | %8.2f %8.2f %8.2f | %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f %6.3f | %9.2f %6.3f" %
              (th, per * th, dt, secs[0], secs[1], tm["dfs"] / total, tm["sort_encode"] / total, tm["merge_thread"] / total, tm["lock_wait"] / total,
               tm["merge_global"] / total, tm["write"] / total, tm["flush"] * th / total, idle / total, rate, rate / (th * base_rate)), flush=True)
        del m
print("columns dfs .. idle: shares of (threads x search wall time); flush = MergeBuffer::flush after the threads have joined (one thread works, the others wait); "
      "idle = threads that found no sequence block left while others were still working; eff = values/s per thread relative to one thread")
