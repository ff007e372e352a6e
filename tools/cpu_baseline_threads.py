import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import _pkg
pkg = _pkg.load()
from bwt_merge_amd import synth
from oracle import oracle as orc
n = 1 << 21
dev = torch.device("cuda", 0)
fm = []
for seed in (1001, 1002):
    sym = synth.leaf_bwt(synth.generate_reads(seed, 0, n, 100, device=dev)).cpu().numpy()
    fm.append(orc.FMI.from_symbols(sym))
a, b = fm
for th in (8, 16, 32, 64, 128, 256):
    t0 = time.perf_counter()
    m, secs = orc.merge(a.clone(), b.clone(), threads=th)
    dt = time.perf_counter() - t0
    print("threads %3d: %.2f s (search %.2f, interleave %.2f) -> %.4f merged Gbases/s" % (th, dt, secs[0], secs[1], 2 * n * 101 / 1e9 / dt), flush=True)
