"""Times the library's builder (bwtm_builder_*) against the tensor-op leaves on the same synthetic set and checks that both
give the same native stream.  usage: python tools/ingest_time.py [reads] [leaf_reads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from bwt_merge_amd import synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
leaves = [int(x) for x in sys.argv[2:]] or [1 << 19]
pkg.init(0)
dev = torch.device("cuda", 0)
workload = os.environ.get("WORKLOAD", "iid")

def run(native, leaf):
    torch.cuda.synchronize(); pkg.synchronize()
    pkg.profile_reset(); pkg.profile_enable(True)
    t0 = time.time()
    ix = synth.build_index(pkg, 1001, reads, 100, leaf_reads=leaf, device=dev, workload=workload, native=native)
    pkg.synchronize()
    dt = time.time() - t0
    prof = pkg.profile_read(); pkg.profile_enable(False)
    ix.encode()
    data = ix.data()
    ix.free()
    return dt, data, prof

ref = None
for leaf in leaves:
    for native in (True, False):
        run(native, leaf)                                   # warm the pool
        dt, data, prof = run(native, leaf)
        if ref is None:
            ref = data
        same = bool(np.array_equal(ref, data))
        top = sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]
        print("reads=%d leaf=%d %s: %.2f s  (%.1f Mbases/s)  same_stream=%s" % (reads, leaf, "native" if native else "torch ", dt, reads * 101 / dt / 1e6, same))
        print("   " + ", ".join("%s %.0f ms" % (k, v[0]) for k, v in top))
