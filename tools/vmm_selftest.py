"""GPU self-test of basic runtime operations on the pool's VMM-backed blocks (run with small BWTM_POOL_VMM_* values)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import _pkg
pkg = _pkg.load()
from oracle import oracle as orc
pkg.init(0)
rng = np.random.default_rng(1)
# upload / download round trips of growing size, rank structure checks
for nruns in (1000, 100000, 3000000):
    syms = rng.integers(0, 6, nruns).astype(np.uint8)
    lens = rng.choice([1, 2, 3, 41, 42, 170], nruns)
    sym = np.repeat(syms, lens)
    f = orc.FMI.from_symbols(sym)
    ix = pkg.Index.upload(f.data, f.sequences, f.bases)
    ok1 = np.array_equal(ix.data(), f.data)
    ok2 = np.array_equal(ix.extract(0, min(sym.size, 1 << 20)), sym[: 1 << 20])
    be, cum = ix.samples(); obe, ocum = f.samples
    ok3 = np.array_equal(be, obe) and np.array_equal(cum, ocum)
    ix.drop_native(); ix.encode()
    ok4 = np.array_equal(ix.data(), f.data)
    print("nruns", nruns, "bytes", f.nbytes, ok1, ok2, ok3, ok4, flush=True)
    ix.free()
