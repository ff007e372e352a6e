"""Pool behaviour that the default configuration only reaches after hours: every buffer through the map / unmap path, and the
hipMalloc fallback once the reserved address range is used up (addresses are consumed, never reused: DESIGN.md section 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import _pkg
from oracle import oracle as orc
pkg = _pkg.load()
pkg.init(0)
ta = orc.generate_reads(1001, 3000, 100); tb = orc.generate_reads(1002, 2500, 100)
a = orc.FMI.from_text(ta); b = orc.FMI.from_text(tb)
m, _ = orc.merge(a.clone(), b.clone(), threads=2)
before = pkg.pool_stats()
for round in range(%(rounds)d):
    A = pkg.Index.upload(a.data, a.sequences, a.bases); B = pkg.Index.upload(b.data, b.sequences, b.bases)
    for algo in (2, 1):
        pkg.tune("search_algo", algo)
        M = pkg.merge(A, B)
        assert np.array_equal(M.data(), m.data) and np.array_equal(M.C, m.C), (round, algo)
        be, cum = M.samples(); obe, ocum = m.samples
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum), (round, algo)
        M.free()
    A.free(); B.free()
    if round %% 3 == 2:
        pkg.trim()
after = pkg.pool_stats()
print("STATS", before, after)
assert after["mapped_blocks"] > 0 or after["address_space_exhausted"] == 1
%(extra)s
print("OK")
'''


def run(env, rounds, extra=""):
    code = SCRIPT % {"root": ROOT, "rounds": rounds, "extra": extra}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


def test_every_buffer_through_map_and_unmap(bwtm):
    """Blocks of 64 KiB and more take the mapped path in 2-MiB chunks: every buffer of a merge is mapped, recycled across sizes
    and unmapped (the default thresholds send only blocks of 128 MiB and more that way)."""
    run({"BWTM_POOL_VMM_CHUNK": "2097152", "BWTM_POOL_VMM_MIN": "65536"}, 4)


def test_address_space_exhaustion_falls_back_to_hipmalloc(bwtm):
    """A reserved range of 96 MiB is used up within a few merges; from then on large blocks come from hipMalloc: the results stay
    exact and bwtm_pool_stats() reports the state."""
    out = run({"BWTM_POOL_VMM_CHUNK": "2097152", "BWTM_POOL_VMM_MIN": "65536", "BWTM_POOL_VA_SEGMENT": "33554432", "BWTM_POOL_VA_LIMIT": "100663296"}, 9,
              extra='assert after["address_space_exhausted"] == 1 and after["hipmalloc_fallbacks"] > 0 and after["address_bytes_reserved"] <= 100663296, after')
    assert "STATS" in out


def test_parity_suite_subset_with_small_chunks(bwtm):
    """A slice of the parity suite in a process whose pool maps everything in 2-MiB chunks (what the builder ran by hand in round 2)."""
    env = dict(os.environ, BWTM_POOL_VMM_CHUNK="2097152", BWTM_POOL_VMM_MIN="65536")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_parity.py", "tests/test_gpu_slices.py", "-k",
                          "stages or chaining or config1 or slices or emit_rounds"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
