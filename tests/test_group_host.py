"""Host side of the merge over partitioned records (include/bwtm.h: bwtm_group_*, bwtm_partition_cuts_host, bwtm_window_blocks), on the CPU:
the parts' shared control block between PROCESSES (barrier, all-gathers of any size, abort), and the cuts / byte shares against the
oracle's FM-index.  No GPU call is made: the functions under test are host arithmetic and shared memory."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group_worker(name, part, parts, q, mode):
    sys.path.insert(0, ROOT)
    import _pkg
    pkg = _pkg.load()
    try:
        g = pkg.Group(name, part, parts)
        out = {}
        if mode == "gather":
            # a small blob, one of exactly the bank size, one that needs several pieces, an empty one; barriers in between
            for n in (3, 2048, 5000, 0, 1):
                mine = (np.arange(n, dtype=np.uint64) * np.uint64(part + 1) + np.uint64(1000 * part))
                got = g.allgather(mine)
                out[n] = [int(got[h].sum()) if n else 0 for h in range(parts)]
                g.barrier()
            for it in range(200):                                     # many rounds: the two banks alternate, nobody overtakes by two
                got = g.allgather(np.array([it * 10 + part], dtype=np.uint64))
                assert [int(v) for v in got[:, 0]] == [it * 10 + h for h in range(parts)]
        elif mode == "abort":
            if part == 1:
                g.abort()
                out["aborted"] = True
            else:
                try:
                    g.barrier(); g.barrier()
                    out["error"] = None
                except pkg.BwtmError as e:
                    out["error"] = str(e)
        g.free()
        q.put((part, out))
    except Exception as e:                                            # noqa: BLE001
        q.put((part, {"exception": repr(e)}))


def _run_group(parts, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/bwtm-test-%d-%s-%d" % (os.getpid(), mode, parts)
    procs = [ctx.Process(target=_group_worker, args=(name, p, parts, q, mode)) for p in range(parts)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(parts))
    for p in procs:
        p.join(timeout=30)
    return res


@pytest.mark.parametrize("parts", [2, 3])
def test_group_allgather_between_processes(parts):
    res = _run_group(parts, "gather")
    for part in range(parts):
        assert "exception" not in res[part], res[part]
        for n in (3, 2048, 5000, 1):
            expect = [int((np.arange(n, dtype=np.uint64) * np.uint64(h + 1) + np.uint64(1000 * h)).sum()) for h in range(parts)]
            assert res[part][n] == expect, (part, n)


def test_group_abort_wakes_the_others():
    res = _run_group(3, "abort")
    assert res[1].get("aborted")
    for part in (0, 2):
        assert res[part].get("error") and "bwtm error 5" in res[part]["error"], res[part]


def test_group_of_one_needs_no_name(bwtm):
    g = bwtm.Group(None, 0, 1)
    got = g.allgather(np.array([7, 8], dtype=np.uint64))
    assert got.shape == (1, 2) and int(got[0][1]) == 8
    g.barrier()
    g.free()
    with pytest.raises(bwtm.BwtmError):
        bwtm.Group("no-slash", 0, 2)
    with pytest.raises(bwtm.BwtmError):
        bwtm.Group(None, 0, 17)


def _host(bwtm, x):
    return bwtm.host_index(x.data, x.samples[1], x.sequences, x.bases)


def test_cuts_are_insertion_points_of_kmers(bwtm, oracle):
    """bwtm_partition_cuts_host against the oracle's own rank queries and against the rank array."""
    ta = oracle.generate_reads(9601, 300, 40); tb = oracle.generate_reads(9602, 200, 55)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    ha, hb = _host(bwtm, a), _host(bwtm, b)
    assert [int(v) for v in ha.C] == [int(v) for v in a.C]
    for parts, k in ((4, 3), (2, 1), (8, 4), (3, 0)):
        I, R = bwtm.partition_cuts_host(ha, hb, parts, k)
        assert I[0] == 0 and R[0] == 0 and I[-1] == a.bases and R[-1] == b.bases
        assert all(I[g] <= I[g + 1] and R[g] <= R[g + 1] for g in range(parts))
        kk = k if k > 0 else 4
        # every cut is the insertion point of ONE k-mer in both indexes
        def points(x):
            sp = [0]
            for _ in range(kk):
                sp = [int(x.C[c]) + x.rank(p, c) for c in range(1, 6) for p in sp]
            return sp
        pa, pb = points(a), points(b)
        pairs = set(zip(pa, pb))
        for g in range(1, parts):
            assert (I[g], R[g]) in pairs, (parts, g)
    ranks, counts, _ = oracle.search(a, b, threads=2)
    ra = oracle.ra_from_runs(ranks, counts)
    I, R = bwtm.partition_cuts_host(ha, hb, 4, 3)
    for g in range(1, 4):                                               # a point of the merged order
        if R[g] < b.bases:
            assert int(ra[R[g]]) >= I[g]
        if R[g] > 0:
            assert int(ra[R[g] - 1]) <= I[g]


def test_window_blocks_cover_the_records(bwtm, oracle):
    a = oracle.FMI.from_text(oracle.generate_reads(9901, 3000, 80))
    ha = _host(bwtm, a)
    starts = a.samples[1].sum(axis=0).astype(np.uint64)
    for first, last in [(0, 5000), (12345, 54321), (a.bases - 3000, a.bases), (70000, 70001), (0, a.bases)]:
        b0, b1, fp, before = bwtm.window_blocks(ha, first, last)
        assert fp == int(starts[b0]) and fp <= (first & ~127)
        assert b1 == a.blocks or int(starts[b1]) >= min(a.bases, (last | 127) + 1)
        assert [int(v) for v in before] == [int(a.samples[1][c][b0]) for c in range(6)]
        assert sum(int(v) for v in before) == fp
    with pytest.raises(bwtm.BwtmError):
        bwtm.window_blocks(ha, 10, 5)
