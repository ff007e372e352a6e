"""Oracle self-consistency: restated reference algorithm vs the brute-force ground truth."""
import numpy as np
import pytest


def make_sets(oracle, na, nb, la, lb, seed=1):
    ta = oracle.generate_reads(1000 + seed, na, la)
    tb = oracle.generate_reads(2000 + seed, nb, lb)
    return ta, tb


@pytest.mark.parametrize("na,nb,la,lb", [(1, 1, 5, 7), (3, 2, 1, 1), (50, 70, 30, 20), (400, 300, 100, 100), (200, 200, 100, 150)])
def test_merge_equals_bwt_of_concatenation(oracle, na, nb, la, lb):
    ta, tb = make_sets(oracle, na, nb, la, lb)
    a = oracle.FMI.from_text(ta); b = oracle.FMI.from_text(tb)
    ab = oracle.FMI.from_text(np.concatenate([ta, tb]))
    for kw in (dict(threads=1), dict(threads=3, sequence_blocks=7), dict(threads=2, run_buffer_size=5, thread_buffer_size=16, merge_buffers=2)):
        m, _ = oracle.merge(a.clone(), b.clone(), **kw)
        assert np.array_equal(m.data, ab.data)
        assert (m.sequences, m.bases) == (ab.sequences, ab.bases)
        assert np.array_equal(m.C, ab.C)
        be1, cum1 = m.samples; be2, cum2 = ab.samples
        assert np.array_equal(be1, be2) and np.array_equal(cum1, cum2)
        assert m.hash == ab.hash


def test_search_is_sorted_multiset_of_walk_ranks(oracle):
    ta, tb = make_sets(oracle, 120, 90, 40, 40, seed=3)
    a = oracle.FMI.from_text(ta); b = oracle.FMI.from_text(tb)
    ranks, counts, stats = oracle.search(a, b, threads=1)
    ra = oracle.ra_from_runs(ranks, counts)
    assert ra.size == b.bases
    # per-sequence LF walk: RA[i] = r
    walk = np.zeros(b.bases, dtype=np.uint64)
    for j in range(b.sequences):
        i, r = j, a.sequences
        walk[i] = r
        while True:
            nxt, c = b.LF(i)
            if c == 0:
                break
            i = nxt; r = a.LF(r, c)
            walk[i] = r
    assert np.array_equal(walk, ra)      # indexed by B position it is non-decreasing
    assert np.all(np.diff(ra.astype(np.int64)) >= 0)
    assert int(stats.sum()) > 0


def test_rank_queries_against_plain_symbols(oracle):
    rng = np.random.default_rng(5)
    # long runs exercise the varint + block rule paths
    pieces = []
    for _ in range(300):
        pieces.append(np.full(int(rng.choice([1, 2, 3, 41, 42, 43, 170, 5000])), rng.integers(0, 6), dtype=np.uint8))
    sym = np.concatenate(pieces)
    f = oracle.FMI.from_symbols(sym)
    assert np.array_equal(f.symbols, sym)
    assert f.bases == sym.size
    cum = np.zeros((6, sym.size + 1), dtype=np.int64)
    for c in range(6):
        cum[c, 1:] = np.cumsum(sym == c)
    for i in list(rng.integers(0, sym.size, 300)) + [0, sym.size - 1]:
        i = int(i)
        c = int(sym[i])
        assert f.at(i) == c
        assert f.inverse_select(i) == (int(cum[c, i]), c)
        for cc in range(6):
            assert f.rank(i, cc) == int(cum[cc, i])
        r = f.ranks_at(i)
        assert all(int(r[cc]) == int(cum[cc, i]) for cc in range(1, 6))
    assert f.rank(sym.size, 2) == int(cum[2, sym.size])
    assert f.rank(sym.size + 10, 2) == int(cum[2, sym.size])
    for c in range(6):
        n = int(cum[c, -1])
        for k in [1, n // 2, n]:
            if k >= 1:
                pos = f.select(k, c)
                assert sym[pos] == c and int(cum[c, pos]) == k - 1
    assert list(f.character_counts) == [int(cum[c, -1]) for c in range(6)]


def test_runbuffer_semantics(oracle):
    # utils.h:121-142: a leading value 0 merges into the initial (0, 0) state
    assert oracle.runbuffer([(0, 2), (0, 1), (3, 1), (3, 4), (1, 1)]) == [(0, 3), (3, 5), (1, 1)]
    assert oracle.runbuffer([(4, 1), (4, 1), (2, 7)]) == [(4, 2), (2, 7)]
