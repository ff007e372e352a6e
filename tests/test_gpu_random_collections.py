"""Randomized parity: merges of odd little collections (empty sequences, one-symbol reads, skewed alphabets,
very unbalanced sets) through the C ABI against the oracle's BWT of the concatenated collection, with both forms
of the search.  hypothesis drives the shapes; every comparison is bit-exact."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

pytestmark = pytest.mark.gpu


def collection(draw, max_seqs):
    nseq = draw(st.integers(0, max_seqs))
    kind = draw(st.sampled_from(["uniform", "skewed", "homopolymer", "two_symbols"]))
    seed = draw(st.integers(0, 2 ** 31))
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(nseq):
        L = int(rng.choice([0, 1, 2, 7, 63, 64, 65, 100, 150, 300], p=[.08, .07, .05, .1, .1, .1, .1, .2, .1, .1]))
        if kind == "uniform":
            s = rng.integers(1, 6, L)
        elif kind == "skewed":
            s = rng.choice([1, 2, 3, 4, 5], L, p=[.7, .1, .1, .05, .05])
        elif kind == "homopolymer":
            s = np.full(L, int(rng.integers(1, 6)))
        else:
            s = rng.choice([2, 5], L)
        parts.append(s.astype(np.uint8)); parts.append(np.zeros(1, dtype=np.uint8))
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)


@st.composite
def two_collections(draw):
    return collection(draw, 40), collection(draw, 40)


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(two_collections())
def test_random_collections_merge_like_the_oracle(bwtm, oracle, ab):
    ta, tb = ab
    bwtm.init(0)
    fa, fb = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    direct = oracle.FMI.from_text(np.concatenate([ta, tb]))
    try:
        for algo in (2, 1, 0):
            bwtm.tune("search_algo", algo)
            A = bwtm.Index.upload(fa.data, fa.sequences, fa.bases)
            B = bwtm.Index.upload(fb.data, fb.sequences, fb.bases)
            M = bwtm.merge(A, B)
            assert (M.sequences, M.bases) == (direct.sequences, direct.bases)
            assert np.array_equal(M.data(), direct.data), algo
            assert np.array_equal(M.C, direct.C)
            be, cum = M.samples(); obe, ocum = direct.samples
            assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
            for x in (A, B, M):
                x.free()
    finally:
        bwtm.tune("search_algo", 0)


@st.composite
def run_stream(draw):
    pool = [1, 1, 1, 2, 3, 5, 31, 32, 33, 41, 42, 43, 63, 64, 65, 100, 127, 128, 129, 500, 4095, 4096, 4097, 8191, 8192, 8193, 40000, 1 << 17]
    lengths = draw(st.lists(st.sampled_from(pool), min_size=1, max_size=6))
    nruns = draw(st.integers(1, 6000))
    seed = draw(st.integers(0, 2 ** 31))
    return lengths, nruns, seed


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(run_stream())
def test_random_run_streams_transcode_and_reencode(bwtm, oracle, spec):
    """Upload of random native streams (all densities: every LDS window size and the long-run fill of the
    transcode): symbols, ranks and samples of the device index equal the oracle's, and re-encoding the device
    index reproduces the stream."""
    lengths, nruns, seed = spec
    rng = np.random.default_rng(seed)
    syms = rng.integers(0, 6, nruns)
    for k in range(1, nruns):
        if syms[k] == syms[k - 1]:
            syms[k] = (syms[k] + 1) % 6
    lens = rng.choice(lengths, nruns)
    if int(lens.sum()) > 30_000_000:
        lens = np.minimum(lens, 4097)
    sym = np.repeat(syms.astype(np.uint8), lens)
    bwtm.init(0)
    f = oracle.FMI.from_symbols(sym)
    ix = bwtm.Index.upload(f.data, f.sequences, f.bases)
    assert (ix.sequences, ix.nbytes, ix.blocks) == (f.sequences, f.nbytes, f.blocks)
    assert np.array_equal(ix.extract(0, sym.size), sym)
    pos = np.unique(np.concatenate([rng.integers(0, sym.size + 1, 300), [0, sym.size]])).astype(np.uint64)
    for c in range(6):
        expect = np.concatenate([[0], np.cumsum(sym == c)])[pos.astype(np.int64)]
        assert np.array_equal(ix.rank(pos, np.full(pos.size, c, dtype=np.uint8)).astype(np.int64), expect), c
    be, cum = ix.samples(); obe, ocum = f.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
    ix.drop_native(); ix.encode()
    assert np.array_equal(ix.data(), f.data)
    be, cum = ix.samples()
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
    ix.free()
