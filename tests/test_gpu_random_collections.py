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
