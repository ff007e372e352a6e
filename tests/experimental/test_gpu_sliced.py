"""The sliced frontier search (dense multi-GPU form of buildRA, include/bwtm_experimental.h: bwtm_fslice_*): G contexts of one GPU stand in for
G GPUs, each advancing a contiguous slice of the sorted frontier and pulling its next slice from all the others' outputs.
The union of their rank arrays must be the oracle's rank array, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    from bwt_merge_amd import experimental
    assert experimental.loaded(), "these tests need BWTM_LIB=libbwtm_experimental.so"
    yield bwtm
    bwtm.make_default_current()
    bwtm.trim()


def oracle_ra(oracle, a, b):
    ranks, counts, _ = oracle.search(a, b, threads=2)
    return oracle.ra_from_runs(ranks, counts)


@pytest.mark.parametrize("parts,contexts", [(1, False), (2, False), (3, True), (5, True), (8, False)])
def test_sliced_search_equals_oracle(gpu, oracle, parts, contexts):
    """Reads of mixed lengths (slices shrink and become empty at different steps), more parts than some steps have elements."""
    from bwt_merge_amd.experimental import search_sliced
    rng = np.random.default_rng(7)
    ta = oracle.generate_reads(9100, 2200, 90)
    tb = np.concatenate([oracle.generate_reads(9200 + k, 300, int(n)) for k, n in enumerate([1, 17, 60, 100, 139, 33])])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    expect = oracle_ra(oracle, a, b)
    ctxs = [gpu.Context(0) for _ in range(parts)] if contexts else []

    def enter(g):
        ctxs[g].make_current()

    idx, ras = [], []
    for g in range(parts):
        if contexts:
            enter(g)
        A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
        idx.append((A, B)); ras.append(gpu.RankArray(A, B))
    steps = search_sliced(gpu, idx, ras, b.sequences, enter if contexts else None)
    assert steps == 140                                   # the longest read (139 symbols) + its endmarker
    if contexts:
        enter(0)
    for g in range(1, parts):                               # the one exchange: OR of the disjoint bit sets (all-reduce across real GPUs)
        ras[0].or_from(ras[g])
    ras[0].finalize()
    assert ras[0].values == b.bases
    assert np.array_equal(ras[0].download(), expect)
    # every GPU's share: disjoint and, on read-like inputs, balanced (slices are equal shares of the frontier at every step)
    ones = []
    for g in range(1, parts):
        if contexts:
            enter(g)
        ras[g].finalize(); ones.append(ras[g].values)
    assert sum(ones) <= b.bases
    for g in range(parts):
        if contexts:
            enter(g)
        ras[g].free(); idx[g][0].free(); idx[g][1].free()
    gpu.make_default_current()
    for c in ctxs:
        c.destroy()


def test_sliced_search_wide_coordinates_and_epochs(gpu, oracle):
    """Coordinates beyond 2^32 (the high bytes travel through the gather) and epochs of a few steps in every slice."""
    from bwt_merge_amd.experimental import search_sliced
    small_a = oracle.FMI.from_text(oracle.generate_reads(9301, 600, 60)); small_b = oracle.FMI.from_text(oracle.generate_reads(9302, 500, 70))
    a = oracle.FMI.from_runs(small_a.symbols.astype(np.uint64), np.full(small_a.symbols.size, 120000, dtype=np.uint64))
    assert a.bases > (1 << 32)
    b = oracle.FMI.from_runs(small_b.symbols.astype(np.uint64), np.full(small_b.symbols.size, 2000, dtype=np.uint64))
    ranks, counts, _ = oracle.search(a, b, capacity=1 << 20, threads=4)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    parts = 3
    ras = [gpu.RankArray(A, B) for _ in range(parts)]
    gpu.tune("frontier_epoch", 5)
    try:
        search_sliced(gpu, [(A, B)] * parts, ras, b.sequences)
    finally:
        gpu.tune("frontier_epoch", 0)
    for g in range(1, parts):
        ras[0].or_from(ras[g])
    ras[0].finalize()
    got_r, got_c = ras[0].runs()
    assert np.array_equal(got_r, ranks) and np.array_equal(got_c, counts)
    for r in ras:
        r.free()
    A.free(); B.free()
