"""The two-plane search view (include/bwtm_experimental.h, bwtm_view.h): exact against the oracle.  Runs only in a process that has
loaded libbwtm_experimental.so (tests/test_gpu_experimental.py starts one)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    from bwt_merge_amd import experimental
    assert experimental.loaded(), "these tests need BWTM_LIB=libbwtm_experimental.so"
    bwtm.tune("search_algo", 2)
    yield bwtm
    bwtm.tune("search_algo", 0)


def test_view_settings_of_the_frontier_search(gpu, oracle):
    """The frontier search on the view, with and without the node phase in front of it, with short epochs; view off in the same build."""
    rng = np.random.default_rng(5)
    cases = [(oracle.FMI.from_text(oracle.generate_reads(1001, 3000, 100)), oracle.FMI.from_text(oracle.generate_reads(1002, 2000, 100))),
             (oracle.FMI.from_text(oracle.generate_reads(11, 700, 37)), oracle.FMI.from_text(oracle.generate_reads(12, 900, 211)))]
    try:
        for a, b in cases:
            ranks, counts, _ = oracle.search(a, b, threads=2)
            ora = oracle.ra_from_runs(ranks, counts)
            A = gpu.Index.upload(a.data, a.sequences, a.bases)
            B = gpu.Index.upload(b.data, b.sequences, b.bases)
            for st in (dict(search_view=1), dict(search_view=1, range_ratio=0), dict(search_view=0), dict(search_view=1, frontier_epoch=9, range_ratio=40)):
                for k in ("frontier_epoch", "range_ratio", "search_view"):
                    gpu.tune(k, {"range_ratio": -1, "search_view": -1}.get(k, 0))
                for k, v in st.items():
                    gpu.tune(k, v)
                ra = gpu.RankArray(A, B)
                ra.search(A, B, 0, b.sequences - 1)
                ra.finalize()
                assert ra.values == b.bases, st
                assert np.array_equal(ra.download(), ora), st
                ra.free()
            A.free(); B.free()
    finally:
        gpu.tune("frontier_epoch", 0); gpu.tune("range_ratio", -1); gpu.tune("search_view", -1)


@pytest.mark.parametrize("ratio", [0, -1])
def test_search_view_exceptions_and_overflow(gpu, oracle, ratio):
    """The search view stores endmarkers and N as exceptions, seven per 160 positions: collections of very short reads (a quarter of
    the BWT is endmarkers), reads that are mostly N, and ordinary reads next to them -- view records without exceptions, with a few, and
    overflowed ones (their elements read the ordinary records) in one search; positions at the very end of an index."""
    rng = np.random.default_rng(78)
    def reads(n, length, p_n):
        out = []
        for _ in range(n):
            r = rng.integers(1, 5, length)
            r[rng.random(length) < p_n] = 5
            out.append(np.concatenate([r, [0]]))
        return np.concatenate(out).astype(np.uint8)
    tb = np.concatenate([reads(900, 3, 0.0), reads(300, 80, 0.7), reads(1200, 100, 0.004), reads(5, 159, 0.0), reads(1, 160, 1.0)])
    ta = np.concatenate([reads(700, 2, 0.1), reads(1500, 90, 0.01), reads(200, 120, 0.5)])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    ranks, counts, _ = oracle.search(a, b, threads=2)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    gpu.tune("range_ratio", ratio); gpu.tune("search_view", 1)
    try:
        ra = gpu.RankArray(A, B)
        ra.search(A, B, 0, b.sequences - 1)
        ra.finalize()
        assert ra.values == b.bases
        assert np.array_equal(ra.download(), oracle.ra_from_runs(ranks, counts))
        rb = gpu.RankArray(B, A)                                 # the other way round: A's reads searched in B
        rb.search(B, A, 0, a.sequences - 1)
        rb.finalize()
        r2, c2, _ = oracle.search(b, a, threads=2)
        assert np.array_equal(rb.download(), oracle.ra_from_runs(r2, c2))
    finally:
        gpu.tune("range_ratio", -1); gpu.tune("search_view", -1)
    for x in (ra, rb, A, B):
        x.free()




def test_view_with_40_bit_coordinates(gpu, oracle):
    """Both indexes beyond 2^32 positions (4096 reads x 10 500 copies, stated as runs): the view path of the step kernel with high bytes."""
    def repeated(seed):
        small = oracle.FMI.from_text(oracle.generate_reads(seed, 4096, 100))
        sym = small.symbols.astype(np.uint64)
        return oracle.FMI.from_runs(sym, np.full(sym.size, 10500, dtype=np.uint64))
    a, b = repeated(501), repeated(502)
    assert a.bases > (1 << 32) and b.bases > (1 << 32)
    oranks, ocounts, _ = oracle.search(a, b, capacity=1 << 21, threads=8)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    gpu.tune("range_ratio", 0); gpu.tune("search_view", 1)
    gpu.profile_enable(True); gpu.profile_reset()
    try:
        ra = gpu.RankArray(A, B)
        ra.search(A, B, 0, b.sequences - 1)
        ra.finalize()
        prof = gpu.profile_read()
        assert prof.get("view_build", (0, 0))[1] > 0 and "range_step" not in prof, sorted(prof)
        ranks, counts = ra.runs()
        assert np.array_equal(ranks, oranks) and np.array_equal(counts, ocounts)
    finally:
        gpu.profile_enable(False)
        gpu.tune("range_ratio", -1); gpu.tune("search_view", -1)
    for x in (ra, A, B):
        x.free()
