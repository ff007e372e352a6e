"""The frontier search over PARTITIONED records (include/bwtm_experimental.h: bwtm_x_index_window, bwtm_fslice_set_cuts, bwtm_fslice_gather_cut;
DESIGN.md section 6.3): G contexts of one GPU stand in for G GPUs.  Every one holds a window of A's and of B's records between fixed cuts
(k-mer boundaries of the merged order) and advances the elements whose coordinates fall into its windows; elements travel between the
GPUs at every step.  The union of the rank arrays must be the oracle's rank array, bit for bit, and every GPU's bits must lie inside its
own output range."""
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


def set_positions(words):
    nz = np.nonzero(words)[0]
    if nz.size == 0:
        return np.zeros(0, dtype=np.uint64)
    sh = np.arange(64, dtype=np.uint64)
    on = ((words[nz][:, None] >> sh[None, :]) & np.uint64(1)).astype(bool)
    return (nz.astype(np.uint64)[:, None] * np.uint64(64) + sh[None, :])[on]


def run_partitioned(gpu, oracle, a, b, parts, k, contexts=True, combine=False, node_ratio=0, capacity=None):
    from bwt_merge_amd.experimental import index_record_bytes, index_window, partition_cuts, search_partitioned
    ctxs = [gpu.Context(0) for _ in range(parts)] if contexts else []

    def enter(g):
        if contexts:
            ctxs[g].make_current()

    enter(0)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    I, R = partition_cuts(A, B, parts, k)
    assert I[0] == 0 and R[0] == 0 and I[-1] == a.bases and R[-1] == b.bases
    assert all(I[g] <= I[g + 1] and R[g] <= R[g + 1] for g in range(parts))
    whole_bytes = index_record_bytes(A) + index_record_bytes(B)
    A.free(); B.free()
    windows, ras, held = [], [], []
    for g in range(parts):
        enter(g)
        A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
        wa, wb = index_window(A, I[g], I[g + 1]), index_window(B, R[g], R[g + 1])
        A.free(); B.free()                                              # only the windows stay
        windows.append((wa, wb)); ras.append(gpu.RankArray(wa, wb)); held.append(index_record_bytes(wa) + index_record_bytes(wb))
    steps, levels, largest, work = search_partitioned(gpu, windows, ras, b.sequences, R, enter if contexts else None, capacity=capacity, node_ratio=node_ratio)
    bits, runs = [], None
    if combine:                                                         # one context: OR of the disjoint bit sets, then the reference's own form
        for g in range(1, parts):
            ras[0].or_from(ras[g])
        ras[0].finalize()
        runs = ras[0].runs()
    else:
        for g in range(parts):
            enter(g)
            bits.append(set_positions(ras[g].bits()))                   # positions of this GPU's set bits in the merged order
    for g in range(parts):
        enter(g)
        ras[g].free(); windows[g][0].free(); windows[g][1].free()
    gpu.make_default_current()
    for c in ctxs:
        c.destroy()
    return dict(I=I, R=R, bits=bits, runs=runs, steps=steps, levels=levels, largest=largest, work=work, held=held, whole=whole_bytes)


@pytest.mark.parametrize("parts,k,node_ratio", [(1, 2, 0), (2, 1, 0), (3, 3, 0), (5, 4, 0), (8, 4, 0), (16, 3, 0),
                                                # the first levels on trie nodes, routed like the elements: a few levels, many, the whole search
                                                (1, 2, 8), (2, 1, 8), (3, 3, 40), (5, 4, 8), (8, 4, 3), (16, 3, 8), (4, 2, 1)])
def test_partitioned_search_equals_oracle(gpu, oracle, parts, k, node_ratio):
    ta = oracle.generate_reads(9400, 2500, 90)
    tb = np.concatenate([oracle.generate_reads(9500 + j, 400, int(n)) for j, n in enumerate([1, 17, 60, 100, 139, 33])])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    expect = oracle_ra(oracle, a, b)
    out = run_partitioned(gpu, oracle, a, b, parts, k, node_ratio=node_ratio)
    assert out["steps"] + out["levels"] == 140 or (node_ratio == 1 and out["steps"] == 0)
    assert (out["levels"] > 0) == (node_ratio > 0)
    got = np.sort(np.concatenate(out["bits"]))
    assert got.size == b.bases
    assert np.array_equal(got, expect + np.arange(b.bases, dtype=np.uint64))          # position of b's suffix r in the merged order = RA[r] + r
    # every GPU set exactly the bits of its own output range [I_g + R_g, I_g+1 + R_g+1): nothing to exchange afterwards
    for g in range(parts):
        lo, hi = out["I"][g] + out["R"][g], out["I"][g + 1] + out["R"][g + 1]
        assert out["bits"][g].size == out["R"][g + 1] - out["R"][g], g
        if out["bits"][g].size:
            assert lo <= int(out["bits"][g].min()) and int(out["bits"][g].max()) < hi, g
    # records are partitioned, not replicated: all windows together hold the records once (+ at most two boundary records per window)
    assert sum(out["held"]) <= out["whole"] + parts * 4 * 64
    if node_ratio == 0:
        assert sum(out["work"]) == b.bases                            # every element of every step was advanced by exactly one GPU


def test_cuts_are_insertion_points_of_kmers(gpu, oracle):
    """partition_cuts' ranks against a direct count over the sorted suffixes of a small collection."""
    from bwt_merge_amd.experimental import partition_cuts
    ta = oracle.generate_reads(9601, 300, 40); tb = oracle.generate_reads(9602, 200, 55)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    I, R = partition_cuts(A, B, 4, 3)
    ra = oracle_ra(oracle, a, b)                                        # ra[r] = number of a's suffixes below b's suffix of rank r
    assert len(set(zip(I, R))) >= 4                                      # 125 candidate 3-mers: four parts get four different cuts
    for g in range(1, 4):
        # (I_g, R_g) is a point of the merged order: b's suffixes from R_g on lie above I_g of a's suffixes, those before it below
        if R[g] < b.bases:
            assert int(ra[R[g]]) >= I[g]
        if R[g] > 0:
            assert int(ra[R[g] - 1]) <= I[g]
    A.free(); B.free()


@pytest.mark.parametrize("nodes", [0, 4])
def test_partitioned_search_wide_coordinates(gpu, oracle, nodes):
    """Coordinates beyond 2^32: the high bytes travel through the cut counts and the gather."""
    small_a = oracle.FMI.from_text(oracle.generate_reads(9301, 600, 60)); small_b = oracle.FMI.from_text(oracle.generate_reads(9302, 500, 70))
    a = oracle.FMI.from_runs(small_a.symbols.astype(np.uint64), np.full(small_a.symbols.size, 120000, dtype=np.uint64))
    assert a.bases > (1 << 32)
    b = oracle.FMI.from_runs(small_b.symbols.astype(np.uint64), np.full(small_b.symbols.size, 2000, dtype=np.uint64))
    ranks, counts, _ = oracle.search(a, b, capacity=1 << 20, threads=4)
    gpu.tune("frontier_epoch", 5)
    try:
        out = run_partitioned(gpu, oracle, a, b, 3, 2, contexts=False, combine=True, node_ratio=nodes)
    finally:
        gpu.tune("frontier_epoch", 0)
    assert np.array_equal(out["runs"][0], ranks) and np.array_equal(out["runs"][1], counts)


def test_window_handles_are_refused_elsewhere(gpu, oracle):
    from bwt_merge_amd.experimental import index_window
    a = oracle.FMI.from_text(oracle.generate_reads(9701, 200, 50))
    A = gpu.Index.upload(a.data, a.sequences, a.bases)
    w = index_window(A, 1000, 5000)
    with pytest.raises(gpu.BwtmError):
        w.rank(np.array([1500], dtype=np.uint64), np.array([2], dtype=np.uint8))
    with pytest.raises(gpu.BwtmError):
        w.extract(1000, 10)
    with pytest.raises(gpu.BwtmError):
        gpu.merge(w, A)
    w.free(); A.free()


class HostIndex:
    """What partition_cuts asks of an index, answered by the oracle's FM-index on the host."""

    def __init__(self, x):
        self.x = x; self.bases = x.bases

    def find(self, patterns):
        c = int(patterns[0][0])
        return np.array([self.x.C[c]], dtype=np.uint64), None

    def rank(self, positions, comps):
        return np.array([self.x.rank(int(p), int(c)) for p, c in zip(positions, comps)], dtype=np.uint64)


def partitioned_merge(gpu, a, b, parts, k, node_ratio, from_bytes=False):
    """experimental.merge_partitioned on G contexts of one GPU: windows (from byte shares: no context ever holds a whole index, the cuts come
    from the host's copy, the oracle here), search with routing, then every part finalizes, interleaves and encodes its own range."""
    from bwt_merge_amd.experimental import merge_partitioned, partition_cuts
    if from_bytes:
        cuts = partition_cuts(HostIndex(a), HostIndex(b), parts, k)
    else:
        A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
        cuts = partition_cuts(A, B, parts, k)
        A.free(); B.free()
    out = merge_partitioned(gpu, a, b, parts, cuts, from_bytes=from_bytes, node_ratio=node_ratio)
    try:
        if from_bytes:
            assert sum(out["ra_bytes"]) <= (a.bases + b.bases) // 8 + parts * 4 * 8192 + 8192      # the bitvector is held once, too
        return np.concatenate(out["data"]), np.concatenate(out["block_end"]), np.concatenate(out["cum"], axis=1), out["held"], out["bounds"]
    finally:
        out["release"]()


@pytest.mark.parametrize("parts,k,node_ratio,from_bytes", [(1, 2, 8, False), (2, 1, 8, False), (3, 3, 0, False), (4, 4, 8, False), (8, 4, 8, False),
                                                           (1, 2, 8, True), (2, 1, 0, True), (3, 3, 8, True), (8, 4, 8, True)])
def test_partitioned_merge_equals_oracle(gpu, oracle, parts, k, node_ratio, from_bytes):
    """Search, interleave and encode from windows only: the concatenation of the parts' bytes and samples is the oracle's merged stream.
    Inputs large enough for several encoder segments per part (2.4 M + 1.9 M positions: 66 segments)."""
    ta = oracle.generate_reads(9801, 24000, 100); tb = oracle.generate_reads(9802, 19000, 100)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    data, be, cum, held, bounds = partitioned_merge(gpu, a, b, parts, k, node_ratio, from_bytes)
    m, _ = oracle.merge(a, b, threads=2)
    assert np.array_equal(data, m.data)
    obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum[:, :-1])
    if parts == 8:
        assert all(x[1] > x[0] for x in bounds)                          # every part produced a piece of the output
        whole = 64 * ((a.bases >> 7) + 1 + (b.bases >> 7) + 1)
        assert max(held) < whole / 2                                     # a part holds its windows (+ margins), not the indexes


def test_window_from_bytes_equals_window_of_whole_records(gpu, oracle):
    """bwtm_x_index_upload_window against bwtm_x_index_window: the searches over both kinds of windows set the same bits (above); here the
    transcode itself -- shares that begin and end in the middle of records, at the very beginning and at the very end of the stream, a share of
    one block -- through the one query a window answers: the element steps of a search that stays inside it."""
    from bwt_merge_amd.experimental import index_record_bytes, index_upload_window, window_blocks
    a = oracle.FMI.from_text(oracle.generate_reads(9901, 3000, 80))
    be, cum = a.samples
    starts = cum.sum(axis=0).astype(np.uint64)
    for first, last in [(0, 5000), (12345, 54321), (a.bases - 3000, a.bases), (70000, 70001), (0, a.bases)]:
        w = index_upload_window(a.data, cum, a.bases, a.sequences, first, last)
        b0, b1 = window_blocks(starts, a.data.size, first, last, a.bases)
        assert int(starts[b0]) <= (first & ~127) and int(starts[b1]) >= min(a.bases, (last | 127) + 1)
        held = index_record_bytes(w) // 64
        assert held >= (last >> 7) - (first >> 7) + 1                                    # every record of the range is there
        assert held <= (int(starts[b1]) - int(starts[b0])) // 128 + 2                    # and nothing but the share's records
        with pytest.raises(gpu.BwtmError):
            w.extract(first, 1)                                                           # a window answers no queries of its own
        w.free()
    with pytest.raises(gpu.BwtmError):                                                    # counts that do not add up to the position
        from bwt_merge_amd import experimental as X
        import ctypes as C
        bad = (C.c_uint64 * 6)(1, 2, 3, 4, 5, 6)
        Cs = (C.c_uint64 * 7)(*[0] * 7)
        out = C.c_void_p()
        X.check(X.lib().bwtm_x_index_upload_window(a.data.ctypes.data_as(C.c_void_p), 64, 5, bad, int(a.bases), int(a.sequences), Cs, C.byref(out)))


def test_windows_of_rank_arrays_are_refused_elsewhere(gpu, oracle):
    from bwt_merge_amd.experimental import ra_bytes, ra_or_range, rank_array_range
    a = oracle.FMI.from_text(oracle.generate_reads(9911, 3000, 100)); b = oracle.FMI.from_text(oracle.generate_reads(9912, 3000, 100))
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    w = rank_array_range(A, B, 200000, 300000)
    assert ra_bytes(w) == 4 * 8192                                                       # tiles 3 and 4 and one on either side
    whole = gpu.RankArray(A, B)
    for call in (lambda: w.search(A, B, 0, 10), lambda: w.finalize(), lambda: w.bits(), lambda: w.device_buffer(), lambda: whole.or_from(w),
                 lambda: w.range_counts(0, 512), lambda: ra_or_range(w, whole, 0, 1000), lambda: ra_or_range(whole, w, 500000, 500100)):
        with pytest.raises(gpu.BwtmError):
            call()
    ra_or_range(w, whole, 200000, 300000)                                                 # inside both: fine
    w.free(); whole.free(); A.free(); B.free()


def repetitive_reads(seed, genome_len, nreads, readlen):
    """Reads of one random genome at high coverage: a BWT of long runs (the other window sizes of k_build_recs and its fill path)."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(1, 5, genome_len, dtype=np.uint8)
    starts = rng.integers(0, genome_len - readlen, nreads)
    out = np.zeros((nreads, readlen + 1), dtype=np.uint8)
    for k, s in enumerate(starts):
        out[k, :readlen] = genome[s: s + readlen]
    return out.reshape(-1)


@pytest.mark.parametrize("glen,coverage", [(300000, 2), (200000, 4), (6000, 60)])
def test_partitioned_merge_of_repetitive_reads(gpu, oracle, glen, coverage):
    """Windows transcoded from byte shares of compressible streams, cuts inside long runs: ~150, ~250 and ~760 positions per 64-byte block are
    the three other window sizes of k_build_recs (16 384 positions, 32 768, 32 768 with the cooperative fill of long runs)."""
    nreads = coverage * glen // 100
    a = oracle.FMI.from_text(repetitive_reads(31, glen, nreads, 100)); b = oracle.FMI.from_text(repetitive_reads(32, glen, nreads * 3 // 4, 100))
    per_block = a.bases / a.blocks
    assert {2: 105 < per_block < 225, 4: 225 < per_block < 400, 60: per_block > 400}[coverage], per_block
    data, be, cum, held, bounds = partitioned_merge(gpu, a, b, 3, 3, 8, from_bytes=True)
    m, _ = oracle.merge(a, b, threads=2)
    assert np.array_equal(data, m.data)
    obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum[:, :-1])


@pytest.mark.parametrize("case", ["short", "one_base", "with_n", "tiny_b", "unequal"])
def test_partitioned_merge_of_odd_collections(gpu, oracle, case):
    """Collections on which most parts end up with nothing: reads shorter than k, a single symbol, N's, a b of three sequences, inputs of very
    different sizes -- all smaller than one encoder segment, so all but one part have an empty range of the output.  Windows may be a single
    record, cuts may coincide; the merge is still the oracle's."""
    rng = np.random.default_rng({"short": 1, "one_base": 2, "with_n": 3, "tiny_b": 4, "unequal": 5}[case])

    def reads(n, lo, hi, alphabet):
        out = []
        for _ in range(n):
            out.append(rng.choice(alphabet, rng.integers(lo, hi + 1)).astype(np.uint8)); out.append(np.zeros(1, dtype=np.uint8))
        return np.concatenate(out)

    if case == "short":
        ta, tb = reads(400, 0, 3, [1, 2, 3, 4]), reads(300, 0, 4, [1, 2, 3, 4])
    elif case == "one_base":
        ta, tb = reads(200, 5, 40, [3]), reads(150, 1, 60, [3])
    elif case == "with_n":
        ta, tb = reads(300, 20, 50, [1, 2, 3, 4, 5, 5]), reads(250, 10, 70, [1, 2, 3, 4, 5])
    elif case == "tiny_b":
        ta, tb = reads(500, 30, 60, [1, 2, 3, 4]), reads(3, 5, 9, [1, 2, 3, 4])
    else:
        ta, tb = reads(40, 10, 20, [1, 2, 3, 4]), reads(900, 40, 80, [1, 2, 3, 4])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    for parts, k, node_ratio in ((3, 2, 8), (5, 3, 0)):
        data, be, cum, held, bounds = partitioned_merge(gpu, a, b, parts, k, node_ratio, from_bytes=True)
        m, _ = oracle.merge(a.clone(), b.clone(), threads=1)
        assert np.array_equal(data, m.data), (case, parts)
        obe, ocum = m.samples
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum[:, :-1]), (case, parts)
