"""The merge over PARTITIONED records through the C ABI (include/bwtm.h: bwtm_group_*, bwtm_part_*): `parts` threads of
this process, one library context of the one GPU each, stand in for GPUs.  Every part transcodes its windows from its own share of the native
bytes, searches in lock step with the others -- its step kernel reads its input straight out of the other parts' output buffers --, and
finalizes / interleaves / encodes its own range of the output.  The concatenation of the parts' bytes and samples must be the oracle's merged
stream, bit for bit; no part ever holds a whole index or the whole bitvector."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    yield bwtm
    bwtm.make_default_current()
    bwtm.trim()


def host(gpu, x):
    return gpu.host_index(x.data, x.samples[1], x.sequences, x.bases)


def merge_parts(gpu, a, b, parts, kmer=0):
    """-> (data, block_end, cum, stats, cuts): the parts' bytes and samples laid end to end."""
    from bwt_merge_amd import partitioned
    out = partitioned.merge_parts(gpu, host(gpu, a), host(gpu, b), parts, kmer=kmer, collect=lambda g, s: partitioned.slice_arrays(s))
    try:
        got = out["collected"]
        total = out["slices"][0].total_nbytes
        assert all(s.total_nbytes == total for s in out["slices"])
        offsets = [s.byte_offset for s in out["slices"]]
        assert offsets == sorted(offsets) and offsets[0] == 0
        data = np.concatenate([g[0] for g in got]); be = np.concatenate([g[1] for g in got]); cum = np.concatenate([g[2] for g in got], axis=1)
        assert data.size == total
        return data, be, cum, out["stats"], out["cuts"]
    finally:
        out["release"]()


def check_against_oracle(oracle, a, b, data, be, cum, threads=2):
    m, _ = oracle.merge(a.clone(), b.clone(), threads=threads)
    assert np.array_equal(data, m.data)
    obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum[:, :-1])


@pytest.mark.parametrize("parts,kmer,range_ratio", [(1, 2, 8), (2, 1, 8), (3, 3, 0), (4, 4, 8), (8, 4, 8), (16, 3, 8), (2, 0, 0), (5, 4, 3), (4, 2, 1)])
def test_parts_merge_equals_oracle(gpu, oracle, parts, kmer, range_ratio):
    """Inputs large enough for several encoder segments per part (2.4 M + 1.9 M positions: 66 segments); node phase of a few levels, of many,
    none at all (elements from the roots on), and the whole search on nodes (range_ratio = 1)."""
    ta = oracle.generate_reads(9801, 24000, 100); tb = oracle.generate_reads(9802, 19000, 100)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    gpu.tune("range_ratio", range_ratio)
    try:
        data, be, cum, stats, cuts = merge_parts(gpu, a, b, parts, kmer)
    finally:
        gpu.tune("range_ratio", 8)
    check_against_oracle(oracle, a, b, data, be, cum)
    if range_ratio == 0:
        assert all(s["node_levels"] == 0 for s in stats)
        assert sum(s["elements"] for s in stats) == b.bases              # every element of every step was advanced by exactly one part
        assert all(s["steps"] == 101 for s in stats)
    elif range_ratio == 8:
        assert all(s["node_levels"] > 0 for s in stats) and all(s["steps"] + s["node_levels"] == 101 for s in stats)
    # records and bitvector are partitioned, not replicated: all parts together hold them once (+ margins and boundary tiles)
    whole = 64 * ((a.bases >> 7) + 1 + (b.bases >> 7) + 1)
    margins = parts * 2 * (2 * 2 * 65536 // 128 + 4) * 64
    assert sum(s["record_bytes"] for s in stats) <= whole + margins
    assert sum(s["bitvector_bytes"] for s in stats) <= (a.bases + b.bases) // 8 + parts * 4 * 8192 + 8192
    if parts == 8:
        assert max(s["record_bytes"] for s in stats) < whole / 2


def test_parts_with_mixed_read_lengths(gpu, oracle):
    """Chains end at different steps: frontiers shrink, parts run empty steps, classes lose their elements at different times."""
    ta = oracle.generate_reads(9400, 2500, 90)
    tb = np.concatenate([oracle.generate_reads(9500 + j, 400, int(n)) for j, n in enumerate([1, 17, 60, 100, 139, 33])])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    # (unfused = 2: the pulled table by k_pull_tables and the two-launch scan instead of k_pull_scan1; 1: the generic scan + k_frontier_prep)
    for parts, kmer, rr, unfused in ((3, 3, 0, 0), (5, 4, 8, 0), (8, 4, 3, 0), (4, 3, 8, 2), (3, 2, 0, 1)):
        gpu.tune("range_ratio", rr); gpu.tune("frontier_unfused", unfused)
        try:
            data, be, cum, stats, _ = merge_parts(gpu, a, b, parts, kmer)
        finally:
            gpu.tune("range_ratio", 8); gpu.tune("frontier_unfused", 0)
        check_against_oracle(oracle, a, b, data, be, cum)
        assert all(s["steps"] + s["node_levels"] == 140 for s in stats)


@pytest.mark.parametrize("rr", [0, 4])
def test_parts_wide_coordinates(gpu, oracle, rr):
    """Coordinates beyond 2^32: the high bytes travel through the cut search and the pulled tables."""
    small_a = oracle.FMI.from_text(oracle.generate_reads(9301, 600, 60)); small_b = oracle.FMI.from_text(oracle.generate_reads(9302, 500, 70))
    a = oracle.FMI.from_runs(small_a.symbols.astype(np.uint64), np.full(small_a.symbols.size, 120000, dtype=np.uint64))
    assert a.bases > (1 << 32)
    b = oracle.FMI.from_runs(small_b.symbols.astype(np.uint64), np.full(small_b.symbols.size, 2000, dtype=np.uint64))
    gpu.tune("frontier_epoch", 5); gpu.tune("range_ratio", rr)
    try:
        data, be, cum, stats, _ = merge_parts(gpu, a, b, 3, 2)
    finally:
        gpu.tune("frontier_epoch", 0); gpu.tune("range_ratio", 8)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    M = gpu.merge(A, B)
    assert np.array_equal(data, M.data())
    mbe, mcum = M.samples()
    assert np.array_equal(be, mbe) and np.array_equal(cum, mcum[:, :-1])
    M.free(); A.free(); B.free()


def repetitive_reads(seed, genome_len, nreads, readlen):
    rng = np.random.default_rng(seed)
    genome = rng.integers(1, 5, genome_len, dtype=np.uint8)
    starts = rng.integers(0, genome_len - readlen, nreads)
    out = np.zeros((nreads, readlen + 1), dtype=np.uint8)
    for k, s in enumerate(starts):
        out[k, :readlen] = genome[s: s + readlen]
    return out.reshape(-1)


@pytest.mark.parametrize("glen,coverage", [(300000, 2), (200000, 4), (6000, 60)])
def test_parts_merge_of_repetitive_reads(gpu, oracle, glen, coverage):
    """Windows transcoded from byte shares of compressible streams, cuts inside long runs (the other window sizes of k_build_recs)."""
    nreads = coverage * glen // 100
    a = oracle.FMI.from_text(repetitive_reads(31, glen, nreads, 100)); b = oracle.FMI.from_text(repetitive_reads(32, glen, nreads * 3 // 4, 100))
    data, be, cum, _, _ = merge_parts(gpu, a, b, 3, 3)
    check_against_oracle(oracle, a, b, data, be, cum)


@pytest.mark.parametrize("case", ["short", "one_base", "with_n", "tiny_b", "unequal", "empty_b"])
def test_parts_merge_of_odd_collections(gpu, oracle, case):
    """Collections on which most parts end up with nothing: windows of a single record, coinciding cuts, empty output ranges."""
    rng = np.random.default_rng({"short": 1, "one_base": 2, "with_n": 3, "tiny_b": 4, "unequal": 5, "empty_b": 6}[case])

    def reads(n, lo, hi, alphabet):
        out = []
        for _ in range(n):
            out.append(rng.choice(alphabet, rng.integers(lo, hi + 1)).astype(np.uint8)); out.append(np.zeros(1, dtype=np.uint8))
        return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)

    if case == "short":
        ta, tb = reads(400, 0, 3, [1, 2, 3, 4]), reads(300, 0, 4, [1, 2, 3, 4])
    elif case == "one_base":
        ta, tb = reads(200, 5, 40, [3]), reads(150, 1, 60, [3])
    elif case == "with_n":
        ta, tb = reads(300, 20, 50, [1, 2, 3, 4, 5, 5]), reads(250, 10, 70, [1, 2, 3, 4, 5])
    elif case == "tiny_b":
        ta, tb = reads(500, 30, 60, [1, 2, 3, 4]), reads(3, 5, 9, [1, 2, 3, 4])
    elif case == "unequal":
        ta, tb = reads(40, 10, 20, [1, 2, 3, 4]), reads(900, 40, 80, [1, 2, 3, 4])
    else:
        ta, tb = reads(300, 20, 50, [1, 2, 3, 4]), reads(1, 0, 0, [1])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    for parts, kmer, rr in ((3, 2, 8), (5, 3, 0)):
        gpu.tune("range_ratio", rr)
        try:
            data, be, cum, _, _ = merge_parts(gpu, a, b, parts, kmer)
        finally:
            gpu.tune("range_ratio", 8)
        check_against_oracle(oracle, a, b, data, be, cum, threads=1)


def test_a_group_serves_a_chain_of_merges(gpu, oracle):
    """One group, three merges of growing size (the exported buffers are re-allocated and re-mapped when a merge needs more): every part of
    every merge gives the oracle's bytes."""
    import threading
    from bwt_merge_amd import capi, partitioned
    sets = [oracle.FMI.from_text(oracle.generate_reads(9100 + k, n, 80)) for k, n in enumerate((1500, 4000, 12000, 30000))]
    parts = 3
    name = partitioned.unique_group_name("chain")
    ctxs = [gpu.Context(0) for _ in range(parts)]
    results = [[None] * parts for _ in range(3)]
    errors = []

    def worker(g):
        try:
            ctxs[g].make_current()
            group = capi.Group(name, g, parts)
            for k in range(3):
                a, b = sets[k], sets[k + 1]
                S, _ = partitioned.merge_part(group, host(gpu, a), host(gpu, b), kmer=3)
                results[k][g] = partitioned.slice_arrays(S)
                S.free()
            group.free()
        except Exception as e:                                          # noqa: BLE001
            errors.append(e)
            raise

    threads = [threading.Thread(target=worker, args=(g,)) for g in range(parts)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    gpu.make_default_current()
    for c in ctxs:
        c.destroy()
    assert not errors, errors
    for k in range(3):
        data = np.concatenate([r[0] for r in results[k]]); be = np.concatenate([r[1] for r in results[k]]); cum = np.concatenate([r[2] for r in results[k]], axis=1)
        check_against_oracle(oracle, sets[k], sets[k + 1], data, be, cum)


def test_a_part_that_runs_out_of_room_stops_every_part(gpu, oracle):
    """Cuts that balance positions do not bound the elements a part holds in a step.  With a capacity of 2000 elements per part (the knob
    `part_capacity`) some part overflows -- at the expansion of the node levels or in an element step -- says so in the step's exchange, and
    every part returns from the same collective call: the one that overflowed with BWTM_ENOMEM, the others with BWTM_EPEER; nobody waits."""
    from bwt_merge_amd import partitioned
    a = oracle.FMI.from_text(oracle.generate_reads(9801, 24000, 100)); b = oracle.FMI.from_text(oracle.generate_reads(9802, 19000, 100))
    ha, hb = host(gpu, a), host(gpu, b)
    # at the expansion of the node levels / at the roots (part 0 begins with all 19 000 of them) / in an element step (a negative value holds only
    # the steps to it: the part says so in the step's exchange and idles through the step, every part stops at the next exchange)
    for rr, capacity in ((8, 2000), (0, 5000), (8, -3000), (0, -6000)):
        gpu.tune("part_capacity", capacity); gpu.tune("range_ratio", rr)
        try:
            with pytest.raises(gpu.BwtmError) as e:
                partitioned.merge_parts(gpu, ha, hb, 4, kmer=3)
            assert "bwtm error 3" in str(e.value) and "capacity" in str(e.value), str(e.value)
        finally:
            gpu.tune("part_capacity", 0); gpu.tune("range_ratio", 8)
    data, be, cum, _, _ = merge_parts(gpu, a, b, 4, 3)                   # and the group-less state is clean: the next merge is the oracle's
    check_against_oracle(oracle, a, b, data, be, cum)


def test_part_errors_reach_every_part(gpu, oracle):
    """A part that cannot go on (here: cuts that are not k-mer boundaries, so a trie node crosses one) stops the others with BWTM_EPEER instead
    of leaving them in a barrier; handles of windows are refused by the entry points that walk whole indexes."""
    from bwt_merge_amd import partitioned
    a = oracle.FMI.from_text(oracle.generate_reads(9701, 3000, 50)); b = oracle.FMI.from_text(oracle.generate_reads(9702, 3000, 50))
    ha, hb = host(gpu, a), host(gpu, b)
    I, R = gpu.partition_cuts_host(ha, hb, 2, 2)
    bad = ([0, I[1] + 1000, a.bases], [0, R[1] + 777, b.bases])
    with pytest.raises(gpu.BwtmError):
        partitioned.merge_parts(gpu, ha, hb, 2, cuts=bad)
    # a window of an index answers no queries of its own
    out = gpu.capi.vp()
    b0, b1, fp, before = gpu.window_blocks(ha, 1000, 50000)
    import ctypes as C
    Cs = (C.c_uint64 * 7)(*[int(v) for v in a.C])
    gpu.capi.check(gpu.capi.lib().bwtm_index_upload_window(a.data.ctypes.data + 64 * b0, 64 * (b1 - b0), fp, before, int(a.bases), int(a.sequences), Cs, 0, C.byref(out)))
    w = gpu.Index(out)
    assert 0 < int(gpu.capi.lib().bwtm_index_record_bytes(w.h)) <= 64 * ((50000 >> 7) - (1000 >> 7) + 4)
    with pytest.raises(gpu.BwtmError):
        w.extract(2000, 10)
    with pytest.raises(gpu.BwtmError):
        w.rank(np.array([1500], dtype=np.uint64), np.array([2], dtype=np.uint8))
    A = gpu.Index.upload(a.data, a.sequences, a.bases)
    with pytest.raises(gpu.BwtmError):
        gpu.merge(w, A)
    w.free(); A.free()
