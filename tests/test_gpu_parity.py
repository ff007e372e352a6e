"""Parity of the HIP path against the CPU oracle, stage by stage, through the C ABI.
Integer / byte / index work throughout: every comparison is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    """The library with the frontier search forced: at test sizes the dispatch would pick the per-chain walk
    (bwtm_search chooses by shard size), and the frontier search is what large inputs run.  The walk and the
    automatic choice are covered by test_partitioned_emit_rounds_fallbacks_and_variants."""
    bwtm.init(0)
    bwtm.tune("search_algo", 2)
    yield bwtm
    bwtm.tune("search_algo", 0)


def cum_counts(sym):
    cum = np.zeros((6, sym.size + 1), dtype=np.int64)
    for c in range(6):
        cum[c, 1:] = np.cumsum(sym == c)
    return cum


def run_symbols(rng, nruns, lengths):
    syms = rng.integers(0, 6, nruns)
    for k in range(1, nruns):                      # adjacent runs differ (maximal runs)
        if syms[k] == syms[k - 1]:
            syms[k] = (syms[k] + 1) % 6
    lens = rng.choice(lengths, nruns)
    return np.repeat(syms.astype(np.uint8), lens)


def check_index(ix, sym, rng, nq=4000):
    n = sym.size
    assert ix.bases == n
    got = ix.extract(0, n)
    assert np.array_equal(got, sym)
    cum = cum_counts(sym)
    pos = np.concatenate([rng.integers(0, n + 1, nq), [0, n, max(n - 1, 0), min(n, 127), min(n, 128), min(n, 129)]]).astype(np.uint64)
    for c in range(6):
        r = ix.rank(pos, np.full(pos.size, c, dtype=np.uint8))
        assert np.array_equal(r.astype(np.int64), cum[c, pos.astype(np.int64)]), c
    # positions past the end clamp (bwt.cpp:322); comp >= 6 ranks 0 (bwt.cpp:321)
    assert int(ix.rank([n + 5], [2])[0]) == int(cum[2, n])
    assert int(ix.rank([3], [7])[0]) == 0
    if n > 0:
        ipos = rng.integers(0, n, nq).astype(np.uint64)
        r, c = ix.inverse_select(ipos)
        assert np.array_equal(c, sym[ipos.astype(np.int64)])
        assert np.array_equal(r.astype(np.int64), cum[sym[ipos.astype(np.int64)], ipos.astype(np.int64)])


@pytest.mark.parametrize("deposit", ["by_density", "straight_line", "branching"])
def test_upload_builds_exact_rank_structure(gpu, oracle, deposit):
    """k_build_recs against plain symbols.  `deposit`: the kernel's two ways to lay a block's runs into the bit planes -- chosen by the stream's
    density, or forced on every stream (all three window sizes take both)."""
    rng = np.random.default_rng(1)
    gpu.tune("recs_uniform", {"by_density": 0, "straight_line": 1, "branching": -1}[deposit])
    try:
        upload_cases(gpu, oracle, rng)
    finally:
        gpu.tune("recs_uniform", 0)


def upload_cases(gpu, oracle, rng):
    # run-length mixes chosen so that all three LDS window sizes of the transcode are used (positions per 62-block group); the last three
    # are streams of the long-window kind: runs around the one-byte limit (9 .. 41), long-event mixes, and a few giant runs between short ones
    for lengths in ([1, 1, 1, 2, 3], [2, 3, 4], [1, 2, 41, 42, 43, 169, 170, 5000], [16425, 16426, 100000, 1, 7], [9, 12, 17, 23, 31, 32, 33, 41],
                    [5, 20, 42, 60, 83, 90, 168, 169, 170, 400], [3, 8, 40000, 14, 70000, 21], [2, 3, 4, 5, 6, 8, 10]):
        sym = run_symbols(rng, 3000 if max(lengths) > 4 else 60000, lengths)
        if max(lengths) in (41, 400, 10):
            sym = run_symbols(rng, 40000, lengths)                  # several groups, windows and records per group
        f = oracle.FMI.from_symbols(sym)
        ix = gpu.Index.upload(f.data, f.sequences, f.bases)
        assert (ix.sequences, ix.nbytes, ix.blocks) == (f.sequences, f.nbytes, f.blocks)
        assert np.array_equal(ix.C, f.C)
        check_index(ix, sym, rng)
        be, cum = ix.samples()
        obe, ocum = f.samples
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
        ix.free()


def test_index_from_borrowed_device_buffer(gpu, oracle):
    """bwtm_index_from_device_borrowed reads the caller's buffer in place: same index as the copying
    constructors, also when the bytes after the stream are garbage and the stream ends mid-block."""
    import torch
    rng = np.random.default_rng(11)
    for nruns in (1, 50, 777, 3000, 4099):
        sym = run_symbols(rng, nruns, [1, 1, 2, 3, 41, 42, 170, 5000])
        f = oracle.FMI.from_symbols(sym)
        host = np.full(f.nbytes + 64, 0xFF, dtype=np.uint8)         # 0xFF = continuation bytes of a long run
        host[:f.nbytes] = f.data
        buf = torch.from_numpy(host).to("cuda:0")
        ix = gpu.Index.from_device(buf.data_ptr(), f.nbytes, f.sequences, f.bases, borrow=True)
        assert (ix.sequences, ix.nbytes, ix.blocks) == (f.sequences, f.nbytes, f.blocks)
        check_index(ix, sym, rng, nq=500)
        be, cum = ix.samples()
        obe, ocum = f.samples
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
        assert np.array_equal(ix.data(), f.data)
        ptr, nb = ix.device_data()
        assert (ptr, nb) == (buf.data_ptr(), f.nbytes)
        ix.drop_native()                                             # releases the claim on `buf`
        del buf
        assert np.array_equal(ix.extract(0, sym.size), sym)
        ix.free()
    with pytest.raises(gpu.BwtmError):
        buf = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
        gpu.Index.from_device(buf.data_ptr() + 1, 8, 0, 8, borrow=True)   # misaligned


def test_upload_rejects_inconsistent_header(gpu, oracle):
    f = oracle.FMI.from_symbols(np.array([1, 2, 0, 3, 0], dtype=np.uint8))
    with pytest.raises(gpu.BwtmError):
        gpu.Index.upload(f.data, f.sequences, f.bases + 1)
    with pytest.raises(gpu.BwtmError):
        gpu.Index.upload(f.data, f.sequences + 1, f.bases)
    # A full block that encodes fewer than 64 positions (redundant varint continuation bytes) cannot come
    # from Run::write (support.h:256-282); the transcode kernels rely on that and the upload refuses it.
    padded = np.array([247] + [0x80] * 62 + [0x00] + [2], dtype=np.uint8)      # (1, 42) in 64 bytes, then (2, 1)
    with pytest.raises(gpu.BwtmError, match="canonical"):
        gpu.Index.upload(padded, 0, 43)


def test_rank_structure_across_super_blocks(gpu, oracle):
    # > 2^25 positions so that several super-table entries and 25-bit relative counts are used
    rng = np.random.default_rng(2)
    sym = run_symbols(rng, 200000, [1, 2, 3, 50, 400, 1500])
    assert sym.size > (1 << 25) + 1000
    f = oracle.FMI.from_symbols(sym)
    ix = gpu.Index.upload(f.data, f.sequences, f.bases)
    check_index(ix, sym, rng, nq=20000)
    ix.free()


def reads_pair(oracle, na, nb, la, lb, seed=0):
    ta = oracle.generate_reads(1000 + seed, na, la)
    tb = oracle.generate_reads(2000 + seed, nb, lb)
    return oracle.FMI.from_text(ta), oracle.FMI.from_text(tb), ta, tb


@pytest.mark.parametrize("na,nb,la,lb", [(3, 2, 4, 5), (1, 1, 1, 1), (200, 300, 30, 45), (2000, 1500, 100, 100), (800, 800, 100, 150)])
def test_stages_match_oracle(gpu, oracle, na, nb, la, lb):
    a, b, ta, tb = reads_pair(oracle, na, nb, la, lb, seed=na)
    ranks, counts, _ = oracle.search(a, b, threads=2)
    ora = oracle.ra_from_runs(ranks, counts)
    A = gpu.Index.upload(a.data, a.sequences, a.bases)
    B = gpu.Index.upload(b.data, b.sequences, b.bases)

    # search (split in two ranges: results accumulate like shards from two GPUs)
    ra = gpu.RankArray(A, B)
    half = b.sequences // 2
    if half > 0:
        ra.search(A, B, 0, half - 1)
    ra.search(A, B, half, b.sequences - 1)
    ra.finalize()
    assert ra.values == b.bases
    assert np.array_equal(ra.download(), ora)
    bits = np.unpackbits(ra.bits().view(np.uint8), bitorder="little")[: a.bases + b.bases]
    expect = np.zeros(a.bases + b.bases, dtype=np.uint8)
    expect[np.arange(b.bases, dtype=np.uint64) + ora] = 1
    assert np.array_equal(bits, expect)

    # interleave
    merged_sym = oracle.interleave_symbols(a.symbols, b.symbols, ora)
    M = gpu.interleave(A, B, ra)
    rng = np.random.default_rng(3)
    check_index(M, merged_sym, rng, nq=2000)
    assert (M.sequences, M.bases) == (a.sequences + b.sequences, a.bases + b.bases)

    # encode + samples == FMI::FMI(a, b) of the oracle == BWT of the concatenation
    m, _ = oracle.merge(a.clone(), b.clone(), threads=2)
    ab = oracle.FMI.from_text(np.concatenate([ta, tb]))
    assert np.array_equal(m.data, ab.data)
    M.encode()
    assert M.nbytes == m.nbytes
    assert np.array_equal(M.data(), m.data)
    assert np.array_equal(M.C, m.C)
    be, cum = M.samples(); obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
    for x in (A, B, M):
        x.free()
    ra.free()


def test_merge_entry_point_and_chaining(gpu, oracle):
    sets = [oracle.generate_reads(3000 + k, 400 + 100 * k, 100 if k != 1 else 150) for k in range(3)]
    fm = [oracle.FMI.from_text(t) for t in sets]
    ix = [gpu.Index.upload(f.data, f.sequences, f.bases) for f in fm]
    m01 = gpu.merge(ix[0], ix[1])
    m012 = gpu.merge(m01, ix[2])                    # bwt_merge.cpp:167-173: result is the next `a`
    direct = oracle.FMI.from_text(np.concatenate(sets))
    assert np.array_equal(m012.data(), direct.data)
    assert np.array_equal(m012.C, direct.C)
    assert (m012.sequences, m012.bases) == (direct.sequences, direct.bases)
    # device-resident chaining without the native form of the intermediate
    m01.drop_native()
    again = gpu.merge(m01, ix[2])
    assert np.array_equal(again.data(), direct.data)


def mixed_length_text(oracle, seed, nreads):
    """A read set in which reads j with j % 5 < 3 are 100 bp and the others 150 bp (BASELINE config 5: each
    length contributes half of the bases), sequences in generation order."""
    long_reads = oracle.generate_reads(seed, nreads, 150).reshape(nreads, 151)
    parts = []
    for j in range(nreads):
        L = 100 if j % 5 < 3 else 150
        parts.append(long_reads[j, :L]); parts.append(np.zeros(1, dtype=np.uint8))
    return np.concatenate(parts)


def test_config4_and_config5_shapes(gpu, oracle):
    """BASELINE config 4 (a small set inserted into a 4x larger one) and config 5 (four sets of mixed
    100 / 150 bp reads merged in command-line order, bwt_merge.cpp:167-173) at a size the oracle can check:
    every result equals the oracle's BWT of the concatenated collection."""
    big = oracle.generate_reads(7001, 2400, 100); small = oracle.generate_reads(7002, 600, 100)
    fb, fs = oracle.FMI.from_text(big), oracle.FMI.from_text(small)
    m = gpu.merge(gpu.Index.upload(fb.data, fb.sequences, fb.bases), gpu.Index.upload(fs.data, fs.sequences, fs.bases))
    direct = oracle.FMI.from_text(np.concatenate([big, small]))
    assert np.array_equal(m.data(), direct.data) and np.array_equal(m.C, direct.C)
    be, cum = m.samples(); obe, ocum = direct.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)

    texts = [mixed_length_text(oracle, 7100 + k, 500) for k in range(4)]
    running = None
    for k, t in enumerate(texts):
        f = oracle.FMI.from_text(t)
        ix = gpu.Index.upload(f.data, f.sequences, f.bases)
        if running is None:
            running = ix
        else:
            nxt = gpu.merge(running, ix)
            nxt.drop_native() if k < len(texts) - 1 else None       # chained merges keep only the rank structure
            running = nxt
    direct = oracle.FMI.from_text(np.concatenate(texts))
    assert (running.sequences, running.bases) == (direct.sequences, direct.bases)
    assert np.array_equal(running.data(), direct.data) and np.array_equal(running.C, direct.C)


def test_empty_increment_and_empty_base(gpu, oracle):
    a = oracle.FMI.from_text(oracle.generate_reads(7, 50, 20))
    e = oracle.FMI.from_symbols(np.zeros(0, dtype=np.uint8))
    A = gpu.Index.upload(a.data, a.sequences, a.bases)
    E = gpu.Index.upload(e.data, 0, 0)
    assert np.array_equal(gpu.merge(A, E).data(), a.data)
    assert np.array_equal(gpu.merge(E, A).data(), a.data)


@pytest.mark.parametrize("case", ["mixed", "boundaries", "sparse_long", "genome_like", "halves", "giant", "tiny", "alternating"])
def test_encoder_block_rule(gpu, oracle, case):
    """Run::write's offset-dependent forms (support.h:256-282), including runs that end in
    chunks far from where they start, against the oracle's encoder."""
    import torch
    rng = np.random.default_rng(11)
    if case == "mixed":
        syms = [run_symbols(rng, 20000, [1, 2, 3, 5, 41, 42, 43, 169, 170, 16425, 16426, 100000])]
    elif case == "boundaries":
        # many long runs so that every offset mod 64 is met by 41/42/43/169/170-length runs
        syms = [run_symbols(rng, 60000, [41, 42, 43, 44, 168, 169, 170, 171, 1])]
    elif case == "sparse_long":
        # chunks of mostly one-byte events with a few runs >= 42 among them: the parallel long-event path
        # (offsets of the short events shifted by the extra bytes of the long ones before them)
        syms = [run_symbols(rng, 400000, [1] * 30 + [2] * 10 + [3, 7, 42, 45, 100, 171, 200, 3000]),
                run_symbols(rng, 100000, [1] * 6 + [41, 42, 50, 64, 65, 127, 128, 129])]
    elif case == "genome_like":
        # the run lengths of reads of a genome at high coverage: most chunks hold runs of 42 .. 82 (two bytes at every offset: a wave scan gives
        # their shifts), a few of 83 and more (the ordered walk), one-byte runs around them -- at every offset mod 64, in both encoder kernels
        syms = [run_symbols(rng, 300000, [3, 5, 9, 14, 20, 30, 41, 42, 50, 64, 82, 83, 84, 100, 169, 170, 300]),
                run_symbols(rng, 200000, [41, 42, 43, 81, 82, 83, 84, 1, 2]),
                run_symbols(rng, 150000, [12] * 8 + [20] * 8 + [45, 60, 82] * 2 + [83, 126, 211, 5000])]
    elif case == "halves":
        # k_enc_emit writes the long event that opens a 32-position half of a tile on its own and the one-byte events behind it through the
        # wave-uniform walk: long events in the low half, in the high half (as the tile's first head and behind a head of the low half), tiles
        # of more than 64 bytes (sixty one-byte events and a long one: two block starts in one tile), every offset mod 64
        syms = [run_symbols(rng, 500000, [1, 1, 1, 1, 2, 42, 43, 44, 50, 63, 64, 65]),
                run_symbols(rng, 400000, [1] * 40 + [42, 83, 200]),
                run_symbols(rng, 300000, [1] * 12 + [2, 3, 31, 32, 33, 42, 52, 62, 82, 83, 90, 1000, 70000])]
    elif case == "giant":
        syms = [np.full(5_000_000, 3, dtype=np.uint8),
                np.concatenate([np.full(70000, 1, np.uint8), np.full(1, 2, np.uint8), np.full(4096 * 64 * 16 + 5, 4, np.uint8)])]
    elif case == "tiny":
        syms = [np.array([4], np.uint8), np.array([0, 0], np.uint8), np.full(63, 2, np.uint8), np.full(64, 2, np.uint8),
                np.full(4096, 5, np.uint8), run_symbols(rng, 3, [1])] + [run_symbols(rng, 40, [1, 2])[:k] for k in (64, 127, 128, 129)]
    else:
        syms = [np.tile(np.array([1, 2], np.uint8), 100000), np.tile(np.array([1, 1, 2], np.uint8), 65536)]
    for sym in syms:
        f = oracle.FMI.from_symbols(sym)
        d = torch.from_numpy(sym).cuda()
        ix = gpu.Index.from_symbols_device(d.data_ptr(), sym.size)
        assert ix.sequences == f.sequences and np.array_equal(ix.C, f.C)
        ix.encode()
        assert ix.nbytes == f.nbytes, case
        assert np.array_equal(ix.data(), f.data), case
        be, cum = ix.samples(); obe, ocum = f.samples
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
        ix.free()


def test_config1_shape_merge(gpu, oracle):
    """BASELINE.json configs[0] shape at reduced read count (the oracle's suffix sort bounds it)."""
    a, b, ta, tb = reads_pair(oracle, 20000, 20000, 100, 100, seed=42)
    A = gpu.Index.upload(a.data, a.sequences, a.bases)
    B = gpu.Index.upload(b.data, b.sequences, b.bases)
    M = gpu.merge(A, B)
    m, _ = oracle.merge(a, b, threads=4)
    assert np.array_equal(M.data(), m.data)
    be, cum = M.samples(); obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)


def test_partitioned_emit_rounds_fallbacks_and_variants(gpu, oracle):
    """The search's partition machinery under stress: many rounds, tiny regions (overflow ->
    exact fallback), skewed insert positions, short epochs, and every product dispatch give the same rank array.
    (Timing-only kernel variants exist only in -DBWTM_DIAGNOSTICS builds and are not part of this suite.)"""
    # B's suffixes all sort into two narrow places of A: emits concentrate in a few tiles
    rng = np.random.default_rng(21)
    a_reads = np.concatenate([np.concatenate([rng.choice([1, 2], 60), [0]]) for _ in range(1500)]).astype(np.uint8)
    b_reads = np.concatenate([np.concatenate([rng.choice([3, 4], 60), [0]]) for _ in range(1200)]).astype(np.uint8)
    cases = [(oracle.FMI.from_text(a_reads), oracle.FMI.from_text(b_reads))]
    ta = oracle.generate_reads(31, 3000, 100); tb = oracle.generate_reads(32, 2500, 100)
    cases.append((oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)))
    try:
        for a, b in cases:
            ranks, counts, _ = oracle.search(a, b, threads=2)
            ora = oracle.ra_from_runs(ranks, counts)
            A = gpu.Index.upload(a.data, a.sequences, a.bases)
            B = gpu.Index.upload(b.data, b.sequences, b.bases)
            settings = [dict(search_algo=2), dict(search_algo=2, l1_cap=5000), dict(search_algo=2, frontier_unfused=1), dict(search_algo=2, frontier_unfused=2), dict(search_algo=2, frontier_epoch=9),
                        dict(search_algo=2, emit_budget=4096), dict(search_algo=0), dict(search_algo=1), dict(search_algo=1, round_emits=20000),
                        dict(search_algo=1, round_emits=3000), dict(emit_path=1), dict(search_algo=1, l1_cap=256),
                        # the node phase (fmi.cpp:304-322): never, for every level it can take (ratio 1: the whole search on trie nodes when
                        # no level has more nodes than sequences), a few levels then elements, with short epochs behind it
                        dict(search_algo=2, range_ratio=0), dict(search_algo=2, range_ratio=1), dict(search_algo=2, range_ratio=40),
                        dict(search_algo=2, range_ratio=3, frontier_epoch=9), dict(search_algo=2, range_ratio=2, l1_cap=5000)]
            for st in settings:
                for k in ("round_emits", "emit_path", "l1_cap", "search_algo", "frontier_unfused", "frontier_epoch", "emit_budget", "range_ratio"):
                    gpu.tune(k, {"round_emits": 1 << 33, "range_ratio": -1}.get(k, 0))
                for k, v in st.items():
                    gpu.tune(k, v)
                ra = gpu.RankArray(A, B)
                ra.search(A, B, 0, b.sequences - 1)
                ra.finalize()
                assert ra.values == b.bases, st
                assert np.array_equal(ra.download(), ora), st
                ra.free()
    finally:
        for k in ("emit_path", "l1_cap", "frontier_unfused", "frontier_epoch", "emit_budget"):
            gpu.tune(k, 0)
        gpu.tune("round_emits", 1 << 33); gpu.tune("range_ratio", -1)
        gpu.tune("search_algo", 2)


@pytest.mark.parametrize("ratio", [0, -1])
def test_frontier_grid_follows_a_shrinking_frontier(gpu, oracle, ratio):
    """Reads of twelve lengths: the frontier shrinks in steps, the step kernel's grid follows it (a quarter of idle blocks triggers a
    shrink) and the segment entries of the wider grids are cleared -- stale entries would resurrect dead chains."""
    rng = np.random.default_rng(77)
    lengths = [3, 9, 20, 33, 47, 60, 75, 90, 110, 140, 170, 200]
    tb = np.concatenate([oracle.generate_reads(7000 + k, 600, n) for k, n in enumerate(lengths)])
    ta = oracle.generate_reads(7100, 4000, 120)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    ranks, counts, _ = oracle.search(a, b, threads=2)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    gpu.tune("range_ratio", ratio)
    try:
        ra = gpu.RankArray(A, B)
        ra.search(A, B, 0, b.sequences - 1)
        ra.finalize()
        assert ra.values == b.bases
        assert np.array_equal(ra.download(), oracle.ra_from_runs(ranks, counts))
        M = gpu.merge(A, B)
        assert np.array_equal(M.data(), oracle.FMI.from_text(np.concatenate([ta, tb])).data)
    finally:
        gpu.tune("range_ratio", -1)
    for x in (ra, M, A, B):
        x.free()


def test_find_batch_matches_oracle_backward_search(gpu, oracle):
    """FMI::find on the device (bwt_merge -v) against the oracle's backward search."""
    t = oracle.generate_reads(77, 600, 80)
    f = oracle.FMI.from_text(t)
    ix = gpu.Index.upload(f.data, f.sequences, f.bases)
    rng = np.random.default_rng(4)
    pats = [np.zeros(0, dtype=np.uint8)]
    for _ in range(300):
        p = int(rng.integers(0, t.size - 30))
        s = t[p:p + int(rng.integers(1, 25))]
        pats.append(s[s != 0][:20] if rng.random() < 0.7 else rng.integers(1, 6, int(rng.integers(1, 12))).astype(np.uint8))
    pats = [p for p in pats]
    sp, ep = ix.find(pats)
    for k, p in enumerate(pats):
        osp, oep = oracle.FMI.find(f, p)
        empty_o = (osp + 1) % (1 << 64) > (oep + 1) % (1 << 64)
        empty_g = (int(sp[k]) + 1) % (1 << 64) > (int(ep[k]) + 1) % (1 << 64)
        assert empty_o == empty_g, k
        if not empty_o:
            assert (int(sp[k]), int(ep[k])) == (osp, oep), k


def test_mid_size_properties_merge_tree_associativity(gpu, oracle):
    """Size-independent properties at a size the brute force cannot reach (2 x 3e5 reads = 60 Mbase):
    the same collection merged along different trees gives the same bytes, the counts add up, and
    reads extracted from the merged index by LF walk equal the generator's."""
    import torch
    from bwt_merge_amd import synth
    dev = torch.device("cuda", 0)
    n, L = 300_000, 100

    def leaf(seed, first, count):
        sym = synth.leaf_bwt(synth.generate_reads(seed, first, count, L, device=dev)).contiguous()
        torch.cuda.synchronize()
        return gpu.Index.from_symbols_device(sym.data_ptr(), sym.numel())

    quarters = [(1001, 0, n // 2), (1001, n // 2, n - n // 2), (1002, 0, n // 2), (1002, n // 2, n - n // 2)]
    # ((q0 + q1) + (q2 + q3))
    a = synth.merge_indexes(gpu, leaf(*quarters[0]), leaf(*quarters[1]))
    b = synth.merge_indexes(gpu, leaf(*quarters[2]), leaf(*quarters[3]))
    m1 = gpu.merge(a, b)
    # (((q0 + q1) + q2) + q3)
    c = synth.merge_indexes(gpu, synth.merge_indexes(gpu, leaf(*quarters[0]), leaf(*quarters[1])), leaf(*quarters[2]))
    m2 = gpu.merge(c, leaf(*quarters[3]))
    d1, d2 = m1.data(), m2.data()
    assert np.array_equal(d1, d2)
    assert m1.bases == 2 * n * (L + 1) and m1.sequences == 2 * n
    assert np.array_equal(m1.C, a.C + b.C)
    be1, cum1 = m1.samples(); be2, cum2 = m2.samples()
    assert np.array_equal(be1, be2) and np.array_equal(cum1, cum2)
    # the stream decodes back to the same index (header check in upload) and sampled reads come out
    r = gpu.Index.upload(d1, m1.sequences, m1.bases)
    ids = np.sort(np.random.default_rng(8).integers(0, 2 * n, 64))
    got = synth.extract_sequences(gpu, r, ids, max_len=L + 8)
    for j, seq in zip(ids, got):
        seed, idx = (1001, int(j)) if j < n else (1002, int(j - n))
        assert seq == synth.generate_reads(seed, idx, 1, L)[0].tolist()


def test_sharded_search_with_caller_owned_buffers(gpu, oracle):
    """What bench.py does at N > 1, on one GPU: every 'rank' searches its shard into its own zeroed
    tensor, the tensors are summed (= all_reduce(SUM) = OR, the bits are disjoint), and the result is
    finalized, interleaved and encoded."""
    import torch
    from bwt_merge_amd.dist import shard_range
    a, b, ta, tb = reads_pair(oracle, 1500, 1300, 90, 110, seed=5)
    A = gpu.Index.upload(a.data, a.sequences, a.bases)
    B = gpu.Index.upload(b.data, b.sequences, b.bases)
    world = 3
    nbytes = gpu.ra_buffer_bytes(A, B)
    bufs = []
    for rank in range(world):
        buf = torch.zeros(nbytes // 8, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        ra = gpu.RankArray(A, B, buf.data_ptr(), nbytes)
        first, last = shard_range(b.sequences, rank, world)
        ra.search(A, B, first, last)
        gpu.synchronize()
        ra.free()
        bufs.append(buf)
    total = bufs[0] + bufs[1] + bufs[2]
    torch.cuda.synchronize()
    ra = gpu.RankArray(A, B, total.data_ptr(), nbytes)
    ra.finalize()
    assert ra.values == b.bases
    M = gpu.interleave(A, B, ra).encode()
    m, _ = oracle.merge(a, b, threads=2)
    assert np.array_equal(M.data(), m.data)
    ra.free()


def test_frontier_scan_forms_agree_on_a_multi_tile_segment_table(gpu):
    """The per-step scan of the segment lengths has three forms (one launch with tagged tile totals: the default since round 5; two
    launches; the generic scan).  Small collections have a one-tile segment table, where the tiles never wait for each other; here the
    table has several tiles (6 x 10^5 sequences: 11 720 segments), and all forms must give the rank array of the per-chain walk."""
    import torch
    from bwt_merge_amd import synth
    dev = torch.device("cuda", 0)
    A = synth.build_index(gpu, 1001, 200_000, 40, leaf_reads=1 << 17, device=dev)
    B = synth.build_index(gpu, 1002, 600_000, 25, leaf_reads=1 << 17, device=dev)
    bits = {}
    try:
        for name, st in (("walk", dict(search_algo=1)), ("one_launch", dict(search_algo=2, frontier_unfused=0)), ("two_launches", dict(search_algo=2, frontier_unfused=2)),
                         ("generic", dict(search_algo=2, frontier_unfused=1)), ("one_launch_no_nodes", dict(search_algo=2, frontier_unfused=0, range_ratio=0))):
            gpu.tune("range_ratio", -1); gpu.tune("frontier_unfused", 0)
            for k, v in st.items():
                gpu.tune(k, v)
            gpu.profile_enable(True); gpu.profile_reset()
            ra = gpu.RankArray(A, B)
            ra.search(A, B, 0, B.sequences - 1)
            ra.finalize()
            prof = gpu.profile_read(); gpu.profile_enable(False)
            assert ra.values == B.bases, name
            assert ("frontier_step" in prof) == (name != "walk"), (name, sorted(prof))
            if name == "one_launch_no_nodes":
                assert prof["frontier_scan"][1] == prof["frontier_step"][1], sorted(prof)     # one scan launch per step
            bits[name] = ra.bits()
            ra.free()
    finally:
        gpu.tune("search_algo", 2); gpu.tune("frontier_unfused", 0); gpu.tune("range_ratio", -1)
    for name in bits:
        assert np.array_equal(bits[name], bits["walk"]), name
    A.free(); B.free()
