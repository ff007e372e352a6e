"""Product branches that small inputs never reach, each compared with the CPU oracle through the C ABI:
40-bit coordinates (positions beyond 2^32 in both indexes), epoch rollover of the frontier search, the dispatch of
collections of long sequences to the per-chain walk, the host-to-host entry points, the consuming merge, contexts.

Inputs of billions of positions are stated as a few megabytes of runs: the BWT of a collection in which every read
occurs `copies` times in a row is the BWT of the distinct reads with every symbol repeated `copies` times (equal
suffixes are ordered by sequence index), and the reference's trie DFS handles the equal
sequences as ranges (fmi.cpp:304-322), so the oracle finishes in seconds."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    bwtm.tune("search_algo", 0)
    yield bwtm
    for k in ("search_algo", "frontier_epoch", "emit_budget", "l1_cap", "eager_cum_budget", "upload_chunk", "download_chunk"):
        bwtm.tune(k, 0)
    bwtm.trim()


def repeated_collection(oracle, seed, nreads, readlen, copies):
    """Oracle FMI of the collection (read 0 x copies, read 1 x copies, ...)."""
    small = oracle.FMI.from_text(oracle.generate_reads(seed, nreads, readlen))
    sym = small.symbols.astype(np.uint64)
    return oracle.FMI.from_runs(sym, np.full(sym.size, copies, dtype=np.uint64))


def upload(gpu, f):
    return gpu.Index.upload(f.data, f.sequences, f.bases)


def search_runs(gpu, A, B, b, algo, range_ratio=-1):
    gpu.tune("search_algo", algo); gpu.tune("range_ratio", range_ratio)   # -1 = the default
    gpu.profile_enable(True); gpu.profile_reset()
    ra = gpu.RankArray(A, B)
    ra.search(A, B, 0, b.sequences - 1)
    ra.finalize()
    prof = gpu.profile_read()
    gpu.profile_enable(False)
    gpu.tune("search_algo", 0); gpu.tune("range_ratio", -1)
    assert ra.values == b.bases
    ranks, counts = ra.runs()
    ra.free()
    return ranks, counts, prof


def test_coordinates_beyond_32_bits(gpu, oracle):
    """Both indexes hold > 2^32 positions, so the high bytes of the frontier's 40-bit coordinates (i in B, r in A), bit
    positions beyond 2^33 and tiles past 65 536 are all exercised, in the frontier search, in the walk, in the
    interleave and in the encoder -- with the sizes of BASELINE config 2 (n = 5.05e9 > 2^32) in mind."""
    copies = 10500
    a = repeated_collection(oracle, 501, 4096, 100, copies)
    b = repeated_collection(oracle, 502, 4096, 100, copies)
    assert a.bases > (1 << 32) and b.bases > (1 << 32)
    oranks, ocounts, _ = oracle.search(a, b, capacity=1 << 21, threads=8)
    A, B = upload(gpu, a), upload(gpu, b)
    # BWT::extract of MORE than 2^32 positions in one call (one thread per position and a grid holds fewer than 2^32 threads: until round 5 the
    # call returned zeros behind the first 2^32 -- and tools/scale_cli.sh wrote config 2's inputs through it)
    sym = A.extract(0, a.bases)
    for first in (0, (1 << 32) - 70000, (1 << 32) + 123457, a.bases - 100000):
        probe = list(range(first, min(first + 100000, a.bases), 997))
        assert [int(sym[i]) for i in probe] == [a.at(i) for i in probe], first
    assert int(np.count_nonzero(sym[1 << 32:])) > (a.bases - (1 << 32)) // 2
    del sym
    # the frontier search alone (one element per sequence), the node phase alone (every level of this collection has at most 4096
    # trie nodes: fmi.cpp:304-322, the reference's own form), five levels of nodes expanded into 4.3e7 elements, and the walk
    for algo, kernel, ratio, absent in ((2, "frontier_step", 0, "range_step"), (2, "range_step", -1, "frontier_step"), (2, "frontier_step", 71680, None),
                                        (1, "lf_walk", -1, "range_step")):
        ranks, counts, prof = search_runs(gpu, A, B, b, algo, ratio)
        assert prof.get(kernel, (0, 0))[1] > 0, (algo, sorted(prof))
        assert absent is None or absent not in prof, (algo, ratio, sorted(prof))
        if ratio == 71680:
            assert prof["range_step"][1] == 5 and prof["range_expand"][1] == 1, prof
        assert np.array_equal(ranks, oranks) and np.array_equal(counts, ocounts), (algo, ratio)
        assert int(ranks.max()) > (1 << 32)
    # the whole path at this size: bytes, C and samples of the oracle's merge
    M = gpu.merge(A, B)
    m, _ = oracle.merge(a, b, threads=8)
    assert (M.sequences, M.bases) == (m.sequences, m.bases)
    assert np.array_equal(M.data(), m.data) and np.array_equal(M.C, m.C)
    be, cum = M.samples(); obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
    # random access into the merged rank structure beyond 2^32
    rng = np.random.default_rng(5)
    pos = rng.integers(1 << 32, M.bases, 2000).astype(np.uint64)
    comps = rng.integers(0, 6, pos.size).astype(np.uint8)
    got = M.rank(pos, comps)
    expect = np.array([m.rank(int(p), int(c)) for p, c in zip(pos, comps)], dtype=np.uint64)
    assert np.array_equal(got, expect)
    for x in (A, B, M):
        x.free()
    gpu.trim()


@pytest.mark.parametrize("epoch,budget,ratio", [(0, 0, 0), (64, 0, 0), (7, 0, 0), (0, 3000, 0), (7, 0, -1), (64, 0, 2)])
def test_frontier_epoch_rollover(gpu, oracle, epoch, budget, ratio):
    """Sequences longer than an epoch of the frontier search (512 steps by default; fewer when the dense emits of an
    epoch would exceed the emit budget): tiles are built at every epoch boundary and the search goes on."""
    ta = oracle.generate_reads(61, 300, 400); tb = oracle.generate_reads(62, 250, 1300)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    oranks, ocounts, _ = oracle.search(a, b, threads=2)
    A, B = upload(gpu, a), upload(gpu, b)
    gpu.tune("frontier_epoch", epoch); gpu.tune("emit_budget", budget)
    try:
        ranks, counts, prof = search_runs(gpu, A, B, b, 2, ratio)
    finally:
        gpu.tune("frontier_epoch", 0); gpu.tune("emit_budget", 0)
    assert prof["frontier_step"][1] + prof.get("range_step", (0, 0))[1] >= 1300 and (ratio == 0) == ("range_step" not in prof)
    assert prof["tile_build"][1] >= (3 if epoch == 0 and budget == 0 else 20)        # one tile build per epoch
    assert np.array_equal(ranks, oranks) and np.array_equal(counts, ocounts)
    M = gpu.merge(A, B)
    assert np.array_equal(M.data(), oracle.FMI.from_text(np.concatenate([ta, tb])).data)


def test_long_sequences_take_the_walk(gpu, oracle):
    """bwtm_search sends collections whose average sequence is longer than 4096 symbols to the per-chain walk even
    when there are more than 2^21 of them; the walk then runs several rounds (round_emits)."""
    b = repeated_collection(oracle, 71, 16, 5000, 135000)
    a = oracle.FMI.from_text(oracle.generate_reads(72, 3000, 100))
    assert b.sequences >= (1 << 21) and b.bases // b.sequences > 4096
    oranks, ocounts, _ = oracle.search(a, b, capacity=1 << 20, threads=8)
    A, B = upload(gpu, a), upload(gpu, b)
    ranks, counts, prof = search_runs(gpu, A, B, b, 0)
    assert "frontier_step" not in prof and prof["lf_walk"][1] >= 2, sorted(prof)
    assert np.array_equal(ranks, oranks) and np.array_equal(counts, ocounts)
    A.free(); B.free()
    gpu.trim()


def test_rank_array_runs_form(gpu, oracle):
    """bwtm_ra_download_runs returns what the reference's RankArray iterates: maximal (rank, count) runs."""
    for na, nb, L in ((1, 1, 1), (3, 2, 5), (700, 900, 60), (2500, 2000, 100)):
        a = oracle.FMI.from_text(oracle.generate_reads(900 + na, na, L))
        b = oracle.FMI.from_text(oracle.generate_reads(950 + nb, nb, L))
        oranks, ocounts, _ = oracle.search(a, b, threads=2)
        A, B = upload(gpu, a), upload(gpu, b)
        ra = gpu.RankArray(A, B)
        ra.search(A, B, 0, b.sequences - 1)
        ra.finalize()
        ranks, counts = ra.runs()
        assert np.array_equal(ranks, oranks) and np.array_equal(counts, ocounts)
        assert np.array_equal(np.repeat(ranks, counts.astype(np.int64)), ra.download())
        ra.free()
    # an empty increment has no runs
    e = oracle.FMI.from_symbols(np.zeros(0, dtype=np.uint8))
    E = gpu.Index.upload(e.data, 0, 0)
    ra = gpu.RankArray(A, E).finalize()
    assert ra.runs()[0].size == 0


def test_shards_in_separate_buffers_of_one_gpu(gpu, oracle):
    """bwtm_ra_or_from: shards searched into separate rank arrays on one device are united (what the multi-GPU host does
    when its "GPUs" are contexts of one GPU; across devices the same union is an all-reduce)."""
    a = oracle.FMI.from_text(oracle.generate_reads(8300, 1500, 80)); b = oracle.FMI.from_text(oracle.generate_reads(8301, 1700, 90))
    oranks, ocounts, _ = oracle.search(a, b, threads=2)
    A, B = upload(gpu, a), upload(gpu, b)
    parts = []
    for first, last in ((0, 499), (500, 1199), (1200, 1699)):
        ra = gpu.RankArray(A, B)
        ra.search(A, B, first, last)
        parts.append(ra)
    parts[0].or_from(parts[1]); parts[0].or_from(parts[2])
    parts[0].finalize()
    ranks, counts = parts[0].runs()
    assert np.array_equal(ranks, oranks) and np.array_equal(counts, ocounts)
    with pytest.raises(gpu.BwtmError):
        parts[0].or_from(parts[1])                              # already finalized


def test_upload_rejects_inconsistent_alphabet(gpu, oracle):
    """A caller-supplied C must agree with the symbol counts of the stream (it feeds every LF step)."""
    f = oracle.FMI.from_text(oracle.generate_reads(5, 40, 30))
    ix = gpu.Index.upload(f.data, f.sequences, f.bases, f.C)
    assert np.array_equal(ix.C, f.C)
    bad = f.C.copy(); bad[3] += 1
    with pytest.raises(gpu.BwtmError, match="alphabet"):
        gpu.Index.upload(f.data, f.sequences, f.bases, bad)


@pytest.mark.parametrize("chunks", [(0, 0), (1, 1), (8192, 4096)])
def test_host_to_host_merge(gpu, oracle, chunks):
    """bwtm_merge_host: page-locked native bytes in, page-locked native bytes + samples out, with the uploads, the
    device work and the download pipelined -- same bytes as the oracle, also with tiny chunks (many pipeline stages),
    and chained with the device-resident result of the previous call (bwt_merge.cpp:167-173)."""
    sets = [oracle.generate_reads(4000 + k, 3000 + 500 * k, 100 if k != 1 else 150) for k in range(3)]
    fm = [oracle.FMI.from_text(t) for t in sets]
    pinned = []
    for f in fm:
        hb = gpu.HostBuffer(f.nbytes)
        hb.array[:] = f.data
        pinned.append(hb)
    gpu.tune("upload_chunk", chunks[0]); gpu.tune("download_chunk", chunks[1])
    try:
        r01 = gpu.merge_host((pinned[0].array, fm[0].sequences, fm[0].bases), (pinned[1].array, fm[1].sequences, fm[1].bases), keep=True)
        m01, _ = oracle.merge(fm[0].clone(), fm[1].clone(), threads=2)
        assert (r01.out.sequences, r01.out.bases, r01.out.nbytes, r01.out.blocks) == (m01.sequences, m01.bases, m01.nbytes, m01.blocks)
        assert np.array_equal(r01.data, m01.data) and np.array_equal(r01.C, m01.C)
        obe, ocum = m01.samples
        assert np.array_equal(r01.block_end, obe) and np.array_equal(r01.cum, ocum)
        assert r01.times["ms_total"] > 0
        kept, r01.keep = r01.keep, None
        r012 = gpu.merge_host(None, (pinned[2].array, fm[2].sequences, fm[2].bases), samples=False, chained=kept)
        direct = oracle.FMI.from_text(np.concatenate(sets))
        assert np.array_equal(r012.data, direct.data) and np.array_equal(r012.C, direct.C)
        assert 1 not in r012.buffers and 2 not in r012.buffers           # no samples requested
        r01.free(); r012.free()
        # a header that does not match the stream is refused after the pipeline has drained
        with pytest.raises(gpu.BwtmError, match="header says"):
            gpu.merge_host((pinned[0].array, fm[0].sequences + 1, fm[0].bases), (pinned[1].array, fm[1].sequences, fm[1].bases))
        # ... and so is a header that UNDERSTATES the bases of input2: its transcode sizes the record array from the header, so the
        # header must be checked against the stream before that kernel is queued (it used to run first)
        for wrong in (fm[1].bases // 2, 128, fm[1].bases + 4096):
            with pytest.raises(gpu.BwtmError, match="header says"):
                gpu.merge_host((pinned[0].array, fm[0].sequences, fm[0].bases), (pinned[1].array, fm[1].sequences, wrong))
        with pytest.raises(gpu.BwtmError, match="header says"):
            gpu.merge_host((pinned[0].array, fm[0].sequences, fm[0].bases // 3), (pinned[1].array, fm[1].sequences, fm[1].bases))
        # a stream that no Run::write produced (a full block of one-position runs of alternating symbols decodes to 64 positions: fine;
        # extension bytes >= 0x80 forever: not canonical) is refused as well, whatever its header says
        bad = gpu.HostBuffer(4096)
        bad.array[:] = 0xFB                                   # run heads with basic length 42 whose extension never ends inside a block
        with pytest.raises(gpu.BwtmError):
            gpu.merge_host((pinned[0].array, fm[0].sequences, fm[0].bases), (bad.array, 1, 64))
        bad.free()
        r = gpu.merge_host((pinned[0].array, fm[0].sequences, fm[0].bases), (pinned[1].array, fm[1].sequences, fm[1].bases))
        assert np.array_equal(r.data, m01.data)                 # the context is still healthy after the refused calls
        r.free()
        # pageable inputs work too (slower copies)
        r = gpu.merge_host((fm[0].data, fm[0].sequences, fm[0].bases), (fm[1].data, fm[1].sequences, fm[1].bases))
        assert np.array_equal(r.data, m01.data)
        r.free()
    finally:
        gpu.tune("upload_chunk", 0); gpu.tune("download_chunk", 0)
        for hb in pinned:
            hb.free()


@pytest.mark.parametrize("chunk", [0, 4096])
def test_pipelined_chain_of_host_merges(gpu, oracle, chunk):
    """bwt_merge in0 in1 in2 in3 as three bwtm_merge_host_pipelined calls: every merge announces the input of the next one, whose
    bytes travel under its search; intermediate results stay on the device (BWTM_RESULT_ON_DEVICE), the last merge downloads.
    The bytes and samples equal the oracle's BWT of the whole collection; a pending upload can also be finished on its own
    (bwtm_upload_begin / _finish == bwtm_index_upload) or dropped."""
    sets = [oracle.generate_reads(5000 + k, 2500 + 400 * k, 100 if k % 2 == 0 else 150) for k in range(4)]
    fm = [oracle.FMI.from_text(t) for t in sets]
    pinned = []
    for f in fm:
        hb = gpu.HostBuffer(f.nbytes)
        hb.array[:] = f.data
        pinned.append(hb)
    inp = [(pinned[k].array, fm[k].sequences, fm[k].bases) for k in range(4)]
    gpu.tune("upload_chunk", chunk)
    try:
        r, pend = gpu.merge_host_pipelined(a=inp[0], b=inp[1], next=inp[2], samples=gpu.RESULT_ON_DEVICE, keep=True)
        assert r.out.nbytes == 0 and 0 not in r.buffers and r.out.bases == fm[0].bases + fm[1].bases
        kept, r.keep = r.keep, None
        r, pend = gpu.merge_host_pipelined(chained=kept, pending=pend, next=inp[3], samples=gpu.RESULT_ON_DEVICE, keep=True)
        kept, r.keep = r.keep, None
        r, none = gpu.merge_host_pipelined(chained=kept, pending=pend, samples=2)
        assert none is None
        direct = oracle.FMI.from_text(np.concatenate(sets))
        assert np.array_equal(r.data, direct.data) and np.array_equal(r.C, direct.C)
        be, cum = r.expanded_samples(); obe, ocum = direct.samples
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
        r.free()
        # the announcement on its own
        u = gpu.upload_begin(*inp[1])
        X = u.finish()
        Y = gpu.Index.upload(fm[1].data, fm[1].sequences, fm[1].bases)
        w0 = np.arange(0, fm[1].bases, 37, dtype=np.uint64); cs = (w0 % 6).astype(np.uint8)
        assert np.array_equal(X.rank(w0, cs), Y.rank(w0, cs)) and np.array_equal(X.C, Y.C)
        X.free(); Y.free()
        gpu.upload_begin(*inp[2]).free()                                 # never consumed
        with pytest.raises(gpu.BwtmError, match="header says"):
            gpu.upload_begin(pinned[3].array, fm[3].sequences, fm[3].bases - 5).finish()
        # a pending upload with a wrong header is refused by the merge that consumes it
        kept = gpu.Index.upload(fm[0].data, fm[0].sequences, fm[0].bases)
        bad = gpu.upload_begin(pinned[1].array, fm[1].sequences + 2, fm[1].bases)
        with pytest.raises(gpu.BwtmError, match="header says"):
            gpu.merge_host_pipelined(chained=kept, pending=bad, samples=False)
        with pytest.raises(gpu.BwtmError):
            gpu.merge_host_pipelined(a=inp[0], b=inp[1], samples=gpu.RESULT_ON_DEVICE)      # a device-only result needs `keep`
    finally:
        gpu.tune("upload_chunk", 0)
        for hb in pinned:
            hb.free()


def test_compact_samples(gpu, oracle):
    """The 6 / 12 / 24-byte form of the samples (fields + anchors) expands to exactly the arrays BWT::build computes, at both
    widths, and the library falls back to the full arrays when a block encodes 2^32 - 1 positions or more."""
    rng = np.random.default_rng(23)
    cases = []
    syms = rng.integers(0, 6, 40000); lens = rng.choice([1, 1, 2, 3, 41, 42, 170], 40000)
    cases.append((oracle.FMI.from_symbols(np.repeat(syms.astype(np.uint8), lens)), 2))                 # 16-bit fields
    syms = rng.integers(0, 6, 3000).astype(np.uint64); lens = rng.choice([1, 2, 50, 70000, 300000], 3000).astype(np.uint64)
    cases.append((oracle.FMI.from_runs(syms, lens), 4))                                               # runs >= 65535: 32-bit fields
    cases.append((oracle.FMI.from_runs(np.array([1, 2, 3], dtype=np.uint64), np.array([10, 5_000_000_000, 7], dtype=np.uint64)), 8))
    cases.append((oracle.FMI.from_symbols(np.array([3], dtype=np.uint8)), 1))
    cases.append((oracle.FMI.from_text(oracle.generate_reads(77, 4000, 100)), 1))                    # a read collection: 8-bit fields
    syms = rng.integers(0, 6, 60000); lens = rng.choice([1, 2, 3], 60000)
    cases.append((oracle.FMI.from_symbols(np.repeat(syms.astype(np.uint8), lens)), 1))
    for f, expect_width in cases:
        ix = upload(gpu, f)
        width, fields, anchors = ix.samples_compact()
        assert width == expect_width
        obe, ocum = f.samples
        if width != 8:
            be, cum = gpu.capi.expand_samples(width, fields, anchors, f.blocks, f.bases)
            assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
            if width < 4:
                for wider in (2, 4)[(width == 2):]:                               # a wider form may always be asked for
                    w = ix.samples_compact(wider)
                    be, cum = gpu.capi.expand_samples(wider, w[1], w[2], f.blocks, f.bases)
                    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
                if width == 2:
                    with pytest.raises(gpu.BwtmError, match="too many positions"):
                        ix.samples_compact(1)
            else:
                with pytest.raises(gpu.BwtmError, match="too many positions"):
                    ix.samples_compact(2)
        be, cum = ix.samples()
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum)
        # through the host-to-host call
        e = np.zeros(0, dtype=np.uint8)
        r = gpu.merge_host((f.data, f.sequences, f.bases), (e, 0, 0), samples=2)
        assert r.out.sample_width == expect_width
        be, cum = r.expanded_samples()
        assert np.array_equal(be, obe) and np.array_equal(cum, ocum) and np.array_equal(r.data, f.data)
        r.free(); ix.free()
    gpu.trim()


def test_host_to_host_empty_inputs(gpu, oracle):
    a = oracle.FMI.from_text(oracle.generate_reads(7, 50, 20))
    e = np.zeros(0, dtype=np.uint8)
    r = gpu.merge_host((a.data, a.sequences, a.bases), (e, 0, 0))
    assert np.array_equal(r.data, a.data)
    obe, ocum = a.samples
    assert np.array_equal(r.block_end, obe) and np.array_equal(r.cum, ocum)
    r.free()
    r = gpu.merge_host((e, 0, 0), (a.data, a.sequences, a.bases))
    assert np.array_equal(r.data, a.data)
    r.free()
    r = gpu.merge_host((e, 0, 0), (e, 0, 0))
    assert r.out.nbytes == 0 and r.out.bases == 0
    r.free()


def test_consuming_merge_and_lazy_samples(gpu, oracle):
    """bwtm_merge_consume ("merges a and b, destroying them", fmi.h:107-109) and the samples produced in chunks at
    download time when they are not materialized by the encoder (eager_cum_budget)."""
    a, b = (oracle.FMI.from_text(oracle.generate_reads(8100 + k, 3000, 100)) for k in range(2))
    A, B = upload(gpu, a), upload(gpu, b)
    gpu.tune("eager_cum_budget", 1)                  # nothing fits: cumulative counts are answered at download time
    try:
        M = gpu.merge_consume(A, B)
    finally:
        gpu.tune("eager_cum_budget", 0)
    assert A.h is None and B.h is None
    m, _ = oracle.merge(a, b, threads=2)
    assert np.array_equal(M.data(), m.data)
    be, cum = M.samples(); obe, ocum = m.samples
    assert np.array_equal(be, obe) and np.array_equal(cum, ocum)


def test_two_contexts_two_threads(gpu, oracle):
    """One host thread per GPU (here: two contexts on one GPU): concurrent merges in different contexts do not
    interfere, and a handle is usable from a thread other than the one that created it."""
    jobs = []
    for k in range(2):
        a = oracle.FMI.from_text(oracle.generate_reads(8200 + 2 * k, 2500, 100))
        b = oracle.FMI.from_text(oracle.generate_reads(8201 + 2 * k, 2000, 100))
        m, _ = oracle.merge(a.clone(), b.clone(), threads=2)
        jobs.append((a, b, m))
    results, errors = [None, None], []

    def work(k):
        try:
            ctx = gpu.Context(0)
            ctx.make_current()
            a, b, _ = jobs[k]
            out = []
            for _ in range(5):
                A, B = upload(gpu, a), upload(gpu, b)
                M = gpu.merge(A, B)
                out.append(M.data())
                for x in (A, B, M):
                    x.free()
            results[k] = (out, upload(gpu, a), ctx)
        except Exception as e:          # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        out, handle, ctx = results[k]
        for d in out:
            assert np.array_equal(d, jobs[k][2].data)
        # the handle created by the worker thread in its own context, used from this thread
        assert np.array_equal(handle.extract(0, 64), jobs[k][0].symbols[:64])
        handle.free()
        ctx.destroy()
    # mixing handles of different contexts is refused
    ctx = gpu.Context(0); ctx.make_current()
    X = upload(gpu, jobs[0][0])
    gpu.make_default_current()
    Y = upload(gpu, jobs[0][1])
    with pytest.raises(gpu.BwtmError, match="context"):
        gpu.merge(X, Y)
    X.free(); Y.free(); ctx.destroy()
