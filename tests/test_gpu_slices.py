"""The result of a merge left sharded by output range (bwtm_interleave_range / bwtm_slice_*): produced in 1, 2, 3 and 8
slices on one GPU, with the encoder's carries exchanged the way dist.py exchanges them between ranks, the concatenation
of the slices' bytes and samples must equal the oracle's merged stream."""
import numpy as np
import pytest

from test_gpu_parity import run_symbols

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    return bwtm


def sliced_result(gpu, A, B, ra, parts):
    from bwt_merge_amd.dist import fold_offsets
    nrecs = gpu.merged_records(A, B)
    slices = [gpu.Slice(A, B, ra, *gpu.slice_bounds(nrecs, parts, g)) for g in range(parts)]
    heads = [s.lasthead() for s in slices]
    tables = [s.size_table(max(heads[:g], default=0)) for g, s in enumerate(slices)]
    offsets = fold_offsets(tables)
    assert [int(x) for x in gpu.fold_offsets(np.array(tables))] == offsets
    for s, off in zip(slices, offsets):
        s.encode(off)
        assert s.byte_first == off
    data = np.concatenate([s.data() for s in slices])
    assert data.size == offsets[-1]
    starts = [s.first_block_start() for s in slices]
    be, cum = [], []
    n = A.bases + B.bases
    for g, s in enumerate(slices):
        nxt = next((p for p in starts[g + 1:] if p is not None), n)
        b, c = s.samples(nxt)
        be.append(b); cum.append(c)
    blocks = sum(s.blocks for s in slices)
    assert [s.block_first for s in slices] == list(np.cumsum([0] + [s.blocks for s in slices[:-1]]))
    for s in slices:
        s.free()
    return data, np.concatenate(be), np.concatenate(cum, axis=1), blocks


@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_sliced_merge_of_read_sets(gpu, oracle, parts):
    ta = oracle.generate_reads(9001, 5000, 100); tb = oracle.generate_reads(9002, 4000, 100)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    ra = gpu.RankArray(A, B)
    ra.search(A, B, 0, b.sequences - 1)
    ra.finalize()
    data, be, cum, blocks = sliced_result(gpu, A, B, ra, parts)
    m, _ = oracle.merge(a, b, threads=2)
    assert np.array_equal(data, m.data)
    obe, ocum = m.samples
    assert blocks == m.blocks and np.array_equal(be, obe) and np.array_equal(cum, ocum[:, :-1])
    # a slice answers for its own positions
    nrecs = gpu.merged_records(A, B)
    f, l = gpu.slice_bounds(nrecs, max(parts, 2), 1)
    s = gpu.Slice(A, B, ra, f, l)
    sym = m.symbols
    assert np.array_equal(s.extract(f * 128, 1000), sym[f * 128: f * 128 + 1000])
    with pytest.raises(gpu.BwtmError):
        s.extract(f * 128 - 1, 2)
    with pytest.raises(gpu.BwtmError):
        gpu.Slice(A, B, ra, 100, 612)                       # not on a segment boundary
    ra.free()


@pytest.mark.parametrize("parts", [1, 2, 3, 5, 8])
def test_range_finalize_after_a_reduce_scatter(gpu, oracle, parts):
    """The exchange the merge runs on several GPUs: every part owns an EQUAL range of the output and a bitvector that is complete only
    inside it (what a reduce-scatter leaves: here a caller-owned copy of the full bitvector with every byte outside the range
    overwritten).  bwtm_ra_range_counts of every part -> the small exchange (dist.combine_range_counts' arithmetic) ->
    bwtm_ra_finalize_range -> bwtm_interleave_range; the slices' bytes and samples equal the oracle's merge.  Large enough for
    several segments per part."""
    import torch
    from bwt_merge_amd.dist import fold_offsets, super_owners
    ta = oracle.generate_reads(9101, 9000, 100); tb = oracle.generate_reads(9102, 7000, 100)
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
    whole = gpu.RankArray(A, B)
    whole.search(A, B, 0, b.sequences - 1)
    full_bits = whole.finalize().bits()                                  # n_out bits as 64-bit words
    whole.free()
    nrecs = gpu.merged_records(A, B)
    bounds = [gpu.slice_bounds_equal(nrecs, parts, g) for g in range(parts)]
    shard_bytes = bounds[0][2]
    dev = torch.device("cuda", 0)
    ras, bufs, counts = [], [], []
    for g, (first, last, _) in enumerate(bounds):
        # what a reduce-scatter delivers: this part's equal share [g, g + 1) x shard_bytes of the summed bitvector (zero behind the
        # last output position, like everybody's contribution there); everything outside the share is garbage
        sw = shard_bytes // 8
        words = np.full(parts * sw, 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
        words[g * sw: (g + 1) * sw] = 0
        lo, hi = min(g * sw, full_bits.size), min((g + 1) * sw, full_bits.size)
        words[lo: hi] = full_bits[lo: hi]
        t = torch.from_numpy(words.view(np.int64)).to(dev)
        ra = gpu.RankArray(A, B, t.data_ptr(), t.numel() * 8)
        counts.append(ra.range_counts(first, last))
        ras.append(ra); bufs.append(t)
    totals = [c[0] for c in counts]
    assert sum(totals) == b.bases
    nsup = counts[0][1].size
    owner = super_owners(nsup, [(x[0], x[1]) for x in bounds])
    prefix = np.concatenate([[0], np.cumsum(totals)]).astype(np.uint64)
    super_boff = prefix[owner] + sum(c[1] for c in counts)
    slices = []
    for g, (first, last, _) in enumerate(bounds):
        halo = next((counts[h][2] for h in range(g - 1, -1, -1) if bounds[h][1] > bounds[h][0]), None)
        ras[g].finalize_range(first, last, int(prefix[g]), int(prefix[parts]), super_boff, halo)
        assert ras[g].values == b.bases
        with pytest.raises(gpu.BwtmError):
            gpu.interleave(A, B, ras[g])                                  # a ranged array serves its range only
        if parts > 1:
            with pytest.raises(gpu.BwtmError):
                gpu.Slice(A, B, ras[g], *gpu.slice_bounds(nrecs, parts + 1, 0))
        slices.append(gpu.Slice(A, B, ras[g], first, last))
    heads = [s.lasthead() for s in slices]
    tables = [s.size_table(max(heads[:g], default=0)) for g, s in enumerate(slices)]
    offsets = fold_offsets(tables)
    for s, off in zip(slices, offsets):
        s.encode(off)
    m, _ = oracle.merge(a, b, threads=2)
    assert np.array_equal(np.concatenate([s.data() for s in slices]), m.data)
    starts = [s.first_block_start() for s in slices]
    be, cum = [], []
    for g, s in enumerate(slices):
        nxt = next((p for p in starts[g + 1:] if p is not None), a.bases + b.bases)
        x, y = s.samples(nxt)
        be.append(x); cum.append(y)
    obe, ocum = m.samples
    assert np.array_equal(np.concatenate(be), obe) and np.array_equal(np.concatenate(cum, axis=1), ocum[:, :-1])
    for x in slices + ras + [A, B]:
        x.free()


@pytest.mark.parametrize("case", ["long_runs", "one_run", "runs_on_cuts", "tiny"])
def test_slices_across_long_runs(gpu, oracle, case):
    """Runs that cross one or several slice boundaries (slices without any run head, runs that end exactly at a cut,
    blocks opened by a run that began slices earlier): merging with an empty increment makes the interleave a copy, so
    any string can be put through the sliced encoder."""
    rng = np.random.default_rng(17)
    if case == "long_runs":
        sym = run_symbols(rng, 3000, [1, 2, 3, 41, 42, 43, 170, 3000, 16426, 100000, 400000])
    elif case == "one_run":
        sym = np.concatenate([np.full(7, 2, np.uint8), np.full(3_000_000, 4, np.uint8), np.full(5, 1, np.uint8)])
    elif case == "runs_on_cuts":
        # runs that end exactly on multiples of 65 536 positions (segment = possible slice boundary)
        sym = np.concatenate([np.full(65536, 1 + (k % 5), np.uint8) if k % 3 else run_symbols(rng, 1, [65536]) for k in range(40)])
        sym = np.concatenate([sym, run_symbols(rng, 50000, [1, 2, 3])])
    else:
        sym = run_symbols(rng, 30, [1, 2, 50])
    f = oracle.FMI.from_symbols(sym)
    e = oracle.FMI.from_symbols(np.zeros(0, dtype=np.uint8))
    A = gpu.Index.upload(f.data, f.sequences, f.bases); E = gpu.Index.upload(e.data, 0, 0)
    ra = gpu.RankArray(A, E).finalize()
    obe, ocum = f.samples
    for parts in (1, 2, 5, 8, 16):
        data, be, cum, blocks = sliced_result(gpu, A, E, ra, parts)
        assert np.array_equal(data, f.data), (case, parts)
        assert blocks == f.blocks and np.array_equal(be, obe) and np.array_equal(cum, ocum[:, :-1]), (case, parts)
    ra.free()
