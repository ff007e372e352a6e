"""Host-side multi-GPU logic on CPU: partitioning and the one exchange step, world_size 2, gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_ranges_match_reference_bounds(bwtm, golden, oracle):
    from bwt_merge_amd.dist import get_bounds, shard_range
    rows = golden["get_bounds"]["rows"]
    assert get_bounds(rows[0]["first"], rows[0]["last"], rows[0]["blocks"]) == [tuple(x) for x in rows[0]["bounds"]]
    for (first, last, blocks) in [(0, 99999, 32), (0, 6, 8), (5, 5, 3), (0, 1000, 7)]:
        assert get_bounds(first, last, blocks) == oracle.get_bounds(first, last, blocks)
    assert shard_range(0, 0, 2) == (1, 0)
    assert shard_range(1, 1, 2) == (1, 0)            # fewer sequences than ranks: empty shard
    cover = [shard_range(1001, r, 8) for r in range(8)]
    assert cover[0][0] == 0 and cover[-1][1] == 1000
    assert all(cover[k][1] + 1 == cover[k + 1][0] for k in range(7))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, result_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import _pkg
    _pkg.load()
    from bwt_merge_amd.dist import exchange_bitvector, shard_range
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ta = orc.generate_reads(1001, 60, 40); tb = orc.generate_reads(1002, 45, 40)
    a = orc.FMI.from_text(ta); b = orc.FMI.from_text(tb)
    first, last = shard_range(b.sequences, rank, world)
    # stand-in for k_lf_walk on this rank's shard: the oracle's per-sequence walk
    n_out = a.bases + b.bases
    bits = np.zeros((n_out + 63) // 64, dtype=np.uint64)
    for j in range(first, last + 1):
        i, r = j, a.sequences
        while True:
            p = i + r
            bits[p >> 6] |= np.uint64(1) << np.uint64(p & 63)
            nxt, c = b.LF(i)
            if c == 0:
                break
            i = nxt; r = a.LF(r, c)
    words = torch.from_numpy(bits.view(np.int64).copy())
    exchange_bitvector(words, dist)
    np.save(os.path.join(result_dir, "bits_%d.npy" % rank), words.numpy().view(np.uint64))
    dist.destroy_process_group()


def test_two_rank_exchange_equals_full_rank_array(tmp_path, oracle):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    ta = oracle.generate_reads(1001, 60, 40); tb = oracle.generate_reads(1002, 45, 40)
    a = oracle.FMI.from_text(ta); b = oracle.FMI.from_text(tb)
    ranks, counts, _ = oracle.search(a, b, threads=1)
    ra = oracle.ra_from_runs(ranks, counts)
    expect = np.zeros(a.bases + b.bases, dtype=np.uint8)
    expect[np.arange(b.bases, dtype=np.uint64) + ra] = 1
    for r in range(world):
        got = np.unpackbits(np.load(tmp_path / ("bits_%d.npy" % r)).view(np.uint8), bitorder="little")[: expect.size]
        assert np.array_equal(got, expect)


# ---------------------------------------------------------------------------------------------------------
# The exchange as the merge runs it: reduce-scatter of the bitvector by EQUAL output ranges + one small all-reduce
# (bwt-merge_amd/dist.py: exchange_bitvector_ranges, combine_range_counts), world_size 2 and 3 over gloo.  What the library computes
# on a GPU for a range (bwtm_ra_range_counts) is restated with numpy here.

def _walk_bits(orc, a, b, first, last):
    n_out = a.bases + b.bases
    bits = np.zeros((n_out + 63) // 64, dtype=np.uint64)
    for j in range(first, last + 1):
        i, r = j, a.sequences
        while True:
            p = i + r
            bits[p >> 6] |= np.uint64(1) << np.uint64(p & 63)
            nxt, c = b.LF(i)
            if c == 0:
                break
            i = nxt; r = a.LF(r, c)
    return bits


def _popcount(words):
    return int(np.unpackbits(np.ascontiguousarray(words).view(np.uint8)).sum())


def _range_worker(rank, world, port, result_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import _pkg
    pkg = _pkg.load()
    from bwt_merge_amd.dist import combine_range_counts, exchange_bitvector_ranges, shard_range
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ta = orc.generate_reads(1001, 900, 100); tb = orc.generate_reads(1002, 700, 100)
    a = orc.FMI.from_text(ta); b = orc.FMI.from_text(tb)
    n_out = a.bases + b.bases
    nrecs = (n_out >> 7) + 1
    bounds = [pkg.slice_bounds_equal(nrecs, world, g) for g in range(world)]
    rec_first, rec_last, shard_bytes = bounds[rank]
    shard_words = shard_bytes // 8
    first, last = shard_range(b.sequences, rank, world)
    mine = _walk_bits(orc, a, b, first, last)
    words = np.zeros(world * shard_words, dtype=np.uint64)
    words[: mine.size] = mine
    t = torch.from_numpy(words.view(np.int64).copy())
    exchange_bitvector_ranges(t, shard_words, rank, world, dist, torch)
    got = t.numpy().view(np.uint64)
    own = got[rank * shard_words: (rank + 1) * shard_words]
    # bwtm_ra_range_counts on this rank's range, in numpy
    nsup = (n_out >> 25) + 1
    c0, c1 = rec_first >> 6, -(-rec_last // 64)
    ones = _popcount(got[2 * rec_first: 128 * c1]) if rec_last > rec_first else 0
    local = np.zeros(nsup, dtype=np.uint64)                   # super 0 starts at record 0: local offset 0 for its owner
    tail = (got[128 * (c1 - 1): 128 * c1].copy() if rec_last > rec_first else np.zeros(128, dtype=np.uint64))
    before, total, super_boff, halo = combine_range_counts(ones, local, tail, [(x[0], x[1]) for x in bounds], rank, world, dist, torch, torch.device("cpu"))
    np.savez(os.path.join(result_dir, "range_%d.npz" % rank), own=own, rec_first=rec_first, rec_last=rec_last, before=before, total=total,
             super_boff=super_boff, halo=(halo if halo is not None else np.zeros(0, dtype=np.uint64)), shard_words=shard_words)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_by_output_range(tmp_path, oracle, bwtm, world):
    mp.start_processes(_range_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    ta = oracle.generate_reads(1001, 900, 100); tb = oracle.generate_reads(1002, 700, 100)
    a = oracle.FMI.from_text(ta); b = oracle.FMI.from_text(tb)
    ranks, counts, _ = oracle.search(a, b, threads=1)
    ra = oracle.ra_from_runs(ranks, counts)
    n_out = a.bases + b.bases
    expect = np.zeros(((n_out + 63) // 64) * 64, dtype=np.uint8)
    expect[np.arange(b.bases, dtype=np.uint64) + ra] = 1
    nonempty = 0
    for r in range(world):
        z = np.load(tmp_path / ("range_%d.npz" % r))
        p0, p1 = int(z["rec_first"]) * 128, min(int(z["rec_last"]) * 128, expect.size)
        own_bits = np.unpackbits(z["own"].view(np.uint8), bitorder="little")
        assert np.array_equal(own_bits[: p1 - p0], expect[p0: p1]), r                         # the range holds the union of all shards
        assert int(z["before"]) == int(expect[: p0].sum()) and int(z["total"]) == b.bases, r   # set bits before the range / of all ranges
        assert z["super_boff"].size == 1 and int(z["super_boff"][0]) == 0
        if r > 0 and p1 > p0:
            halo_bits = np.unpackbits(z["halo"].view(np.uint8), bitorder="little")
            assert np.array_equal(halo_bits, expect[p0 - 8192: p0]), r                          # the chunk of 64 records before the range
        nonempty += (p1 > p0)
    assert nonempty >= 2                                                                        # the input is large enough for two real ranges


def test_super_owners_and_equal_bounds(bwtm):
    """Pure host arithmetic: equal output ranges (what a reduce-scatter wants) and the owner of every super block of the output."""
    from bwt_merge_amd.dist import super_owners
    for nrecs, world in ((1, 2), (511, 3), (513, 2), (1 << 20, 8), (789012345, 8), (5 * (1 << 18) + 7, 4)):
        bounds = [bwtm.slice_bounds_equal(nrecs, world, g) for g in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == nrecs
        assert all(bounds[g][1] == bounds[g + 1][0] for g in range(world - 1))
        assert all(b[0] == b[1] or (b[0] % 512 == 0 and (b[1] % 512 == 0 or b[1] == nrecs)) for b in bounds)
        assert len({b[2] for b in bounds}) == 1 and world * bounds[0][2] >= -(-nrecs // 64) * 1024    # equal shares cover the bitvector
        sizes = [b[1] - b[0] for b in bounds]
        full = bounds[0][2] // 16                                           # records per full range
        last_nonempty = max(g for g in range(world) if sizes[g] > 0)
        assert all(sizes[g] == full for g in range(last_nonempty)) and 0 < sizes[last_nonempty] <= full
        n_out = (nrecs - 1) << 7
        nsup = (n_out >> 25) + 1
        owner = super_owners(nsup, [(b[0], b[1]) for b in bounds])
        for s in range(nsup):
            q = s << 18
            assert bounds[owner[s]][0] <= q < bounds[owner[s]][1], (nrecs, world, s)


# ---------------------------------------------------------------------------------------------------------
# The encoder's carries across output slices (bwt-merge_amd/dist.py: exchange_encoder_carries), with a plain CPU
# statement of what a slice computes: its run heads, the bytes Run::write emits for the runs that END in it.

def _slice_events(sym, p0, p1, heads_before):
    """Runs (symbol, length) that end at a head inside [p0, p1) (the virtual head at len(sym) belongs to the last slice),
    and the slice's (last head) + 1."""
    n = sym.size
    heads = [h for h in range(p0, min(p1, n + 1)) if h == 0 or h == n or sym[h] != sym[h - 1]]
    events, prev1 = [], heads_before
    for h in heads:
        if h > 0:
            events.append((int(sym[h - 1]), h + 1 - prev1 if prev1 > 0 else h))
        prev1 = h + 1
    return events, (heads[-1] + 1 if heads else 0)


def _slice_table(orc, events):
    table = np.zeros(64, dtype=np.uint64)
    for o in range(64):
        off = o
        for comp, length in events:
            off += len(orc.run_write(comp, length, prefill=off % 64))
        table[o] = off - o
    return table


def _slice_symbols():
    rng = np.random.default_rng(99)
    syms = rng.integers(0, 6, 400)
    lens = rng.choice([1, 1, 2, 3, 41, 42, 43, 170, 3000], 400)
    return np.repeat(syms.astype(np.uint8), lens)


def _carry_worker(rank, world, port, result_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import _pkg
    _pkg.load()
    from bwt_merge_amd.dist import exchange_encoder_carries
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sym = _slice_symbols()
    cut = [0, 37 * 128, sym.size + 1][rank: rank + 2]                 # two slices of whole records, the cut inside a long run or not
    _, lasthead = _slice_events(sym, cut[0], cut[1], 0)
    def table_fn(heads_before):
        return _slice_table(orc, _slice_events(sym, cut[0], cut[1], heads_before)[0])
    heads_before, offset, total = exchange_encoder_carries(lasthead, table_fn, rank, world, dist, torch, "cpu")
    # this slice's bytes, written at its offset
    data = bytearray()
    off = offset
    for comp, length in _slice_events(sym, cut[0], cut[1], heads_before)[0]:
        piece = orc.run_write(comp, length, prefill=off % 64)
        data += piece; off += len(piece)
    np.save(os.path.join(result_dir, "slice_%d.npy" % rank), np.frombuffer(bytes(data), dtype=np.uint8))
    np.save(os.path.join(result_dir, "meta_%d.npy" % rank), np.array([offset, total], dtype=np.uint64))
    dist.destroy_process_group()


def test_two_rank_encoder_carries_concatenate_to_the_whole_stream(tmp_path, oracle):
    world = 2
    mp.start_processes(_carry_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    whole = oracle.FMI.from_symbols(_slice_symbols()).data
    parts = [np.load(tmp_path / ("slice_%d.npy" % r)) for r in range(world)]
    metas = [np.load(tmp_path / ("meta_%d.npy" % r)) for r in range(world)]
    assert int(metas[0][0]) == 0 and int(metas[1][0]) == parts[0].size
    assert all(int(m[1]) == whole.size for m in metas)
    assert np.array_equal(np.concatenate(parts), whole)


def test_fold_offsets_binding_matches_host_logic(bwtm):
    from bwt_merge_amd.dist import fold_offsets
    rng = np.random.default_rng(3)
    tables = rng.integers(0, 1000, (5, 64)).astype(np.uint64)
    assert [int(x) for x in bwtm.fold_offsets(tables)] == fold_offsets(tables)
    assert bwtm.slice_bounds(5000, 3, 1) == (1536, 3072) and bwtm.slice_bounds(300, 4, 0) == (0, 0) and bwtm.slice_bounds(300, 4, 3) == (0, 300)


# ---------------------------------------------------------------------------------------------------------
# The sliced frontier search (bwt-merge_amd/experimental.py: search_sliced, include/bwtm_experimental.h: bwtm_fslice_*) as a plain CPU model: every part
# advances a contiguous slice of the sorted frontier and the next frontier is read in the order (class, part, position inside the
# part's output).  The claim the GPU code rests on: that order IS the suffix order, so the union of the parts' emits is the rank array.

def test_sliced_frontier_model_on_cpu(oracle, bwtm):
    bwtm.build(experimental=True)                                # bwt_merge_amd.experimental imports only next to libbwtm_experimental.so
    from bwt_merge_amd.dist import shard_range
    from bwt_merge_amd.experimental import slice_range
    ta = oracle.generate_reads(1001, 50, 30)
    tb = np.concatenate([oracle.generate_reads(1002 + k, 12, n) for k, n in enumerate([1, 7, 19, 40, 33])])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    ranks, counts, _ = oracle.search(a, b, threads=1)
    expect = oracle.ra_from_runs(ranks, counts)
    for parts in (1, 2, 3, 7):
        # outputs of "step -1": every part's block of sequences, in class 0
        outputs = [[[] for _ in range(6)] for _ in range(parts)]
        for g in range(parts):
            first, last = shard_range(b.sequences, g, parts)
            outputs[g][0] = [(j, a.sequences) for j in range(first, last + 1)]
        got = np.zeros(b.bases, dtype=np.uint64)
        seen = np.zeros(b.bases, dtype=bool)
        while True:
            frontier = [e for c in range(6) for g in range(parts) for e in outputs[g][c]]        # logical order: (class, part, inside)
            if not frontier:
                break
            assert all(frontier[k][0] < frontier[k + 1][0] and frontier[k][1] <= frontier[k + 1][1] for k in range(len(frontier) - 1))   # sorted by suffix
            outputs = [[[] for _ in range(6)] for _ in range(parts)]
            for g in range(parts):
                lo, hi = slice_range(len(frontier), g, parts)
                for i, r in frontier[lo:hi]:                                                  # one LF step on the slice, stable split by class
                    assert not seen[i]
                    got[i] = r; seen[i] = True
                    nxt, c = b.LF(i)
                    if c != 0:
                        outputs[g][c].append((nxt, a.LF(r, c)))
        assert seen.all() and np.array_equal(got, expect), parts


# ---------------------------------------------------------------------------------------------------------
# Sharded upload (bwt-merge_amd/dist.py: gather_native_bytes): every rank contributes its part of the native bytes, all end up with the whole.

def _gather_worker(rank, world, port, result_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import _pkg
    _pkg.load()
    from bwt_merge_amd.dist import gather_native_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for n in (0, 1, 255, 256, 257, 100003):
        data = (np.arange(n, dtype=np.uint64) * 2654435761 % 251).astype(np.uint8)
        mine = data.copy()
        chunk = ((n + world - 1) // world + 255) // 256 * 256
        own = slice(min(rank * chunk, n), min((rank + 1) * chunk, n))
        mask = np.ones(n, dtype=bool); mask[own] = False
        mine[mask] = 0xEE
        full, took = gather_native_bytes(mine, rank, world, dist, torch, "cpu")
        got = full.numpy()
        out.append(bool(np.array_equal(got[:n], data) and not got[n:].any() and took == own.stop - own.start and got.size >= n + 16))
    np.save(os.path.join(result_dir, "gather_%d.npy" % rank), np.array(out))
    dist.destroy_process_group()


def test_two_rank_sharded_upload_gathers_the_whole_stream(tmp_path):
    world = 2
    mp.start_processes(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    for r in range(world):
        assert np.load(tmp_path / ("gather_%d.npy" % r)).all()
