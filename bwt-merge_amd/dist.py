"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL).

The path shards over the sequences of the second input (every LF chain is independent; the
reference hands out sequence blocks to threads the same way, fmi.cpp:355-357).  Every rank
searches its contiguous block, then ONE bulk exchange combines the rank-array shards: each rank
has set a disjoint subset of the bits of the interleaving bitvector, so SUM over 64-bit words
equals the bitwise OR (no carries).  Every rank consumes only its own range of the OUTPUT afterwards,
so the exchange is a REDUCE-SCATTER by output range (exchange_bitvector_ranges): half the bytes of
the all-reduce of the first versions, and no rank counts bits it never reads; what the ranks need
from each other beyond their own range -- the number of set bits before it, the offsets of the
output's super blocks, the one chunk of bits before it -- is a few KB in one small all-reduce.

After the exchange every rank interleaves and encodes only its own range of the OUTPUT (mergeBWT
cut by output position, bwt.cpp:215-282).  Two facts cross a slice boundary and travel as two tiny
all-gathers (exchange_encoder_carries): the run that is still open at the boundary, and the byte
offset modulo 64 that Run::write's block rule depends on (support.h:256-282).

(The C++ host in csrc/host/multi_gpu.h does the same with one thread per GPU and librccl directly.)
"""
import numpy as np


def get_bounds(first, last, blocks):
    """Split the closed range [first, last] into at most `blocks` near-equal closed ranges
    (same arithmetic as getBounds, utils.cpp:169-187)."""
    if first + 1 > last + 1:
        return []
    length = last + 1 - first
    blocks = max(1, min(blocks, length))
    bounds = []
    start = first
    for block in range(blocks):
        lo = start
        if start <= last:
            start += max(1, (last + 1 - start) // (blocks - block))
        bounds.append((lo, start - 1))
    return bounds


def shard_range(sequences, rank, world):
    """Closed range of sequence ids of input2 searched by `rank`; (1, 0) = empty."""
    if sequences == 0:
        return (1, 0)
    bounds = get_bounds(0, sequences - 1, world)
    return bounds[rank] if rank < len(bounds) else (1, 0)


def exchange_bitvector(words, dist=None):
    """All-reduce of the rank-array bitvector shards (an int64 tensor viewed as 64-bit words): the form in which EVERY rank ends up
    with the whole bitvector (kept for callers that want that; the merge uses exchange_bitvector_ranges)."""
    if dist is None:
        import torch.distributed as dist
    dist.all_reduce(words, op=dist.ReduceOp.SUM)
    return words


CHUNK_WORDS = 128                  # 64-bit words of the bitvector per chunk of 64 output records (kernels/interleave.hip.h)
SUPER_SHIFT = 25                   # positions per super block of the rank structure (bwtm_device.h)


def exchange_bitvector_ranges(words, shard_words, rank, world, dist, torch):
    """Reduce-scatter of the bitvector by EQUAL output ranges: `words` (int64, world * shard_words entries, this rank's bits set, zero
    behind the bitvector) -> afterwards words[rank * shard_words : (rank + 1) * shard_words] holds the sum (= or) over all ranks and
    the rest of `words` is unspecified.  Every rank sends and receives (world - 1) / world of ONE bitvector: half of an all-reduce."""
    assert words.numel() == world * shard_words
    if dist is None:                                   # no process group at all; with one the collective runs on a single rank too (--force-dist)
        return words
    mine = torch.empty(shard_words, dtype=words.dtype, device=words.device)
    dist.reduce_scatter_tensor(mine, words, op=dist.ReduceOp.SUM)
    words[rank * shard_words: (rank + 1) * shard_words].copy_(mine)
    return words


def super_owners(nsup, bounds):
    """owner[s] = the rank whose range of output records holds the start of super block s (record s << 18); the ranges are consecutive
    and cover all records, and every super starts at a record below the total."""
    starts = np.arange(nsup, dtype=np.uint64) << np.uint64(SUPER_SHIFT - 7)
    owner = np.zeros(nsup, dtype=np.int64)
    for g, (first, last) in enumerate(bounds):
        owner[(starts >= np.uint64(first)) & (starts < np.uint64(last))] = g
    return owner


def combine_range_counts(ones, super_local, tail_words, bounds, rank, world, dist, torch, device):
    """The small exchange behind the reduce-scatter.  ones / super_local / tail_words: what bwtm_ra_range_counts returned for this rank's
    range; bounds[g] = (rec_first, rec_last) of every rank.  One all-reduce (every entry is written by exactly one rank) of
    [world range totals | nsup local super offsets | world x 128 tail words] -> (ones_before, ones_total, super_boff[nsup], halo_words or
    None) for bwtm_ra_finalize_range."""
    nsup = int(super_local.size)
    vec = np.zeros(world + nsup + world * CHUNK_WORDS, dtype=np.uint64)
    vec[rank] = ones
    vec[world: world + nsup] = super_local
    vec[world + nsup + rank * CHUNK_WORDS: world + nsup + (rank + 1) * CHUNK_WORDS] = tail_words
    if dist is not None and world > 1:
        t = torch.from_numpy(vec.view(np.int64).copy()).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        vec = t.cpu().numpy().view(np.uint64)
    totals = vec[:world]
    prefix = np.concatenate([[0], np.cumsum(totals)]).astype(np.uint64)
    super_boff = prefix[super_owners(nsup, bounds)] + vec[world: world + nsup]
    halo = None
    for g in range(rank - 1, -1, -1):                        # the nearest earlier range that is not empty
        if bounds[g][1] > bounds[g][0]:
            halo = vec[world + nsup + g * CHUNK_WORDS: world + nsup + (g + 1) * CHUNK_WORDS].copy()
            break
    return int(prefix[rank]), int(prefix[world]), super_boff, halo


def fold_offsets(tables):
    """offsets[g] = byte offset at which slice g starts, offsets[parts] = size of the stream (bwtm_fold_offsets):
    tables[g][o] = bytes slice g emits when the stream is at offset o (mod 64) where it starts."""
    off = 0
    out = []
    for row in tables:
        out.append(off)
        off += int(row[off & 63])
    out.append(off)
    return out


def all_gather_u64(values, rank, world, dist, torch, device):
    """All-gather of a small vector of unsigned 64-bit integers per rank -> [world, len(values)] numpy uint64."""
    mine = np.asarray(values, dtype=np.uint64).reshape(-1)
    if dist is None or world == 1:
        return mine.reshape(1, -1)
    t = torch.from_numpy(mine.view(np.int64).copy()).to(device)
    out = torch.empty(world * mine.size, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().view(np.uint64).reshape(world, mine.size)


def exchange_encoder_carries(lasthead, size_table_fn, rank, world, dist, torch, device):
    """The encoder's two carries across output slices.
    lasthead: this slice's (position of the last run head) + 1, 0 if none; size_table_fn(heads_before) -> 64 sizes.
    Returns (heads_before, byte offset of this slice, size of the whole stream)."""
    heads = all_gather_u64([lasthead], rank, world, dist, torch, device)[:, 0]
    heads_before = int(heads[:rank].max()) if rank > 0 else 0
    table = size_table_fn(heads_before)
    tables = all_gather_u64(table, rank, world, dist, torch, device)
    offsets = fold_offsets(tables)
    return heads_before, offsets[rank], offsets[world]


def gather_native_bytes(host_array, rank, world, dist, torch, device):
    """Every rank copies its 1 / world of the page-locked native bytes to its device and the parts are all-gathered in place (RCCL over
    xGMI; gloo in the CPU tests): returns (the full-size buffer: the whole stream followed by zeros, bytes this rank took from the host)."""
    nbytes = int(host_array.size)
    chunk = ((nbytes + world - 1) // world + 255) // 256 * 256
    full = torch.empty(chunk * world + 16, dtype=torch.uint8, device=device)
    off = min(rank * chunk, nbytes)
    length = min(chunk, nbytes - off)
    if length > 0:
        full[off: off + length].copy_(torch.from_numpy(host_array[off: off + length]), non_blocking=True)
    if dist is not None and world > 1:
        dist.all_gather_into_tensor(full[: chunk * world], full[rank * chunk: (rank + 1) * chunk].clone() if full.device.type == "cpu" else full[rank * chunk: (rank + 1) * chunk])
    full[nbytes:].zero_()                                    # the borrowed form wants readable zeros behind the stream
    return full, length


def upload_sharded(pkg, host_array, sequences, bases, rank, world, dist, torch, device):
    """BWT::load of one input on `world` GPUs with every native byte crossing PCIe ONCE (gather_native_bytes), then every rank decodes /
    transcodes its complete device copy.  Returns (index, bytes this rank received from the host)."""
    full, length = gather_native_bytes(host_array, rank, world, dist, torch, device)
    torch.cuda.synchronize()
    ix = pkg.Index.from_device(full.data_ptr(), int(host_array.size), sequences, bases, borrow=True)
    ix.drop_native()                                         # synchronizes: `full` may go
    del full
    return ix, length


_bitvector_cache = {}


def bitvector_buffer(words, torch, device):
    """The caller-owned bitvector of a sharded merge, zeroed: ONE tensor per device and size, kept across merges (a fresh torch.zeros per
    merge went through the caching allocator's search and, the first time, a device allocation of up to 12.6 GB inside the timed merge)."""
    key = (str(device), int(words))
    buf = _bitvector_cache.get(key)
    if buf is None:
        _bitvector_cache.clear()                       # one size at a time: a chain's merges grow
        buf = torch.empty(int(words), dtype=torch.int64, device=device)
        _bitvector_cache[key] = buf
    buf.zero_()
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()                       # the library's streams do not order themselves behind torch's
    return buf


def release_buffers():
    """Drops the cached bitvector (bench.py calls it before measurements that need the memory)."""
    _bitvector_cache.clear()


def merge_sharded(pkg, A, B, rank, world, dist, torch, device, times=None):
    """FMI::FMI(a, b) on `world` GPUs, as this rank sees it: search of its block of b's sequences, reduce-scatter of the rank-array
    bitvector by output range (RCCL over xGMI) + one small all-reduce, then interleave + encode of its own range of the output.
    Returns the encoded pkg.Slice (total_nbytes = size of the whole merged stream).  times (a dict, optional) receives this rank's phases
    in milliseconds (search / exchange = reduce-scatter + the small all-reduce / interleave_encode incl. the carries) and the bytes of
    bitvector this rank sends and receives in the reduce-scatter."""
    import time as _time
    t0 = _time.perf_counter()
    nrecs = pkg.merged_records(A, B)
    bounds = [pkg.slice_bounds_equal(nrecs, world, g) for g in range(world)]
    rec_first, rec_last, shard_bytes = bounds[rank]
    assert world * shard_bytes >= pkg.ra_buffer_bytes(A, B)            # equal shares for the collective: zero words behind the bitvector
    buf = bitvector_buffer(world * shard_bytes // 8, torch, device)
    ra = pkg.RankArray(A, B, buf.data_ptr(), buf.numel() * 8)
    first, last = shard_range(B.sequences, rank, world)
    if first <= last:
        ra.search(A, B, first, last)
    pkg.synchronize()
    t1 = _time.perf_counter()
    exchange_bitvector_ranges(buf, shard_bytes // 8, rank, world, dist, torch)
    torch.cuda.synchronize()
    ones, local, tail = ra.range_counts(rec_first, rec_last)
    before, total, super_boff, halo = combine_range_counts(ones, local, tail, [(b[0], b[1]) for b in bounds], rank, world, dist, torch, device)
    t2 = _time.perf_counter()
    ra.finalize_range(rec_first, rec_last, before, total, super_boff, halo)
    S = pkg.Slice(A, B, ra, rec_first, rec_last)
    _, offset, total_bytes = exchange_encoder_carries(S.lasthead(), S.size_table, rank, world, dist, torch, device)
    S.encode(offset)
    S.total_nbytes = total_bytes
    pkg.synchronize()
    ra.free()
    if times is not None:
        t3 = _time.perf_counter()
        for k, v in (("ms_search", t1 - t0), ("ms_exchange", t2 - t1), ("ms_interleave_encode", t3 - t2)):
            times[k] = times.get(k, 0.0) + v * 1e3
        times["merges"] = times.get("merges", 0) + 1
        times["exchange_bytes_per_gpu"] = (world - 1) * shard_bytes      # sent and received by every rank: (N - 1) / N of one bitvector
    return S
