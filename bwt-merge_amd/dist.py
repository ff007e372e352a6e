"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL).

The path shards over the sequences of the second input (every LF chain is independent; the
reference hands out sequence blocks to threads the same way, fmi.cpp:355-357).  Every rank
searches its contiguous block, then ONE bulk exchange combines the rank-array shards: each rank
has set a disjoint subset of the bits of the interleaving bitvector, so an all-reduce with
SUM over 64-bit words equals the bitwise OR (no carries).

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
    """All-reduce of the rank-array bitvector shards (an int64 tensor viewed as 64-bit words)."""
    if dist is None:
        import torch.distributed as dist
    dist.all_reduce(words, op=dist.ReduceOp.SUM)
    return words


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


def slice_range(total, part, parts):
    """Contiguous share `part` of `total` frontier elements: [first, last)."""
    per = (total + parts - 1) // parts
    return min(total, part * per), min(total, (part + 1) * per)


def search_sliced(pkg, indexes, ras, sequences, enter=None):
    """The sliced frontier search driven from ONE host thread over `parts` GPUs (or contexts of one GPU): indexes[g] = (A, B) as
    GPU g holds them, ras[g] = its rank array; enter(g) makes GPU g's context current for the calling thread (None: one context).
    Every GPU ends up with the bits of the elements it advanced; combine the rank arrays as after bwtm_search().  Returns the
    number of LF steps.  (A host thread or process per GPU would run the same loop with barriers where this one switches GPUs.)"""
    parts = len(indexes)
    cap = (sequences + parts - 1) // parts + 1
    views = (pkg.FSliceView * parts)()
    fs = []
    for g in range(parts):
        if enter:
            enter(g)
        f = pkg.FSlice(indexes[g][0], indexes[g][1], ras[g], cap, parts)
        first, last = shard_range(sequences, g, parts)
        f.seed(first, (last - first + 1) if first <= last else 0)
        f.export(views[g])
        fs.append(f)
    steps = 0
    while True:
        total = sum(int(views[h].totals[c]) for h in range(parts) for c in range(5))
        if total == 0:
            break
        for g in range(parts):                              # every GPU pulls its slice of the frontier from all GPUs' outputs ...
            if enter:
                enter(g)
            first, last = slice_range(total, g, parts)
            fs[g].gather(views, parts, first, last)
        for g in range(parts):                              # ... and only then overwrites its own outputs
            if enter:
                enter(g)
            fs[g].advance()
            fs[g].export(views[g])
        steps += 1
    for g in range(parts):
        if enter:
            enter(g)
        fs[g].finish()
        fs[g].free()
    return steps


def merge_sharded(pkg, A, B, rank, world, dist, torch, device):
    """FMI::FMI(a, b) on `world` GPUs, as this rank sees it: search of its block of b's sequences, all-reduce of the
    rank-array bitvector (RCCL over xGMI), then interleave + encode of its own slice of the output.  Returns the
    encoded pkg.Slice (total_nbytes = size of the whole merged stream)."""
    nbytes = pkg.ra_buffer_bytes(A, B)
    buf = torch.zeros(nbytes // 8, dtype=torch.int64, device=device)
    torch.cuda.synchronize()
    ra = pkg.RankArray(A, B, buf.data_ptr(), nbytes)
    first, last = shard_range(B.sequences, rank, world)
    if first <= last:
        ra.search(A, B, first, last)
    pkg.synchronize()
    if dist is not None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)              # disjoint bits: sum == or
        torch.cuda.synchronize()
    ra.finalize()
    rec_first, rec_last = pkg.slice_bounds(pkg.merged_records(A, B), world, rank)
    S = pkg.Slice(A, B, ra, rec_first, rec_last)
    _, offset, total = exchange_encoder_carries(S.lasthead(), S.size_table, rank, world, dist, torch, device)
    S.encode(offset)
    S.total_nbytes = total
    pkg.synchronize()
    ra.free()
    del buf
    return S
