"""Multi-GPU host logic: one process per GPU (torch.distributed, backend "nccl" = RCCL).

The path shards over the sequences of the second input (every LF chain is independent; the
reference hands out sequence blocks to threads the same way, fmi.cpp:355-357).  Every rank
searches its contiguous block, then ONE exchange combines the rank-array shards: each rank
has set a disjoint subset of the bits of the interleaving bitvector, so an all-reduce with
SUM over 64-bit words equals the bitwise OR (no carries).
"""


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
