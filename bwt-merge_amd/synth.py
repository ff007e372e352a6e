"""Synthetic inputs at scale: the reference cannot build a BWT from
reads (it relies on RopeBWT / SGA, README.md:5,20), so benchmark inputs are made here:

  generate_reads   counter-based splitmix64 reads, identical to oracle/bwtm_oracle.cpp
  leaf_bwt         multi-string BWT of a batch of reads by LSD radix sort of the suffixes
  build_index      leaves grown into one index by a merge tree that uses the GPU merger itself

This is input tooling, not the hot path: it uses PyTorch tensor ops (device agnostic, so the
CPU test-suite checks it against the oracle) and hands device pointers to the C ABI.
"""
import numpy as np
import torch

_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB


def _s64(x):
    """Python int -> two's complement int64 value."""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _mix(z):
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


def read_length(j, mixed):
    """Length of read j: 100, or the 100/150 mix of BASELINE config 5 (3 x 100 bp per 2 x 150 bp,
    so each length contributes half of the bases)."""
    if not mixed:
        return 100
    return 100 if (j % 5) < 3 else 150


def generate_reads(seed, first_read, nreads, readlen, device="cpu"):
    """[nreads, readlen] uint8 of comp values 1..5: base t of read j is N (5) when
    z1 % 256 == 0 and 1 + z2 % 4 otherwise, with z1, z2 = splitmix64 outputs at counters
    2 (256 j + t) + 1 and + 2 (same stream as orc_generate_reads)."""
    j = torch.arange(first_read, first_read + nreads, dtype=torch.int64, device=device).unsqueeze(1)
    t = torch.arange(readlen, dtype=torch.int64, device=device).unsqueeze(0)
    idx = j * 256 + t
    z1 = _mix(idx * _s64(2 * _GAMMA) + _s64(seed + _GAMMA))
    z2 = _mix(idx * _s64(2 * _GAMMA) + _s64(seed + 2 * _GAMMA))
    base = 1 + (z2 & 3)
    base = torch.where((z1 & 255) == 0, torch.full_like(base, 5), base)
    return base.to(torch.uint8)


GENOME_SEED = 7777


def generate_genome_reads(seed, first_read, nreads, readlen, genome_len, device="cpu", error_percent=1):
    """Secondary distribution of SURVEY.md 8(d): reads sampled uniformly from a random ACGT genome (the same
    genome for every set, base k = 1 + splitmix64(k) % 4) with `error_percent` % substitutions.  Counter-based
    like generate_reads, so any read can be regenerated on its own.  [nreads, readlen] uint8, comp values 1..4."""
    j = torch.arange(first_read, first_read + nreads, dtype=torch.int64, device=device).unsqueeze(1)
    t = torch.arange(readlen, dtype=torch.int64, device=device).unsqueeze(0)
    span = max(1, genome_len - readlen + 1)
    start = _lsr(_mix(j * _s64(_GAMMA) + _s64(seed * 3 + 1)), 1) % span
    pos = start + t
    g = _mix(pos * _s64(_GAMMA) + _s64(GENOME_SEED)) & 3
    z = _mix((j * 256 + t) * _s64(2 * _GAMMA) + _s64(seed + 5 * _GAMMA))
    is_sub = (_lsr(z, 8) % 100) < error_percent
    other = (g + 1 + ((z & 0xFF) % 3)) & 3                                  # one of the three other bases
    return (1 + torch.where(is_sub, other, g)).to(torch.uint8)


def make_reads(workload, seed, first_read, nreads, readlen, total_reads, device="cpu", coverage=30, error_percent=1):
    """Reads first_read .. first_read + nreads - 1 of set `seed` for a named workload: "iid" (the headline
    distribution) or "genome" (30x coverage of a shared random genome, 1 % substitutions)."""
    if workload == "iid":
        return generate_reads(seed, first_read, nreads, readlen, device=device)
    if workload == "genome":
        return generate_genome_reads(seed, first_read, nreads, readlen, max(readlen, total_reads * readlen // coverage), device=device,
                                     error_percent=error_percent)
    raise ValueError("unknown workload %r" % workload)


def leaf_bwt(reads):
    """BWT (one comp value per byte, endmarkers 0) of the collection reads[0], reads[1], ...
    Suffixes are compared symbol by symbol with the endmarker smallest; equal suffixes (both
    ended) are ordered by sequence index -- the order bwt_merge produces."""
    m, L = reads.shape
    dev = reads.device
    W = 21                                   # 3-bit symbols per 63-bit key word
    nw = max(1, -(-L // W))
    padded = torch.zeros((m, L + 1 + nw * W), dtype=torch.uint8, device=dev)
    padded[:, :L] = reads
    perm = torch.arange(m * (L + 1), dtype=torch.int64, device=dev)
    for w in reversed(range(nw)):
        key = torch.zeros((m, L + 1), dtype=torch.int64, device=dev)
        for t in range(W):
            key = key * 8 + padded[:, w * W + t: w * W + t + L + 1]
        key = key.reshape(-1)[perm]
        order = torch.sort(key, stable=True).indices
        perm = perm[order]
        del key, order
    s = torch.div(perm, L + 1, rounding_mode="floor")
    o = perm - s * (L + 1)
    prev = reads.reshape(-1)[(s * L + o - 1).clamp_(min=0)]
    return torch.where(o > 0, prev, torch.zeros_like(prev))


def leaf_bwt_ragged(reads_list):
    """leaf_bwt for reads of different lengths: list of [m_k, L_k] tensors in collection order
    is not needed by the benchmark's fixed patterns; mixed sets are built from uniform leaves
    of each length merged in sequence order (see build_index)."""
    raise NotImplementedError


def merge_indexes(pkg, a, b, free_inputs=True):
    """Device-resident merge without the native encode (what a merge tree needs)."""
    ra = pkg.RankArray(a, b)
    if b.sequences > 0:
        ra.search(a, b, 0, b.sequences - 1)
    ra.finalize()
    out = pkg.interleave(a, b, ra)
    pkg.synchronize()
    ra.free()
    if free_inputs:
        a.free(); b.free()
    return out


def build_index(pkg, seed, nreads, readlen=100, leaf_reads=1 << 19, device="cuda", progress=None, workload="iid", **workload_args):
    """Index of the synthetic set `seed` (reads 0 .. nreads-1 in generation order)."""
    stack = []                               # (level, index); adjacent entries are adjacent read ranges
    for first in range(0, nreads, leaf_reads):
        count = min(leaf_reads, nreads - first)
        reads = make_reads(workload, seed, first, count, readlen, nreads, device=device, **workload_args)
        sym = leaf_bwt(reads).contiguous()
        if sym.is_cuda:
            torch.cuda.synchronize()
        leaf = pkg.Index.from_symbols_device(sym.data_ptr(), sym.numel())
        del sym, reads
        stack.append((0, leaf))
        while len(stack) >= 2 and stack[-1][0] == stack[-2][0]:
            lb, b = stack.pop()
            la, a = stack.pop()
            stack.append((la + 1, merge_indexes(pkg, a, b)))
        if progress:
            progress(first + count, nreads)
    while len(stack) >= 2:
        lb, b = stack.pop()
        la, a = stack.pop()
        stack.append((max(la, lb) + 1, merge_indexes(pkg, a, b)))
    return stack[0][1]


def extract_sequences(pkg, index, seq_ids, max_len=256):
    """Walks LF from the endmarker positions seq_ids and returns the sequences (lists of comp
    values, in forward order) -- a size-independent end-to-end check of a merged index."""
    C = index.C
    pos = np.asarray(seq_ids, dtype=np.uint64).copy()
    alive = np.ones(pos.size, dtype=bool)
    out = [[] for _ in range(pos.size)]
    for _ in range(max_len + 1):
        if not alive.any():
            break
        r, c = index.inverse_select(pos)
        for k in np.nonzero(alive)[0]:
            if c[k] == 0:
                alive[k] = False
            else:
                out[k].append(int(c[k]))
        pos = np.where(alive, C[c.astype(np.int64)] + r, pos).astype(np.uint64)
    return [list(reversed(x)) for x in out]
