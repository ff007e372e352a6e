"""Synthetic inputs at scale: the reference cannot build a BWT from
reads (it relies on RopeBWT / SGA, README.md:5,20), so benchmark inputs are made here:

  generate_reads   counter-based splitmix64 reads, identical to oracle/bwtm_oracle.cpp
  leaf_bwt         multi-string BWT of a batch of reads by LSD radix sort of the suffixes (tensor ops; the CPU suite
                   pins it to the oracle, the GPU suite pins the library's builder to it and to the oracle)
  build_index      reads -> index: on a GPU through the library's builder (bwtm_builder_*: suffix-sorted leaves grown
                   by a merge tree that uses the merger itself); native=False drives the same tree from here

The generators are input tooling, not the hot path: PyTorch tensor ops (device agnostic, so the
CPU test-suite checks them against the oracle) that hand device pointers to the C ABI.
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


def _read_ids(first_read, nreads, device):
    """Read indexes as an int64 column: a contiguous range, or an explicit tensor / array of indexes."""
    if nreads is None:
        return torch.as_tensor(np.asarray(first_read, dtype=np.int64), device=device).reshape(-1, 1)
    return torch.arange(first_read, first_read + nreads, dtype=torch.int64, device=device).unsqueeze(1)


def generate_reads(seed, first_read, nreads, readlen, device="cpu"):
    """[nreads, readlen] uint8 of comp values 1..5: base t of read j is N (5) when
    z1 % 256 == 0 and 1 + z2 % 4 otherwise, with z1, z2 = splitmix64 outputs at counters
    2 (256 j + t) + 1 and + 2 (same stream as orc_generate_reads).  nreads=None: first_read is a list of read indexes."""
    j = _read_ids(first_read, nreads, device)
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
    j = _read_ids(first_read, nreads, device)
    t = torch.arange(readlen, dtype=torch.int64, device=device).unsqueeze(0)
    span = max(1, genome_len - readlen + 1)
    start = _lsr(_mix(j * _s64(_GAMMA) + _s64(seed * 3 + 1)), 1) % span
    pos = start + t
    g = _mix(pos * _s64(_GAMMA) + _s64(GENOME_SEED)) & 3
    z = _mix((j * 256 + t) * _s64(2 * _GAMMA) + _s64(seed + 5 * _GAMMA))
    is_sub = (_lsr(z, 8) % 100) < error_percent
    other = (g + 1 + ((z & 0xFF) % 3)) & 3                                  # one of the three other bases
    return (1 + torch.where(is_sub, other, g)).to(torch.uint8)


MIXED_LONG = 150


def make_reads(workload, seed, first_read, nreads, readlen, total_reads, device="cpu", coverage=30, error_percent=1):
    """Reads first_read .. first_read + nreads - 1 of set `seed` for a named workload: "iid" (the headline
    distribution), "genome" (30x coverage of a shared random genome, 1 % substitutions) or "mixed" (BASELINE config 5:
    iid reads, read j is 100 bp when j % 5 < 3 and 150 bp otherwise; returned 150 wide with zeros after the end of the
    short reads -- see read_lengths)."""
    if workload == "iid":
        return generate_reads(seed, first_read, nreads, readlen, device=device)
    if workload == "mixed":
        reads = generate_reads(seed, first_read, nreads, MIXED_LONG, device=device)
        lengths = read_lengths(workload, first_read, nreads, readlen, device)
        t = torch.arange(MIXED_LONG, dtype=torch.int64, device=device).unsqueeze(0)
        return torch.where(t < lengths.unsqueeze(1), reads, torch.zeros_like(reads))
    if workload == "genome":
        return generate_genome_reads(seed, first_read, nreads, readlen, max(readlen, total_reads * readlen // coverage), device=device,
                                     error_percent=error_percent)
    raise ValueError("unknown workload %r" % workload)


def read_lengths(workload, first_read, nreads, readlen, device="cpu"):
    """Lengths of the reads make_reads returns (int64 tensor), or None when they all have `readlen` symbols."""
    if workload != "mixed":
        return None
    j = _read_ids(first_read, nreads, device).reshape(-1)
    return torch.where((j % 5) < 3, torch.full_like(j, 100), torch.full_like(j, MIXED_LONG))


def leaf_bwt(reads, lengths=None):
    """BWT (one comp value per byte, endmarkers 0) of the collection reads[0], reads[1], ...
    Suffixes are compared symbol by symbol with the endmarker smallest; equal suffixes (both
    ended) are ordered by sequence index -- the order bwt_merge produces.
    lengths (optional int64 [m]): read k has lengths[k] symbols and reads[k, lengths[k]:] is zero -- the ragged
    collections PlainData::read ingests (formats.cpp:133-161)."""
    m, L = reads.shape
    dev = reads.device
    W = 21                                   # 3-bit symbols per 63-bit key word
    nw = max(1, -(-L // W))
    padded = torch.zeros((m, L + 1 + nw * W), dtype=torch.uint8, device=dev)
    padded[:, :L] = reads
    if lengths is None:
        perm = torch.arange(m * (L + 1), dtype=torch.int64, device=dev)
    else:
        # only the suffixes that exist: offsets 0 .. lengths[k] of read k (the last one is its endmarker)
        o = torch.arange(L + 1, dtype=torch.int64, device=dev).unsqueeze(0)
        perm = torch.nonzero((o <= lengths.unsqueeze(1)).reshape(-1)).reshape(-1)
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


def leaf_symbols(workload, seed, first_read, nreads, readlen, total_reads, device, **workload_args):
    """BWT symbols of the leaf collection reads first_read .. first_read + nreads - 1 of a workload."""
    reads = make_reads(workload, seed, first_read, nreads, readlen, total_reads, device=device, **workload_args)
    return leaf_bwt(reads, read_lengths(workload, first_read, nreads, readlen, device)).contiguous()


def reads_matrix(workload, seed, read_ids, readlen, total_reads, width, **workload_args):
    """The reads with the given indexes as a [len(read_ids), width] uint8 matrix, zero after each read's end (CPU)."""
    reads = make_reads(workload, seed, read_ids, None, readlen, total_reads, device="cpu", **workload_args).numpy()
    out = np.zeros((reads.shape[0], width), dtype=np.uint8)
    out[:, : reads.shape[1]] = reads[:, :width]
    return out


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


def build_index(pkg, seed, nreads, readlen=100, leaf_reads=1 << 19, device="cuda", progress=None, workload="iid", native=True,
                **workload_args):
    """Index of the synthetic set `seed` (reads 0 .. nreads-1 in generation order).  native=True: the library's builder
    (bwtm_builder_*: suffix-sorted leaves + merge tree inside the library); native=False: the tensor-op leaves below with
    the same merge tree driven from here (what the CPU suite checks against the oracle; kept as a cross-check)."""
    if native and str(device).startswith("cuda"):
        builder = pkg.Builder(leaf_reads)
        for first in range(0, nreads, leaf_reads):
            count = min(leaf_reads, nreads - first)
            reads = make_reads(workload, seed, first, count, readlen, nreads, device=device, **workload_args).contiguous()
            lengths = read_lengths(workload, first, count, readlen, device)
            if lengths is not None:
                lengths = lengths.to(torch.int32).contiguous()
            torch.cuda.synchronize()
            builder.add_device(reads.data_ptr(), count, reads.shape[1], lengths_ptr=None if lengths is None else lengths.data_ptr())
            del reads, lengths
            if progress:
                progress(first + count, nreads)
        return builder.finish()
    stack = []                               # (level, index); adjacent entries are adjacent read ranges
    for first in range(0, nreads, leaf_reads):
        count = min(leaf_reads, nreads - first)
        sym = leaf_symbols(workload, seed, first, count, readlen, nreads, device, **workload_args)
        if sym.is_cuda:
            torch.cuda.synchronize()
        leaf = pkg.Index.from_symbols_device(sym.data_ptr(), sym.numel())
        del sym
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


def extract_sequences_matrix(index, seq_ids, max_len=256):
    """The same for many sequences at once: [len(seq_ids), max_len] uint8, forward order, zero after each sequence's end
    (a sequence longer than max_len comes out truncated at its START, which the comparison with the generator catches)."""
    C = index.C
    pos = np.asarray(seq_ids, dtype=np.uint64).copy()
    alive = np.ones(pos.size, dtype=bool)
    back = np.zeros((pos.size, max_len), dtype=np.uint8)          # back[k, t] = symbol t steps before the end
    length = np.zeros(pos.size, dtype=np.int64)
    for t in range(max_len):
        if not alive.any():
            break
        r, c = index.inverse_select(pos)
        alive &= (c != 0)
        back[alive, t] = c[alive]
        length[alive] = t + 1
        pos = np.where(alive, C[c.astype(np.int64)] + r, pos).astype(np.uint64)
    t = np.arange(max_len, dtype=np.int64)[None, :]
    src = np.clip(length[:, None] - 1 - t, 0, max_len - 1)
    return np.where(t < length[:, None], np.take_along_axis(back, src, axis=1), 0).astype(np.uint8)


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
