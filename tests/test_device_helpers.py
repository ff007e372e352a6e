"""The pure helpers shared by all kernels (bwtm_device.h), compiled for the host and checked
against the reference's known-answer vectors and the oracle's codec."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "libshim.so")
    src = os.path.join(HERE, "host_shim.cpp")
    hdr = os.path.join(HERE, "..", "bwt-merge_amd", "csrc", "bwtm_device.h")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", lib, src])
    L = C.CDLL(lib)
    u64, u32 = C.c_uint64, C.c_uint32
    L.shim_long_run_bytes.restype = u64; L.shim_long_run_bytes.argtypes = [u64, u64]
    L.shim_long_run_write.restype = u64; L.shim_long_run_write.argtypes = [C.c_void_p, u64, u32, u64]
    L.shim_pack_header.argtypes = [C.c_void_p, C.c_void_p]
    L.shim_rec_header.restype = u32; L.shim_rec_header.argtypes = [C.c_void_p, u32]
    L.shim_rec_count.restype = u32; L.shim_rec_count.argtypes = [C.c_void_p, u32, u32]
    L.shim_rec_symbol.restype = u32; L.shim_rec_symbol.argtypes = [C.c_void_p, u32]
    L.shim_range_mask128.argtypes = [u32, u32, C.c_void_p, C.c_void_p]
    L.shim_encode_runs.restype = u64; L.shim_encode_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u64, C.c_void_p]
    L.shim_deposit64.argtypes = [u64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.shim_bit_merge32.argtypes = [u32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.shim_run_decode.restype = u64; L.shim_run_decode.argtypes = [C.c_void_p, u64, C.c_void_p, C.c_void_p]
    L.shim_view_build.restype = C.c_int; L.shim_view_build.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p]
    L.shim_view_symbol.restype = u32; L.shim_view_symbol.argtypes = [C.c_void_p, u32]
    L.shim_view_count.restype = u32; L.shim_view_count.argtypes = [C.c_void_p, u32, u32]
    L.shim_view_header.restype = u32; L.shim_view_header.argtypes = [C.c_void_p, u32]
    L.shim_view_overflow.restype = C.c_int; L.shim_view_overflow.argtypes = [C.c_void_p]
    return L


def test_long_run_encoder_matches_reference_vectors(shim, golden, oracle):
    g = golden["run_write"]
    for row in g["rows"]:
        for off in (0, 61, 62, 63):
            expect = bytes.fromhex(row["off%d" % off].replace(" ", ""))
            buf = np.zeros(256, dtype=np.uint8)
            n = shim.shim_long_run_write(buf.ctypes.data, 64 * 2 + off, g["comp"], row["len"])
            assert bytes(buf[128 + off: 128 + off + n]) == expect
            assert shim.shim_long_run_bytes(off, row["len"]) == len(expect)


def test_long_run_encoder_matches_oracle_everywhere(shim, oracle):
    rng = np.random.default_rng(0)
    lengths = [1, 2, 40, 41, 42, 43, 44, 127, 128, 168, 169, 170, 171, 300, 16424, 16425, 16426, 16427, 2 ** 21, 2 ** 21 + 41,
               2 ** 28 + 5, 2 ** 35 + 9, 2 ** 40 + 1] + [int(x) for x in rng.integers(42, 10 ** 7, 40)]
    for length in lengths:
        for off in range(64):
            ref = oracle.run_write(5, length, prefill=off)
            assert shim.shim_long_run_bytes(off, length) == len(ref), (length, off)
            buf = np.zeros(128, dtype=np.uint8)
            n = shim.shim_long_run_write(buf.ctypes.data, off, 5, length)
            assert bytes(buf[off: off + n]) == ref, (length, off)


def test_record_helpers(shim):
    rng = np.random.default_rng(1)
    for _ in range(200):
        sym = rng.integers(0, 6, 128).astype(np.uint32)
        rel = np.zeros(6, dtype=np.uint32); rel[1:] = rng.integers(0, 1 << 25, 5)
        h = np.zeros(4, dtype=np.uint32)
        shim.shim_pack_header(rel.ctypes.data, h.ctypes.data)
        w = np.zeros(16, dtype=np.uint32)
        for k in range(4):
            for plane in range(3):
                bits = (sym[32 * k: 32 * k + 32] >> plane) & 1
                w[4 * k + plane] = int(sum(int(b) << t for t, b in enumerate(bits)))
            w[4 * k + 3] = h[k]
        for c in range(1, 6):
            assert shim.shim_rec_header(w.ctypes.data, c) == rel[c]
        for j in list(rng.integers(0, 129, 8)) + [0, 128, 31, 32, 33, 64, 96]:
            j = int(j)
            for c in range(6):
                assert shim.shim_rec_count(w.ctypes.data, c, j) == int(np.sum(sym[:j] == c))
            if j < 128:
                assert shim.shim_rec_symbol(w.ctypes.data, j) == sym[j]


def test_search_view_record_helpers(shim):
    """The two-plane search view (bwtm_device.h): symbol and rank at every position of records with 0 .. 7 exceptions (endmarkers
    and N in any mix, also at positions 0 and 159), the header fields, and the overflow flag from the eighth exception on."""
    rng = np.random.default_rng(3)
    for trial in range(300):
        count = 160 if trial % 5 else int(rng.integers(1, 161))                      # the last record of an index is shorter
        sym = rng.integers(1, 5, 160).astype(np.uint8)
        nexc = int(rng.integers(0, 10)) if trial % 3 else int(rng.integers(0, 8))
        where = rng.choice(count, size=min(nexc, count), replace=False)
        if trial % 7 == 0 and count == 160 and nexc >= 2:
            where[0], where[1] = 0, 159
        sym[where] = rng.choice([0, 5], size=where.size)
        rel = np.zeros(6, dtype=np.uint32); rel[1:] = rng.integers(0, 1 << 25, 5)
        v = np.zeros(16, dtype=np.uint32)
        got_exc = shim.shim_view_build(sym.ctypes.data, count, rel.ctypes.data, v.ctypes.data)
        true_exc = int(np.sum((sym[:count] == 0) | (sym[:count] == 5)))
        assert got_exc == true_exc and shim.shim_view_overflow(v.ctypes.data) == (1 if true_exc > 7 else 0)
        for c in range(1, 6):
            assert shim.shim_view_header(v.ctypes.data, c) == rel[c]
        if true_exc > 7:
            continue                                                                    # such records are answered by the ordinary ones
        for j in range(count + 1):
            for c in range(1, 6):
                assert shim.shim_view_count(v.ctypes.data, c, j) == int(np.sum(sym[:j] == c)), (trial, j, c)
            if j < count:
                assert shim.shim_view_symbol(v.ctypes.data, j) == sym[j], (trial, j)


def test_range_mask(shim):
    for a in range(0, 128, 7):
        for n in (1, 2, 31, 32, 33, 63, 64, 65, 128 - a):
            if n <= 0 or a + n > 128:
                continue
            lo = C.c_uint64(0); hi = C.c_uint64(0)
            shim.shim_range_mask128(a, n, C.byref(lo), C.byref(hi))
            v = lo.value | (hi.value << 64)
            assert v == ((1 << n) - 1) << a


def test_encoder_block_starts_match_oracle_samples(shim, oracle):
    """Run::write splits long runs at block boundaries (support.h:256-282); the block start positions the
    encoder records while writing must be the reference's block_boundaries (bwt.cpp:496)."""
    rng = np.random.default_rng(3)
    for lengths in ([1, 2, 3], [1, 41, 42, 43, 169, 170, 171], [1, 5000, 16425, 16426, 2 ** 21, 2 ** 35 + 7], [41, 42]):
        nruns = 700
        syms = rng.integers(0, 6, nruns)
        for k in range(1, nruns):
            if syms[k] == syms[k - 1]:
                syms[k] = (syms[k] + 1) % 6
        syms = syms.astype(np.uint8)
        lens = rng.choice(np.array(lengths, dtype=np.uint64), nruns)
        out = np.zeros(16 * nruns, dtype=np.uint8)
        bs = np.full(16 * nruns // 64 + 2, 2 ** 64 - 1, dtype=np.uint64)
        nbytes = shim.shim_encode_runs(out.ctypes.data, syms.ctypes.data, lens.ctypes.data, nruns, bs.ctypes.data)
        bases = int(lens.sum()); sequences = int(lens[syms == 0].sum())
        f = oracle.FMI.from_native(out[:nbytes], sequences, bases)       # the oracle scans the stream (BWT::build)
        assert (f.bases, f.nbytes) == (bases, nbytes)
        obe, _ = f.samples
        assert bs[0] == 0 and np.array_equal(bs[1:f.blocks] - 1, obe[:-1])
        assert np.all(bs[f.blocks:] == 2 ** 64 - 1)


def test_deposit64_is_the_bitwise_interleave(shim):
    """mergeBWT (bwt.cpp:215-282) restricted to 64 output positions: position t takes the next
    symbol of B where the mask bit is set and the next symbol of A otherwise."""
    rng = np.random.default_rng(7)
    masks = [0, (1 << 64) - 1, 1, 1 << 63, 0xFFFFFFFF, 0xFFFFFFFF00000000, 0xAAAAAAAAAAAAAAAA]
    masks += [int(x) for x in rng.integers(0, 1 << 63, 300, dtype=np.uint64)]
    masks += [int(x) & int(y) & int(z) for x, y, z in rng.integers(0, 1 << 63, (100, 3), dtype=np.uint64)]   # sparse
    masks += [(int(x) | int(y) | int(z) | (1 << 63)) for x, y, z in rng.integers(0, 1 << 63, (100, 3), dtype=np.uint64)]   # dense
    for mask in masks:
        a = rng.integers(0, 1 << 63, 3, dtype=np.uint64) * 2 + rng.integers(0, 2, 3, dtype=np.uint64)
        b = rng.integers(0, 1 << 63, 3, dtype=np.uint64) * 2 + rng.integers(0, 2, 3, dtype=np.uint64)
        o = np.zeros(3, dtype=np.uint64)
        shim.shim_deposit64(mask, a.ctypes.data, b.ctypes.data, o.ctypes.data)
        for plane in range(3):
            av, bv, expect = int(a[plane]), int(b[plane]), 0
            for t in range(64):
                if (mask >> t) & 1:
                    expect |= (bv & 1) << t; bv >>= 1
                else:
                    expect |= (av & 1) << t; av >>= 1
            assert int(o[plane]) == expect, (hex(mask), plane)


def test_bit_merge32_is_the_bitwise_interleave(shim):
    """The bit merge of k_interleave (bwtm_bitmerge.h: bit-sliced prefix counts + five pull rounds stated at the destination) against the
    definition of mergeBWT on 32 output positions, for every shape of mask: empty, full, single bits, runs, sparse, dense, random."""
    rng = np.random.default_rng(11)
    masks = [0, 0xFFFFFFFF, 1, 1 << 31, 0xFFFF, 0xFFFF0000, 0xAAAAAAAA, 0x55555555, 0x7FFFFFFF, 0xFFFFFFFE, 0x80000001]
    masks += [(1 << k) - 1 for k in range(33)] + [((1 << k) - 1) << (32 - k) for k in range(33)] + [1 << k for k in range(32)]
    masks += [int(x) for x in rng.integers(0, 1 << 32, 3000, dtype=np.uint64)]
    masks += [int(x) & int(y) & int(z) for x, y, z in rng.integers(0, 1 << 32, (500, 3), dtype=np.uint64)]
    masks += [(int(x) | int(y) | int(z)) & 0xFFFFFFFF for x, y, z in rng.integers(0, 1 << 32, (500, 3), dtype=np.uint64)]
    for mask in masks:
        mask &= 0xFFFFFFFF
        a = rng.integers(0, 1 << 32, 3, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 1 << 32, 3, dtype=np.uint64).astype(np.uint32)
        o = np.zeros(3, dtype=np.uint32)
        shim.shim_bit_merge32(mask, a.ctypes.data, b.ctypes.data, o.ctypes.data)
        for plane in range(3):
            av, bv, expect = int(a[plane]), int(b[plane]), 0
            for t in range(32):
                if (mask >> t) & 1:
                    expect |= (bv & 1) << t; bv >>= 1
                else:
                    expect |= (av & 1) << t; av >>= 1
            assert int(o[plane]) == expect, (hex(mask), plane)


def test_run_decode_matches_oracle(shim, oracle):
    data = b"".join(oracle.run_write(c, l, prefill=0) for c, l in [(1, 1), (5, 41), (0, 42), (3, 1000000), (2, 169)])
    ref = oracle.run_decode(data)
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    pos, got = 0, []
    while pos < buf.size:
        sym = C.c_uint32(0); ln = C.c_uint64(0)
        pos = shim.shim_run_decode(buf.ctypes.data, pos, C.byref(sym), C.byref(ln))
        got.append((sym.value, ln.value))
    assert got == ref
