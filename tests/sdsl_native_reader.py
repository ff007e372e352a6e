"""An independent reader of the reference's NATIVE file (what FMI::serialize<NativeFormat> writes: fmi.cpp:87-98 -> bwt.cpp:111-130 ->
support.cpp:160-171, 296-324, 442-454), written in Python from the published layout of sdsl-lite's containers --
int_vector, bit_vector, sd_vector (Elias-Fano: low bits + unary-coded high bits) and select_support_mcl (superblocks of 4096 arguments,
mini blocks of every 64th, long blocks of all positions) -- and NOT from the C++ writer in bwt-merge_amd/csrc/host/sdsl_compat.h.

Purpose: SDSL is not installed here and the reference ships no files, so no byte of this framing has ever met a
file that SDSL wrote.  Two implementations that were written separately and agree on every file raise the confidence that the
framing is what the description says; they cannot prove that the description matches the SDSL revision the reference links.  Test
infrastructure only: nothing under bwt-merge_amd/ imports it.

Every container is parsed by its own size fields (the reader never assumes a width the file does not state), the select supports are
checked against the bit vectors they index, and parse_native() insists that the file ends exactly where the last member ends.
"""
import struct

import numpy as np

NATIVE_TAG = 0x54574221          # formats.h:35
BLOCK_BYTES = 8 * 1048576        # BlockArray::BLOCK_SIZE, support.h:125-128 (serialized as whole blocks, support.cpp:296-309)
SUPER = 4096                     # select_support_mcl: arguments per superblock
MINI = 64                        # ... per mini block


class Cursor:
    def __init__(self, raw):
        self.raw, self.pos = raw, 0

    def take(self, n):
        if self.pos + n > len(self.raw):
            raise ValueError("file ends inside a member (wanted %d bytes at offset %d of %d)" % (n, self.pos, len(self.raw)))
        out = self.raw[self.pos: self.pos + n]
        self.pos += n
        return out

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def u8(self):
        return self.take(1)[0]


def _unpack(words, count, width):
    """`count` integers of `width` bits from little-endian 64-bit words (entry j = bits [j width, (j + 1) width))."""
    if count == 0:
        return np.zeros(0, dtype=np.uint64)
    if width == 64:
        return words[:count].copy()
    if width == 8:
        return words.view(np.uint8)[:count].astype(np.uint64)
    bits = np.unpackbits(words.view(np.uint8), bitorder="little")[: count * width].reshape(count, width).astype(np.uint64)
    return (bits << np.arange(width, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)


def read_int_vector(cur, fixed_width):
    """sdsl::int_vector<w>: u64 size in BITS, (w == 0: one byte with the width), then ceil(bits / 64) 64-bit words.  Returns (width, values)."""
    bits = cur.u64()
    width = fixed_width if fixed_width else cur.u8()
    if width == 0 or width > 64 or bits % width != 0:
        raise ValueError("int_vector of %d bits with width %d" % (bits, width))
    words = np.frombuffer(cur.take(8 * ((bits + 63) // 64)), dtype="<u8")
    return width, _unpack(words, bits // width, width)


def read_bit_vector(cur):
    bits = cur.u64()
    words = np.frombuffer(cur.take(8 * ((bits + 63) // 64)), dtype="<u8")
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:bits]


def read_select_mcl(cur, vector_bits, wanted):
    """sdsl::select_support_mcl<wanted>: u64 number of arguments; if there are any: the superblock array (position of every 4096th
    argument), a bit vector that says which superblocks are MINI (empty = all of them), then one int_vector<0> per superblock: a
    long block holds the positions of all its arguments, a mini block the offset of every 64th from the superblock's first.
    Checked against the vector it indexes."""
    args = cur.u64()
    where = np.nonzero(vector_bits == wanted)[0].astype(np.uint64)
    if args != where.size:
        raise ValueError("select support for %d-bits counts %d arguments, the vector has %d" % (wanted, args, where.size))
    if args == 0:
        return {"arguments": 0, "superblocks": 0, "long": 0}
    _, superblock = read_int_vector(cur, 0)
    sb = (args + SUPER - 1) // SUPER
    if superblock.size != sb or not np.array_equal(superblock, where[::SUPER]):
        raise ValueError("select support: superblock array does not hold the position of every 4096th argument")
    mini_or_long = read_bit_vector(cur)
    if mini_or_long.size not in (0, sb):
        raise ValueError("select support: mini_or_long has %d entries for %d superblocks" % (mini_or_long.size, sb))
    nlong = 0
    for i in range(sb):
        _, block = read_int_vector(cur, 0)
        mine = where[i * SUPER: (i + 1) * SUPER]
        is_mini = (mini_or_long.size == 0 or mini_or_long[i] == 1)
        if is_mini:
            expect = mine[::MINI] - mine[0]
            if block.size != MINI or not np.array_equal(block[: expect.size], expect):
                raise ValueError("select support: mini block %d does not hold the offset of every 64th argument" % i)
        else:
            nlong += 1
            if block.size != SUPER or not np.array_equal(block[: mine.size], mine):
                raise ValueError("select support: long block %d does not hold the positions of its arguments" % i)
    return {"arguments": int(args), "superblocks": int(sb), "long": nlong}


def hi(x):
    return x.bit_length() - 1 if x > 0 else 0          # sdsl::bits::hi


def read_sd_vector(cur):
    """sdsl::sd_vector<>: u64 size n, u8 wl, low (int_vector<0>: m entries of wl bits), high (bit_vector: item j sets bit (p_j >> wl) + j),
    select_support_mcl<1> and <0> on high.  Returns (n, positions of the m ones, what the two select supports reported)."""
    n = cur.u64()
    wl = cur.u8()
    width, low = read_int_vector(cur, 0)
    high = read_bit_vector(cur)
    ones = np.nonzero(high)[0].astype(np.uint64)
    m = int(ones.size)
    if low.size != m:
        raise ValueError("sd_vector: %d low parts for %d ones in high" % (low.size, m))
    if width != wl:
        raise ValueError("sd_vector: low parts are %d bits wide, wl = %d" % (width, wl))
    # the width rule of the builder: logm = hi(m) + 1, logn = hi(n) + 1, if they are equal logm -= 1; wl = logn - logm
    logm, logn = hi(m) + 1, hi(n) + 1
    if logm == logn:
        logm -= 1
    if wl != logn - logm:
        raise ValueError("sd_vector: wl = %d, the builder's rule gives %d for n = %d, m = %d" % (wl, logn - logm, n, m))
    if high.size != m + (1 << logm):
        raise ValueError("sd_vector: high has %d bits, expected m + 2^logm = %d" % (high.size, m + (1 << logm)))
    positions = ((ones - np.arange(m, dtype=np.uint64)) << np.uint64(wl)) | (low & np.uint64((1 << wl) - 1) if wl else np.zeros(m, dtype=np.uint64))
    if m > 0 and (np.any(np.diff(positions.astype(np.int64)) <= 0) or int(positions[-1]) >= n):
        raise ValueError("sd_vector: positions are not strictly increasing below n")
    s1 = read_select_mcl(cur, high, 1)
    s0 = read_select_mcl(cur, high, 0)
    return n, positions, (s1, s0)


def read_cumulative_array(cur):
    """CumulativeArray (support.h:290-380, support.cpp:442-454): sd_vector, three support structures that serialize as nothing, u64 size.
    Element k is stored as that many 0-bits followed by a 1-bit.  Returns the elements."""
    n, positions, supports = read_sd_vector(cur)
    size = cur.u64()
    if size != positions.size:
        raise ValueError("CumulativeArray: size %d, %d ones" % (size, positions.size))
    prev = np.concatenate([[np.uint64(0)], positions[:-1] + np.uint64(1)]) if size else positions
    return positions - prev, supports


def parse_native(path):
    """The whole file.  Returns a dict: header fields, data (uint8), per-block counts [6][blocks], block_end [blocks], alphabet arrays."""
    raw = open(path, "rb").read()
    cur = Cursor(raw)
    tag, flags, sequences, bases = struct.unpack("<IIQQ", cur.take(24))
    if tag != NATIVE_TAG:
        raise ValueError("not a native file: tag %08x" % tag)
    nbytes = cur.u64()
    stored = (nbytes + BLOCK_BYTES - 1) // BLOCK_BYTES * BLOCK_BYTES
    body = np.frombuffer(cur.take(stored), dtype=np.uint8)
    if np.any(body[nbytes:] != 0):
        raise ValueError("BlockArray: the tail of the last 8 MiB block is not zero")
    counts, supports = [], []
    for c in range(6):
        elements, sup = read_cumulative_array(cur)
        counts.append(elements); supports.append(sup)
    n, boundaries, bsup = read_sd_vector(cur)
    _, char2comp = read_int_vector(cur, 8)
    _, comp2char = read_int_vector(cur, 8)
    _, C = read_int_vector(cur, 64)
    sigma = cur.u64()
    if cur.pos != len(raw):
        raise ValueError("%d bytes left behind the alphabet" % (len(raw) - cur.pos))
    return {"flags": flags, "sequences": sequences, "bases": bases, "data": body[:nbytes].copy(), "counts": np.array(counts), "block_end": boundaries,
            "boundaries_size": n, "char2comp": char2comp, "comp2char": comp2char, "C": C, "sigma": sigma, "select_supports": supports + [bsup]}
