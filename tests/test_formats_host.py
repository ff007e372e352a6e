"""The foreign BWT file formats of the host facade (reference formats.h:64-156,
formats.cpp:100-445) through the bwt_convert / bwt_inspect tools.  CPU only.

The expected bytes are restated here from the format definitions (one byte per run with 3 + 5 or 5 + 3
bit fields, SDSL int_vector<8> = 64-bit bit count + bytes padded to 8, ...), independently of the C++
codecs, and the native side is checked against the oracle's canonical encoder.  The reference itself
cannot be built in this image (SDSL is missing), so these layouts are pinned to the reference's source
text, not to files it produced."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "bwt-merge_amd", "csrc", "host")
DEFAULT = b"$ACGTN"
SORTED = b"$ACGNT"
NATIVE_DATA_OFFSET = 24 + 8          # NativeHeader (4 + 4 + 8 + 8) + BlockArray size (Appendix B)


@pytest.fixture(scope="module")
def tools(bwtm):
    bwtm.build()
    subprocess.check_call(["make", "-C", HOST, "-s"])
    return HOST


def convert(tools, src, dst, i, o):
    out = subprocess.run([os.path.join(tools, "bwt_convert"), "-i", i, "-o", o, str(src), str(dst)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    return out


def native_parts(path):
    raw = open(path, "rb").read()
    tag, flags, sequences, bases = struct.unpack("<IIQQ", raw[:24])
    (size,) = struct.unpack("<Q", raw[24:32])
    return tag, flags, sequences, bases, np.frombuffer(raw[NATIVE_DATA_OFFSET:NATIVE_DATA_OFFSET + size], dtype=np.uint8)


def native_pieces(data):
    """(comp, length) of every encoded run of a native byte stream (support.h:236-250): the writers of the
    run-byte formats go through the native pieces (a long run is cut at 64-byte block boundaries)."""
    out, k = [], 0
    data = bytes(data)
    while k < len(data):
        c, n = data[k] % 6, data[k] // 6 + 1
        k += 1
        if n >= 42:
            shift = 0
            while True:
                v = data[k]; k += 1
                n += (v & 0x7F) << shift; shift += 7
                if not v & 0x80:
                    break
        out.append((c, n))
    return out


def split31(runs):
    out = []
    for c, n in runs:
        while n > 31:
            out.append((c, 31)); n -= 31
        out.append((c, n))
    return out


@pytest.fixture(scope="module")
def sample():
    rng = np.random.default_rng(5)
    syms = rng.integers(0, 6, 4000)
    lens = rng.choice([1, 1, 2, 3, 31, 32, 33, 62, 63, 100, 5000], 4000)
    return np.repeat(syms, lens).astype(np.uint8)


def test_usage_and_format_list(tools):
    out = subprocess.run([os.path.join(tools, "bwt_convert")], capture_output=True, text=True)
    assert out.returncode == 0
    for tag in ("native", "plain_default", "plain_sorted", "rfm", "sdsl", "ropebwt", "sga"):
        assert ("  " + tag) in out.stderr
    assert "Formats using sorted alphabet:" in out.stderr
    bad = subprocess.run([os.path.join(tools, "bwt_convert"), "-i", "nonsense", "a", "b"], capture_output=True, text=True)
    assert bad.returncode != 0 and "Invalid input format: nonsense" in bad.stderr


def test_default_alphabet_formats(tools, oracle, sample, tmp_path):
    """plain_default -> native is the oracle's canonical encoding; native -> ropebwt / sga are the documented
    one-byte-per-run layouts; every format converts back to the identical native file."""
    f = oracle.FMI.from_symbols(sample)
    plain = tmp_path / "x.plain"
    np.frombuffer(DEFAULT, dtype=np.uint8)[sample].tofile(plain)
    native = tmp_path / "x.native"
    convert(tools, plain, native, "plain_default", "native")
    tag, flags, sequences, bases, data = native_parts(native)
    assert (tag, flags & 0xFF, sequences, bases) == (0x54574221, 0, f.sequences, f.bases)
    assert np.array_equal(data, f.data)

    assert sum(n for _, n in native_pieces(f.data)) == sample.size
    codes = split31(native_pieces(f.data))
    convert(tools, native, tmp_path / "x.rope", "native", "ropebwt")
    raw = open(tmp_path / "x.rope", "rb").read()
    assert raw[:4] == struct.pack("<I", 0x06454C52)
    assert raw[4:] == bytes((n << 3) | c for c, n in codes)

    convert(tools, native, tmp_path / "x.sga", "native", "sga")
    raw = open(tmp_path / "x.sga", "rb").read()
    assert raw[:30] == struct.pack("<HQQQI", 0xCACA, f.sequences, f.bases, len(codes), 0)
    assert raw[30:] == bytes((c << 5) | n for c, n in codes)

    convert(tools, native, tmp_path / "x.plain2", "native", "plain_default")
    assert open(tmp_path / "x.plain2", "rb").read() == open(plain, "rb").read()
    for fmt, name in (("ropebwt", "x.rope"), ("sga", "x.sga"), ("plain_default", "x.plain2")):
        back = tmp_path / ("back." + fmt)
        convert(tools, tmp_path / name, back, fmt, "native")
        assert open(back, "rb").read() == open(native, "rb").read(), fmt

    out = subprocess.run([os.path.join(tools, "bwt_inspect"), str(native), str(tmp_path / "x.sga"), str(tmp_path / "x.rope"), str(plain)],
                         capture_output=True, text=True)
    assert "Native format: %d sequences, %d bases, default alphabet" % (f.sequences, f.bases) in out.stdout
    assert "SGA format: %d sequences, %d bases, %d bytes" % (f.sequences, f.bases, len(codes)) in out.stdout
    assert "RopeBWT format" in out.stdout and "Unknown format" in out.stdout
    assert "Total: %d sequences, %d bases" % (2 * f.sequences, 2 * f.bases) in out.stdout


def test_sorted_alphabet_formats(tools, oracle, sample, tmp_path):
    """Sorted alphabet ($ACGNT): comp 4 is N and comp 5 is T.  rfm stores comp values, sdsl characters, both
    as SDSL int_vector<8> files."""
    chars = np.frombuffer(SORTED, dtype=np.uint8)[sample]
    plain = tmp_path / "s.plain"
    chars.tofile(plain)
    native = tmp_path / "s.native"
    convert(tools, plain, native, "plain_sorted", "native")
    f = oracle.FMI.from_symbols(sample)              # same comp values, whatever the characters are
    tag, flags, sequences, bases, data = native_parts(native)
    assert (flags & 0xFF, sequences, bases) == (1, f.sequences, f.bases) and np.array_equal(data, f.data)

    pad = bytes((8 - sample.size % 8) % 8)
    convert(tools, native, tmp_path / "s.rfm", "native", "rfm")
    assert open(tmp_path / "s.rfm", "rb").read() == struct.pack("<Q", 8 * sample.size) + sample.tobytes() + pad
    convert(tools, native, tmp_path / "s.sdsl", "native", "sdsl")
    assert open(tmp_path / "s.sdsl", "rb").read() == struct.pack("<Q", 8 * sample.size) + chars.tobytes() + pad
    for fmt, name in (("rfm", "s.rfm"), ("sdsl", "s.sdsl"), ("plain_sorted", "s.plain")):
        back = tmp_path / ("back." + fmt)
        convert(tools, tmp_path / name, back, fmt, "native")
        assert open(back, "rb").read() == open(native, "rb").read(), fmt

    # a format of the other alphabetic order only warns (reference fmi.h:117-122)
    out = convert(tools, native, tmp_path / "s.sga", "native", "sga")
    assert "SGA format is not compatible with sorted alphabets" in out.stderr


def test_plain_runs_are_formed_before_mapping(tools, oracle, tmp_path):
    """formats.cpp:147-156: 'a' next to 'A' stays two runs, unknown characters become N, and comp values
    outside the alphabet of an rfm file become 0."""
    text = b"AAaaACGTTtXXNn$$"
    plain = tmp_path / "m.plain"
    open(plain, "wb").write(text)
    convert(tools, plain, tmp_path / "m.native", "plain_default", "native")
    _, _, sequences, bases, data = native_parts(tmp_path / "m.native")
    expect = [(1, 2), (1, 2), (1, 1), (2, 1), (3, 1), (4, 2), (4, 1), (5, 2), (5, 1), (5, 1), (0, 2)]
    assert (sequences, bases) == (2, len(text))
    assert data.tobytes() == bytes(c + 6 * (n - 1) for c, n in expect)
    rfm = tmp_path / "m.rfm"
    vals = bytes([1, 1, 7, 7, 2, 200, 0, 5])
    open(rfm, "wb").write(struct.pack("<Q", 64) + vals)
    convert(tools, rfm, tmp_path / "m2.native", "rfm", "native")
    _, flags, sequences, bases, data = native_parts(tmp_path / "m2.native")
    assert (flags & 0xFF, sequences, bases) == (1, 4, 8)
    assert data.tobytes() == bytes(c + 6 * (n - 1) for c, n in [(1, 2), (0, 2), (2, 1), (0, 1), (0, 1), (5, 1)])


@pytest.mark.parametrize("which", ["runs", "reads"])
def test_native_file_through_the_independent_reader(tools, oracle, sample, tmp_path, which):
    """a22: the native file the facade writes, parsed by a second implementation of the SDSL framing (tests/sdsl_native_reader.py:
    written from the published container layouts, not from the C++ writer).  Every container is walked by its own size fields, the
    select supports are checked against the bit vectors they index, the file must end where the alphabet ends, and everything the
    path computes -- header, data, per-block counts, block boundaries, C -- equals the oracle's.  `reads`: 9000 reads = 2.6 superblocks
    of 4096 blocks in the select support of block_boundaries."""
    import sdsl_native_reader as reader
    sym = sample if which == "runs" else oracle.FMI.from_text(oracle.generate_reads(77, 9000, 100)).symbols
    f = oracle.FMI.from_symbols(sym)
    plain, native = tmp_path / "in.plain", tmp_path / "out.native"
    np.frombuffer(DEFAULT, dtype=np.uint8)[sym].tofile(plain)
    convert(tools, plain, native, "plain_default", "native")
    got = reader.parse_native(native)
    assert (got["flags"] & 0xFF, got["sequences"], got["bases"]) == (0, f.sequences, f.bases)
    assert np.array_equal(got["data"], f.data)
    be, cum = f.samples                                                    # block_end [blocks], cum [6][blocks + 1]
    assert np.array_equal(got["block_end"], be) and got["boundaries_size"] == f.bases
    assert np.array_equal(got["counts"], np.diff(cum, axis=1))
    assert np.array_equal(got["C"], f.C) and got["sigma"] == 6
    assert bytes(got["comp2char"].astype(np.uint8)) == DEFAULT
    c2c = got["char2comp"]
    assert c2c.size == 256 and all(c2c[ch] == k for k, ch in enumerate(DEFAULT))
    for s1, s0 in got["select_supports"]:
        assert s1["arguments"] == f.blocks                                   # every vector has one 1-bit per block
    if which == "reads":
        assert got["select_supports"][-1][0]["superblocks"] == 3
    # and back: the file the reader accepted loads into the facade again (native -> plain == the input)
    back = tmp_path / "back.plain"
    convert(tools, native, back, "native", "plain_default")
    assert back.read_bytes() == plain.read_bytes()


def test_foreign_file_check_tool(tools, oracle, tmp_path):
    """tools/check_foreign_files.py -- the one command that closes a22 / f4 the day a file written by SDSL, RopeBWT or SGA is available -- on the
    only files there are here, the repo's own: it must find the format of each and see it round-trip, and it must fail on a damaged file."""
    x = oracle.FMI.from_text(oracle.generate_reads(5, 400, 50))
    chars = np.frombuffer(b"$ACGTN", dtype=np.uint8)
    d = tmp_path / "files"; d.mkdir()
    plain = str(d / "x.plain")
    with open(plain, "wb") as f:
        f.write(chars[x.symbols].tobytes())
    for fmt, name in (("native", "x.native"), ("sga", "x.sga"), ("ropebwt", "x.rope"), ("rfm", "x.rfm"), ("sdsl", "x.sdsl")):
        convert(tools, plain, str(d / name), "plain_default", fmt)
    tool = os.path.join(ROOT, "tools", "check_foreign_files.py")
    out = subprocess.run([sys.executable, tool, str(d)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for name, fmt in (("x.native", "native"), ("x.sga", "sga"), ("x.rope", "ropebwt"), ("x.rfm", "rfm"), ("x.sdsl", "sdsl")):
        assert "| %s | %s | True | True |" % (name, fmt) in out.stdout, out.stdout
    raw = bytearray(open(str(d / "x.sga"), "rb").read())
    raw[40] ^= 0x55                                                     # a run byte of the SGA file
    bad = tmp_path / "bad"; bad.mkdir()
    with open(str(bad / "y.sga"), "wb") as f:
        f.write(bytes(raw[:-3]))                                        # and a truncated tail
    out = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert out.returncode == 1 and "round-tripped byte for byte in at least one format: False" in out.stdout
