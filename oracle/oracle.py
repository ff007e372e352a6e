"""ctypes front end of the CPU oracle (oracle/bwtm_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py -- never by the product package.  Nothing here touches a GPU or reads
/root/reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")
SIGMA = 6

u64 = C.c_uint64
p_u8 = C.POINTER(C.c_uint8)
p_u64 = C.POINTER(C.c_uint64)


class OrcParams(C.Structure):
    _fields_ = [("run_buffer_size", u64), ("thread_buffer_size", u64), ("merge_buffers", u64),
                ("threads", u64), ("sequence_blocks", u64)]


def build(force=False):
    """Compile liborc.so with g++ (seconds)."""
    src = os.path.join(_HERE, "bwtm_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liborc.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        def sig(name, res, *args):
            f = getattr(L, name); f.restype = res; f.argtypes = list(args)
        sig("orc_run_write", u64, u64, u64, u64, p_u8, u64)
        sig("orc_run_decode", u64, p_u8, u64, p_u64, p_u64, u64)
        sig("orc_bytecode_write", u64, u64, p_u8, u64)
        sig("orc_bytecode_read", u64, p_u8, p_u64)
        sig("orc_get_bounds", u64, u64, u64, u64, p_u64, p_u64, u64)
        sig("orc_fnv1a_bytes", u64, p_u8, u64)
        sig("orc_runbuffer", u64, p_u64, p_u64, u64, p_u64, p_u64)
        sig("orc_generate_reads", None, u64, u64, u64, u64, p_u8)
        sig("orc_fmi_from_text", C.c_void_p, p_u8, u64)
        sig("orc_fmi_from_symbols", C.c_void_p, p_u8, u64)
        sig("orc_fmi_from_native", C.c_void_p, p_u8, u64, u64, u64)
        sig("orc_fmi_from_runs", C.c_void_p, p_u64, p_u64, u64)
        sig("orc_fmi_clone", C.c_void_p, C.c_void_p)
        sig("orc_fmi_free", None, C.c_void_p)
        for n in ("bases", "sequences", "bytes", "blocks", "hash"):
            sig("orc_fmi_" + n, u64, C.c_void_p)
        sig("orc_fmi_C", None, C.c_void_p, p_u64)
        sig("orc_fmi_data", None, C.c_void_p, p_u8)
        sig("orc_fmi_symbols", None, C.c_void_p, p_u8)
        sig("orc_fmi_character_counts", None, C.c_void_p, p_u64)
        sig("orc_fmi_samples", None, C.c_void_p, p_u64, p_u64)
        sig("orc_rank", u64, C.c_void_p, u64, u64)
        sig("orc_select", u64, C.c_void_p, u64, u64)
        sig("orc_at", u64, C.c_void_p, u64)
        sig("orc_inverse_select", None, C.c_void_p, u64, p_u64, p_u64)
        sig("orc_ranks_at", None, C.c_void_p, u64, p_u64)
        sig("orc_ranks_range", None, C.c_void_p, u64, u64, p_u64, p_u64)
        sig("orc_LF", None, C.c_void_p, u64, p_u64, p_u64)
        sig("orc_LF_c", u64, C.c_void_p, u64, u64)
        sig("orc_find", None, C.c_void_p, p_u8, u64, p_u64, p_u64)
        sig("orc_search", u64, C.c_void_p, C.c_void_p, C.POINTER(OrcParams), p_u64, p_u64, u64, p_u64)
        sig("orc_merge", C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OrcParams), C.POINTER(C.c_double))
        sig("orc_merge_timed", C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OrcParams), C.POINTER(C.c_double), C.POINTER(C.c_double))
        sig("orc_interleave_symbols", None, p_u8, u64, p_u8, u64, p_u64, p_u8)
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(p_u8)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(p_u64)


# ---------------------------------------------------------------------------- codecs

def run_write(comp, length, prefill=0):
    buf = np.zeros(64, dtype=np.uint8)
    n = lib().orc_run_write(prefill, comp, length, buf.ctypes.data_as(p_u8), buf.size)
    assert n <= buf.size
    return bytes(buf[:n])


def run_decode(data):
    data, dp = _u8(np.frombuffer(bytes(data), dtype=np.uint8))
    cap = max(1, data.size)
    comps = np.zeros(cap, dtype=np.uint64); lens = np.zeros(cap, dtype=np.uint64)
    n = lib().orc_run_decode(dp, data.size, comps.ctypes.data_as(p_u64), lens.ctypes.data_as(p_u64), cap)
    return [(int(comps[k]), int(lens[k])) for k in range(n)]


def bytecode_write(value):
    buf = np.zeros(16, dtype=np.uint8)
    n = lib().orc_bytecode_write(value, buf.ctypes.data_as(p_u8), buf.size)
    return bytes(buf[:n])


def bytecode_read(data):
    data, dp = _u8(np.frombuffer(bytes(data), dtype=np.uint8))
    used = u64(0)
    v = lib().orc_bytecode_read(dp, C.byref(used))
    return int(v), int(used.value)


def get_bounds(first, last, blocks):
    cap = max(1, blocks)
    f = np.zeros(cap, dtype=np.uint64); l = np.zeros(cap, dtype=np.uint64)
    n = lib().orc_get_bounds(first, last, blocks, f.ctypes.data_as(p_u64), l.ctypes.data_as(p_u64), cap)
    return [(int(f[k]), int(l[k])) for k in range(n)]


def fnv1a(data):
    data, dp = _u8(np.frombuffer(bytes(data), dtype=np.uint8))
    return int(lib().orc_fnv1a_bytes(dp, data.size))


def runbuffer(pairs):
    v, vp = _u64([p[0] for p in pairs]); l, lp = _u64([p[1] for p in pairs])
    ov = np.zeros(len(pairs) + 1, dtype=np.uint64); ol = np.zeros(len(pairs) + 1, dtype=np.uint64)
    n = lib().orc_runbuffer(vp, lp, len(pairs), ov.ctypes.data_as(p_u64), ol.ctypes.data_as(p_u64))
    return [(int(ov[k]), int(ol[k])) for k in range(n)]


# ---------------------------------------------------------------------------- reads

def generate_reads(seed, nreads, readlen, first_read=0):
    """Synthetic reads as a flat uint8 array of comp values, each read 0-terminated."""
    out = np.zeros(nreads * (readlen + 1), dtype=np.uint8)
    lib().orc_generate_reads(seed, first_read, nreads, readlen, out.ctypes.data_as(p_u8))
    return out


COMP = {"$": 0, "A": 1, "C": 2, "G": 3, "T": 4, "N": 5}
CHARS = "$ACGTN"


def text_from_strings(strings):
    out = []
    for s in strings:
        out.extend(COMP[ch] for ch in s)
        out.append(0)
    return np.array(out, dtype=np.uint8)


# ---------------------------------------------------------------------------- FMI

class FMI:
    """Handle on an oracle FMI (run-length BWT + samples + C)."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.orc_fmi_free(self.h); self.h = None

    @staticmethod
    def from_text(text):
        text, tp = _u8(text)
        return FMI(lib().orc_fmi_from_text(tp, text.size))

    @staticmethod
    def from_symbols(symbols):
        symbols, sp = _u8(symbols)
        return FMI(lib().orc_fmi_from_symbols(sp, symbols.size))

    @staticmethod
    def from_runs(symbols, lengths):
        """From (symbol, length) pairs; adjacent pairs of one symbol coalesce (RunBuffer)."""
        symbols, sp = _u64(symbols); lengths, lp = _u64(lengths)
        return FMI(lib().orc_fmi_from_runs(sp, lp, symbols.size))

    @staticmethod
    def from_native(data, sequences, bases):
        data, dp = _u8(data)
        return FMI(lib().orc_fmi_from_native(dp, data.size, sequences, bases))

    def clone(self):
        return FMI(lib().orc_fmi_clone(self.h))

    bases = property(lambda s: int(lib().orc_fmi_bases(s.h)))
    sequences = property(lambda s: int(lib().orc_fmi_sequences(s.h)))
    nbytes = property(lambda s: int(lib().orc_fmi_bytes(s.h)))
    blocks = property(lambda s: int(lib().orc_fmi_blocks(s.h)))
    hash = property(lambda s: int(lib().orc_fmi_hash(s.h)))

    @property
    def C(self):
        out = np.zeros(SIGMA + 1, dtype=np.uint64)
        lib().orc_fmi_C(self.h, out.ctypes.data_as(p_u64))
        return out

    @property
    def data(self):
        out = np.zeros(self.nbytes, dtype=np.uint8)
        lib().orc_fmi_data(self.h, out.ctypes.data_as(p_u8))
        return out

    @property
    def symbols(self):
        out = np.zeros(self.bases, dtype=np.uint8)
        lib().orc_fmi_symbols(self.h, out.ctypes.data_as(p_u8))
        return out

    @property
    def character_counts(self):
        out = np.zeros(SIGMA, dtype=np.uint64)
        lib().orc_fmi_character_counts(self.h, out.ctypes.data_as(p_u64))
        return out

    @property
    def samples(self):
        """(block_end[blocks], cum[6][blocks + 1])"""
        nb = self.blocks
        be = np.zeros(nb, dtype=np.uint64); cum = np.zeros((SIGMA, nb + 1), dtype=np.uint64)
        lib().orc_fmi_samples(self.h, be.ctypes.data_as(p_u64), cum.ctypes.data_as(p_u64))
        return be, cum

    def rank(self, i, c): return int(lib().orc_rank(self.h, i, c))
    def select(self, i, c): return int(lib().orc_select(self.h, i, c))
    def at(self, i): return int(lib().orc_at(self.h, i))

    def inverse_select(self, i):
        r = u64(0); c = u64(0)
        lib().orc_inverse_select(self.h, i, C.byref(r), C.byref(c))
        return int(r.value), int(c.value)

    def ranks_at(self, i):
        out = np.zeros(SIGMA, dtype=np.uint64)
        lib().orc_ranks_at(self.h, i, out.ctypes.data_as(p_u64))
        return out

    def ranks_range(self, sp, ep):
        f = np.zeros(SIGMA, dtype=np.uint64); s = np.zeros(SIGMA, dtype=np.uint64)
        lib().orc_ranks_range(self.h, sp, ep, f.ctypes.data_as(p_u64), s.ctypes.data_as(p_u64))
        return f, s

    def LF(self, i, c=None):
        if c is not None:
            return int(lib().orc_LF_c(self.h, i, c))
        n = u64(0); cc = u64(0)
        lib().orc_LF(self.h, i, C.byref(n), C.byref(cc))
        return int(n.value), int(cc.value)

    def find(self, pattern):
        pattern, pp = _u8(pattern)
        sp = u64(0); ep = u64(0)
        lib().orc_find(self.h, pp, pattern.size, C.byref(sp), C.byref(ep))
        return int(sp.value), int(ep.value)


def _params(threads=1, sequence_blocks=0, run_buffer_size=0, thread_buffer_size=0, merge_buffers=0):
    return OrcParams(run_buffer_size, thread_buffer_size, merge_buffers, threads, sequence_blocks)


def search(a, b, capacity=None, **kw):
    """Rank array of inserting b into a, as maximal (rank, count) runs; also branch stats.
    capacity: upper bound of the number of runs (default: one per position of b)."""
    cap = (b.bases + 1 if capacity is None else capacity)
    r = np.zeros(cap, dtype=np.uint64); c = np.zeros(cap, dtype=np.uint64)
    stats = np.zeros(3, dtype=np.uint64)
    p = _params(**kw)
    n = lib().orc_search(a.h, b.h, C.byref(p), r.ctypes.data_as(p_u64), c.ctypes.data_as(p_u64), cap,
                         stats.ctypes.data_as(p_u64))
    assert n <= cap, "orc_search: %d runs, capacity %d" % (n, cap)
    return r[:n].copy(), c[:n].copy(), stats


def merge(a, b, **kw):
    """FMI::FMI(a, b, params). Consumes a and b. Returns (merged FMI, (search_s, interleave_s))."""
    secs = (C.c_double * 2)()
    p = _params(**kw)
    h = lib().orc_merge(a.h, b.h, C.byref(p), secs)
    return FMI(h), (secs[0], secs[1])


TIMING_FIELDS = ("dfs", "sort_encode", "merge_thread", "lock_wait", "merge_global", "write", "flush", "threads_wall", "threads", "sequence_blocks")


def merge_timed(a, b, **kw):
    """merge() with the time accounting of the search phase: returns (merged FMI, (search_s, interleave_s), {field: seconds summed over
    threads (flush: one thread's wall clock), threads, sequence_blocks})."""
    secs = (C.c_double * 2)()
    timing = (C.c_double * len(TIMING_FIELDS))()
    p = _params(**kw)
    h = lib().orc_merge_timed(a.h, b.h, C.byref(p), secs, timing)
    return FMI(h), (secs[0], secs[1]), {k: timing[i] for i, k in enumerate(TIMING_FIELDS)}


def interleave_symbols(a, b, ra):
    a, ap = _u8(a); b, bp = _u8(b); ra, rp = _u64(ra)
    out = np.zeros(a.size + b.size, dtype=np.uint8)
    lib().orc_interleave_symbols(ap, a.size, bp, b.size, rp, out.ctypes.data_as(p_u8))
    return out


def ra_from_runs(ranks, counts):
    """Expand (rank, count) runs to one rank per B position."""
    return np.repeat(np.asarray(ranks, dtype=np.uint64), np.asarray(counts, dtype=np.int64))
