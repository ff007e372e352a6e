"""ctypes binding of the C ABI in include/bwtm.h (the same stub a maintainer of another host
language would write).  There is no CPU fallback: if libbwtm.so is
missing or no GPU is usable, calls raise."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# BWTM_LIB selects another build of the same library (A/B measurements of kernel variants on one GPU box)
LIB_PATH = os.environ.get("BWTM_LIB") or os.path.join(HERE, "libbwtm.so")
SIGMA = 6

u64 = C.c_uint64
p_u8 = C.POINTER(C.c_uint8)
p_u64 = C.POINTER(C.c_uint64)
vp = C.c_void_p



class HostInput(C.Structure):
    """bwtm_host_input"""
    _fields_ = [("data", C.c_void_p), ("nbytes", u64), ("sequences", u64), ("bases", u64), ("C", p_u64)]


class HostOutput(C.Structure):
    """bwtm_host_output"""
    _fields_ = [("data", C.c_void_p), ("nbytes", u64), ("blocks", u64), ("sequences", u64), ("bases", u64),
                ("C", u64 * (SIGMA + 1)), ("block_end", C.c_void_p), ("cum", C.c_void_p),
                ("sample_width", C.c_int), ("fields", C.c_void_p), ("anchors", C.c_void_p),
                ("ms_upload", C.c_double), ("ms_search", C.c_double), ("ms_interleave", C.c_double),
                ("ms_encode_download", C.c_double), ("ms_samples", C.c_double), ("ms_total", C.c_double)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, u64)


class HostIndex(C.Structure):
    """bwtm_host_index"""
    _fields_ = [("data", C.c_void_p), ("nbytes", u64), ("blocks", u64), ("sequences", u64), ("bases", u64), ("C", u64 * (SIGMA + 1)), ("cum", C.c_void_p)]


class IndexHeader(C.Structure):
    """bwtm_index_header"""
    _fields_ = [("bases", u64), ("sequences", u64), ("C", u64 * (SIGMA + 1))]


class PartInfo(C.Structure):
    """bwtm_part_info"""
    _fields_ = [("steps", u64), ("node_levels", u64), ("elements", u64), ("largest", u64), ("pulled_bytes", u64), ("boundary_bytes", u64),
                ("record_bytes", u64), ("bitvector_bytes", u64), ("ms_search", C.c_double), ("ms_search_wait", C.c_double), ("ms_finish", C.c_double)]

# Every symbol include/bwtm.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bwtm_init", C.c_int, [C.c_int]),
    ("bwtm_context_create", C.c_int, [C.c_int, C.POINTER(vp)]),
    ("bwtm_context_make_current", C.c_int, [vp]),
    ("bwtm_context_destroy", None, [vp]),
    ("bwtm_device_bytes_peak", u64, [C.c_int]),
    ("bwtm_host_alloc", C.c_int, [u64, C.POINTER(vp)]),
    ("bwtm_host_free", None, [vp]),
    ("bwtm_ra_download_runs", C.c_int, [vp, p_u64, p_u64, u64, p_u64]),
    ("bwtm_ra_or_from", C.c_int, [vp, vp, u64]),
    ("bwtm_merge_consume", C.c_int, [vp, vp, C.POINTER(vp)]),
    ("bwtm_merged_records", u64, [vp, vp]),
    ("bwtm_slice_bounds", C.c_int, [u64, C.c_int, C.c_int, p_u64, p_u64]),
    ("bwtm_slice_bounds_equal", C.c_int, [u64, C.c_int, C.c_int, p_u64, p_u64, p_u64]),
    ("bwtm_ra_range_counts", C.c_int, [vp, u64, u64, p_u64, p_u64, p_u64]),
    ("bwtm_ra_finalize_range", C.c_int, [vp, u64, u64, u64, u64, p_u64, p_u64]),
    ("bwtm_interleave_range", C.c_int, [vp, vp, vp, u64, u64, C.POINTER(vp)]),
    ("bwtm_slice_free", None, [vp]),
    ("bwtm_slice_lasthead", C.c_int, [vp, p_u64]),
    ("bwtm_slice_size_table", C.c_int, [vp, u64, p_u64]),
    ("bwtm_fold_offsets", C.c_int, [p_u64, C.c_int, p_u64]),
    ("bwtm_slice_encode", C.c_int, [vp, u64]),
    ("bwtm_slice_byte_first", u64, [vp]),
    ("bwtm_slice_bytes", u64, [vp]),
    ("bwtm_slice_block_first", u64, [vp]),
    ("bwtm_slice_blocks", u64, [vp]),
    ("bwtm_slice_first_block_start", C.c_int, [vp, p_u64]),
    ("bwtm_slice_download_data", C.c_int, [vp, p_u8, u64]),
    ("bwtm_slice_download_samples", C.c_int, [vp, u64, p_u64, p_u64]),
    ("bwtm_slice_extract", C.c_int, [vp, u64, u64, p_u8]),
    ("bwtm_merge_host", C.c_int, [C.POINTER(HostInput), C.POINTER(HostInput), ALLOC_FN, vp, C.c_int, C.POINTER(HostOutput), C.POINTER(vp)]),
    ("bwtm_merge_host_chained", C.c_int, [vp, C.POINTER(HostInput), ALLOC_FN, vp, C.c_int, C.POINTER(HostOutput), C.POINTER(vp)]),
    ("bwtm_merge_host_pipelined", C.c_int, [vp, C.POINTER(HostInput), C.POINTER(HostInput), vp, C.POINTER(HostInput), C.POINTER(vp), ALLOC_FN, vp, C.c_int,
                                            C.POINTER(HostOutput), C.POINTER(vp)]),
    ("bwtm_upload_begin", C.c_int, [C.POINTER(HostInput), C.POINTER(vp)]),
    ("bwtm_upload_finish", C.c_int, [vp, C.POINTER(vp)]),
    ("bwtm_upload_free", None, [vp]),
    ("bwtm_last_error", C.c_char_p, []),
    ("bwtm_synchronize", C.c_int, []),
    ("bwtm_trim", C.c_int, []),
    ("bwtm_tune", C.c_int, [C.c_char_p, C.c_longlong]),
    ("bwtm_index_upload", C.c_int, [p_u8, u64, u64, u64, p_u64, C.POINTER(vp)]),
    ("bwtm_index_from_device", C.c_int, [vp, u64, u64, u64, p_u64, C.POINTER(vp)]),
    ("bwtm_index_from_device_borrowed", C.c_int, [vp, u64, u64, u64, p_u64, C.POINTER(vp)]),
    ("bwtm_index_from_symbols_device", C.c_int, [vp, u64, C.POINTER(vp)]),
    ("bwtm_index_free", None, [vp]),
    ("bwtm_index_bases", u64, [vp]),
    ("bwtm_index_sequences", u64, [vp]),
    ("bwtm_index_bytes", u64, [vp]),
    ("bwtm_index_blocks", u64, [vp]),
    ("bwtm_index_C", None, [vp, p_u64]),
    ("bwtm_index_encode", C.c_int, [vp]),
    ("bwtm_index_drop_native", C.c_int, [vp]),
    ("bwtm_index_device_data", C.c_int, [vp, C.POINTER(vp), p_u64]),
    ("bwtm_index_download_data", C.c_int, [vp, p_u8, u64]),
    ("bwtm_index_download_samples", C.c_int, [vp, p_u64, p_u64]),
    ("bwtm_index_samples_width", C.c_int, [vp, C.POINTER(C.c_int)]),
    ("bwtm_index_download_samples_compact", C.c_int, [vp, C.c_int, vp, p_u64]),
    ("bwtm_rank_batch", C.c_int, [vp, p_u64, p_u8, u64, p_u64]),
    ("bwtm_inverse_select_batch", C.c_int, [vp, p_u64, u64, p_u64, p_u8]),
    ("bwtm_find_batch", C.c_int, [vp, p_u8, p_u64, u64, p_u64, p_u64]),
    ("bwtm_extract", C.c_int, [vp, u64, u64, p_u8]),
    ("bwtm_ra_create", C.c_int, [vp, vp, C.POINTER(vp)]),
    ("bwtm_ra_buffer_bytes", u64, [vp, vp]),
    ("bwtm_ra_create_on", C.c_int, [vp, vp, vp, u64, C.POINTER(vp)]),
    ("bwtm_ra_free", None, [vp]),
    ("bwtm_search", C.c_int, [vp, vp, u64, u64, vp]),
    ("bwtm_ra_device_buffer", C.c_int, [vp, C.POINTER(vp), p_u64]),
    ("bwtm_ra_finalize", C.c_int, [vp]),
    ("bwtm_ra_subset_check", C.c_int, [vp, vp, p_u64, p_u64]),
    ("bwtm_ra_values", u64, [vp]),
    ("bwtm_ra_download", C.c_int, [vp, p_u64, u64]),
    ("bwtm_ra_download_bits", C.c_int, [vp, p_u64, u64]),
    ("bwtm_interleave", C.c_int, [vp, vp, vp, C.POINTER(vp)]),
    ("bwtm_merge", C.c_int, [vp, vp, C.POINTER(vp)]),
    ("bwtm_builder_create", C.c_int, [u64, C.POINTER(vp)]),
    ("bwtm_builder_add", C.c_int, [vp, vp, u64, C.c_uint32, u64, vp, C.c_int]),
    ("bwtm_builder_reads", u64, [vp]),
    ("bwtm_builder_finish", C.c_int, [vp, C.POINTER(vp)]),
    ("bwtm_builder_free", None, [vp]),
    ("bwtm_pool_stats", C.c_int, [vp]),
    ("bwtm_group_create", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]),
    ("bwtm_group_free", None, [vp]),
    ("bwtm_group_part", C.c_int, [vp]),
    ("bwtm_group_parts", C.c_int, [vp]),
    ("bwtm_group_barrier", C.c_int, [vp]),
    ("bwtm_group_allgather", C.c_int, [vp, vp, u64, vp]),
    ("bwtm_group_abort", None, [vp]),
    ("bwtm_partition_cuts_host", C.c_int, [C.POINTER(HostIndex), C.POINTER(HostIndex), C.c_int, C.c_int, p_u64, p_u64]),
    ("bwtm_window_blocks", C.c_int, [C.POINTER(HostIndex), u64, u64, p_u64, p_u64, p_u64, p_u64]),
    ("bwtm_index_upload_window", C.c_int, [vp, u64, u64, p_u64, u64, u64, p_u64, C.c_int, C.POINTER(vp)]),
    ("bwtm_index_record_bytes", u64, [vp]),
    ("bwtm_ra_create_range", C.c_int, [vp, vp, u64, u64, C.POINTER(vp)]),
    ("bwtm_ra_bytes", u64, [vp]),
    ("bwtm_part_create", C.c_int, [vp, C.POINTER(IndexHeader), C.POINTER(IndexHeader), p_u64, p_u64, C.POINTER(vp)]),
    ("bwtm_part_free", None, [vp]),
    ("bwtm_part_window", C.c_int, [vp, C.c_int, p_u64, p_u64]),
    ("bwtm_part_upload", C.c_int, [vp, C.c_int, vp, u64, u64, p_u64, C.c_int]),
    ("bwtm_part_search", C.c_int, [vp]),
    ("bwtm_part_finish", C.c_int, [vp, C.POINTER(vp), p_u64, p_u64, p_u64]),
    ("bwtm_part_stats", C.c_int, [vp, C.POINTER(PartInfo)]),
    ("bwtm_profile_enable", C.c_int, [C.c_int]),
    ("bwtm_profile_only", C.c_int, [C.c_char_p]),
    ("bwtm_profile_reset", C.c_int, []),
    ("bwtm_profile_read", C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_double), p_u64, C.c_int]),
]


class BwtmError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libbwtm.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BwtmError("HIP extension %s is missing: run `python __graft_entry__.py build` "
                            "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Processes that use both
        # (bench.py, torch.distributed) must share ONE HIP runtime, so torch is loaded first and
        # libbwtm.so's NEEDED libamdhip64.so.7 then resolves to the already loaded copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)       # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise BwtmError("bwtm error %d: %s" % (rc, lib().bwtm_last_error().decode()))


def init(device=0):
    check(lib().bwtm_init(device))


def synchronize():
    check(lib().bwtm_synchronize())


def tune(key, value):
    check(lib().bwtm_tune(key.encode(), int(value)))


def trim():
    check(lib().bwtm_trim())


def device_bytes_peak(reset=False):
    """Peak bytes of device memory the library's context has held since the last reset."""
    return int(lib().bwtm_device_bytes_peak(1 if reset else 0))


class Context:
    """An additional library context (device + streams + memory pool); make_current() binds the calling thread."""

    def __init__(self, device=0):
        out = vp()
        check(lib().bwtm_context_create(device, C.byref(out)))
        self.h = out

    def make_current(self):
        check(lib().bwtm_context_make_current(self.h))

    def destroy(self):
        if self.h:
            lib().bwtm_context_destroy(self.h)
            self.h = None


def make_default_current():
    check(lib().bwtm_context_make_current(None))


class HostBuffer:
    """Page-locked host memory (bwtm_host_alloc) viewed as a numpy array."""

    def __init__(self, nbytes, dtype=np.uint8, ptr=None):
        self.nbytes = int(nbytes)
        if ptr is None:
            out = vp()
            check(lib().bwtm_host_alloc(self.nbytes, C.byref(out)))
            ptr = out.value
        self.ptr = ptr
        raw = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(raw, dtype=np.uint8)[: self.nbytes].view(dtype)

    def free(self):
        if self.ptr:
            self.array = None
            lib().bwtm_host_free(vp(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostMerge:
    """Result of merge_host: page-locked data / block_end / cum arrays + the phase times of the call."""

    def __init__(self):
        self.buffers = {}
        self.out = HostOutput()
        self.keep = None

    def free(self):
        for b in self.buffers.values():
            b.free()
        self.buffers = {}
        if self.keep is not None:
            self.keep.free()
            self.keep = None

    def expanded_samples(self):
        """(block_end, cum) whatever form the call returned."""
        if self.out.sample_width == 8:
            return self.block_end, self.cum
        return expand_samples(self.out.sample_width, self.fields, self.anchors, self.out.blocks, self.out.bases)

    fields = property(lambda s: s.buffers[3].array[: SIGMA * s.out.blocks * s.out.sample_width].view(FIELD_DTYPES[s.out.sample_width]).reshape(SIGMA, s.out.blocks))
    anchors = property(lambda s: s.buffers[4].array.view(np.uint64)[: SIGMA * ((s.out.blocks + 63) // 64)].reshape(SIGMA, (s.out.blocks + 63) // 64))
    data = property(lambda s: s.buffers[0].array[: s.out.nbytes])
    block_end = property(lambda s: s.buffers[1].array.view(np.uint64)[: s.out.blocks])
    cum = property(lambda s: s.buffers[2].array.view(np.uint64)[: SIGMA * (s.out.blocks + 1)].reshape(SIGMA, s.out.blocks + 1))
    C = property(lambda s: np.array(list(s.out.C), dtype=np.uint64))
    times = property(lambda s: {k: getattr(s.out, k) for k in ("ms_upload", "ms_search", "ms_interleave", "ms_encode_download", "ms_samples", "ms_total")})


FIELD_DTYPES = {1: np.uint8, 2: np.uint16, 4: np.uint32}


def expand_samples(width, fields, anchors, blocks, bases):
    """Compact samples -> (block_end[blocks], cum[6][blocks + 1]) as bwtm_index_download_samples returns them."""
    f = fields.astype(np.uint64)
    k = np.arange(blocks, dtype=np.int64)
    group = k // 64
    excl = np.cumsum(f, axis=1) - f                                  # exclusive prefix over all blocks ...
    first = excl[:, (group * 64)]                                    # ... minus the prefix at the block's anchor
    at = anchors[:, group] + (excl - first)                          # row 0: block starts; rows 1..5: counts before the block
    starts = at[0]
    block_end = np.concatenate([starts[1:], np.array([bases], dtype=np.uint64)]) - np.uint64(1) if blocks > 0 else np.zeros(0, dtype=np.uint64)
    cum = np.zeros((SIGMA, blocks + 1), dtype=np.uint64)
    if blocks > 0:
        cum[1:, :blocks] = at[1:]
        cum[1:, blocks] = at[1:, -1] + f[1:, -1]
        cum[0, :blocks] = starts - at[1:].sum(axis=0)
        cum[0, blocks] = np.uint64(bases) - cum[1:, blocks].sum()
    return block_end, cum


def _host_input(data, sequences, bases):
    """data: numpy uint8 array (ideally the .array of a HostBuffer)."""
    assert data.dtype == np.uint8 and data.flags["C_CONTIGUOUS"]
    return HostInput(data.ctypes.data, data.size, sequences, bases, None)


def merge_host(a, b, samples=True, keep=False, chained=None, buffers=None):
    """FMI::FMI(a, b) from host-resident inputs to a host-resident result (bwtm_merge_host).
    a, b: (data uint8 array, sequences, bases); chained: a device Index (consumed) instead of a.
    samples: False / 0 = none, True / 1 = block_end + cum, 2 = the compact form (fields + anchors).
    buffers: a dict that keeps the page-locked output buffers between calls (they are reused when large enough)."""
    res = HostMerge()
    if buffers is not None:
        res.buffers = buffers

    def alloc(user, what, nbytes):
        buf = res.buffers.get(what)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = HostBuffer(nbytes)
            res.buffers[what] = buf
        return buf.ptr

    cb = ALLOC_FN(alloc)
    hb = _host_input(*b)
    kp = vp()
    if chained is not None:
        h, chained.h = chained.h, None                     # consumed by the call
        rc = lib().bwtm_merge_host_chained(h, C.byref(hb), cb, None, int(samples), C.byref(res.out), C.byref(kp) if keep else None)
    else:
        ha = _host_input(*a)
        rc = lib().bwtm_merge_host(C.byref(ha), C.byref(hb), cb, None, int(samples), C.byref(res.out), C.byref(kp) if keep else None)
    if rc != 0:
        if buffers is None:
            res.free()
        check(rc)
    if keep:
        res.keep = Index(kp)
    return res


RESULT_ON_DEVICE = -1


class Upload:
    """An input on its way to the device (bwtm_upload): keeps the host array alive until it is consumed."""

    def __init__(self, handle, source):
        self.h, self.source = handle, source

    def finish(self):
        """-> Index (bwtm_upload_finish: waits for the copies, decodes, validates, transcodes)."""
        h, self.h = self.h, None
        out = vp()
        check(lib().bwtm_upload_finish(h, C.byref(out)))
        self.source = None
        return Index(out)

    def free(self):
        if self.h is not None:
            lib().bwtm_upload_free(self.h)
            self.h = None
        self.source = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def upload_begin(data, sequences, bases):
    hi = _host_input(data, sequences, bases)
    out = vp()
    check(lib().bwtm_upload_begin(C.byref(hi), C.byref(out)))
    return Upload(out, data)


def merge_host_pipelined(a=None, b=None, chained=None, pending=None, next=None, samples=True, keep=False, buffers=None):
    """bwtm_merge_host_pipelined: a = (data, sequences, bases) or chained = device Index (consumed); b likewise or pending = Upload
    (consumed); next = (data, sequences, bases) of the FOLLOWING merge's input, whose copies run under this merge's search.
    samples = RESULT_ON_DEVICE: nothing is encoded or downloaded (needs keep).  Returns (HostMerge, Upload or None)."""
    res = HostMerge()
    if buffers is not None:
        res.buffers = buffers

    def alloc(user, what, nbytes):
        buf = res.buffers.get(what)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = HostBuffer(nbytes)
            res.buffers[what] = buf
        return buf.ptr

    cb = ALLOC_FN(alloc)
    ha = _host_input(*a) if a is not None else None
    hb = _host_input(*b) if b is not None else None
    hn = _host_input(*next) if next is not None else None
    dev = None
    if chained is not None:
        dev, chained.h = chained.h, None
    pend = None
    if pending is not None:
        pend, pending.h = pending.h, None
    kp, np_ = vp(), vp()
    rc = lib().bwtm_merge_host_pipelined(dev, C.byref(ha) if ha else None, C.byref(hb) if hb else None, pend, C.byref(hn) if hn else None,
                                         C.byref(np_) if hn else None, cb, None, int(samples), C.byref(res.out), C.byref(kp) if keep else None)
    if pending is not None:
        pending.source = None
    if rc != 0:
        if buffers is None:
            res.free()
        check(rc)
    if keep:
        res.keep = Index(kp)
    return res, (Upload(np_, next[0]) if hn else None)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(p_u8)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(p_u64)


class Index:
    """Device-resident FM-index (handle on bwtm_index)."""

    def __init__(self, handle):
        self.h = vp(handle) if not isinstance(handle, vp) else handle

    def free(self):
        if self.h:
            lib().bwtm_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @staticmethod
    def upload(data, sequences, bases, C_array=None):
        data, dp = _u8(data)
        out = vp()
        cp = None
        if C_array is not None:
            C_array, cp = _u64(C_array)
        check(lib().bwtm_index_upload(dp, data.size, sequences, bases, cp, C.byref(out)))
        return Index(out)

    @staticmethod
    def from_device(ptr, nbytes, sequences, bases, C_array=None, borrow=False):
        """borrow=True: the index reads the caller's buffer in place (it must outlive the index)."""
        out = vp()
        cp = None
        if C_array is not None:
            C_array, cp = _u64(C_array)
        f = lib().bwtm_index_from_device_borrowed if borrow else lib().bwtm_index_from_device
        check(f(vp(ptr), nbytes, sequences, bases, cp, C.byref(out)))
        return Index(out)

    @staticmethod
    def from_symbols_device(ptr, bases):
        out = vp()
        check(lib().bwtm_index_from_symbols_device(vp(ptr), bases, C.byref(out)))
        return Index(out)

    bases = property(lambda s: int(lib().bwtm_index_bases(s.h)))
    sequences = property(lambda s: int(lib().bwtm_index_sequences(s.h)))
    nbytes = property(lambda s: int(lib().bwtm_index_bytes(s.h)))
    blocks = property(lambda s: int(lib().bwtm_index_blocks(s.h)))

    @property
    def C(self):
        out = np.zeros(SIGMA + 1, dtype=np.uint64)
        lib().bwtm_index_C(self.h, out.ctypes.data_as(p_u64))
        return out

    def encode(self):
        check(lib().bwtm_index_encode(self.h))
        return self

    def drop_native(self):
        check(lib().bwtm_index_drop_native(self.h))
        return self

    def device_data(self):
        """(device pointer, nbytes) of the native byte stream."""
        ptr = vp()
        n = u64(0)
        check(lib().bwtm_index_device_data(self.h, C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    total_nbytes = nbytes            # a slice of a sharded result reports the size of the whole stream here

    def data(self):
        out = np.zeros(self.nbytes, dtype=np.uint8)
        check(lib().bwtm_index_download_data(self.h, out.ctypes.data_as(p_u8), out.size))
        return out

    def download_into(self, out):
        """Native bytes into a caller's uint8 array (e.g. page-locked memory)."""
        assert out.dtype == np.uint8 and out.size >= self.nbytes
        check(lib().bwtm_index_download_data(self.h, out.ctypes.data_as(p_u8), out.size))
        return out

    def samples(self):
        nb = self.blocks
        be = np.zeros(nb, dtype=np.uint64)
        cum = np.zeros((SIGMA, nb + 1), dtype=np.uint64)
        check(lib().bwtm_index_download_samples(self.h, be.ctypes.data_as(p_u64), cum.ctypes.data_as(p_u64)))
        return be, cum

    def samples_compact(self, width=None):
        """(width, fields [6][blocks] of uint8 / uint16 / uint32, anchors [6][ceil(blocks / 64)] uint64); see include/bwtm.h."""
        if width is None:
            w = C.c_int(0)
            check(lib().bwtm_index_samples_width(self.h, C.byref(w)))
            width = w.value
        if width == 8:
            return 8, None, None
        nb = self.blocks
        fields = np.zeros((SIGMA, nb), dtype=FIELD_DTYPES[width])
        anchors = np.zeros((SIGMA, (nb + 63) // 64), dtype=np.uint64)
        check(lib().bwtm_index_download_samples_compact(self.h, width, fields.ctypes.data_as(vp), anchors.ctypes.data_as(p_u64)))
        return width, fields, anchors

    def rank(self, positions, comps):
        positions, pp = _u64(positions)
        comps, cp = _u8(comps)
        out = np.zeros(positions.size, dtype=np.uint64)
        check(lib().bwtm_rank_batch(self.h, pp, cp, positions.size, out.ctypes.data_as(p_u64)))
        return out

    def inverse_select(self, positions):
        positions, pp = _u64(positions)
        r = np.zeros(positions.size, dtype=np.uint64)
        c = np.zeros(positions.size, dtype=np.uint8)
        check(lib().bwtm_inverse_select_batch(self.h, pp, positions.size, r.ctypes.data_as(p_u64), c.ctypes.data_as(p_u8)))
        return r, c

    def find(self, patterns):
        """Backward search of a list of patterns (sequences of comp values); returns (sp, ep) arrays."""
        lens = np.array([len(p) for p in patterns], dtype=np.uint64)
        offsets = np.zeros(len(patterns) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        text = np.concatenate([np.asarray(p, dtype=np.uint8) for p in patterns] + [np.zeros(0, dtype=np.uint8)])
        text, tp = _u8(text)
        sp = np.zeros(len(patterns), dtype=np.uint64)
        ep = np.zeros(len(patterns), dtype=np.uint64)
        check(lib().bwtm_find_batch(self.h, tp, offsets.ctypes.data_as(p_u64), len(patterns), sp.ctypes.data_as(p_u64), ep.ctypes.data_as(p_u64)))
        return sp, ep

    def extract(self, first, count):
        out = np.zeros(count, dtype=np.uint8)
        check(lib().bwtm_extract(self.h, first, count, out.ctypes.data_as(p_u8)))
        return out


class RankArray:
    """Device-resident rank array (handle on bwtm_ra)."""

    def __init__(self, a, b, device_buffer=None, nbytes=0):
        out = vp()
        if device_buffer is None:
            check(lib().bwtm_ra_create(a.h, b.h, C.byref(out)))
        else:
            check(lib().bwtm_ra_create_on(a.h, b.h, vp(device_buffer), nbytes, C.byref(out)))
        self.h = out
        self.n_out = a.bases + b.bases
        self.nb = b.bases

    def free(self):
        if self.h:
            lib().bwtm_ra_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def search(self, a, b, seq_first, seq_last):
        check(lib().bwtm_search(a.h, b.h, seq_first, seq_last, self.h))

    def device_buffer(self):
        ptr = vp()
        n = u64(0)
        check(lib().bwtm_ra_device_buffer(self.h, C.byref(ptr), C.byref(n)))
        return int(ptr.value), int(n.value)

    def or_from(self, other):
        """self |= other's bits (another rank array for the same inputs on the same device)."""
        ptr, n = other.device_buffer()
        check(lib().bwtm_ra_or_from(self.h, vp(ptr), n))

    def subset_check(self, whole):
        """(set bits of this array, 64-bit words in which it has a bit that `whole` lacks)."""
        ones, bad = u64(0), u64(0)
        check(lib().bwtm_ra_subset_check(self.h, whole.h, C.byref(ones), C.byref(bad)))
        return int(ones.value), int(bad.value)

    def finalize(self):
        check(lib().bwtm_ra_finalize(self.h))
        return self

    def range_counts(self, rec_first, rec_last):
        """Output-range finalize, step 1: (set bits of the range, per-super local counts [nsup], the words of the range's last chunk [128])."""
        nsup = (self.n_out >> 25) + 1
        ones = u64(0)
        local = np.zeros(nsup, dtype=np.uint64)
        tail = np.zeros(128, dtype=np.uint64)
        check(lib().bwtm_ra_range_counts(self.h, rec_first, rec_last, C.byref(ones), local.ctypes.data_as(p_u64), tail.ctypes.data_as(p_u64)))
        return int(ones.value), local, tail

    def finalize_range(self, rec_first, rec_last, ones_before, ones_total, super_boff, halo_words=None):
        """Output-range finalize, step 2 (include/bwtm.h: bwtm_ra_finalize_range)."""
        super_boff = np.ascontiguousarray(super_boff, dtype=np.uint64)
        assert super_boff.size == (self.n_out >> 25) + 1
        halo = None if halo_words is None else np.ascontiguousarray(halo_words, dtype=np.uint64)
        assert halo is None or halo.size == 128
        check(lib().bwtm_ra_finalize_range(self.h, rec_first, rec_last, ones_before, ones_total, super_boff.ctypes.data_as(p_u64),
                                           None if halo is None else halo.ctypes.data_as(p_u64)))
        return self

    values = property(lambda s: int(lib().bwtm_ra_values(s.h)))

    def download(self):
        out = np.zeros(self.nb, dtype=np.uint64)
        check(lib().bwtm_ra_download(self.h, out.ctypes.data_as(p_u64), out.size))
        return out

    def runs(self):
        """The rank array as maximal (rank, count) runs (the reference's own form)."""
        n = u64(0)
        check(lib().bwtm_ra_download_runs(self.h, None, None, 0, C.byref(n)))
        ranks = np.zeros(n.value, dtype=np.uint64)
        counts = np.zeros(n.value, dtype=np.uint64)
        check(lib().bwtm_ra_download_runs(self.h, ranks.ctypes.data_as(p_u64), counts.ctypes.data_as(p_u64), n.value, C.byref(n)))
        return ranks, counts

    def bits(self):
        words = (self.n_out + 63) // 64
        out = np.zeros(words, dtype=np.uint64)
        check(lib().bwtm_ra_download_bits(self.h, out.ctypes.data_as(p_u64), out.size))
        return out


class Slice:
    """One GPU's share of a merged index: the output records [rec_first, rec_last) (handle on bwtm_slice)."""

    def __init__(self, a, b, ra, rec_first, rec_last):
        out = vp()
        check(lib().bwtm_interleave_range(a.h, b.h, ra.h, rec_first, rec_last, C.byref(out)))
        self.h = out
        self.total_nbytes = 0

    def free(self):
        if self.h:
            lib().bwtm_slice_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def lasthead(self):
        v = u64(0)
        check(lib().bwtm_slice_lasthead(self.h, C.byref(v)))
        return int(v.value)

    def size_table(self, heads_before):
        t = np.zeros(64, dtype=np.uint64)
        check(lib().bwtm_slice_size_table(self.h, int(heads_before), t.ctypes.data_as(p_u64)))
        return t

    def encode(self, byte_offset):
        check(lib().bwtm_slice_encode(self.h, int(byte_offset)))
        return self

    byte_first = property(lambda s: int(lib().bwtm_slice_byte_first(s.h)))
    nbytes = property(lambda s: int(lib().bwtm_slice_bytes(s.h)))
    block_first = property(lambda s: int(lib().bwtm_slice_block_first(s.h)))
    blocks = property(lambda s: int(lib().bwtm_slice_blocks(s.h)))

    def first_block_start(self):
        """Sequence position at which the slice's first block starts; None if it has no block."""
        v = u64(0)
        check(lib().bwtm_slice_first_block_start(self.h, C.byref(v)))
        return None if v.value == (1 << 64) - 1 else int(v.value)

    def data(self):
        out = np.zeros(self.nbytes, dtype=np.uint8)
        check(lib().bwtm_slice_download_data(self.h, out.ctypes.data_as(p_u8), out.size))
        return out

    def data_into(self, out):
        """The slice's bytes into a caller-provided uint8 array (e.g. page-locked: HostBuffer.array)."""
        check(lib().bwtm_slice_download_data(self.h, out.ctypes.data_as(p_u8), out.size))
        return out[: self.nbytes]

    def samples(self, next_block_start):
        nb = self.blocks
        be = np.zeros(nb, dtype=np.uint64)
        cum = np.zeros((SIGMA, nb), dtype=np.uint64)
        check(lib().bwtm_slice_download_samples(self.h, int(next_block_start), be.ctypes.data_as(p_u64), cum.ctypes.data_as(p_u64)))
        return be, cum

    def extract(self, first, count):
        out = np.zeros(count, dtype=np.uint8)
        check(lib().bwtm_slice_extract(self.h, first, count, out.ctypes.data_as(p_u8)))
        return out


class Group:
    """The parts of a merge over partitioned records (bwtm_group): `name` = a POSIX shared-memory name unique to the group (None for one part)."""

    def __init__(self, name, part, parts):
        out = vp()
        check(lib().bwtm_group_create(name.encode() if name else None, int(part), int(parts), C.byref(out)))
        self.h = out
        self.part, self.parts = int(part), int(parts)

    def free(self):
        if self.h:
            lib().bwtm_group_free(self.h)
            self.h = None

    def barrier(self):
        check(lib().bwtm_group_barrier(self.h))

    def allgather(self, mine):
        """mine: a numpy array; returns [parts, ...] of the same dtype."""
        mine = np.ascontiguousarray(mine)
        out = np.zeros((self.parts,) + mine.shape, dtype=mine.dtype)
        check(lib().bwtm_group_allgather(self.h, mine.ctypes.data_as(vp), mine.nbytes, out.ctypes.data_as(vp)))
        return out

    def abort(self):
        if self.h:
            lib().bwtm_group_abort(self.h)


def host_index(data, cum, sequences, bases):
    """bwtm_host_index over numpy arrays (kept alive by the returned object): data = the native bytes, cum[6][blocks + 1] = the samples."""
    assert data.dtype == np.uint8 and data.flags["C_CONTIGUOUS"]
    cum = np.ascontiguousarray(cum, dtype=np.uint64)
    blocks = cum.shape[1] - 1
    x = HostIndex()
    x.data = data.ctypes.data; x.nbytes = data.size; x.blocks = blocks; x.sequences = int(sequences); x.bases = int(bases)
    totals = [int(cum[c][blocks]) for c in range(SIGMA)]
    for c in range(SIGMA + 1):
        x.C[c] = sum(totals[:c])
    x.cum = cum.ctypes.data
    x._keep = (data, cum)
    return x


def partition_cuts_host(a, b, parts, kmer=0):
    """(cut_a, cut_b): parts + 1 ranks each (bwtm_partition_cuts_host); a, b = host_index() objects."""
    ca = np.zeros(parts + 1, dtype=np.uint64); cb = np.zeros(parts + 1, dtype=np.uint64)
    check(lib().bwtm_partition_cuts_host(C.byref(a), C.byref(b), int(parts), int(kmer), ca.ctypes.data_as(p_u64), cb.ctypes.data_as(p_u64)))
    return [int(v) for v in ca], [int(v) for v in cb]


def window_blocks(x, pos_first, pos_last):
    """(block_first, block_end, first_position, counts_before[6]) of the blocks that cover the records of [pos_first, pos_last] (bwtm_window_blocks)."""
    b0, b1, fp = u64(0), u64(0), u64(0)
    before = (u64 * 6)()
    check(lib().bwtm_window_blocks(C.byref(x), int(pos_first), int(pos_last), C.byref(b0), C.byref(b1), C.byref(fp), before))
    return int(b0.value), int(b1.value), int(fp.value), before


class Part:
    """One part of a merge over partitioned records (bwtm_part): created in the calling thread's context."""

    def __init__(self, group, a, b, cut_a, cut_b):
        """a, b: host_index() objects (or anything with bases / sequences / C)."""
        ha, hb = IndexHeader(), IndexHeader()
        for h, x in ((ha, a), (hb, b)):
            h.bases = int(x.bases); h.sequences = int(x.sequences)
            for c in range(SIGMA + 1):
                h.C[c] = int(x.C[c])
        ca = (u64 * len(cut_a))(*[int(v) for v in cut_a]); cb = (u64 * len(cut_b))(*[int(v) for v in cut_b])
        out = vp()
        check(lib().bwtm_part_create(group.h, C.byref(ha), C.byref(hb), ca, cb, C.byref(out)))
        self.h = out
        self.group = group

    def free(self):
        if self.h:
            lib().bwtm_part_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def window(self, which):
        lo, hi = u64(0), u64(0)
        check(lib().bwtm_part_window(self.h, int(which), C.byref(lo), C.byref(hi)))
        return int(lo.value), int(hi.value)

    def share(self, which, x):
        """(byte offset, byte count, first_position, counts_before) of this part's share of host index x's stream."""
        lo, hi = self.window(which)
        b0, b1, fp, before = window_blocks(x, lo, hi)
        return b0 * 64, min(b1 * 64, int(x.nbytes)) - b0 * 64, fp, before

    def upload(self, which, ptr, nbytes, first_position, counts_before, on_device=False):
        check(lib().bwtm_part_upload(self.h, int(which), vp(ptr), int(nbytes), int(first_position), counts_before, 1 if on_device else 0))

    def upload_host(self, which, x):
        off, count, fp, before = self.share(which, x)
        self.upload(which, x.data + off, count, fp, before, on_device=False)

    def search(self):
        check(lib().bwtm_part_search(self.h))

    def finish(self):
        """The part's encoded Slice (total_nbytes, byte_offset, next_block_start set)."""
        out, off, total, nxt = vp(), u64(0), u64(0), u64(0)
        check(lib().bwtm_part_finish(self.h, C.byref(out), C.byref(off), C.byref(total), C.byref(nxt)))
        s = Slice.__new__(Slice)
        s.h = out; s.total_nbytes = int(total.value); s.byte_offset = int(off.value); s.next_block_start = int(nxt.value)
        return s

    def stats(self):
        info = PartInfo()
        check(lib().bwtm_part_stats(self.h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in PartInfo._fields_}


def merged_records(a, b):
    return int(lib().bwtm_merged_records(a.h, b.h))


def slice_bounds(nrecs, parts, part):
    f, l = u64(0), u64(0)
    check(lib().bwtm_slice_bounds(nrecs, parts, part, C.byref(f), C.byref(l)))
    return int(f.value), int(l.value)


def slice_bounds_equal(nrecs, parts, part):
    """(rec_first, rec_last, bytes of the bitvector per range) of EQUAL output ranges (what a reduce-scatter wants)."""
    f, l, sb = u64(0), u64(0), u64(0)
    check(lib().bwtm_slice_bounds_equal(nrecs, parts, part, C.byref(f), C.byref(l), C.byref(sb)))
    return int(f.value), int(l.value), int(sb.value)


def fold_offsets(tables):
    """tables: [parts, 64] uint64 -> offsets [parts + 1] (byte offset of every slice, then the size of the stream)."""
    tables = np.ascontiguousarray(tables, dtype=np.uint64)
    out = np.zeros(tables.shape[0] + 1, dtype=np.uint64)
    check(lib().bwtm_fold_offsets(tables.ctypes.data_as(p_u64), tables.shape[0], out.ctypes.data_as(p_u64)))
    return out


def ra_buffer_bytes(a, b):
    return int(lib().bwtm_ra_buffer_bytes(a.h, b.h))


def interleave(a, b, ra):
    out = vp()
    check(lib().bwtm_interleave(a.h, b.h, ra.h, C.byref(out)))
    return Index(out)


def merge(a, b):
    """FMI::FMI(a, b): search + finalize + interleave + encode + samples on the device."""
    out = vp()
    check(lib().bwtm_merge(a.h, b.h, C.byref(out)))
    return Index(out)


def merge_consume(a, b):
    """The same with the reference's ownership: a and b are destroyed, their memory released progressively."""
    out = vp()
    ha, hb = a.h, b.h
    a.h = None; b.h = None
    check(lib().bwtm_merge_consume(ha, hb, C.byref(out)))
    return Index(out)


class Builder:
    """Reads -> index on the GPU (bwtm_builder_*): leaves by suffix sort, grown by the merger itself."""

    def __init__(self, leaf_reads=0):
        out = vp()
        check(lib().bwtm_builder_create(leaf_reads, C.byref(out)))
        self.h = out

    def add(self, reads, lengths=None):
        """reads: [n, width] uint8 numpy array of comp values 1..5 (host); lengths: optional uint32 [n]."""
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        n, width = reads.shape
        lp = None
        if lengths is not None:
            lengths = np.ascontiguousarray(lengths, dtype=np.uint32)
            assert lengths.shape == (n,)
            lp = vp(lengths.ctypes.data)
        check(lib().bwtm_builder_add(self.h, vp(reads.ctypes.data), n, width, width, lp, 0))

    def add_device(self, ptr, nreads, width, stride=None, lengths_ptr=None):
        """The same for rows already on the device (lengths_ptr: device uint32 [nreads] or None)."""
        check(lib().bwtm_builder_add(self.h, vp(ptr), nreads, width, width if stride is None else stride,
                                     vp(lengths_ptr) if lengths_ptr else None, 1))

    reads = property(lambda s: int(lib().bwtm_builder_reads(s.h)))

    def finish(self):
        out = vp()
        h = self.h
        self.h = None
        check(lib().bwtm_builder_finish(h, C.byref(out)))
        return Index(out)

    def free(self):
        if self.h:
            lib().bwtm_builder_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PoolInfo(C.Structure):
    _fields_ = [("held_bytes", u64), ("cached_bytes", u64), ("peak_bytes", u64), ("mapped_blocks", u64), ("address_bytes_reserved", u64),
                ("address_space_exhausted", C.c_int), ("hipmalloc_fallbacks", u64)]


def pool_stats():
    info = PoolInfo()
    check(lib().bwtm_pool_stats(C.byref(info)))
    return {k: int(getattr(info, k)) for k, _ in PoolInfo._fields_}


def profile_enable(on=True):
    check(lib().bwtm_profile_enable(1 if on else 0))


def profile_only(name=None):
    check(lib().bwtm_profile_only(name.encode() if name else None))


def profile_reset():
    check(lib().bwtm_profile_reset())


def profile_read():
    """{kernel name: (total_ms, launches)} since the last reset."""
    cap = 64
    names = (C.c_char_p * cap)()
    ms = (C.c_double * cap)()
    n = (u64 * cap)()
    k = lib().bwtm_profile_read(names, ms, n, cap)
    return {names[i].decode(): (ms[i], int(n[i])) for i in range(min(k, cap))}
