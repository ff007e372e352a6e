"""Bindings of include/bwtm_experimental.h: measured-and-rejected or unfinished designs that are NOT part of the product library.

Importable only next to libbwtm_experimental.so (a second build of the same sources with -DBWTM_EXPERIMENTAL), and usable only in a
process that has loaded THAT library as its bwtm library (BWTM_LIB=<path of libbwtm_experimental.so>): the product package
(bwt_merge_amd, capi.py, dist.py) exposes none of these names -- tests/test_c_abi.py checks both directions.

  FSlice, FSliceView, search_sliced   the frontier search in position slices, one per GPU
  partition_cuts, index_window, search_partitioned   the same search over PARTITIONED records: fixed cuts, every GPU holds one window of
                                      each index and the elements travel
  device_scan                         test hook of the library's device scan
  (the two-plane search view has no entry point of its own: it is the `search_view` knob of the experimental build)
"""
import ctypes as C
import os

from . import capi
from .capi import BwtmError, check, lib, u64, vp
from .dist import shard_range

HERE = os.path.dirname(os.path.abspath(__file__))
# Every symbol include/bwtm_experimental.h declares: exported only by libbwtm_experimental.so (-DBWTM_EXPERIMENTAL), which the tests of
# these features select with BWTM_LIB
EXPERIMENTAL_SYMBOLS = [
    ("bwtm_fslice_create", C.c_int, [vp, vp, vp, u64, C.c_int, C.POINTER(vp)]),
    ("bwtm_fslice_free", None, [vp]),
    ("bwtm_fslice_seed", C.c_int, [vp, u64, u64]),
    ("bwtm_fslice_export", C.c_int, [vp, vp]),
    ("bwtm_fslice_gather", C.c_int, [vp, vp, C.c_int, u64, u64]),
    ("bwtm_fslice_advance", C.c_int, [vp]),
    ("bwtm_fslice_finish", C.c_int, [vp]),
    ("bwtm_x_device_scan", C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), u64, u64, C.c_int]),
    ("bwtm_x_index_window", C.c_int, [vp, u64, u64, C.POINTER(vp)]),
    ("bwtm_x_index_record_bytes", u64, [vp]),
    ("bwtm_x_ra_create_range", C.c_int, [vp, vp, u64, u64, C.POINTER(vp)]),
    ("bwtm_x_ra_or_range", C.c_int, [vp, vp, u64, u64]),
    ("bwtm_x_ra_bytes", u64, [vp]),
    ("bwtm_x_ra_read_words", C.c_int, [vp, u64, u64, vp]),
    ("bwtm_x_ra_or_words", C.c_int, [vp, u64, u64, vp]),
    ("bwtm_x_index_upload_window", C.c_int, [C.c_void_p, u64, u64, C.POINTER(u64), u64, u64, C.POINTER(u64), C.POINTER(vp)]),
    ("bwtm_fslice_set_cuts", C.c_int, [vp, C.POINTER(u64), C.c_int]),
    ("bwtm_fslice_gather_cut", C.c_int, [vp, vp, C.c_int, C.c_int]),
    ("bwtm_fslice_input_buffers", C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]),
    ("bwtm_fslice_set_input", C.c_int, [vp, u64]),
    ("bwtm_fslice_nodes_begin", C.c_int, [vp, u64, u64, u64]),
    ("bwtm_fslice_nodes_step", C.c_int, [vp, vp]),
    ("bwtm_fslice_nodes_gather", C.c_int, [vp, vp, C.c_int, C.c_int]),
    ("bwtm_fslice_nodes_expand", C.c_int, [vp]),
    ("bwtm_fslice_nodes_input_buffers", C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]),
    ("bwtm_fslice_nodes_set_input", C.c_int, [vp, u64]),
]
EXPERIMENTAL_LIB_PATH = os.path.join(HERE, "libbwtm_experimental.so")
if not os.path.exists(EXPERIMENTAL_LIB_PATH):
    raise ImportError("bwt_merge_amd.experimental needs %s (python -c 'import _pkg; _pkg.load().build(experimental=True)')" % EXPERIMENTAL_LIB_PATH)




_bound = False


def loaded():
    """True when the process's bwtm library is the experimental build."""
    return hasattr(lib(), EXPERIMENTAL_SYMBOLS[0][0])


def _bind():
    global _bound
    if not loaded():
        raise BwtmError("the loaded bwtm library is the product build: start the process with BWTM_LIB=%s" % EXPERIMENTAL_LIB_PATH)
    if not _bound:
        for name, res, args in EXPERIMENTAL_SYMBOLS:
            f = getattr(lib(), name)
            f.restype = res
            f.argtypes = args
        _bound = True


MAX_PARTS = 16                  # BWTM_X_MAX_PARTS


class FSliceView(C.Structure):
    _fields_ = [("lo", vp), ("hi", vp), ("prefix", vp), ("phys", vp), ("blocks", u64), ("totals", u64 * 5), ("below", (u64 * (MAX_PARTS + 1)) * 5),
                ("dense_lo", vp), ("dense_hi", vp), ("class_first", u64 * 6)]


class FSliceNodesView(C.Structure):
    _fields_ = [("sp", vp), ("r", vp), ("count", vp), ("class_first", u64 * 6), ("below", (u64 * (MAX_PARTS + 1)) * 5)]


class FSlice:
    """One GPU's state of the sliced frontier search (bwtm_fslice; include/bwtm_experimental.h)."""

    def __init__(self, a, b, ra, capacity, parts):
        _bind()
        out = vp()
        check(lib().bwtm_fslice_create(a.h, b.h, ra.h, capacity, parts, C.byref(out)))
        self.h = out

    def free(self):
        if self.h:
            lib().bwtm_fslice_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def seed(self, seq_first, count):
        check(lib().bwtm_fslice_seed(self.h, seq_first, count))

    def export(self, view):
        check(lib().bwtm_fslice_export(self.h, C.byref(view)))

    def gather(self, views, parts, first, last):
        check(lib().bwtm_fslice_gather(self.h, C.byref(views), parts, first, last))

    def set_cuts(self, r_cuts):
        arr = (u64 * len(r_cuts))(*[int(x) for x in r_cuts])
        check(lib().bwtm_fslice_set_cuts(self.h, arr, len(r_cuts) - 1))

    def gather_cut(self, views, parts, part):
        check(lib().bwtm_fslice_gather_cut(self.h, C.byref(views), parts, part))

    def input_buffers(self):
        """(device address of the input coordinates, of their high bytes or None, capacity in elements): where an exchange between processes
        delivers this slice's next input (experimental_dist.py)."""
        lo, hi, cap = vp(), vp(), u64(0)
        check(lib().bwtm_fslice_input_buffers(self.h, C.byref(lo), C.byref(hi), C.byref(cap)))
        return lo.value, hi.value, int(cap.value)

    def set_input(self, count):
        check(lib().bwtm_fslice_set_input(self.h, int(count)))

    def nodes_begin(self, seq_first, count, node_capacity):
        check(lib().bwtm_fslice_nodes_begin(self.h, seq_first, count, node_capacity))

    def nodes_step(self, view):
        check(lib().bwtm_fslice_nodes_step(self.h, C.byref(view)))

    def nodes_gather(self, views, parts, part):
        check(lib().bwtm_fslice_nodes_gather(self.h, C.byref(views), parts, part))

    def nodes_input_buffers(self):
        sp, r, cnt, cap = vp(), vp(), vp(), u64(0)
        check(lib().bwtm_fslice_nodes_input_buffers(self.h, C.byref(sp), C.byref(r), C.byref(cnt), C.byref(cap)))
        return sp.value, r.value, cnt.value, int(cap.value)

    def nodes_set_input(self, nodes):
        check(lib().bwtm_fslice_nodes_set_input(self.h, int(nodes)))

    def nodes_expand(self):
        check(lib().bwtm_fslice_nodes_expand(self.h))

    def advance(self):
        check(lib().bwtm_fslice_advance(self.h))

    def finish(self):
        check(lib().bwtm_fslice_finish(self.h))


def slice_range(total, part, parts):
    """Contiguous share `part` of `total` frontier elements: [first, last)."""
    per = (total + parts - 1) // parts
    return min(total, part * per), min(total, (part + 1) * per)


def device_scan(values, narrays=1, op=0):
    """Exclusive scan (op 0 = sum, 1 = max) of `narrays` equally long u64 arrays laid end to end, by the library's device scan."""
    import numpy as np
    _bind()
    v = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty_like(v)
    check(lib().bwtm_x_device_scan(v.ctypes.data_as(C.POINTER(C.c_uint64)), out.ctypes.data_as(C.POINTER(C.c_uint64)), v.size // narrays, narrays, op))
    return out


def search_sliced(pkg, indexes, ras, sequences, enter=None):
    """The sliced frontier search driven from ONE host thread over `parts` GPUs (or contexts of one GPU): indexes[g] = (A, B) as
    GPU g holds them, ras[g] = its rank array; enter(g) makes GPU g's context current for the calling thread (None: one context).
    Every GPU ends up with the bits of the elements it advanced; combine the rank arrays as after bwtm_search().  Returns the
    number of LF steps.  (A host thread or process per GPU would run the same loop with barriers where this one switches GPUs.)"""
    parts = len(indexes)
    cap = (sequences + parts - 1) // parts + 1
    views = (FSliceView * parts)()
    fs = []
    for g in range(parts):
        if enter:
            enter(g)
        f = FSlice(indexes[g][0], indexes[g][1], ras[g], cap, parts)
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




def index_window(index, pos_first, pos_last):
    """The records of the positions [pos_first, pos_last] of `index` as a handle of their own (bwtm_x_index_window): only for FSlice / RankArray."""
    _bind()
    out = vp()
    check(lib().bwtm_x_index_window(index.h, int(pos_first), int(pos_last), C.byref(out)))
    return capi.Index(out)


def index_record_bytes(index):
    _bind()
    return int(lib().bwtm_x_index_record_bytes(index.h))


def partition_cuts(a, b, parts, k=4):
    """Cuts of the merged order at k-mer boundaries that balance the POSITIONS (a's + b's) of the parts: returns (I, R), two lists of
    parts + 1 ranks with I[0] = R[0] = 0, I[parts] = a.bases, R[parts] = b.bases.  (I[g], R[g]) = the number of a's / b's suffixes below the
    g-th chosen k-mer w: sp(c w) = C[c] + rank_c(sp(w)), the insertion point of backward search, also for k-mers that do not occur."""
    import numpy as np

    def insertion_points(x):
        C_of = [int(x.find([[c]])[0][0]) for c in range(1, 6)]
        sp = np.zeros(1, dtype=np.uint64)                           # the empty string
        for _ in range(k):
            nxt = []
            for c in range(1, 6):                                    # c w for every w, c-major: lexicographic order is kept
                r = x.rank(sp, np.full(sp.size, c, dtype=np.uint8))
                nxt.append(np.uint64(C_of[c - 1]) + r)
            sp = np.concatenate(nxt)
        return sp

    spa, spb = insertion_points(a), insertion_points(b)
    na, nb = int(a.bases), int(b.bases)
    both = spa.astype(np.float64) + spb.astype(np.float64)
    I, R = [0], [0]
    for g in range(1, parts):
        j = int(np.argmin(np.abs(both - g * (na + nb) / parts)))
        I.append(max(I[-1], int(spa[j]))); R.append(max(R[-1], int(spb[j])))
    I.append(na); R.append(nb)
    return I, R


def search_partitioned(pkg, windows, ras, sequences, r_cuts, enter=None, capacity=None, node_ratio=8, node_capacity=None):
    """The frontier search over partitioned records, driven from ONE host thread: windows[g] = (window of A, window of B) as GPU g holds
    them (index_window over the cuts of partition_cuts), ras[g] = its rank array, r_cuts = the B ranks of the cuts.  The first levels run
    on trie nodes (while a level has at most sequences / node_ratio of them; node_ratio = 0: elements from the roots on), every node on
    the GPU that owns its range; then every GPU advances the elements whose coordinates fall into its windows.  Every GPU ends up with the
    bits of its own output range.  capacity / node_capacity = elements / nodes a GPU can hold in a step / level (default: what ONE GPU
    would need for all of them; with balanced cuts a GPU needs 1 / parts of that and some slack).  Returns (element steps, node
    levels, the largest number of elements any GPU held in a step, elements advanced per GPU)."""
    parts = len(windows)
    cap = int(capacity) if capacity else sequences + 1
    limit = sequences // node_ratio if node_ratio > 0 else 0
    node_cap = int(node_capacity) if node_capacity else min(5 * max(limit, 1), sequences) + 1       # nodes (and children) a GPU can hold in a level
    views = (FSliceView * parts)()
    nviews = (FSliceNodesView * parts)()
    fs = []
    roots = []
    for g in range(parts):
        if enter:
            enter(g)
        f = FSlice(windows[g][0], windows[g][1], ras[g], cap, parts)
        f.set_cuts(r_cuts)
        first, last = min(int(r_cuts[g]), sequences), min(int(r_cuts[g + 1]), sequences)
        roots.append((first, last - first))
        fs.append(f)
    levels = 0
    if limit >= 1 and sum(1 for _, n in roots if n > 0) == 1:         # the root node "$" must lie on one GPU (k-mer cuts: the first)
        for g in range(parts):
            if enter:
                enter(g)
            fs[g].nodes_begin(roots[g][0], roots[g][1], node_cap)
        level_nodes = 1
        while 0 < level_nodes <= limit:
            for g in range(parts):
                if enter:
                    enter(g)
                fs[g].nodes_step(nviews[g])
            level_nodes = 0
            for g in range(parts):
                if enter:
                    enter(g)
                fs[g].nodes_gather(nviews, parts, g)
                level_nodes += sum(int(nviews[h].below[c][g + 1]) - int(nviews[h].below[c][g]) for h in range(parts) for c in range(5))
            levels += 1
        for g in range(parts):
            if enter:
                enter(g)
            fs[g].nodes_expand()
            fs[g].export(views[g])
    else:
        for g in range(parts):
            if enter:
                enter(g)
            fs[g].seed(roots[g][0], roots[g][1])
            fs[g].export(views[g])
    steps, largest, work = 0, 0, [0] * parts
    while True:
        total = sum(int(views[h].totals[c]) for h in range(parts) for c in range(5))
        if total == 0:
            break
        for g in range(parts):                              # every GPU pulls the elements of its position range from all GPUs' outputs ...
            if enter:
                enter(g)
            fs[g].gather_cut(views, parts, g)
            held = sum(int(views[h].below[c][g + 1]) - int(views[h].below[c][g]) for h in range(parts) for c in range(5))
            largest = max(largest, held); work[g] += held
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
    return steps, levels, largest, work


def window_blocks(block_starts, nbytes, pos_first, pos_last, bases):
    """The 64-byte blocks [b0, b1) of a native stream whose records cover the positions [pos_first, pos_last]: block_starts[b] = the position
    block b begins at (block_starts[blocks] = bases; the cumulative sample arrays of a native file summed over the symbols)."""
    import numpy as np
    nb = len(block_starts) - 1
    lo = int(pos_first) & ~127
    hi = min(int(bases), (int(pos_last) | 127) + 1)
    b0 = max(0, int(np.searchsorted(block_starts, lo, side="right")) - 1)
    b1 = min(nb, int(np.searchsorted(block_starts, hi, side="left")))
    return b0, max(b1, b0 + 1)


def block_starts(cum):
    """The position every block of a native stream begins at, from its cumulative sample arrays (blocks + 1 entries; the last = bases)."""
    import numpy as np
    return cum.sum(axis=0).astype(np.uint64)


def index_upload_window(data, cum, bases, sequences, pos_first, pos_last, starts=None):
    """A window of an index transcoded from its own share of the native bytes: data = the whole native stream in host memory (only the
    share is copied to the device), cum[6][blocks + 1] = its cumulative sample arrays (counts of every symbol before every block);
    starts = block_starts(cum) when the caller has it already (a pass over 48 bytes per block otherwise)."""
    import numpy as np
    _bind()
    if starts is None:
        starts = block_starts(cum)
    b0, b1 = window_blocks(starts, data.size, pos_first, pos_last, bases)
    share = np.ascontiguousarray(data[b0 * 64: min(b1 * 64, data.size)])
    before = (u64 * 6)(*[int(cum[c][b0]) for c in range(6)])
    totals = [int(cum[c][-1]) for c in range(6)]
    Cs = (u64 * 7)(*[sum(totals[:c]) for c in range(7)])
    out = vp()
    check(lib().bwtm_x_index_upload_window(share.ctypes.data_as(C.c_void_p), share.size, int(starts[b0]), before, int(bases), int(sequences), Cs, C.byref(out)))
    return capi.Index(out)


def rank_array_range(a, b, pos_first, pos_last):
    """The rank array of one part: the bits of the output positions [pos_first, pos_last) and a tile on either side (bwtm_x_ra_create_range)."""
    _bind()
    out = vp()
    check(lib().bwtm_x_ra_create_range(a.h, b.h, int(pos_first), int(pos_last), C.byref(out)))
    ra = capi.RankArray.__new__(capi.RankArray)
    ra.h = out; ra.n_out = a.bases + b.bases; ra.nb = b.bases
    return ra


def ra_or_range(dst, src, pos_first, pos_last):
    _bind()
    check(lib().bwtm_x_ra_or_range(dst.h, src.h, int(pos_first), int(pos_last)))


def ra_bytes(ra):
    _bind()
    return int(lib().bwtm_x_ra_bytes(ra.h))


MERGE_MARGIN = 2 * 65536        # positions of A / B a part may read beyond its cuts: one encoder segment + the halo chunk


def merge_partitioned(pkg, a, b, parts, cuts, from_bytes=True, node_ratio=8, capacity=None, node_capacity=None, profile=False, keep_data=True):
    """The whole merge over partitioned records, driven from ONE host thread over `parts` contexts of one GPU (one GPU each on a real machine):
    a, b = the inputs on the HOST (objects with .data = the native bytes, .samples = (block_end, cum[6][blocks + 1]), .bases, .sequences);
    cuts = (I, R) from partition_cuts.  Every part
      1. transcodes its windows of A and B from its own share of the native bytes (from_bytes = False: cuts them from whole records, which it
         uploads first) and creates the rank array of its own output range,
      2. searches: node phase and element steps with the elements routed by position (search_partitioned),
      3. takes the earlier parts' bits inside its first output segment, and finalizes / interleaves / encodes its range of the output from its
         windows with the product's range entry points (the output ranges are the cuts rounded down to 65 536-position segments).
    Returns a dict: data (the parts' native bytes, in order), block_end / cum (their sample arrays), bounds, held (bytes of records per part),
    ra_bytes, and with profile = True the kernel milliseconds of every part by phase."""
    import numpy as np
    from .dist import fold_offsets, super_owners
    I, R = cuts
    na, nb = int(a.bases), int(b.bases)
    nrecs = ((na + nb) >> 7) + 1
    P = [I[g] + R[g] for g in range(parts + 1)]                           # the parts' ranges of the output
    ctxs = [pkg.Context(0) for _ in range(parts)]
    phases = [dict() for _ in range(parts)]

    def enter(g):
        ctxs[g].make_current()

    def begin(g):
        enter(g)
        if profile:
            pkg.profile_only(None); pkg.profile_reset(); pkg.profile_enable(True)

    def end(g, name):
        if profile:
            enter(g)
            pkg.synchronize()
            phases[g][name] = phases[g].get(name, 0.0) + sum(v[0] for v in pkg.profile_read().values())
            pkg.profile_enable(False)

    windows, ras = [], []
    for g in range(parts):
        begin(g)
        a_lo, a_hi = max(0, I[g] - MERGE_MARGIN), min(na, I[g + 1] + MERGE_MARGIN)
        b_lo, b_hi = max(0, R[g] - MERGE_MARGIN), min(nb, R[g + 1] + MERGE_MARGIN)
        if from_bytes:
            wa = index_upload_window(a.data, a.samples[1], na, a.sequences, a_lo, a_hi, starts=_starts_of(a))
            wb = index_upload_window(b.data, b.samples[1], nb, b.sequences, b_lo, b_hi, starts=_starts_of(b))
        else:
            A = pkg.Index.upload(a.data, a.sequences, na); B = pkg.Index.upload(b.data, b.sequences, nb)
            wa, wb = index_window(A, a_lo, a_hi), index_window(B, b_lo, b_hi)
            A.free(); B.free()
        windows.append((wa, wb))
        ras.append(rank_array_range(wa, wb, P[g], P[g + 1]) if from_bytes else pkg.RankArray(wa, wb))
        end(g, "transcode")
    for g in range(parts):
        begin(g)
    steps, levels, largest, work = search_partitioned(pkg, windows, ras, int(b.sequences), R, enter, capacity=capacity, node_ratio=node_ratio, node_capacity=node_capacity)
    for g in range(parts):
        end(g, "search")
    seg = [0] + [P[g] // 65536 for g in range(1, parts)]
    bounds = [(min(nrecs, seg[g] * 512), nrecs if g == parts - 1 else min(nrecs, seg[g + 1] * 512)) for g in range(parts)]
    for g in range(parts):
        begin(g)
        for h in range(parts):
            if from_bytes:
                first, last = max(seg[g] * 65536, P[h]), min(P[g], P[h + 1])
                if h < g and first < last:
                    ra_or_range(ras[g], ras[h], first, last)
            elif h != g:
                ras[g].or_from(ras[h])
    counts = []
    for g, (first, last) in enumerate(bounds):
        enter(g)
        counts.append(ras[g].range_counts(first, last))
    totals = [c[0] for c in counts]
    if sum(totals) != nb:
        raise BwtmError("the parts' ranges hold %d set bits, b has %d positions" % (sum(totals), nb))
    nsup = counts[0][1].size
    owner = super_owners(nsup, bounds)
    prefix = np.concatenate([[0], np.cumsum(totals)]).astype(np.uint64)
    super_boff = prefix[owner] + sum(c[1] for c in counts)
    slices = []
    for g, (first, last) in enumerate(bounds):
        enter(g)
        halo = next((counts[h][2] for h in range(g - 1, -1, -1) if bounds[h][1] > bounds[h][0]), None)
        ras[g].finalize_range(first, last, int(prefix[g]), int(prefix[parts]), super_boff, halo)
        slices.append(pkg.Slice(windows[g][0], windows[g][1], ras[g], first, last))
    heads = []
    for g, s in enumerate(slices):
        enter(g); heads.append(s.lasthead())
    tables = []
    for g, s in enumerate(slices):
        enter(g); tables.append(s.size_table(max(heads[:g], default=0)))
    offsets = fold_offsets(tables)
    data, starts, nbytes = [], [], []
    for g, (s, off) in enumerate(zip(slices, offsets)):
        enter(g); s.encode(off); starts.append(s.first_block_start()); nbytes.append(s.nbytes)
    for g in range(parts):
        end(g, "finalize_interleave_encode")
    be, cum = [], []
    for g, s in enumerate(slices):
        enter(g)
        if keep_data:
            data.append(s.data())
        nxt = next((p for p in starts[g + 1:] if p is not None), na + nb)
        x, y = s.samples(nxt)
        be.append(x); cum.append(y)
    out = dict(data=data, block_end=be, cum=cum, bounds=bounds, offsets=offsets, nbytes=nbytes, steps=steps, levels=levels, largest=largest, work=work,
               held=[index_record_bytes(w[0]) + index_record_bytes(w[1]) for w in windows], ra_bytes=[ra_bytes(r) if from_bytes else (na + nb) // 8 for r in ras],
               phases=phases, slices=slices)
    out["release"] = lambda: _release_partitioned(pkg, ctxs, slices, ras, windows)
    return out


def _starts_of(x):
    """block_starts of a host input, computed once per object."""
    st = getattr(x, "_block_starts", None)
    if st is None:
        st = block_starts(x.samples[1])
        try:
            x._block_starts = st
        except AttributeError:
            pass
    return st


def _release_partitioned(pkg, ctxs, slices, ras, windows):
    for g in range(len(ctxs)):
        ctxs[g].make_current()
        slices[g].free(); ras[g].free(); windows[g][0].free(); windows[g][1].free()
    pkg.make_default_current()
    for c in ctxs:
        c.destroy()


def ra_read_words(ra, pos_first, pos_last, device_ptr):
    _bind()
    check(lib().bwtm_x_ra_read_words(ra.h, int(pos_first), int(pos_last), vp(device_ptr)))


def ra_or_words(ra, pos_first, pos_last, device_ptr):
    _bind()
    check(lib().bwtm_x_ra_or_words(ra.h, int(pos_first), int(pos_last), vp(device_ptr)))
