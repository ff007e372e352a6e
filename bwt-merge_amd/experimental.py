"""Bindings of include/bwtm_experimental.h: measured-and-rejected or unfinished designs that are NOT part of the product library.

Importable only next to libbwtm_experimental.so (a second build of the same sources with -DBWTM_EXPERIMENTAL), and usable only in a
process that has loaded THAT library as its bwtm library (BWTM_LIB=<path of libbwtm_experimental.so>): the product package
(bwt_merge_amd, capi.py, dist.py) exposes none of these names -- tests/test_c_abi.py checks both directions.

  FSlice, FSliceView, search_sliced   the frontier search in position slices, one per GPU
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


class FSliceView(C.Structure):
    _fields_ = [("lo", vp), ("hi", vp), ("prefix", vp), ("phys", vp), ("blocks", u64), ("totals", u64 * 5)]


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

    def advance(self):
        check(lib().bwtm_fslice_advance(self.h))

    def finish(self):
        check(lib().bwtm_fslice_finish(self.h))


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


