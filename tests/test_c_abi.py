"""The drop-in boundary is the C ABI: every function include/bwtm.h declares is exported by libbwtm.so and bound by the ctypes stub
(bwt-merge_amd/capi.py); include/bwtm_experimental.h is matched by libbwtm_experimental.so only.  No compute calls: runs without a GPU."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)                       # comments quote call sequences
    return sorted(set(re.findall(r"\b(bwtm_[A-Za-z0-9_]+)\s*\(", text)))


def test_every_declared_function_is_exported_and_bound(bwtm):
    bwtm.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "bwt-merge_amd", "libbwtm.so"))
    names = declared("bwtm.h")
    assert len(names) > 70
    bound = {n for n, _, _ in bwtm.capi.SYMBOLS} | {"bwtm_last_error", "bwtm_tune", "bwtm_trim", "bwtm_synchronize"}
    for n in names:
        assert hasattr(lib, n), "include/bwtm.h declares %s, libbwtm.so does not export it" % n
    missing = [n for n in names if n not in bound and not hasattr(bwtm.capi.lib(), n)]
    assert not missing, missing
    for n, _, _ in bwtm.capi.SYMBOLS:
        assert n in names, "capi.py binds %s, which include/bwtm.h does not declare" % n


def test_experimental_header_matches_the_experimental_library(bwtm):
    bwtm.build(experimental=True)
    from bwt_merge_amd import experimental
    names = declared("bwtm_experimental.h")
    assert names == sorted(n for n, _, _ in experimental.EXPERIMENTAL_SYMBOLS)
    exp = ctypes.CDLL(experimental.EXPERIMENTAL_LIB_PATH)
    product = ctypes.CDLL(os.path.join(ROOT, "bwt-merge_amd", "libbwtm.so"))
    for n in names:
        assert hasattr(exp, n) and not hasattr(product, n), n


def test_product_package_exposes_no_experimental_names(bwtm):
    """The binding follows the split of the C side: nothing of include/bwtm_experimental.h is reachable through the product package or
    its modules (capi, dist); the experimental module refuses to bind against the product library."""
    import pytest
    from bwt_merge_amd import capi, dist, experimental
    for mod in (bwtm, capi, dist):
        for name in ("FSlice", "FSliceView", "search_sliced", "slice_range", "EXPERIMENTAL_SYMBOLS", "EXPERIMENTAL_LIB_PATH", "need_experimental"):
            assert not hasattr(mod, name), (mod.__name__, name)
    assert not any(n.startswith("bwtm_fslice") for n, _, _ in capi.SYMBOLS)
    if not os.environ.get("BWTM_LIB"):                                          # the product library is what this process binds
        assert experimental.loaded() is False
        with pytest.raises(bwtm.BwtmError):
            experimental._bind()
