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
    names = declared("bwtm_experimental.h")
    assert names == sorted(n for n, _, _ in bwtm.capi.EXPERIMENTAL_SYMBOLS)
    exp = ctypes.CDLL(bwtm.EXPERIMENTAL_LIB_PATH)
    product = ctypes.CDLL(os.path.join(ROOT, "bwt-merge_amd", "libbwtm.so"))
    for n in names:
        assert hasattr(exp, n) and not hasattr(product, n), n
