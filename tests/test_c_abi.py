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


def test_struct_layouts_of_the_binding_match_the_headers(bwtm, tmp_path):
    """ctypes mirrors of the C structs against the compiler's own layout of include/*.h: a binding (or a host binary) built against an older
    layout hands the library wrong pointers without any error -- the experimental view struct grew in round 5 and a stale host binary faulted
    on the device."""
    import ctypes
    import subprocess
    from bwt_merge_amd import capi, experimental
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bwtm.h"\n#include "bwtm_experimental.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %d %zu %zu %zu %zu %zu\\n", sizeof(bwtm_fslice_view), offsetof(bwtm_fslice_view, totals), offsetof(bwtm_fslice_view, below),\n'
                   '  offsetof(bwtm_fslice_view, class_first), BWTM_X_MAX_PARTS, sizeof(bwtm_host_input), sizeof(bwtm_host_output), offsetof(bwtm_host_output, ms_total),\n'
                   '  sizeof(bwtm_pool_info), offsetof(bwtm_pool_info, hipmalloc_fallbacks)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    (size, off_totals, off_below, off_class_first, max_parts, size_in, size_out, off_ms_total, size_pool,
     off_fallbacks) = (int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert ctypes.sizeof(capi.HostInput) == size_in
    assert ctypes.sizeof(capi.HostOutput) == size_out and capi.HostOutput.ms_total.offset == off_ms_total
    assert ctypes.sizeof(capi.PoolInfo) == size_pool and capi.PoolInfo.hipmalloc_fallbacks.offset == off_fallbacks
    V = experimental.FSliceView
    assert max_parts == experimental.MAX_PARTS
    assert ctypes.sizeof(V) == size
    assert V.totals.offset == off_totals and V.below.offset == off_below and V.class_first.offset == off_class_first
    # the structs of the merge over partitioned records (round 6)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "bwtm.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %d\\n", sizeof(bwtm_host_index), offsetof(bwtm_host_index, cum), offsetof(bwtm_host_index, C),\n'
                   '  sizeof(bwtm_index_header), sizeof(bwtm_part_info), offsetof(bwtm_part_info, ms_search), offsetof(bwtm_part_info, record_bytes), BWTM_MAX_PARTS); return 0; }\n')
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    size_hi, off_cum, off_C, size_hdr, size_info, off_ms, off_rec, max_parts = (int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split())
    assert ctypes.sizeof(capi.HostIndex) == size_hi and capi.HostIndex.cum.offset == off_cum and capi.HostIndex.C.offset == off_C
    assert ctypes.sizeof(capi.IndexHeader) == size_hdr
    assert ctypes.sizeof(capi.PartInfo) == size_info and capi.PartInfo.ms_search.offset == off_ms and capi.PartInfo.record_bytes.offset == off_rec
    assert max_parts == 16
