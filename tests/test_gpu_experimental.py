"""The features of include/bwtm_experimental.h (sliced frontier search, two-plane search view) are not in libbwtm.so.  Their tests
(tests/experimental/) run here in a child process that loads libbwtm_experimental.so -- a second build of the same sources with
-DBWTM_EXPERIMENTAL -- so the product suite and the driver's record of loaded libraries stay about the product."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_has_no_experimental_entry_points(bwtm):
    """CPU check: libbwtm.so exports none of the symbols of include/bwtm_experimental.h, libbwtm_experimental.so exports all of them."""
    import ctypes
    bwtm.build(); bwtm.build(experimental=True)
    from bwt_merge_amd import experimental
    product = ctypes.CDLL(os.path.join(ROOT, "bwt-merge_amd", "libbwtm.so"))
    exp = ctypes.CDLL(experimental.EXPERIMENTAL_LIB_PATH)
    for name, _, _ in experimental.EXPERIMENTAL_SYMBOLS:
        assert not hasattr(product, name), name
        assert hasattr(exp, name), name
    for name, _, _ in bwtm.capi.SYMBOLS:
        assert hasattr(exp, name), name


@pytest.mark.gpu
def test_experimental_suite_in_its_own_process(bwtm):
    bwtm.build(experimental=True)
    from bwt_merge_amd import experimental
    env = dict(os.environ, BWTM_LIB=experimental.EXPERIMENTAL_LIB_PATH, BWTM_EXPERIMENTAL_TESTS="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "experimental")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
