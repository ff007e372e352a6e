import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# tests/experimental/ needs libbwtm_experimental.so (BWTM_LIB): collected only in the child process tests/test_gpu_experimental.py starts
collect_ignore_glob = [] if os.environ.get("BWTM_EXPERIMENTAL_TESTS") else ["experimental/*"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the product package (directory 'bwt-merge_amd') under the name bwt_merge_amd."""
    import _pkg
    return _pkg.load()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def bwtm():
    return load_pkg()
