"""The library's device scan (bwtm_x_device_scan, experimental build only) against numpy: sums and maxima, one and several arrays, sizes
around every boundary of its recursion (one tile of 2048 items, one level of tile totals, two levels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TILE = 2048


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    from bwt_merge_amd import experimental
    assert experimental.loaded(), "these tests need BWTM_LIB=libbwtm_experimental.so"
    yield bwtm
    bwtm.trim()


def expect(values, narrays, op):
    v = values.reshape(narrays, -1)
    out = np.zeros_like(v)
    if op == 0:
        out[:, 1:] = np.cumsum(v[:, :-1], axis=1, dtype=np.uint64)
    else:
        out[:, 1:] = np.maximum.accumulate(v[:, :-1], axis=1)
    return out.reshape(-1)


SIZES = [1, 2, TILE - 1, TILE, TILE + 1, 2 * TILE + 5, 63 * TILE, 64 * TILE, 64 * TILE + 1, 65 * TILE + 7, 129 * TILE + 3, 4097 * TILE + 11]


@pytest.mark.parametrize("op", [0, 1])
def test_scan_equals_numpy(gpu, op):
    from bwt_merge_amd.experimental import device_scan
    rng = np.random.default_rng(100 + op)
    for n in SIZES:
        for narrays in (1, 7):
            if n * narrays > 40_000_000:
                continue
            if op == 0:
                v = rng.integers(0, 1 << 20, n * narrays, dtype=np.uint64)
                v[rng.integers(0, n * narrays, 3)] = np.uint64(1 << 33)          # sums beyond 32 bits
            else:
                v = rng.integers(0, 1 << 39, n * narrays, dtype=np.uint64)
            got = device_scan(v, narrays, op)
            assert np.array_equal(got, expect(v, narrays, op)), (n, narrays, op)


def test_scan_of_sparse_and_constant_tables(gpu):
    """Tables as the path produces them: mostly zero (segment lengths late in a search), all ones (flags), one huge item."""
    from bwt_merge_amd.experimental import device_scan
    n = 300 * TILE + 17
    z = np.zeros(n, dtype=np.uint64); z[[5, n // 2, n - 1]] = [7, 1 << 35, 3]
    assert np.array_equal(device_scan(z), expect(z, 1, 0))
    ones = np.ones(n, dtype=np.uint64)
    assert np.array_equal(device_scan(ones), np.arange(n, dtype=np.uint64))
    assert np.array_equal(device_scan(z, 1, 1), expect(z, 1, 1))
