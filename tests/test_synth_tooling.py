"""The input tooling (torch, CPU here) against the oracle's generator and brute-force BWT."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def synth(bwtm):
    from bwt_merge_amd import synth
    return synth


@pytest.mark.parametrize("seed,first,n,L", [(1001, 0, 50, 100), (1002, 12345, 20, 150), (7, 3, 5, 1)])
def test_generator_matches_oracle(synth, oracle, seed, first, n, L):
    got = synth.generate_reads(seed, first, n, L).numpy()
    ref = oracle.generate_reads(seed, n, L, first_read=first).reshape(n, L + 1)
    assert np.all(ref[:, L] == 0)
    assert np.array_equal(got, ref[:, :L])
    assert got.min() >= 1 and got.max() <= 5


@pytest.mark.parametrize("n,L", [(1, 1), (7, 3), (300, 100), (200, 150), (64, 21), (64, 22), (50, 42)])
def test_leaf_bwt_matches_brute_force(synth, oracle, n, L):
    reads = synth.generate_reads(1001, 0, n, L)
    got = synth.leaf_bwt(reads).numpy()
    text = oracle.generate_reads(1001, n, L)
    ref = oracle.FMI.from_text(text).symbols
    assert np.array_equal(got, ref)


def test_leaf_bwt_with_duplicate_reads(synth, oracle):
    # identical reads: ties between equal suffixes must be broken by sequence index
    base = synth.generate_reads(5, 0, 10, 30)
    reads = torch.cat([base, base[:4], base[2:7]], dim=0)
    got = synth.leaf_bwt(reads).numpy()
    text = np.concatenate([np.concatenate([r.numpy(), [0]]) for r in reads]).astype(np.uint8)
    assert np.array_equal(got, oracle.FMI.from_text(text).symbols)
