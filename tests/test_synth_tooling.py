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


def test_ragged_leaf_bwt_matches_brute_force(synth, oracle):
    """Reads of different lengths in one leaf (what PlainData::read ingests, formats.cpp:133-161): the mixed 100 / 150 bp
    workload of BASELINE config 5 and arbitrary lengths, including empty reads."""
    n = 300
    reads = synth.make_reads("mixed", 77, 40, n, 100, 1000)
    lens = synth.read_lengths("mixed", 40, n, 100)
    assert int((lens == 100).sum()) * 100 == pytest.approx(int((lens == 150).sum()) * 150, rel=0.05)      # half of the bases each
    long_reads = oracle.generate_reads(77, n, 150, first_read=40).reshape(n, 151)
    text = np.concatenate([np.concatenate([long_reads[j, :int(lens[j])], [0]]) for j in range(n)]).astype(np.uint8)
    assert np.array_equal(synth.leaf_bwt(reads, lens).numpy(), oracle.FMI.from_text(text).symbols)
    assert np.array_equal(synth.leaf_symbols("mixed", 77, 40, n, 100, 1000, "cpu").numpy(), oracle.FMI.from_text(text).symbols)
    rng = np.random.default_rng(1)
    lens = torch.from_numpy(rng.integers(0, 31, 120))
    base = synth.generate_reads(9, 0, 120, 30)
    t = torch.arange(30).unsqueeze(0)
    reads = torch.where(t < lens.unsqueeze(1), base, torch.zeros_like(base))
    text = np.concatenate([np.concatenate([base[j, :int(lens[j])].numpy(), [0]]) for j in range(120)]).astype(np.uint8)
    assert np.array_equal(synth.leaf_bwt(reads, lens).numpy(), oracle.FMI.from_text(text).symbols)


def test_reads_matrix_regenerates_single_reads(synth):
    ids = np.array([0, 3, 4, 1000, 7])
    m = synth.reads_matrix("mixed", 1001, ids, 100, 5000, 152)
    assert list((m != 0).sum(axis=1)) == [100, 150, 150, 100, 100]
    full = synth.make_reads("mixed", 1001, 0, 1001, 100, 5000).numpy()
    assert np.array_equal(m[:, :150], full[ids])
    g = synth.reads_matrix("iid", 1002, ids, 100, 5000, 102)
    assert np.array_equal(g[:, :100], synth.generate_reads(1002, 0, 1001, 100).numpy()[ids])
