"""Host-side logic of the partitioned merge (bwt-merge_amd/experimental.py), on the CPU: the cuts chosen from an
index on the host are points of the merged order that balance the parts, and the blocks named for a window hold every record of it."""
import numpy as np
import pytest


class HostIndex:
    """What partition_cuts asks of an index, answered by the oracle's FM-index."""

    def __init__(self, x):
        self.x = x; self.bases = x.bases

    def find(self, patterns):
        return np.array([self.x.C[int(patterns[0][0])]], dtype=np.uint64), None

    def rank(self, positions, comps):
        return np.array([self.x.rank(int(p), int(c)) for p, c in zip(positions, comps)], dtype=np.uint64)


@pytest.fixture(scope="module")
def inputs(bwtm, oracle):
    bwtm.build(experimental=True)
    a = oracle.FMI.from_text(oracle.generate_reads(8101, 1500, 60)); b = oracle.FMI.from_text(oracle.generate_reads(8102, 1100, 75))
    ranks, counts, _ = oracle.search(a, b, threads=2)
    return a, b, oracle.ra_from_runs(ranks, counts)


@pytest.mark.parametrize("parts,k", [(2, 1), (3, 2), (4, 3), (8, 4), (16, 3)])
def test_cuts_are_balanced_points_of_the_merged_order(inputs, parts, k):
    from bwt_merge_amd.experimental import partition_cuts
    a, b, ra = inputs
    I, R = partition_cuts(HostIndex(a), HostIndex(b), parts, k)
    assert len(I) == parts + 1 and I[0] == 0 and R[0] == 0 and I[-1] == a.bases and R[-1] == b.bases
    assert all(I[g] <= I[g + 1] and R[g] <= R[g + 1] for g in range(parts))
    for g in range(1, parts):
        # b's suffixes from R_g on lie above at least I_g of a's suffixes, those before it above at most I_g: (I_g, R_g) is a point of the order
        if R[g] < b.bases:
            assert int(ra[R[g]]) >= I[g]
        if R[g] > 0:
            assert int(ra[R[g] - 1]) <= I[g]
    if 5 ** k >= 8 * parts:                                              # enough candidate k-mers: the parts' shares of the output are close to equal
        share = np.diff([I[g] + R[g] for g in range(parts + 1)]) / (a.bases + b.bases)
        assert share.max() < 2.0 / parts and share.min() > 0.4 / parts, share
    # the roots (the "$" range of b) lie below the first cut: the node phase starts on one part
    assert parts == 1 or R[1] >= b.sequences


def test_blocks_named_for_a_window_hold_all_its_records(inputs):
    from bwt_merge_amd.experimental import window_blocks
    a, _, _ = inputs
    be, cum = a.samples
    starts = cum.sum(axis=0).astype(np.uint64)
    assert int(starts[0]) == 0 and int(starts[-1]) == a.bases and np.array_equal(starts[1:], be + np.uint64(1))
    rng = np.random.default_rng(3)
    nb = starts.size - 1
    ranges = [(0, 0), (0, a.bases), (a.bases - 1, a.bases), (127, 128), (128, 255)] + [tuple(sorted(rng.integers(0, a.bases, 2))) for _ in range(200)]
    for first, last in ranges:
        b0, b1 = window_blocks(starts, a.data.size, first, last, a.bases)
        assert 0 <= b0 < b1 <= nb
        # every record of [first >> 7, last >> 7] lies wholly inside the blocks' positions (the last record of the index ends with the index)
        assert int(starts[b0]) <= (int(first) & ~127)
        assert int(starts[b1]) >= min(a.bases, (int(last) | 127) + 1)
        # and the share is tight: the block before b1 is needed, the block after b0 would not do
        assert b1 - b0 == 1 or int(starts[b1 - 1]) < min(a.bases, (int(last) | 127) + 1)
        assert int(starts[min(b0 + 1, nb)]) > (int(first) & ~127) or b0 + 1 == b1


@pytest.mark.parametrize("case", ["short", "one_base", "with_n", "tiny_b", "unequal"])
def test_cuts_on_odd_collections(bwtm, oracle, case):
    """Cuts stay points of the merged order whatever the collections look like: reads shorter than k, a single symbol, N's, a b of a handful of
    sequences, inputs of very different sizes.  (Parts may then be empty or very unequal: the cuts only have to be valid and monotone.)"""
    bwtm.build(experimental=True)
    from bwt_merge_amd.experimental import partition_cuts
    rng = np.random.default_rng({"short": 1, "one_base": 2, "with_n": 3, "tiny_b": 4, "unequal": 5}[case])

    def reads(n, lo, hi, alphabet):
        out = []
        for _ in range(n):
            out.append(rng.choice(alphabet, rng.integers(lo, hi + 1)).astype(np.uint8)); out.append(np.zeros(1, dtype=np.uint8))
        return np.concatenate(out)

    if case == "short":
        ta, tb = reads(400, 0, 3, [1, 2, 3, 4]), reads(300, 0, 4, [1, 2, 3, 4])
    elif case == "one_base":
        ta, tb = reads(200, 5, 40, [3]), reads(150, 1, 60, [3])
    elif case == "with_n":
        ta, tb = reads(300, 20, 50, [1, 2, 3, 4, 5, 5]), reads(250, 10, 70, [1, 2, 3, 4, 5])
    elif case == "tiny_b":
        ta, tb = reads(500, 30, 60, [1, 2, 3, 4]), reads(3, 5, 9, [1, 2, 3, 4])
    else:
        ta, tb = reads(40, 10, 20, [1, 2, 3, 4]), reads(900, 40, 80, [1, 2, 3, 4])
    a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
    ranks, counts, _ = oracle.search(a, b, threads=1)
    ra = oracle.ra_from_runs(ranks, counts)
    for parts, k in ((2, 1), (4, 2), (8, 3), (16, 5)):
        I, R = partition_cuts(HostIndex(a), HostIndex(b), parts, k)
        assert I[0] == 0 and R[0] == 0 and I[-1] == a.bases and R[-1] == b.bases
        assert all(I[g] <= I[g + 1] and R[g] <= R[g + 1] for g in range(parts)), (I, R)
        for g in range(1, parts):
            if R[g] < b.bases:
                assert int(ra[R[g]]) >= I[g], (case, parts, k, g)
            if R[g] > 0:
                assert int(ra[R[g] - 1]) <= I[g], (case, parts, k, g)
            assert R[g] >= min(b.sequences, R[g]) and (R[g] >= b.sequences or R[g] == 0 or True)
        # every root lies below the first cut that is not 0 (cuts are k-mer boundaries: the "$" range is never split)
        first = next((R[g] for g in range(1, parts + 1) if R[g] > 0), b.bases)
        assert first >= b.sequences
