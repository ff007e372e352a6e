"""The per-step exchange of the partitioned search between processes (bwt-merge_amd/experimental_dist.py: exchange_plan + five
all_to_all_single calls per step) on the CPU over gloo, 2 and 3 ranks.  The GPU work of a rank -- one LF step on its elements, the stable
split by symbol, the counts below the cuts -- is done by the oracle's LF here; the exchange code is the one the GPU path runs.  The union of
the ranks' bits must be the oracle's rank array, and every rank's bits must lie inside its own range of the output."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _inputs(orc):
    ta = orc.generate_reads(1201, 90, 30); tb = np.concatenate([orc.generate_reads(1202 + k, 25, int(n)) for k, n in enumerate([1, 9, 30, 41])])
    return orc.FMI.from_text(ta), orc.FMI.from_text(tb)


class _HostIndex:
    def __init__(self, x):
        self.x = x; self.bases = x.bases

    def find(self, patterns):
        return np.array([self.x.C[int(patterns[0][0])]], dtype=np.uint64), None

    def rank(self, positions, comps):
        return np.array([self.x.rank(int(p), int(c)) for p, c in zip(positions, comps)], dtype=np.uint64)


def _worker(rank, world, port, result_dir, node_limit):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import _pkg
    _pkg.load()
    from bwt_merge_amd.experimental import partition_cuts
    from bwt_merge_amd.experimental_dist import all_to_all_classes, exchange_plan
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = _inputs(orc)
    I, R = partition_cuts(_HostIndex(a), _HostIndex(b), world, 2)
    n_out = a.bases + b.bases
    bits = np.zeros(n_out, dtype=np.uint8)
    first, last = min(R[rank], b.sequences), min(R[rank + 1], b.sequences)
    BIG = np.iinfo(np.int64).max

    def counts_below(keys):
        return torch.tensor([[int(np.searchsorted(keys[c], R[k] if k < world else BIG, side="left")) for k in range(world + 1)] for c in range(5)], dtype=torch.int64)

    def gather_counts(mine):
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        return [t.numpy() for t in everyone]

    levels = 0
    if node_limit > 0:
        # the first levels on trie nodes (sp, count, r), every node on the rank that owns sp; children exchanged like the elements, three arrays
        nodes = [(0, b.sequences, a.sequences)] if first == 0 and last > 0 else []
        level_nodes = 1
        while 0 < level_nodes <= node_limit:
            kids = [[] for _ in range(5)]
            for sp, cnt, r in nodes:
                assert R[rank] <= sp and sp + cnt <= max(R[rank + 1], b.sequences if rank == 0 else 0) and I[rank] <= r <= I[rank + 1]      # the node lies inside this rank's windows
                bits[sp + r: sp + cnt + r] = 1
                for c in range(1, 6):
                    lo_c, hi_c = b.rank(sp, c), b.rank(sp + cnt, c)
                    if hi_c > lo_c:
                        kids[c - 1].append((int(b.C[c]) + lo_c, hi_c - lo_c, int(a.C[c]) + a.rank(r, c)))
            arrays = [[np.array([k[f] for k in kids[c]], dtype=np.int64) for c in range(5)] for f in range(3)]
            class_first = np.concatenate([[0], np.cumsum([x.size for x in arrays[0]])])
            below_all = gather_counts(counts_below(arrays[0]))
            # no child crosses a cut (cuts at k-mer boundaries)
            for c in range(5):
                for sp, cnt, _ in kids[c]:
                    assert not any(sp < R[k] < sp + cnt for k in range(1, world))
            send, recv = exchange_plan(below_all, rank, world)
            n_in = sum(sum(x) for x in recv)
            got = []
            for f in range(3):
                buf = torch.empty(n_in, dtype=torch.int64)
                all_to_all_classes(dist, torch.from_numpy(np.concatenate(arrays[f])), class_first, send, recv, buf)
                got.append(buf.numpy())
            nodes = list(zip(got[0].tolist(), got[1].tolist(), got[2].tolist()))
            assert all(nodes[j][0] + nodes[j][1] <= nodes[j + 1][0] for j in range(len(nodes) - 1))      # sorted, disjoint
            level_nodes = sum(int(x[c][world]) for x in below_all for c in range(5))
            levels += 1
        # every rank expands its own nodes into its own elements: the outputs of a step, class 0
        out_i = [np.array([sp + j for sp, cnt, r in nodes for j in range(cnt)], dtype=np.int64)] + [np.zeros(0, dtype=np.int64)] * 4
        out_r = [np.array([r for sp, cnt, r in nodes for j in range(cnt)], dtype=np.int64)] + [np.zeros(0, dtype=np.int64)] * 4
    else:
        # this rank's outputs of "step -1": the roots it owns, class 0, as the dense send buffer (i = b's coordinate, r = a's)
        out_i = [np.arange(first, last, dtype=np.int64)] + [np.zeros(0, dtype=np.int64)] * 4
        out_r = [np.full(last - first, a.sequences, dtype=np.int64)] + [np.zeros(0, dtype=np.int64)] * 4
    steps = 0
    while True:
        class_first = np.concatenate([[0], np.cumsum([x.size for x in out_i])])
        below_all = gather_counts(counts_below(out_i))
        if sum(int(x[c][world]) for x in below_all for c in range(5)) == 0:
            break
        send, recv = exchange_plan(below_all, rank, world)
        n_in = sum(sum(x) for x in recv)
        in_i, in_r = torch.empty(n_in, dtype=torch.int64), torch.empty(n_in, dtype=torch.int64)
        assert all_to_all_classes(dist, torch.from_numpy(np.concatenate(out_i)), class_first, send, recv, in_i) == n_in
        all_to_all_classes(dist, torch.from_numpy(np.concatenate(out_r)), class_first, send, recv, in_r)
        in_i, in_r = in_i.numpy(), in_r.numpy()
        # what arrives is this rank's range of the sorted frontier, in order
        assert np.all(np.diff(in_i) > 0) and np.all(np.diff(in_r) >= 0)
        assert n_in == 0 or (R[rank] <= in_i[0] and in_i[-1] < max(R[rank + 1], 1) and I[rank] <= in_r[0] and in_r[-1] <= I[rank + 1])
        # one LF step on every element (the step kernel's work), outputs split by symbol, order kept
        out_i = [[] for _ in range(5)]; out_r = [[] for _ in range(5)]
        for i, r in zip(in_i.tolist(), in_r.tolist()):
            bits[i + r] = 1
            nxt, c = b.LF(i)
            if c != 0:
                out_i[c - 1].append(nxt); out_r[c - 1].append(a.LF(r, c))
        out_i = [np.array(x, dtype=np.int64) for x in out_i]; out_r = [np.array(x, dtype=np.int64) for x in out_r]
        steps += 1
    np.save(os.path.join(result_dir, "bits_%d.npy" % rank), bits)
    np.save(os.path.join(result_dir, "cuts_%d.npy" % rank), np.array([I, R], dtype=np.int64))
    np.save(os.path.join(result_dir, "steps_%d.npy" % rank), np.array([steps, levels], dtype=np.int64))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,node_limit", [(2, 0), (3, 0), (2, 12), (3, 40), (3, 10 ** 9)])
def test_exchange_between_ranks_equals_the_rank_array(tmp_path, oracle, bwtm, world, node_limit):
    """node_limit = nodes a level may have to be processed as nodes (0: elements from the roots on; huge: the whole search on nodes)."""
    bwtm.build(experimental=True)
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path), node_limit), nprocs=world, join=True, start_method="spawn")
    a, b = _inputs(oracle)
    ranks, counts, _ = oracle.search(a, b, threads=1)
    ra = oracle.ra_from_runs(ranks, counts)
    expect = np.zeros(a.bases + b.bases, dtype=np.uint8)
    expect[np.arange(b.bases, dtype=np.uint64) + ra] = 1
    I, R = np.load(os.path.join(str(tmp_path), "cuts_0.npy"))
    total = np.zeros_like(expect)
    for r in range(world):
        bits = np.load(os.path.join(str(tmp_path), "bits_%d.npy" % r))
        assert not np.any(total & bits)                                     # disjoint
        on = np.nonzero(bits)[0]
        assert on.size == R[r + 1] - R[r]
        if on.size:
            assert I[r] + R[r] <= on[0] and on[-1] < I[r + 1] + R[r + 1]        # inside the rank's own range of the output
        total |= bits
    assert np.array_equal(total, expect)
    steps, levels = np.load(os.path.join(str(tmp_path), "steps_0.npy"))
    assert (levels > 0) == (node_limit > 0) and (steps + levels == 42 or node_limit > 10 ** 6)      # the longest read (41 symbols) + its endmarker


def test_plan_of_one_rank_is_the_identity(bwtm):
    bwtm.build(experimental=True)
    from bwt_merge_amd.experimental_dist import exchange_plan
    below = [np.array([[0, 7], [0, 0], [0, 3], [0, 1], [0, 0]])]
    send, recv = exchange_plan(below, 0, 1)
    assert send == recv == [[7], [0], [3], [1], [0]]
