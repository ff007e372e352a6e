"""experimental_dist.search_partitioned_dist on ONE rank over RCCL: the library's device buffers handed to torch.distributed without a copy
(the dense send buffer, the receive buffers), the all-gathered counts, five all_to_all_single calls per step -- with itself here.  More
ranks run the same code; their exchange logic is covered over gloo on the CPU (tests/test_partition_dist_host.py)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(bwtm):
    bwtm.init(0)
    from bwt_merge_amd import experimental
    assert experimental.loaded(), "these tests need BWTM_LIB=libbwtm_experimental.so"
    yield bwtm
    bwtm.make_default_current()
    bwtm.trim()


def test_one_rank_over_rccl_equals_oracle(gpu, oracle):
    import torch
    import torch.distributed as dist
    from bwt_merge_amd.experimental_dist import search_partitioned_dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ta = oracle.generate_reads(9951, 2000, 80)
        tb = np.concatenate([oracle.generate_reads(9952 + j, 300, int(n)) for j, n in enumerate([1, 25, 80, 121])])
        a, b = oracle.FMI.from_text(ta), oracle.FMI.from_text(tb)
        ranks, counts, _ = oracle.search(a, b, threads=2)
        A = gpu.Index.upload(a.data, a.sequences, a.bases); B = gpu.Index.upload(b.data, b.sequences, b.bases)
        for node_ratio in (0, 8, 1):                                        # elements from the roots on; a few node levels; the whole search on nodes
            ra = gpu.RankArray(A, B)
            steps, levels = search_partitioned_dist(gpu, A, B, ra, b.sequences, [0, b.bases], 0, 1, dist, torch, dev, node_ratio=node_ratio)
            assert (steps + levels == 122 or node_ratio == 1) and (levels > 0) == (node_ratio > 0), (node_ratio, steps, levels)
            ra.finalize()
            got_r, got_c = ra.runs()
            assert np.array_equal(got_r, ranks) and np.array_equal(got_c, counts), node_ratio
            ra.free()
        A.free(); B.free()
    finally:
        dist.destroy_process_group()


def test_whole_merge_of_one_rank_over_rccl_equals_oracle(gpu, oracle):
    """merge_partitioned_dist with one rank: windows from bytes, ranged bitvector, the search's exchange, the boundary all-gather, the
    product's range exchanges -- every collective runs (with itself); the slice is the oracle's merged stream."""
    import torch
    import torch.distributed as dist
    from bwt_merge_amd.experimental_dist import merge_partitioned_dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        a = oracle.FMI.from_text(oracle.generate_reads(9961, 6000, 100)); b = oracle.FMI.from_text(oracle.generate_reads(9962, 5000, 100))
        S, keep, steps = merge_partitioned_dist(gpu, a, b, ([0, a.bases], [0, b.bases]), 0, 1, dist, torch, dev)
        m, _ = oracle.merge(a, b, threads=2)
        assert steps[0] + steps[1] == 101 and steps[1] > 0 and S.total_nbytes == m.data.size
        assert np.array_equal(S.data(), m.data)
        S.free()
        for x in keep:
            x.free()
    finally:
        dist.destroy_process_group()
