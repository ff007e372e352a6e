"""bench.py end to end on the GPU at a small size: the single-GPU path and the multi-GPU code path
(process group on RCCL, caller-owned bitvector, all-reduce) forced onto one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(cmd):
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_single_gpu_small(bwtm):
    d = run_bench([sys.executable, "bench.py", "--reads", "200000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "20000"])
    assert d["verified"] is True and d["n_gpus"] == 1 and d["value"] > 0
    # at this size bwtm_search picks the per-chain walk; config 2 runs the frontier search
    assert d["roofline"]["kernel"] in ("k_lf_walk_binned", "k_frontier_step") and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["gpu_parity_on_sample"] is True and d["cpu_baseline"]["kind"] == "port"


def test_bench_distributed_path_on_one_rank(bwtm):
    d = run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                   "--master-port", "29517", "bench.py", "--gpus", "1", "--force-dist", "--reads", "200000", "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline"])
    assert d["verified"] is True and d["value"] > 0
