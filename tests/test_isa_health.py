"""Compiler-level regressions that cost more than any algorithmic change did: a hot kernel that starts to
use scratch memory (a select chain turned into a dynamically indexed private array made k_frontier_step 75 % slower) or loses
occupancy.  Compiles the library's device code to ISA text (no GPU needed) and reads the per-kernel metadata."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bwt-merge_amd", "csrc", "bwtm_api.hip")
OUT = os.path.join(ROOT, "tests", "_build", "bwtm_api.s")
HOT = ("k_frontier_step", "k_frontier_scan", "k_tile_build_frontier", "k_build_recs", "k_block_len", "k_interleave", "k_enc_emit", "k_enc_size",
       "k_enc_lasthead", "k_block_cum", "k_ingest_scatter", "k_ingest_hist", "k_ingest_keys", "k_lf_walk_binned", "k_pull_scan1", "k_cut_search_seg")


@pytest.fixture(scope="module")
def isa():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", OUT, SRC],
                          stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for line in open(OUT):
        m = re.match(r"^(_ZN4bwtm\w+):", line)
        if m:
            cur = m.group(1); kernels[cur] = {}
            continue
        m = re.match(r"; (ScratchSize|NumVgprs|Occupancy): (\d+)", line)
        if m and cur:
            kernels[cur].setdefault(m.group(1), int(m.group(2)))
    return kernels


def test_hot_kernels_use_no_scratch(isa):
    seen = set()
    for name, meta in isa.items():
        for hot in HOT:
            if hot in name:
                seen.add(hot)
                assert meta.get("ScratchSize", 0) == 0, "%s uses %d bytes of scratch per lane" % (name, meta["ScratchSize"])
    assert seen == set(HOT), "kernels not found in the ISA: %s" % sorted(set(HOT) - seen)


def test_step_kernel_keeps_full_occupancy(isa):
    steps = [meta for name, meta in isa.items() if "k_frontier_step" in name]
    assert steps and all(m["Occupancy"] == 8 and m["NumVgprs"] <= 64 for m in steps), steps
