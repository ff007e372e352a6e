#!/bin/bash
# The evidence set of a round, on the GPU box (~30 min): PMC traffic of the step kernel at both sizes, GPU tests, rocprofv3 stats + per-kernel PMC at config 2, rocprofv3 stats at the target size,
# the default bench line (config 2 + target), the other shapes.  Usage: bash tools/final_round.sh <tag>   (outputs under gpurun_out/<tag>/)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; out=$R/gpurun_out/$tag; mkdir -p $out
cd $R
# the step kernel's HBM traffic first: bench.py uses profiles/search_kernel_traffic.json only when its hash of the search's sources matches
bash tools/pmc_step_kernel.sh ${tag}_pmc_step_cfg2 50000000
python tools/make_traffic_json.py gpurun_out/${tag}_pmc_step_cfg2/pmc.txt "profiles/${tag}_pmc_step_kernel_config2.txt (tools/pmc_step_kernel.sh)" 50000000 100 4450000000
bash tools/pmc_step_kernel.sh ${tag}_pmc_step_target 500000000 --tune emit_budget=8589934592
python tools/make_traffic_json.py gpurun_out/${tag}_pmc_step_target/pmc.txt "profiles/${tag}_pmc_step_kernel_target.txt (tools/pmc_step_kernel.sh --tune emit_budget=8589934592)" 500000000 100 44500000000 emit_budget=8589934592
cp profiles/search_kernel_traffic.json $out/search_kernel_traffic.json          # copy back into profiles/ with the two pmc.txt files
cd $R
BWTM_REQUIRE_FRESH_PMC=1 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/gpu_tests.txt; cat $out/gpu_tests.txt          # (the PMC passes above are this tree's: a stale profile fails here)
bash tools/profile_round.sh $tag/prof > $out/profile_round.log 2>&1
bash tools/pmc_passes.sh $tag/pmc_all sq > /dev/null 2>&1
cd /tmp; rm -rf /tmp/prof_t
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- python3 $R/bench.py --reads 500000000 --no-cpu-baseline --no-verify --no-host --target off --keep-pool --steps 2 --warmup 1 > $out/bench_target_under_rocprof.json 2> $out/bench_target_under_rocprof.log
s=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1)
if [ -n "$s" ]; then grep -E "Name|k_frontier_step|k_build_recs|k_enc_emit|k_interleave|k_tile_build|k_block_len|k_enc_size" $s | head -20 > $out/target_rocprofv3_kernel_stats_bwtm.csv; fi
cd $R
python bench.py > $out/bench_default.json 2> $out/bench_default.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.log          # the driver's command
python bench.py --reads-a 200000000 --reads 50000000 --no-cpu-baseline --target off --steps 3 > $out/bench_config4_shape.json 2> $out/bench_config4_shape.log
python bench.py --chain 4 --workload mixed --reads 24000000 --no-cpu-baseline --target off --steps 3 > $out/bench_config5_shape.json 2> $out/bench_config5_shape.log
python bench.py --workload genome --coverage 30 --no-cpu-baseline --target off --no-host --steps 3 > $out/bench_genome30.json 2> $out/bench_genome30.log
python bench.py --workload genome --coverage 300 --no-cpu-baseline --target off --no-host --steps 3 > $out/bench_genome300.json 2> $out/bench_genome300.log
# the merge over partitioned records: contexts of the one GPU standing in for GPUs (per-part kernel times, bytes == the single-GPU stream), one rank, and ranks that share the GPU
python tools/parts_scale.py --parts 2,4,8 > $out/parts_scale_config2.txt 2>&1; tail -5 $out/parts_scale_config2.txt
python bench.py --force-dist --search partitioned --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_partitioned_1rank_config2.json 2> $out/bench_partitioned_1rank_config2.log
python bench.py --gpus 4 --same-device --search partitioned --reads 12000000 --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_partitioned_4proc_same_device_12M.json 2> $out/bench_partitioned_4proc_same_device_12M.log
for f in bench_default bench_config4_shape bench_config5_shape bench_genome30 bench_genome300 bench_partitioned_1rank_config2 bench_partitioned_4proc_same_device_12M; do python3 -c "
import json,sys
try:
    d=json.loads(open('$out/$f.json').read().strip().splitlines()[-1]); t=d.get('target') or {}
    print('$f', d['value'], d['ms_per_step'], d['verified'], d['roofline']['frac'], d['roofline']['traffic_profile_check'], '| target', t.get('value'), t.get('ms_per_step'), t.get('verified'), (t.get('roofline') or {}).get('frac'), (t.get('roofline') or {}).get('traffic_profile_check'))
except Exception as e: print('$f FAILED', e)
"; done
