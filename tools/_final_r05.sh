set -x
cd $GRAFT_REPO_ROOT
bash tools/pmc_step_kernel.sh r05_pmc_step_cfg2 50000000
python tools/make_traffic_json.py gpurun_out/r05_pmc_step_cfg2/pmc.txt "profiles/r05_pmc_step_kernel_config2.txt (tools/pmc_step_kernel.sh, round 5)" 50000000 100 4450000000
bash tools/pmc_step_kernel.sh r05_pmc_step_target 500000000 --tune emit_budget=8589934592
python tools/make_traffic_json.py gpurun_out/r05_pmc_step_target/pmc.txt "profiles/r05_pmc_step_kernel_target.txt (tools/pmc_step_kernel.sh --tune emit_budget=8589934592, round 5)" 500000000 100 44500000000 emit_budget=8589934592
cp profiles/search_kernel_traffic.json gpurun_out/search_kernel_traffic_final.json
cd $GRAFT_REPO_ROOT
bash tools/final_round.sh r05f
cd $GRAFT_REPO_ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05f/bench_driver.json 2> gpurun_out/r05f/bench_driver.log; tail -c 300 gpurun_out/r05f/bench_driver.json
python bench.py > gpurun_out/r05f/bench_default_run1.json 2> gpurun_out/r05f/bench_default_run1.log; tail -c 300 gpurun_out/r05f/bench_default_run1.json
