cd $GRAFT_REPO_ROOT
BWTM_REQUIRE_FRESH_PMC=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_command_final.json 2> gpurun_out/r06_bench_driver_command_final.log; tail -1 gpurun_out/r06_bench_driver_command_final.log | cut -c1-400
