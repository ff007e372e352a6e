cd $GRAFT_REPO_ROOT
BWTM_TUNE=recs_uniform=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parts.py tests/test_gpu_branches.py -x -q 2>&1 | tail -8
for cov in 300 30; do for g in 0 -1; do
timeout 500 python bench.py --workload genome --coverage $cov --steps 3 --warmup 1 --no-cpu-baseline --no-host --target off --tune recs_uniform=$g > gpurun_out/r06_genome${cov}_u$g.json 2> gpurun_out/r06_genome${cov}_u$g.log
python -c "
import json; d=json.load(open('gpurun_out/r06_genome${cov}_u$g.json')); k=d['kernel_ms_per_step']; print($cov, $g, d['ms_per_step'], d['verified'], {n:k[n] for n in ('build_recs','block_len','enc_emit','enc_size') if n in k})"
done; done
for g in 1 -1; do
timeout 500 python bench.py --reads 20000000 --steps 3 --warmup 1 --no-cpu-baseline --no-host --target off --tune recs_uniform=$g > gpurun_out/r06_iid20_u$g.json 2> gpurun_out/r06_iid20_u$g.log
python -c "
import json; d=json.load(open('gpurun_out/r06_iid20_u$g.json')); k=d['kernel_ms_per_step']; print('iid20M', $g, d['ms_per_step'], d['verified'], {n:k[n] for n in ('build_recs','block_len','enc_emit','enc_size') if n in k})"
done
for g in 1 -1; do
timeout 500 python bench.py --workload genome --coverage 5 --steps 3 --warmup 1 --no-cpu-baseline --no-host --target off --tune recs_uniform=$g > gpurun_out/r06_genome5_u$g.json 2> gpurun_out/r06_genome5_u$g.log
python -c "
import json; d=json.load(open('gpurun_out/r06_genome5_u$g.json')); k=d['kernel_ms_per_step']; print('genome5', $g, d['ms_per_step'], d['verified'], d['config']['native_bytes'], {n:k[n] for n in ('build_recs','block_len','enc_emit','enc_size') if n in k})"
done
