export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for v in 1 0; do
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pmc_x
  timeout 600 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_x -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-host --target off --steps 1 --warmup 0 --tune search_view=$v > /tmp/pmc_x.log 2>&1
  f=$(find /tmp/pmc_x -name "*counter_collection.csv" | head -1)
  echo "search_view=$v"; python3 $R/tools/pmc_aggregate.py $f | grep -E "k_frontier_step|k_view"
done; done
