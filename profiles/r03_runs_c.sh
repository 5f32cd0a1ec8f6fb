# round-3 evidence, run C (one MI355X): sampler bucket probe, the round's kernel table + bench line + PMC traffic,
# BASELINE config 1's shape with the CPU legs, PMC traffic of the F = 100 gather
set -e
mkdir -p gpurun_out/r03c
export TMPDIR=/tmp
( cd /tmp && hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/profiles/bucket_probe.hip -o /tmp/bucket_probe && /tmp/bucket_probe > $GRAFT_REPO_ROOT/gpurun_out/r03c/bucket_probe.log 2>&1 )
echo probe done
python3 profiles/collect_round.py r03 > gpurun_out/r03c/collect.log 2>&1; echo collect rc=$?
python bench.py --workload products --fanout 25,10 > gpurun_out/r03c/bench_products_2hop.json 2> gpurun_out/r03c/bench_products_2hop.err; echo products rc=$?
python3 profiles/make_pmc_traffic.py r03 products_2hop --workload products --fanout 25,10 > gpurun_out/r03c/pmc_products.log 2>&1; echo pmc rc=$?
cp profiles/r03_* gpurun_out/r03c/ 2>/dev/null || true
