set -e
mkdir -p gpurun_out/r03b
cd /tmp && export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/profiles/bucket_probe.hip -o /tmp/bucket_probe && /tmp/bucket_probe > $GRAFT_REPO_ROOT/gpurun_out/r03b/bucket_probe.log 2>&1
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "presampling_cache_pipeline" > gpurun_out/r03b/pytest_peer.log 2>&1 || true
for fan in 25,10 25,10,5; do
for pitch in auto dense; do
for lanes in 0 1; do
  LEGION_GATHER_ROW_LANES=$lanes python bench.py --workload products --fanout $fan --row-pitch $pitch --headline-only --cpu-baseline-seconds 0 --min-time 1 > gpurun_out/r03b/products_${fan}_${pitch}_lanes${lanes}.json 2> gpurun_out/r03b/products_${fan}_${pitch}_lanes${lanes}.err
done; done; done
echo benches done
LEGION_BENCH_FORCE_DEVICE=0 timeout -k 10 600 python bench.py --gpus 4 --scale 0.2 --steps 20 > gpurun_out/r03b/rehearsal4.json 2> gpurun_out/r03b/rehearsal4.err; echo rehearsal rc=$?
