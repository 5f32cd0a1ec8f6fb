# HISTORICAL RECORD (rounds 3 / 4): this script rebuilds or REPLACES legion-1_amd/csrc/liblegion_amd.so in place -- a killed run leaves an invalid
# library behind.  Since round 5 a variant library is built with `make -C legion-1_amd/csrc variant VARIANT=... VARIANT_FLAGS=...` and selected through
# $LEGION_LIB (profiles/ab_kernels.sh, profiles/r05_runs_robust.sh); the shipped library is never touched.  Kept as the record of what was run.
[ "${LEGION_RUN_HISTORICAL:-0}" = 1 ] || { echo "$0: historical script that overwrites the shipped library; see its header (LEGION_RUN_HISTORICAL=1 to run it anyway)"; exit 1; }
# round 4, call h: SAFE timing probe of k_write's tile-count prefix: a variant library builds the prefix 1 / 2 / 3 times per workgroup (same result,
# bit-exact output -- checked first); the difference between the rows is what one prefix build costs
O=$GRAFT_REPO_ROOT/gpurun_out/r04v
mkdir -p $O
LIB=$GRAFT_REPO_ROOT/legion-1_amd/csrc/liblegion_amd.so
cp $LIB $O/lib_ship.so
cp $GRAFT_REPO_ROOT/profiles/ab/liblegion_amd_probe.so $LIB
cd $GRAFT_REPO_ROOT
LEGION_WRITE_PROBE=2 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "toy_golden or medium_digests or random_graphs or heavy_duplicate or full_batch_bound" > $O/pytest_probe.log 2>&1 || { tail -20 $O/pytest_probe.log; cp $O/lib_ship.so $LIB; exit 1; }
tail -1 $O/pytest_probe.log
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
 for v in 0 1 2; do
  for wl in "papers100M 25,10,5" "products 25,10,5"; do
    set -- $wl
    export LEGION_WRITE_PROBE=$v
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/$v.$round/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 0.2 --steps 20 > $O/$v.$round.$1.json 2>/dev/null || { cp $O/lib_ship.so $LIB; exit 1; }
    python3 $GRAFT_REPO_ROOT/profiles/hop_table.py $O/$v.$round/$1 "prefix builds=$((v+1)) #$round $1" $O/$v.$round.$1.json
  done
 done
done
cp $O/lib_ship.so $LIB
