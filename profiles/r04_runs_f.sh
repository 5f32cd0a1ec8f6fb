# HISTORICAL RECORD (rounds 3 / 4): this script rebuilds or REPLACES legion-1_amd/csrc/liblegion_amd.so in place -- a killed run leaves an invalid
# library behind.  Since round 5 a variant library is built with `make -C legion-1_amd/csrc variant VARIANT=... VARIANT_FLAGS=...` and selected through
# $LEGION_LIB (profiles/ab_kernels.sh, profiles/r05_runs_robust.sh); the shipped library is never touched.  Kept as the record of what was run.
[ "${LEGION_RUN_HISTORICAL:-0}" = 1 ] || { echo "$0: historical script that overwrites the shipped library; see its header (LEGION_RUN_HISTORICAL=1 to run it anyway)"; exit 1; }
# round 4, call f: timing-only probes of k_write (results INVALID; a variant library that is not shipped): which part of the kernel costs what
O=$GRAFT_REPO_ROOT/gpurun_out/r04m
mkdir -p $O
LIB=$GRAFT_REPO_ROOT/legion-1_amd/csrc/liblegion_amd.so
cp $LIB $O/lib_ship.so
cp $GRAFT_REPO_ROOT/profiles/ab/liblegion_amd_probe.so $LIB
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
 for v in 0 1; do   # 1 = no loser links.  (2 = no prefix build writes through uninitialised offsets and hangs: removed)
  for wl in "papers100M 25,10,5" "products 25,10,5"; do
    set -- $wl
    export LEGION_WRITE_PROBE=$v
    rocprofv3 --kernel-trace --output-format csv -d $O/$v.$round/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 0.2 --steps 20 > $O/$v.$round.$1.json 2>/dev/null || { cp $O/lib_ship.so $LIB; exit 1; }
    python3 $GRAFT_REPO_ROOT/profiles/hop_table.py $O/$v.$round/$1 "probe=$v #$round $1" $O/$v.$round.$1.json
  done
 done
done
cp $O/lib_ship.so $LIB
