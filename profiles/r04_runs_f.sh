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
