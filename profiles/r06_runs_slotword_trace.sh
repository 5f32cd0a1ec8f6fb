# r06: where does the sampler's extra time go in the -DLEGION_SLOT_WORD variant?  per-hop kernel medians (rocprofv3 --kernel-trace) of the cached configuration, shipped vs variant
O=$GRAFT_REPO_ROOT/gpurun_out/r06_slotword_trace; mkdir -p $O
V=$GRAFT_REPO_ROOT/legion-1_amd/csrc/variants/liblegion_amd_slotword.so
cd /tmp && export TMPDIR=/tmp
for which in new var; do
  if [ $which = var ]; then export LEGION_LIB=$V; else unset LEGION_LIB; fi
  for cfg in "cached --cache unified --cache-frac 0.25" "products_cached --workload products --cache unified --cache-frac 0.25"; do
    set -- $cfg; name=$1; shift
    rocprofv3 --kernel-trace --output-format csv -d $O/$which/$name -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 0.3 --steps 20 > $O/$which.$name.json 2>/dev/null || { echo "$which $name FAILED"; continue; }
    python3 $GRAFT_REPO_ROOT/profiles/hop_table.py $O/$which/$name "$which $name" $O/$which.$name.json
  done
done
unset LEGION_LIB
