# Same-box sweep of one environment knob: per-hop kernel medians (rocprofv3 --kernel-trace) for each value, twice, interleaved.
#   bash profiles/knob_sweep.sh <outdir-under-gpurun_out> <ENV_NAME> "<v1> <v2> ..." ["papers100M 25,10,5" ...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; NAME=$2; VALS=$3; shift 3
mkdir -p $O
[ $# -eq 0 ] && set -- "papers100M 25,10,5" "products 25,10,5"
WLS=("$@")
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
 for v in $VALS; do
  for wl in "${WLS[@]}"; do
    set -- $wl
    env $NAME=$v true
    export $NAME=$v
    rocprofv3 --kernel-trace --output-format csv -d $O/$v.$round/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 0.3 --steps 20 > $O/$v.$round.$1.json 2>/dev/null || exit 1
    python3 $GRAFT_REPO_ROOT/profiles/hop_table.py $O/$v.$round/$1 "$NAME=$v #$round $1 $2" $O/$v.$round.$1.json
  done
 done
done
