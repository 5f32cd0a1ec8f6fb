# Same-box A/B of two builds of liblegion_amd.so (the shipped one vs profiles/ab/liblegion_amd_old.so, built from an older
# kernels.hip): per-hop kernel medians from rocprofv3 --kernel-trace, alternating new / old / new / old.
#   bash profiles/ab_kernels.sh <outdir-under-gpurun_out> ["papers100M 25,10,5" "products 25,10,5" ...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
LIB=$GRAFT_REPO_ROOT/legion-1_amd/csrc/liblegion_amd.so
cp $LIB $O/lib_new.so
[ $# -eq 0 ] && set -- "papers100M 25,10,5" "products 25,10,5"
WLS=("$@")
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
 for which in new old; do
  if [ $which = old ]; then cp $GRAFT_REPO_ROOT/profiles/ab/liblegion_amd_old.so $LIB; else cp $O/lib_new.so $LIB; fi
  for wl in "${WLS[@]}"; do
    set -- $wl
    rocprofv3 --kernel-trace --output-format csv -d $O/$which$round/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 0.3 --steps 20 > $O/$which$round.$1.json 2>/dev/null || exit 1
    python3 - $O/$which$round/$1 "$which$round $1 $2" $O/$which$round.$1.json <<'PY'
import csv, glob, collections, statistics, sys, json
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
seq = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "legion::k_" not in n or "synth" in n or "copy" in n: continue
    seq.append((int(r["Start_Timestamp"]), n.split("legion::")[1].split("<")[0].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
seq.sort()
per, hop = collections.defaultdict(list), 0
for t, n, d in seq:
    if n == "k_seed": hop = 0
    elif n == "k_sample": hop += 1
    per[(n, hop if n not in ("k_seed", "k_gather") else 0)].append(d)
med = {k: round(statistics.median(v), 1) for k, v in sorted(per.items())}
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[2], "ms/batch", d["ms_per_step"], "sampler sum", round(sum(v for k, v in med.items() if k[0] != "k_gather"), 1), med, flush=True)
PY
  done
 done
done
cp $O/lib_new.so $LIB
