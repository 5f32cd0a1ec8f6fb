# Same-box A/B of two builds of liblegion_amd.so: the shipped one ("new") against a VARIANT library ("old" / "var"), alternating
# new / var / new / var, per-hop kernel medians from rocprofv3 --kernel-trace.  The variant is selected through $LEGION_LIB
# (legion-1_amd/capi.py lib_path): the shipped library is never copied over or replaced (round 4 swapped it in place; a killed
# run left an invalid library behind on that lease).
#   make -C legion-1_amd/csrc variant VARIANT=unbounded VARIANT_FLAGS=-DLEGION_UNBOUNDED_STORES     # -> csrc/variants/liblegion_amd_unbounded.so
#   bash profiles/ab_kernels.sh <outdir-under-gpurun_out> <absolute path of the variant .so> ["papers100M 25,10,5" "products 25,10,5" ...]
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
VAR=$1; shift
[ -f "$VAR" ] || { echo "variant library $VAR missing"; exit 1; }
mkdir -p $O
[ $# -eq 0 ] && set -- "papers100M 25,10,5" "products 25,10,5"
WLS=("$@")
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
 for which in new var; do
  if [ $which = var ]; then export LEGION_LIB=$VAR; else unset LEGION_LIB; fi
  for wl in "${WLS[@]}"; do
    set -- $wl
    D=$1_${2//,/-}      # one trace directory per (workload, fan-out)
    rocprofv3 --kernel-trace --output-format csv -d $O/$which$round/$D -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 0.3 --steps 20 > $O/$which$round.$D.json 2>/dev/null || exit 1
    python3 - $O/$which$round/$D "$which$round $1 $2" $O/$which$round.$D.json <<'PY'
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
unset LEGION_LIB
