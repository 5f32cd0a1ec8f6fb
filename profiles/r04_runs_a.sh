# round 4, call a: (1) cached gather: lookup pass vs fused lookups; (2) products 3-hop per-hop kernel table; (3) SQ + HBM counters at that shape
# run on the GPU box from the repository root:  bash profiles/r04_runs_a.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r04c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "randomised_clique_cache or presampling" > $O/pytest_cache.log 2>&1 || { tail -30 $O/pytest_cache.log; exit 1; }
tail -2 $O/pytest_cache.log
for mode in pass fused1 fused2 fused4 pass fused2; do
  LEGION_GATHER_LOOKUP=$mode timeout -k 10 200 python bench.py --steps 20 --min-time 0.5 --extra-min-time 2 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs cached_gather > $O/cached_$mode.json 2>$O/cached_$mode.err || exit 1
  python3 - $O/cached_$mode.json $mode <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])["extra_legs"]["cached_gather"]
print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "gather_avg_launch_us", "gather_frac_of_hbm_peak", "sampler_us_per_batch", "rows_last_batch")})
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload products --fanout 25,10,5 --headline-only --cpu-baseline-seconds 0 --measure-traffic off --min-time 0.3 --steps 20 > $O/p3.json 2>/dev/null || exit 1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, statistics
f = glob.glob("gpurun_out/r04c/p3/*/*kernel_trace.csv")[0]
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
print("products 25,10,5 median us", {k: round(statistics.median(v), 1) for k, v in sorted(per.items())})
PY
python3 profiles/pmc_sq.py --workload products --fanout 25,10,5 --measure-traffic off > $O/pmc_sq_products_3hop.log 2>&1; tail -14 $O/pmc_sq_products_3hop.log
python3 profiles/make_pmc_traffic.py r04 products_3hop --workload products --fanout 25,10,5 --measure-traffic off > $O/pmc_traffic_products_3hop.log 2>&1; tail -3 $O/pmc_traffic_products_3hop.log
