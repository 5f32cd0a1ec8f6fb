mkdir -p gpurun_out/r03q
cd /tmp && export TMPDIR=/tmp
for wl in "papers100M 25,10,5" "products 25,10,5" "uk-union 25,10,5"; do
  set -- $wl
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03q/$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --min-time 0.3 --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/r03q/$1.json 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, statistics
for wl in ("papers100M", "products", "uk-union"):
    f = glob.glob("gpurun_out/r03q/%s/*/*kernel_trace.csv" % wl)[0]
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
    print(wl, {k: round(statistics.median(v), 1) for k, v in sorted(per.items())})
PY
