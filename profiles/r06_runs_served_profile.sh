# round 6: kernel table of the `legion` SERVER PROCESS while it serves the papers100M {25,10,5} shape to a READING consumer (bench.py --served-consumer ... reading: every served row and both COO arrays summed on the consumer's stream before the pipe goes back) (rocprofv3 --kernel-trace --stats on the server
# binary itself; no counters): what the kernels cost under the server's two-stream schedule, next to the serial table of r05_bench_papers100M_summary.md.
#   bash profiles/r06_runs_served_profile.sh      (GPU box, repository root)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06_served_prof
rm -rf $O; mkdir -p $O
T=$(mktemp -d /tmp/legion_sp6_XXXX)
echo "synth:papers100M 8000 111059956 0 128 11105995 512 512 0 2 0" > $T/meta_config
export LEGION_IPC_NAMESPACE=sp$$_ HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $R/legion-1_amd/csrc/legion 1 0 25,10,5 $T/meta_config > $O/server.log 2>&1 &
SRV=$!
for i in $(seq 1 600); do grep -q "System is ready for serving" $O/server.log && break; kill -0 $SRV 2>/dev/null || { echo "server died"; tail -n 5 $O/server.log; exit 1; }; sleep 0.2; done
python3 $R/bench.py --served-consumer 3 2 -1 reading 128 papers100M 1.0 > $O/consumer.json || { kill $SRV; exit 1; }
wait $SRV
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
d = json.loads(open(o + "/consumer.json").read().strip().splitlines()[-1])      # (the library prints "IPC shared memory opened" first)
t = d["t"]; ts = d["steps"][0]
print("served: %d batches, %.4f ms per training batch over epoch 2 (consumer clock)" % (len(t), (t[2 * ts] - t[ts + 1]) / (ts - 1) * 1e3))
f = glob.glob(o + "/prof/*/*kernel_stats.csv")[0]
print("| kernel | calls | avg us | max us | total ms | % |\n|---|---|---|---|---|---|")
for r in csv.DictReader(open(f)):
    if "legion::" in r["Name"]:
        print("| %s | %s | %.1f | %.1f | %.2f | %s |" % (r["Name"].replace("void ", "").replace("legion::", "").split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
cp $(ls $O/prof/*/*kernel_stats.csv | head -n 1) $O/r06_served_kernel_stats.csv
rm -rf $O/prof $T
