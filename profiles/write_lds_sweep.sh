# NOTE: LEGION_WRITE_ENTRIES was a temporary env knob of the round-2 sweep (profiles/r02_write_lds_sweep.md); the shipped constant is kWriteEntries = 6144.
# k_write: LDS prefix granularity vs occupancy (temporary knob LEGION_WRITE_ENTRIES), kernel averages from rocprofv3
for e in 12288 6144 3072 1536; do
  d=$GRAFT_REPO_ROOT/gpurun_out/r02p/e$e
  (cd /tmp && TMPDIR=/tmp LEGION_WRITE_ENTRIES=$e rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/bench.py --workload $1 --headline-only --cpu-baseline-seconds 0 --min-time 0.1 > $d.log 2>&1)
  python3 - <<PY
import csv,glob
f=glob.glob("$d/*/*kernel_stats.csv")[0]
r={x["Name"].split("(")[0].replace("void ","").replace("legion::","")[:10]:x for x in csv.DictReader(open(f))}
print("$1 entries $e:", " ".join("%s avg %.1f max %.1f"%(k, float(r[k]["AverageNs"])/1e3, float(r[k]["MaxNs"])/1e3) for k in r if k.startswith(("k_write","k_mark","k_sample"))), flush=True)
PY
done
