# r06: the cached-gather experiment of VERDICT r05 next 8 (-DLEGION_SLOT_WORD): parity through the variant library, then a same-box A/B of
#   (a) the cached configuration (25 % of the rows in a Kg = 1 shard: bench.py --cache unified) and (b) the replicated headline, new / var / new / var.
#   make -C legion-1_amd/csrc variant VARIANT=slotword VARIANT_FLAGS=-DLEGION_SLOT_WORD      (build container)
O=gpurun_out/r06_slotword; mkdir -p $O
V=$GRAFT_REPO_ROOT/legion-1_amd/csrc/variants/liblegion_amd_slotword.so
LEGION_LIB=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_shape.py -m gpu -q -x -k "not server and not ipc" > $O/parity.log 2>&1; echo "variant parity rc=$?" | tee -a $O/parity.log
tail -3 $O/parity.log
for round in 1 2; do
 for which in new var; do
  if [ $which = var ]; then export LEGION_LIB=$V; else unset LEGION_LIB; fi
  for cfg in "cached --cache unified --cache-frac 0.25" "replicated --cache replicated" "products_cached --workload products --cache unified --cache-frac 0.25"; do
    set -- $cfg; name=$1; shift
    python3 bench.py "$@" --headline-only --cpu-baseline-seconds 0 --measure-traffic off --extra-legs none --min-time 1.5 --steps 20 > $O/$which$round.$name.json 2>/dev/null || { echo "$which$round $name FAILED"; continue; }
    python3 - $O/$which$round.$name.json "$which$round $name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(sys.argv[2], "ms/batch", d["ms_per_step"], "gather us", r["avg_launch_us"], "sampler us", r["sampler"]["us_per_batch"], "overlap ms", d.get("ms_per_step_overlap"), flush=True)
PY
  done
 done
done
unset LEGION_LIB
