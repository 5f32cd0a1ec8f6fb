line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1: batch %.4f ms  gather %.1f us  frac(alg) %.4f' % (d['ms_per_step'], r['avg_launch_us'], r['frac']))"; }
for fan in 25,10 25,10,5; do
for p in 0 1 0 1; do
  if [ $p = 1 ]; then export LEGION_SPLIT_PROBE=1; else unset LEGION_SPLIT_PROBE; fi
  python bench.py --workload products --fanout $fan --headline-only --cpu-baseline-seconds 0 --min-time 0.5 2>/dev/null | line "products $fan split_probe=$p"
done; done
export LEGION_SPLIT_PROBE=1
python3 profiles/make_pmc_traffic.py r03x products_split --workload products --fanout 25,10 2>&1 | tail -1
