# sweep of the cache-policy bits on k_gather's 16-byte row loads at F = 100 (products shape) and F = 128 (papers100M shape)
# usage: bash profiles/gather_loadmod_sweep.sh > gpurun_out/<dir>/loadmod.log
for wl in "products 25,10" "products 25,10,5" "papers100M 25,10,5"; do
  set -- $wl
  for mod in 0 1 2 3 4 5 6 7 8 18 19 20 21 22 23; do
    LEGION_GATHER_LOADMOD=$mod python bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --min-time 0.5 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1 $2 loadmod $mod: gather %.1f us  frac %.4f  batch %.4f ms' % (r['avg_launch_us'], r['frac'], d['ms_per_step']))"
  done
done
