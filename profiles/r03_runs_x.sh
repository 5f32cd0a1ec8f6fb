mkdir -p gpurun_out/r03x
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1: batch %.4f ms  gather %.1f us  frac %.4f' % (d['ms_per_step'], r['avg_launch_us'], r['frac']))"; }
for fan in 25,10 25,10,5; do
for mod in 0 25 9 0 25; do
  LEGION_GATHER_LOADMOD=$mod python bench.py --workload products --fanout $fan --headline-only --cpu-baseline-seconds 0 --min-time 0.5 2>/dev/null | line "products $fan loadmod=$mod"
done; done
LEGION_GATHER_LOADMOD=25 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pitch or full_size" 2>&1 | tail -1
for mod in 0 25; do
  LEGION_GATHER_LOADMOD=$mod python3 profiles/make_pmc_traffic.py r03x products_mod$mod --workload products --fanout 25,10 2>&1 | tail -1
done
