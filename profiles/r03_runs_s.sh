# blocked-gather experiment ($LEGION_GATHER_BLOCKED=<U>[,<workgroups per CU>]): parity, then A/B at several shapes
mkdir -p gpurun_out/r03s
LEGION_GATHER_BLOCKED=4 python -m pytest tests/test_gpu_parity.py tests/test_gpu_unified_ipc.py -m gpu -q -x > gpurun_out/r03s/pytest_blocked.log 2>&1; echo "parity with blocked gather rc=$?"; tail -1 gpurun_out/r03s/pytest_blocked.log
line() { python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1: batch %.4f ms  gather(+lookup) %.1f us  frac %.4f' % (d['ms_per_step'], r['avg_launch_us'], r['frac']))"; }
for cfg in "" "--cache unified" "--cache unified --cache-frac 1.0" "--workload products --fanout 25,10,5" "--workload uk-union --fanout 25,10"; do
  for b in 0 1 2,8 4,8 8,8 4,4 4,16 4,32 2,16; do
    LEGION_GATHER_BLOCKED=$b python bench.py $cfg --headline-only --cpu-baseline-seconds 0 --min-time 0.5 2>/dev/null | line "[$cfg] blocked=$b"
  done
done
