mkdir -p gpurun_out/r03j
python -m pytest tests -m gpu -q -x > gpurun_out/r03j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03j/pytest.log
for wl in "papers100M 25,10,5" "products 25,10,5" "products 25,10" "uk-union 25,10,5" "uk-union 25,10"; do
  set -- $wl
  for ht in 0 auto 16 32; do
    LEGION_HEAD_TABLE=$ht python bench.py --workload $1 --fanout $2 --headline-only --cpu-baseline-seconds 0 --min-time 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1 $2 head=$ht: batch %.4f ms  %.3f G edges/s  sampler %.1f us  gather %.1f us  overlap %s' % (d['ms_per_step'], d['value']/1e9, r['sampler']['us_per_batch'], r['avg_launch_us'], d.get('ms_per_step_overlap')))"
  done
done
