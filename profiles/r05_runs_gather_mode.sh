#!/bin/bash
# Same-box A/B of the runner's gather formulation ($LEGION_RUNNER_GATHER: level = one FeatureExtractor op per level behind each hop, the reference's op
# list; all = one gather over all rows behind the last hop): bench.py's served legs, alternating level / all / level / all.
#   bash profiles/r05_runs_gather_mode.sh > gpurun_out/r05_runner_gather.log
cd "$(dirname "$0")/.."
for round in 1 2; do
 for spin in level all; do
  LEGION_RUNNER_GATHER=$spin python3 bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --measure-traffic off --extra-legs served,products_2hop,products_3hop 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
def f(name, sv, lv):
    print('gather $spin round $round %-22s served %.4f ms  in-process levels %.4f ms  ratio %.3f  all-batches %.4f ms  equal=%s' % (name, sv['ms_per_step'], lv, sv['ms_per_step'] / lv, sv['all_training_batches_ms_per_step'], sv['served_batches_equal_the_timed_ones']), flush=True)
f('papers100M {25,10,5}', d['extra_legs']['served'], d['alt_schedule_levels']['ms_per_step'])
for k, n in (('products_2hop', 'products {25,10}'), ('products_3hop', 'products {25,10,5}')):
    f(n, d['extra_legs'][k]['served'], d['extra_legs'][k]['ms_per_step_levels'])
"
 done
done
