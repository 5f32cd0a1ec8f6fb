#!/bin/bash
# Same-box A/B of the hand-off polling ($LEGION_HANDOFF_SPIN_US: 0 = block on the semaphore at once, as in rounds 1-4; 200 = poll for up to 200 us
# first, the default since round 5): bench.py's served legs (the `legion` server binary + a null consumer process), alternating 0 / 200 / 0 / 200.
#   bash profiles/r05_runs_handoff.sh > gpurun_out/r05_handoff_spin.log
cd "$(dirname "$0")/.."
for round in 1 2; do
 for spin in 0 200; do
  LEGION_HANDOFF_SPIN_US=$spin python3 bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --measure-traffic off --extra-legs served,products_2hop,products_3hop 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
def f(name, sv, lv):
    print('spin $spin round $round %-22s served %.4f ms  in-process levels %.4f ms  ratio %.3f  all-batches %.4f ms  equal=%s' % (name, sv['ms_per_step'], lv, sv['ms_per_step'] / lv, sv['all_training_batches_ms_per_step'], sv['served_batches_equal_the_timed_ones']), flush=True)
f('papers100M {25,10,5}', d['extra_legs']['served'], d['alt_schedule_levels']['ms_per_step'])
for k, n in (('products_2hop', 'products {25,10}'), ('products_3hop', 'products {25,10,5}')):
    f(n, d['extra_legs'][k]['served'], d['extra_legs'][k]['ms_per_step_levels'])
"
 done
done
