cd $GRAFT_REPO_ROOT
for legs in products_2hop "served,products_2hop" "lp,products_2hop" "cached_gather,products_2hop"; do
 for st in 20 50; do
  python3 bench.py --steps $st --warmup 5 --cpu-baseline-seconds 0 --measure-traffic off --extra-legs $legs 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['extra_legs']['products_2hop']
print('legs=$legs steps=$st', 'serial', p['ms_per_step'], 'overlap', p['ms_per_step_overlap'], 'levels', p['ms_per_step_levels'], 'served', (p.get('served') or {}).get('ms_per_step'), 'headline levels', d['alt_schedule_levels']['ms_per_step'], flush=True)
"
 done
done
