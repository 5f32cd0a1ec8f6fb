# NOTE: LEGION_GATHER_ITERS / LEGION_GATHER_MSHIFT were temporary env knobs of the round-2 sweep (profiles/r02_gather_grid_sweep.md) and no longer exist in launch_gather.
set -e
mkdir -p gpurun_out/r02g
for it in 1 2 3 4; do for ms in 2 4 5; do
  LEGION_GATHER_ITERS=$it LEGION_GATHER_MSHIFT=$ms python bench.py --headline-only --cpu-baseline-seconds 0 --steps 30 --min-time 0.2 > gpurun_out/r02g/g_${it}_${ms}.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r02g/g_${it}_${ms}.log") if l.startswith("{")][-1])
print("iters ${it} mshift ${ms}: ms/step %.4f gather us %.1f frac %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]), flush=True)
PY
done; done
