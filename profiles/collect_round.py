#!/usr/bin/env python3
"""Regenerate the per-round evidence in one go (run on the GPU box from the repository root):

    python3 profiles/collect_round.py r02

writes profiles/<tag>_bench_line.json, <tag>_bench_papers100M_kernel_stats.csv, <tag>_bench_papers100M_summary.md and
<tag>_pmc_hbm_traffic.json (bench.py reads the newest *_pmc_hbm_traffic.json of the default workload).
rocprofv3 is always given the program itself after `--`; PMC passes run without trace domains."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
scratch = os.path.join(ROOT, "gpurun_out", "collect_" + tag)
os.makedirs(scratch, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")


def last_json(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


# 1. kernel table of the timed (serial) schedule
prof_dir = os.path.join(scratch, "stats")
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", prof_dir, "--", "python3",
                    os.path.join(ROOT, "bench.py"), "--headline-only", "--cpu-baseline-seconds", "0"], cwd="/tmp", env=env,
                   capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stdout[-2000:] + r.stderr[-2000:])
profiled = last_json(r.stdout)
stats = glob.glob(os.path.join(prof_dir, "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(ROOT, "profiles", tag + "_bench_papers100M_kernel_stats.csv"))

# 2. the default bench line (all legs, CPU baseline)
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], cwd=ROOT, capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stdout[-2000:] + r.stderr[-2000:])
line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
open(os.path.join(ROOT, "profiles", tag + "_bench_line.json"), "w").write(line + "\n")
d = json.loads(line)
rf = d["roofline"]

out = ["# %s: kernel table + bench line, papers100M shape\n" % tag,
       "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --headline-only --cpu-baseline-seconds 0`",
       "(batch 8000, fan-out 25,10,5; 5 warm-up + 50 census batches + R windows of 50 timed batches of the serial schedule; MI355X)\n",
       "profiled run: %.4f ms/batch, k_gather avg %.2f us by HIP events;" % (profiled["ms_per_step"], profiled["roofline"]["avg_launch_us"]),
       "un-profiled default run (`%s_bench_line.json`): %.4f ms/batch, k_gather %.1f us = %.0f GB/s = %.3f of 8 TB/s, whole batch %.3f, %.2f G edges/s, %.0f GB/s of rows;" % (
           tag, d["ms_per_step"], rf["avg_launch_us"], rf["achieved"], rf["frac"], rf["pipeline_frac"], d["value"] / 1e9, d["feature_GBps"]),
       "overlapped schedule %s ms/batch, hipGraph replay %s ms/batch.\n" % ((d.get("alt_schedule") or {}).get("ms_per_step"), (d.get("graph_replay") or {}).get("ms_per_step")),
       "| kernel | calls | avg us | max us | total ms | % |", "|---|---|---|---|---|---|"]
for row in csv.DictReader(open(stats)):
    n = row["Name"]
    if "legion::" not in n:
        continue
    out.append("| %s | %s | %.1f | %.1f | %.2f | %s |" % (n.replace("void ", "").replace("legion::", "").split("(")[0][:40], row["Calls"],
                                                      float(row["AverageNs"]) / 1e3, float(row["MaxNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e6, row["Percentage"]))
open(os.path.join(ROOT, "profiles", tag + "_bench_papers100M_summary.md"), "w").write("\n".join(out) + "\n")

# 3. HBM traffic from the PMC passes
r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "make_pmc_traffic.py"), tag], cwd=ROOT, capture_output=True, text=True)
print(r.stdout[-600:], r.stderr[-300:])
print("\n".join(out[3:6]))

# gpurun only merges gpurun_out/ back: leave copies of everything written to profiles/ there
for f in glob.glob(os.path.join(ROOT, "profiles", tag + "_*")):
    shutil.copy(f, scratch)
print("copies of profiles/%s_* in %s" % (tag, scratch))
