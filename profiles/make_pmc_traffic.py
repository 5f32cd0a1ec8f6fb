#!/usr/bin/env python3
"""Collect HBM traffic per kernel with rocprofv3 PMC counters and write profiles/<tag>_pmc_hbm_traffic.json.

Run on the GPU box from the repository root:   python3 profiles/make_pmc_traffic.py r01
Another shape:   python3 profiles/make_pmc_traffic.py r03 products_2hop --workload products --fanout 25,10
(writes profiles/r03_pmc_hbm_traffic_products_2hop.json; bench.py only reads the un-suffixed file of the default workload)
One rocprofv3 pass per counter (FETCH_SIZE, WRITE_SIZE), no trace domains, the program itself after `--`.
Units / corrections as prescribed in MI355X_MICROARCH.md (HBM section): rocprofv3 reports both counters in KiB;
on gfx950 FETCH_SIZE counts half of the bytes of wide (16 B/lane) coalesced reads -> x2 for k_gather; the scattered
4/8-byte accesses of the sampler kernels are reported raw (uncalibrated), so their ratio is an upper-bound indicator."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
suffix = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = sys.argv[(3 if suffix else 2):]
out_dir = os.path.join(ROOT, "gpurun_out", "pmc_traffic" + ("_" + suffix if suffix else ""))
os.makedirs(out_dir, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
per = collections.defaultdict(dict)
bench = None
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join(out_dir, counter)
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
           "--steps", "10", "--warmup", "2", "--min-time", "0", "--headline-only", "--cpu-baseline-seconds", "0", "--measure-traffic", "off"] + extra
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit(r.stdout[-2000:] + r.stderr[-2000:])
    bench = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    vals = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "legion::k_" in k and "synth" not in k and "copy" not in k and row["Counter_Name"] == counter:
            vals[k.replace("void ", "").replace("legion::", "").split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        per[k]["launches"] = len(v)
        per[k][counter + "_KiB_avg"] = round(sum(v) / len(v), 1)
batches = next(v["launches"] for k, v in per.items() if k.startswith("k_seed"))   # warm-up + census + one timed window
g = next(k for k in per if k.startswith("k_gather"))
read = per[g]["FETCH_SIZE_KiB_avg"] * 1024 * 2
write = per[g]["WRITE_SIZE_KiB_avg"] * 1024
alg = bench["gather_algorithmic_bytes_per_batch"]
samp = sum((v.get("FETCH_SIZE_KiB_avg", 0) + v.get("WRITE_SIZE_KiB_avg", 0)) * 1024 * v["launches"] for k, v in per.items()
           if not k.startswith("k_gather")) / batches
doc = {
    "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 10 --warmup 2 --min-time 0 --headline-only --cpu-baseline-seconds 0 "
               + " ".join(extra) + " (one pass per counter: FETCH_SIZE, WRITE_SIZE; profiles/make_pmc_traffic.py)",
    "workload": bench["config"]["workload"],
    "units": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction: FETCH_SIZE x2 for the 16 B/lane reads of k_gather; sampler kernels raw",
    "kernels": per,
    "k_gather": {"algorithmic_bytes_per_launch": alg, "hbm_read_bytes_corrected": int(read), "hbm_write_bytes": int(write),
                 "traffic_bytes_per_launch": int(read + write), "traffic_over_algorithmic": round((read + write) / alg, 4)},
    "sampler": {"algorithmic_bytes_per_batch": bench["sampler_algorithmic_bytes_per_batch"], "raw_counter_bytes_per_batch": int(samp),
                "raw_over_algorithmic": round(samp / bench["sampler_algorithmic_bytes_per_batch"], 3),
                "note": "20 N + 28 E + 8 U per hop counts 4/8-byte elements; every scattered element moves a 32/64-byte sector"},
}
path = os.path.join(ROOT, "profiles", tag + "_pmc_hbm_traffic" + ("_" + suffix if suffix else "") + ".json")
json.dump(doc, open(path, "w"), indent=1)
json.dump(doc, open(os.path.join(out_dir, os.path.basename(path)), "w"), indent=1)   # gpurun_out/ travels back from the GPU box, profiles/ does not
print(json.dumps(doc["k_gather"]), json.dumps(doc["sampler"]))
