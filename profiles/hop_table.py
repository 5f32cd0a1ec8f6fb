#!/usr/bin/env python3
"""Per-hop kernel medians (us) of one rocprofv3 --kernel-trace run of bench.py:  python3 profiles/hop_table.py <trace dir> <label> [bench json]"""
import collections
import csv
import glob
import json
import statistics
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
seq = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "legion::k_" not in n or "synth" in n or "copy" in n:
        continue
    seq.append((int(r["Start_Timestamp"]), n.split("legion::")[1].split("<")[0].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
seq.sort()
per, hop = collections.defaultdict(list), 0
for t, n, d in seq:
    if n == "k_seed":
        hop = 0
    elif n == "k_sample":
        hop += 1
    per[(n, hop if n not in ("k_seed", "k_gather", "k_row_ptrs", "k_gather_lookup") else 0)].append(d)
med = {k: round(statistics.median(v), 1) for k, v in sorted(per.items())}
ms = ""
if len(sys.argv) > 3:
    ms = "ms/batch %s " % json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])["ms_per_step"]
print(sys.argv[2], ms + "sampler sum", round(sum(v for k, v in med.items() if not k[0].startswith("k_gather") and k[0] != "k_row_ptrs"), 1),
      " ".join("%s%s=%s" % (k[0][2:], k[1] or "", v) for k, v in med.items()), flush=True)
