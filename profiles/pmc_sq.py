#!/usr/bin/env python3
"""Where does k_sample's time go?  SQ counters per kernel and hop, one rocprofv3 --pmc pass per counter group (no trace domains).
   python3 profiles/pmc_sq.py [l2] [bench.py flags, e.g. --workload products --fanout 25,10,5] > gpurun_out/<dir>/pmc_sq.log
   (on the GPU box, from the repository root; default: the papers100M headline shape.  `l2`: the L2 / memory-side request counters
   -- TCC hits, misses, EA reads / writes / atomics -- instead of the SQ groups)"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
          ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_ANY"],
          ["SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_INT32"],
          ["SQ_WAVES", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SMEM"]]
ARGS = sys.argv[1:]
if ARGS and ARGS[0] == "l2":
    ARGS = ARGS[1:]
    GROUPS = [["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"], ["TCC_EA0_ATOMIC_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"],
              ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum"]]
env = dict(os.environ, TMPDIR="/tmp")
out = collections.defaultdict(dict)
for gi, grp in enumerate(GROUPS):
    d = os.path.join(ROOT, "gpurun_out", "pmc_sq", "g%d" % gi)
    cmd = ["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
                                         "--steps", "6", "--warmup", "2", "--min-time", "0", "--headline-only", "--cpu-baseline-seconds", "0", "--measure-traffic", "off"] + ARGS
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print("group", gi, "failed:", r.stderr[-800:])
        continue
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    # dispatch order -> hop: k_seed starts a batch, every k_sample starts a hop
    disp = {}
    for row in rows:
        disp[int(row["Dispatch_Id"])] = row["Kernel_Name"]
    hop_of, hop = {}, 0
    for did in sorted(disp):
        n = disp[did]
        if "k_seed" in n:
            hop = 0
        elif "k_sample" in n:
            hop += 1
        hop_of[did] = hop
    acc = collections.defaultdict(list)
    for row in rows:
        n = row["Kernel_Name"]
        if "legion::k_" not in n or "synth" in n or "copy" in n:
            continue
        key = (n.split("legion::")[1].split("<")[0].split("(")[0], hop_of[int(row["Dispatch_Id"])] if "k_gather" not in n and "k_seed" not in n else 0)
        acc[(key, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (key, c), v in acc.items():
        out[key][c] = sum(v) / len(v)
names = [c for g in GROUPS for c in g]
print("%-12s %3s " % ("kernel", "hop") + " ".join("%14s" % c.replace("SQ_", "").replace("_sum", "")[-14:] for c in names))
for key in sorted(out):
    print("%-12s %3d " % key + " ".join("%14.4g" % out[key].get(c, float("nan")) for c in names))
