#!/usr/bin/env python3
"""Where does a recorded batch graph lose against the plain two-stream op loop of the `legion` server?  

For each RunOnce variant -- plain pipelined loop, LEGION_BATCH_GRAPH=1 (graph recorded on one stream), LEGION_BATCH_GRAPH=2
(fork/join graph: the two-stream loop as recorded) -- the server binary runs under `rocprofv3 --hip-trace --kernel-trace`
(no counters; the program itself after `--`) with a null consumer attached, and the trace is reduced to, per steady-state
training batch: the period (k_seed to k_seed), the sum of the kernel durations, the time no kernel runs, how many kernels
overlap another one, and the host time of the launch calls.

    python3 profiles/graph_trace.py [--fanout 25,10] [--scale 0.3] > gpurun_out/<dir>/graph_trace.log   (GPU box, repository root)"""
import argparse
import collections
import csv
import glob
import os
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(d, hops):
    kt = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    ht = glob.glob(os.path.join(d, "*", "*hip_api_trace.csv"))
    ks = []
    for r in csv.DictReader(open(kt[0])):
        n = r["Kernel_Name"]
        if "legion::k_" not in n:
            continue
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("legion::")[1].split("<")[0].split("(")[0]))
    ks.sort()
    # batches: from one k_seed to the next; steady state = full training batches (the most common kernel count)
    starts = [i for i, k in enumerate(ks) if k[2] == "k_seed"]
    batches = [ks[a:b] for a, b in zip(starts, starts[1:])]
    batches = [b for b in batches if not any(k[2] == "k_hotness" for k in b)]      # the pre-sampling epoch (sampler + k_hotness, one stream) is not steady state
    common = collections.Counter(len(b) for b in batches).most_common(1)[0][0]
    batches = [b for b in batches if len(b) == common]
    batches = batches[min(20, len(batches) // 4):len(batches) - 2]
    period = [b2[0][0] - b1[0][0] for b1, b2 in zip(batches, batches[1:]) if b2[0][0] - b1[0][0] < 5e6]
    busy, idle, ovl, ksum = [], [], [], []
    for b in batches:
        ev = sorted([(s, 1) for s, e, _ in b] + [(e, -1) for s, e, _ in b])
        depth, last, t_busy, t_ovl = 0, None, 0, 0
        for t, dlt in ev:
            if depth > 0:
                t_busy += t - last
            if depth > 1:
                t_ovl += t - last
            depth += dlt
            last = t
        span = max(e for _, e, _ in b) - b[0][0]
        busy.append(t_busy); idle.append(span - t_busy); ovl.append(t_ovl); ksum.append(sum(e - s for s, e, _ in b))
    per_k = collections.defaultdict(list)
    for b in batches:
        hop = 0
        for s, e, n in b:
            if n == "k_sample":
                hop += 1
            per_k[(n, hop if n in ("k_sample", "k_mark", "k_write") else 0)].append((e - s) / 1e3)
    api = collections.defaultdict(list)
    if ht:
        for r in csv.DictReader(open(ht[0])):
            api[r["Function"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = statistics.median
    out = {"kernels_per_batch": common, "batches": len(batches), "period_us": round(med(period) / 1e3, 1), "kernel_sum_us": round(med(ksum) / 1e3, 1),
           "gpu_busy_us": round(med(busy) / 1e3, 1), "no_kernel_running_us_inside_batch_span": round(med(idle) / 1e3, 1),
           "two_kernels_running_us": round(med(ovl) / 1e3, 1)}
    calls = {f: (len(v), round(med(v), 1), round(sum(v) / max(1, len(batches) + 25) , 1)) for f, v in api.items()
             if f in ("hipGraphLaunch", "hipLaunchKernel", "hipEventRecord", "hipStreamWaitEvent", "hipEventSynchronize", "hipEventQuery", "hipStreamSynchronize", "hipExtLaunchKernel")}
    return out, calls, {k: round(med(v), 1) for k, v in sorted(per_k.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="products")
    ap.add_argument("--scale", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--fanout", default="25,10")
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "graph_trace"))
    ap.add_argument("--variants", default="plain,graph1,graph2,graph3",
                    help="plain (LEGION_RUNNER_GATHER=auto) | level | all (the plain loop with that gather formulation, round 5) | graph1 | graph2 | graph3")
    ap.add_argument("--source", default="files", choices=["files", "synth"], help="synth: the server generates the tables (meta_config `synth:<workload>:<scale>`): full shapes")
    a = ap.parse_args()
    a.out = os.path.abspath(a.out)
    import dataclasses
    import legion1_amd.synth as S
    tmp = tempfile.mkdtemp(prefix="legion_gt_")
    meta = os.path.join(tmp, "meta_config")
    if a.source == "synth":
        spec = S.spec_for(a.workload, scale=a.scale)
        with open(meta, "w") as f:
            f.write("synth:%s:%r %d %d 0 %d %d 512 512 0 %d 0" % (a.workload, a.scale, a.batch, spec.V, spec.F, spec.n_train, a.epochs))
    else:
        ds = S.generate(S.spec_for(a.workload, scale=a.scale))
        ds.valid, ds.test = ds.valid[:512], ds.test[:512]
        ds.spec = dataclasses.replace(ds.spec, n_valid=len(ds.valid), n_test=len(ds.test))
        data = os.path.join(tmp, "ds") + "/"
        S.write_legion_files(ds, data)
        with open(meta, "w") as f:
            f.write(S.meta_config_line(ds, data, a.batch, 1 << 40, a.epochs, 0))
    server = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")
    hops = len(a.fanout.split(","))
    for name, extra, traced in (("level", {"LEGION_RUNNER_GATHER": "level"}, False), ("all", {"LEGION_RUNNER_GATHER": "all"}, False),
                                ("level", {"LEGION_RUNNER_GATHER": "level"}, True), ("all", {"LEGION_RUNNER_GATHER": "all"}, True),
                                ("plain", {}, False), ("graph1", {"LEGION_BATCH_GRAPH": "1"}, False), ("graph2", {"LEGION_BATCH_GRAPH": "2"}, False), ("graph3", {"LEGION_BATCH_GRAPH": "3"}, False),
                                ("plain", {}, True), ("graph1", {"LEGION_BATCH_GRAPH": "1"}, True), ("graph2", {"LEGION_BATCH_GRAPH": "2"}, True), ("graph3", {"LEGION_BATCH_GRAPH": "3"}, True)):
        if name not in a.variants.split(","):
            continue
        ns = "gt%d_%s%d_" % (os.getpid(), name, traced)
        env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", TMPDIR="/tmp", **extra)
        d = os.path.join(a.out, name)
        log = open(os.path.join(tmp, "server_%s%d.log" % (name, traced)), "w")
        cmd = [server, "1", "0", a.fanout, meta]
        if traced:
            cmd = ["rocprofv3", "--hip-trace", "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd
        proc = subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, cwd=tmp)
        t0 = time.time()
        while "System is ready for serving" not in open(log.name).read():
            if proc.poll() is not None or time.time() - t0 > 300:
                raise SystemExit("server died / not ready:\n" + open(log.name).read()[-2000:])
            time.sleep(0.2)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "serve_bench.py"), "--consume", str(a.epochs), "--fanout", a.fanout],
                             env=env, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            raise SystemExit(out.stdout[-2000:] + out.stderr[-2000:])
        total, dt, edges = out.stdout.strip().splitlines()[-1].split()
        proc.wait(timeout=120)
        print("%-7s %s %5d batches  %.3f ms/batch as the consumer sees it" % (name, "traced  " if traced else "untraced", int(total), float(dt) / int(total) * 1e3), flush=True)
        if traced:
            summary, calls, per_k = analyse(d, hops)
            print("   per steady-state batch:", summary)
            print("   kernel medians (us):", per_k)
            print("   HIP calls (count, median us, us per batch):", calls, flush=True)


if __name__ == "__main__":
    main()
