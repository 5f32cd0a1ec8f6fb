#!/usr/bin/env python3
"""Throughput of the `legion` server process as a trainer sees it: a null consumer (wait -> read counters -> post)
drains every batch of the schedule through the C-ABI IPC client, for the two RunOnce variants of the runner:

    (default)                  enqueue batch i, then wait for batch i-1 and post it (sampler i || gathers i-1)
    LEGION_BATCH_GRAPH=1       the sampler side as a recorded hipGraph, the rows gathered by one plain launch on stream 1 behind it
(the reference's synchronous loop and the whole-batch graphs lost at every shape: profiles/r01_server_loop.md, r04_graph_trace.md)

    python examples/serve_bench.py [--workload products --scale 0.3 --batch 8000 --fanout 25,10 --epochs 3]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def consume(epochs, hops):
    import legion1_amd.capi as K
    lib = K.lib()
    lib.legion_ipc_client_open.restype = C.c_void_p
    c = C.c_void_p(lib.legion_ipc_client_open(-1))
    steps = (C.c_int32 * 3)()
    lib.legion_ipc_client_steps(c, steps)
    total = (steps[0] + steps[1]) * epochs + steps[2]
    nc, ec = (C.c_int32 * 16)(), (C.c_int32 * 16)()
    edges = 0
    t0 = time.perf_counter()
    for _ in range(total):
        lib.legion_ipc_client_wait(c)
        lib.legion_ipc_client_read_counters(c, nc, ec)
        edges += ec[2 + hops]
        lib.legion_ipc_client_post(c)
    dt = time.perf_counter() - t0
    lib.legion_ipc_client_close(c)
    print(total, dt, edges)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--consume", type=int, default=0, help="internal: run the null consumer for this many epochs")
    ap.add_argument("--workload", default="products")
    ap.add_argument("--scale", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--fanout", default="25,10")
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--variants", default="all", help="comma list of: pipelined,graph + gather")
    ap.add_argument("--full-eval", action="store_true", help="keep the full validation / test sets (512-seed batches)")
    ap.add_argument("--source", default="files", choices=["files", "synth"],
                    help="files: write the dataset in Legion's raw layout and let the server read it (GPUGraphStore.cu:254-325).  synth: the server generates "
                         "the same tables in its own HBM (meta_config dataset path `synth:<workload>:<scale>`) -- the only way to serve the papers100M / uk-union shapes")
    a = ap.parse_args()
    if a.consume:
        return consume(a.consume, len(a.fanout.split(",")))
    import legion1_amd.synth as S
    if a.source == "synth":
        spec = S.spec_for(a.workload, scale=a.scale)
        n_eval = min(512, spec.n_valid, spec.n_test) if not a.full_eval else None
        tmp = tempfile.mkdtemp(prefix="legion_serve_")
        meta = os.path.join(tmp, "meta_config")
        with open(meta, "w") as f:
            f.write("synth:%s:%r %d %d 0 %d %d %d %d 0 %d 0" % (a.workload, a.scale, a.batch, spec.V, spec.F, spec.n_train,
                                                             n_eval or spec.n_valid, n_eval or spec.n_test, a.epochs))
        return serve_variants(a, tmp, meta)
    ds = S.generate(S.spec_for(a.workload, scale=a.scale))
    if not a.full_eval:   # keep the schedule dominated by full training batches: one validation / test batch each
        import dataclasses
        ds.valid, ds.test = ds.valid[:512], ds.test[:512]
        ds.spec = dataclasses.replace(ds.spec, n_valid=len(ds.valid), n_test=len(ds.test))
    tmp = tempfile.mkdtemp(prefix="legion_serve_")
    data = os.path.join(tmp, "ds") + "/"
    S.write_legion_files(ds, data)
    meta = os.path.join(tmp, "meta_config")
    with open(meta, "w") as f:
        f.write(S.meta_config_line(ds, data, a.batch, 1 << 40, a.epochs, 0))
    return serve_variants(a, tmp, meta)


def serve_variants(a, tmp, meta):
    server = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")
    variants = [("pipelined", {}), ("graph + gather", {"LEGION_BATCH_GRAPH": "1"}), ("pipelined", {}), ("graph + gather", {"LEGION_BATCH_GRAPH": "1"})]
    if a.variants != "all":
        variants = [v for v in variants[:2] if v[0] in a.variants.split(",")]
    for name, extra in variants:
        ns = "sb%d_%s%s_" % (os.getpid(), name[:3], extra.get("LEGION_BATCH_GRAPH", ""))
        env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        log = open(os.path.join(tmp, "server_%s.log" % name[:3]), "w")
        proc = subprocess.Popen([server, "1", "0", a.fanout, meta], stdout=log, stderr=subprocess.STDOUT, env=env, cwd=tmp)
        while "System is ready for serving" not in open(log.name).read():
            if proc.poll() is not None:
                raise SystemExit("server died:\n" + open(log.name).read()[-2000:])
            time.sleep(0.2)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--consume", str(a.epochs), "--fanout", a.fanout],
                             env=env, capture_output=True, text=True, timeout=600)   # one consumer process per server
        if out.returncode != 0:
            raise SystemExit(out.stdout[-2000:] + out.stderr[-2000:])
        total, dt, edges = out.stdout.strip().splitlines()[-1].split()
        total, dt, edges = int(total), float(dt), int(edges)
        proc.wait(timeout=60)
        print("%-15s %5d batches  %.3f ms/batch  %.2f G edges/s" % (name, total, dt / total * 1e3, edges / dt / 1e9), flush=True)


if __name__ == "__main__":
    main()
