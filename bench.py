#!/usr/bin/env python3
"""bench.py -- throughput of the mini-batch hot path (sampler + compaction + feature gather) on a
synthetic graph of the ogbn-papers100M shape (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python bench.py --gpus N ...          # spawns N ranks itself (parent never touches the GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one mini-batch (B seeds, H-hop sampling, COO construction, gather of every unique
node's feature row) with CSR and features already resident in HBM.  Each rank owns the seeds
`tid % N == rank` (GPUGraphStore.cu:332-346) and a full replica of the graph (Kg = 1), so there is
no data-path collective: scaling is weak.  With N > 1 the same processes then run the reference's
unified feature cache (Kg = N: rank-t hot row on GPU t % N, peer shards read in-kernel over xGMI;
GPUCache.cu:88-108,593-607) and report it inside the same line.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LEG_HUNG_EXIT = 3     # exit code of every rank when a leg after the headline hung: the line was printed, "legs_failed" names the leg
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E vendor peak (/opt/skills/guides/MI355X_MICROARCH.md)
XGMI_PEAK_GBPS = 7 * 153.0  # 7 point-to-point links x ~153 GB/s per GPU


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="papers100M", choices=["products", "papers100M", "uk-union"])
    ap.add_argument("--task", default="node", choices=["node", "lp"],
                    help="node: node-classification seeds (the training set).  lp: link-prediction seed batches laid out as "
                         "[src | pos | neg] thirds (lp_sage.py:87-90), generated on the GPU per rank (triples split by src %% N)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink V and seed sets (debug only)")
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--fanout", default="25,10,5")
    ap.add_argument("--gather", default="all", choices=["all", "level"], help="one gather per batch or one per level")
    ap.add_argument("--pipeline", default="serial", choices=["overlap", "serial", "intra"],
                    help="serial: one stream.  intra: the reference's two-stream schedule inside a batch (Server.cu:301-328): "
                         "the rows of level l are gathered on a second stream while hop l+1 is sampled; the last level runs alone.  "
                         "overlap: whole-batch gather of batch i on a second stream while batch i+1 is sampled (depth-2 pipes)")
    ap.add_argument("--cache", default="replicated", choices=["replicated", "unified"],
                    help="headline leg.  replicated: every GPU holds all features (Kg=1).  unified: the clique-wide "
                         "hotness-partitioned feature cache of the reference (rank-t row on GPU t %% N), peer shards read in-kernel over xGMI")
    ap.add_argument("--no-unified-leg", action="store_true", help="N > 1: skip the unified-cache leg that follows the replicated headline")
    ap.add_argument("--no-exchange-leg", action="store_true", help="N > 1: skip the owner-computes exchange variant of the unified-cache gather")
    ap.add_argument("--unified-timeout", type=float, default=240.0,
                    help="N > 1: seconds the unified-cache leg may take before the headline line is printed without it")
    ap.add_argument("--extra-legs", default="auto",
                    help="comma list of further legs run by the same processes after the headline (+ unified) leg and reported in the same "
                         "line: lp (BASELINE config 5: link-prediction seed batches on the papers100M graph, B = 7998), uk_union (config 4: "
                         "uk-union shape, 2-hop {25,10}, CSR sharded over the clique + capped feature cache); N = 1 also served (the `legion` server binary + "
                         "a trainer-side consumer process: the headline batches as a trainer sees them), cached_gather (config 3's FindFeat + "
                         "gather path, 25 %% of the rows in a shard), products_2hop / products_3hop (configs 1 / 2, each with its own served figure), partitioned_csr (= uk_union on one GPU).  "
                         "auto (default workload only) = served, lp, cached_gather, products_2hop, products_3hop, partitioned_csr at N = 1; lp, uk_union at N > 1; "
                         "'none' disables")
    ap.add_argument("--extra-timeout", type=float, default=150.0, help="seconds each extra leg may take")
    ap.add_argument("--extra-min-time", type=float, default=1.0, help="--min-time of the extra legs")
    ap.add_argument("--table", default="device", choices=["device", "host"],
                    help="where the V x F feature table lives: HBM (default) or pinned host memory read over PCIe -- the "
                         "reference's UVA configuration (GPUGraphStore.cu:315); combine with --cache unified for an HBM cache")
    ap.add_argument("--cache-frac", type=float, default=0.25, help="unified: fraction of the V feature rows cached per clique")
    ap.add_argument("--topo-frac", type=float, default=0.0, help="unified: fraction of the V adjacency rows cached as partitioned CSR "
                    "fragments per clique (0: topology stays replicated)")
    ap.add_argument("--presc-steps", type=int, default=8, help="unified: batches of the pre-sampling (hotness) epoch")
    ap.add_argument("--min-time", type=float, default=3.0,
                    help="the K-step timed window is repeated until this many seconds are covered; ms_per_step is the median window")
    ap.add_argument("--max-reps", type=int, default=400)
    ap.add_argument("--skew", type=int, default=205, help="synthetic neighbours: n/256 of them drawn from the Zipf-like skew "
                    "(205 = the spec'd 80 %%; 0 = uniform neighbours, the Infinity-Cache control run)")
    ap.add_argument("--headline-only", action="store_true", help="skip the alt_schedule and graph_replay legs (clean kernel profiles)")
    ap.add_argument("--stream-priority", default="none", choices=["none", "sampler", "gather"],
                    help="overlap schedule: which of the two streams gets the high stream priority (the other the low one)")
    ap.add_argument("--cu-split", type=int, default=0, help="experiment (overlap schedule): of every 8 compute units, this many run "
                    "the sampler stream and the rest the gather stream (hipExtStreamCreateWithCUMask); 0 = unrestricted streams")
    ap.add_argument("--cu-pattern", default="mod", choices=["mod", "block"], help="--cu-split: bit i belongs to the sampler if "
                    "i %% 8 < S (mod) or (i // 32) %% 8 < S (block)")
    ap.add_argument("--row-pitch", default="auto", choices=["auto", "dense"],
                    help="row pitch of the HBM feature table: auto = legion_row_pitch(F) (rows start on a 128-byte line: F = 100 -> 128 "
                         "floats; F = 128 / 256 are dense anyway), dense = F (the reference's file layout)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0,
                    help="CPU time budget of the baseline legs together (60 %% reference-semantics oracle, 40 %% DGL-semantics sampler); 0 disables")
    ap.add_argument("--no-cpu-features", action="store_true", help="CPU baseline: sampler only (skip the 57 GB host copy)")
    ap.add_argument("--measure-traffic", default="auto", choices=["auto", "off"],
                    help="auto (N = 1, tables in HBM, rocprofv3 on PATH, not already under a profiler): after the timed legs, two short child "
                         "runs of this workload under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` measure the HBM bytes of the dominant kernel for "
                         "roofline.traffic; off: quote the newest committed profiles/r*_pmc_hbm_traffic.json instead")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launch: seconds before the parent gives up on its ranks")
    ap.add_argument("--time-budget", type=float, default=480.0,
                    help="seconds the whole command may take (from the start of the first process).  Every leg after the headline gets "
                         "min(its own timeout, what is left minus a reserve); a leg that would get less than it needs is skipped and named in "
                         "\"legs_skipped\", so the JSON line is printed before an outer limit (the driver's 600 s) ends the run.  0 = no budget")
    ap.add_argument("--served-epochs", type=int, default=0, help="`served` leg: epochs the `legion` server runs (0 = sized for about 2 s of serving)")
    return ap.parse_args(argv)


# ======================================================================================================
# self-launch: `python bench.py --gpus N` without torchrun.  The parent never imports torch and never
# touches the GPU; it starts N fresh children (never exec, never a restart of a process that has
# initialised the GPU), waits for them and fails if any of them fails.
# ======================================================================================================
def launch_plan(args, environ):
    """What main() does with this (--gpus, environment): 'worker' (run here), 'spawn' (start N ranks) or an error text."""
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        if args.gpus < 1:
            return "error: --gpus must be >= 1"
        return "spawn" if args.gpus > 1 else "worker"
    if int(ws) != args.gpus:
        return f"error: --gpus {args.gpus} but WORLD_SIZE={ws}: refusing to run a different number of ranks than asked for"
    return "worker"


def child_env(environ, rank, world, port):
    env = dict(environ)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               LEGION_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode this pool's driver supports
    return env


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(args, argv, popen=subprocess.Popen, poll_s=0.2, grace_s=20.0, script=None):
    port = free_port()
    environ = dict(os.environ)
    environ["LEGION_BENCH_T0"] = repr(time.time())                 # the time budget counts from here (never a stale value of an outer shell)
    # stdout carries exactly ONE JSON line (rank 0's): every other rank writes to the parent's stderr, whatever it prints
    procs = [popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=child_env(environ, r, args.gpus, port),
                   stdout=None if r == 0 else sys.stderr)
             for r in range(args.gpus)]
    deadline = time.time() + args.launch_timeout
    first_bad = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [c for c in codes if c not in (None, 0)]
        now = time.time()
        if bad and first_bad is None:
            first_bad = now                       # a dead rank leaves the others in a collective: give them a moment, then stop them
        if (first_bad is not None and now - first_bad > grace_s) or now > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                      # exactly the PIDs started above
            for p in procs:
                p.wait()
            if now > deadline and first_bad is None:
                print(f"bench.py: ranks did not finish within {args.launch_timeout:.0f} s", file=sys.stderr)
                return 124
            break
        time.sleep(poll_s)
    codes = [p.returncode for p in procs]
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        if all(c in (0, LEG_HUNG_EXIT) or (c is not None and c < 0) for c in codes) and LEG_HUNG_EXIT in codes:
            print("bench.py: the headline line was printed, but a later leg did not finish in time (see \"legs_failed\" in the line)",
                  file=sys.stderr)
            return LEG_HUNG_EXIT
        # a rank's own exit code (one that died for its own reasons first); 1 if only stragglers were stopped
        return next((c for c in codes if c and c > 0 and c != LEG_HUNG_EXIT), next((c for c in codes if c and c > 0), 1))
    return 0


# ======================================================================================================
# worker: one rank
# ======================================================================================================
def build_graph_on_gpu(K, spec, dev, skew=205, pitch=0):
    """Synthetic dataset generated on the GPU by csrc/synth.hip (spec: legion-1_amd/synth.py).  pitch > F: the feature
    rows are laid out with that many floats between two rows (same values)."""
    import torch
    L = K.lib()
    V, F = spec.V, spec.F
    ladder = np.ascontiguousarray(spec.ladder, dtype=np.int32)
    deg = torch.empty(V, dtype=torch.int64, device=dev)
    L.legion_synth_degrees(None, deg.data_ptr(), 0, V, ladder.ctypes.data)
    indptr = torch.zeros(V + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=indptr[1:])
    del deg
    E = int(indptr[-1].item())
    indices = torch.empty(E, dtype=torch.int32, device=dev)
    L.legion_synth_neighbors_skew(None, indices.data_ptr(), 0, E, V, spec.M, spec.C, skew)
    if pitch > F:
        feats = torch.zeros((V, pitch), dtype=torch.float32, device=dev)
        L.legion_synth_features_pitched(None, feats.data_ptr(), 0, V, F, pitch)
    else:
        feats = torch.empty((V, F), dtype=torch.float32, device=dev)
        L.legion_synth_features(None, feats.data_ptr(), 0, V, F)
    torch.cuda.synchronize()
    K.check()
    return indptr, indices, feats, E


class Ctx:
    """Everything the legs share: the rank's device, the synthetic graph and this rank's seed list."""


_LINE_OUT = None      # where the ONE JSON line goes (the process' original stdout); None: sys.stdout


def claim_stdout():
    """From here on file descriptor 1 of this process IS its stderr: whatever the HIP library, RCCL, torch or a child process prints
    (std::cout chatter of the C++ side, "xGMI Clique ...", "Feature Cache Hit ...") can no longer land inside -- or next to -- the JSON
    line, which is written to a private duplicate of the original stdout by emit_line()."""
    global _LINE_OUT
    if _LINE_OUT is not None:
        return
    sys.stdout.flush()
    _LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    os.environ.setdefault("LEGION_LOG", "stderr")      # the library's own prints go to stderr at the source, too (runtime.cpp)


def emit_line(line):
    out = _LINE_OUT if _LINE_OUT is not None else sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


class Budget:
    """--time-budget: what is left of the command's wall-clock allowance.  T0 is the start of the FIRST process of the command (the
    self-launching parent hands it down in LEGION_BENCH_T0)."""
    RESERVE_S = 15.0          # kept back for printing the line, the final barrier and process teardown

    def __init__(self, total, t0=None, now=time.time):
        self.total, self.now = float(total), now
        self.t0 = float(t0) if t0 is not None else now()

    def left(self):
        return float("inf") if self.total <= 0 else self.total - (self.now() - self.t0)

    def grant(self, want, least):
        """Seconds a leg may take: min(want, left - reserve), or 0.0 when that is less than `least` (the leg is skipped)."""
        got = min(float(want), self.left() - self.RESERVE_S)
        return got if got >= least else 0.0


LEG_LEAST_S = {"unified_cache": 60.0, "served": 40.0, "served_all": 60.0, "partitioned_csr_host_spill": 90.0}     # a leg that cannot get this much is skipped (default: 30 s)


def run_budgeted(c, line, guard, name, want, fn):
    """Run a leg under the watchdog with what the time budget grants it (rank 0 decides, every rank follows); a leg the budget
    does not admit is named in "legs_skipped" and the run goes on to print its line."""
    least = min(LEG_LEAST_S.get(name, 30.0), float(want))     # a leg asked to run in less than its usual minimum (a test) is not skipped for that
    grant = c.budget.grant(want, least)
    if c.world > 1:
        grant = c.D.allgather_object(grant, c.world)[0]
    if grant <= 0.0:
        line.setdefault("legs_skipped", []).append({"leg": name, "why": "time budget: %.0f s of --time-budget %.0f left, the leg needs %.0f + %.0f reserve"
                                                                       % (max(0.0, c.budget.left()), c.budget.total, least, Budget.RESERVE_S)})
        return {"skipped": "time budget"}
    return guard.run(name, grant, fn)


def worker(args):
    import torch
    if os.environ.get("LEGION_BENCH_WATCHDOG"):  # debugging aid: dump all Python stacks and exit if stuck
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LEGION_BENCH_WATCHDOG"]), exit=True)
    claim_stdout()
    c = Ctx()
    c.args = args
    # the budget counts from the start of the self-launching parent (only its own children believe LEGION_BENCH_T0), else from here
    c.budget = Budget(args.time_budget, os.environ.get("LEGION_BENCH_T0") if os.environ.get("LEGION_BENCH_SPAWNED") == "1" else None)
    c.children = []       # child processes a leg started (the `served` leg's server and consumer): the watchdog stops them
    c.rank = rank = int(os.environ.get("RANK", "0"))
    c.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a one-GPU box: LEGION_BENCH_FORCE_DEVICE=0 puts every rank on that device and uses gloo
    # (RCCL refuses two ranks on one GPU).  Never set on a real multi-GPU run.
    forced = os.environ.get("LEGION_BENCH_FORCE_DEVICE")
    c.shared_device = forced is not None and world > 1
    if forced is not None:
        local_rank = int(forced)
    n_vis = torch.cuda.device_count()
    if local_rank >= n_vis:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {n_vis} GPU(s) are visible "
                         f"(--gpus {args.gpus}): refusing to run on fewer GPUs than asked for")
    c.local_rank = local_rank
    torch.cuda.set_device(local_rank)
    c.dev = dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced is not None:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import legion1_amd.capi as K
    import legion1_amd.dist as D
    import legion1_amd.synth as S
    c.K, c.D, c.S = K, D, S
    c.L = L = K.lib()
    c.fan = [int(x) for x in args.fanout.split(",")]
    c.head_fan = list(c.fan)      # the headline's fan-out (later legs change c.fan on copies of the context)
    c.H = len(c.fan)
    c.B = args.batch
    L.legion_set_device_map(0, local_rank)
    L.SetGPUDevice(0)
    load_workload(c, args.workload)

    head = run_leg(c, unified=(args.cache == "unified"), headline=True)
    c.head = head
    line = headline_line(c, head) if rank == 0 else {"legs_failed": [], "legs_skipped": [], "extra_legs": {}}
    c.guard = guard = LegGuard(c, line)
    # Everything behind the headline is evidence beside it: whatever goes wrong in the glue between the legs (a leg's own failures are
    # caught and reported by LegGuard), rank 0 still prints the line it has -- with the error named -- before the process ends non-zero.
    try:

        def budgeted(name, want, fn):
            return run_budgeted(c, line, guard, name, want, fn)

        if world > 1 and args.cache == "replicated" and not args.no_unified_leg and args.table == "device":
            line["unified_cache"] = budgeted("unified_cache", args.unified_timeout,
                                             lambda: unified_summary(c, run_leg(c, unified=True, headline=False, min_time=args.extra_min_time)))

        # measured streaming-copy rate of this box (float4 copy kernel sized like the gather, read + write bytes)
        if rank == 0 and line.get("roofline") is not None:
            try:
                line["roofline"]["measured_copy_GBps"] = measure_copy(c)
            except Exception as ex:   # noqa: BLE001 -- evidence beside the headline: never lose the line over it (e.g. no 8 GiB left at N > 1 behind the unified leg)
                line["roofline"]["measured_copy_GBps"] = None
                line["roofline"]["measured_copy_error"] = repr(ex)[:200]
            if (world == 1 and args.measure_traffic == "auto" and not args.headline_only and args.table == "device" and args.cache == "replicated"
                    and args.gather == "all" and args.pipeline == "serial" and c.budget.left() > 240.0):   # two passes of <= 90 s each
                got = measure_traffic_in_run(args)
                if got is not None:
                    line["roofline"].update(got)
        if rank == 0 and world > 1 and args.cpu_baseline_seconds > 0 and c.budget.left() > 120.0:
            # N > 1: the same baseline, short -- sampler + COO only (no 57 GB host copy of the table), <= 5 s of CPU time, on rank 0
            # while the other ranks wait in the next collective
            import copy
            a1 = copy.copy(args)
            a1.no_cpu_features, a1.cpu_baseline_seconds = True, min(5.0, args.cpu_baseline_seconds)
            try:
                line["cpu_baseline"] = run_cpu_baseline(a1, c.spec, c.indptr, c.indices, None, c.mine, c.my_labels, c.B, c.fan, c.steps_avail)
            except Exception as ex:   # noqa: BLE001 -- reported baseline only
                line["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(ex)[:200]}
        if rank == 0 and world == 1 and args.cpu_baseline_seconds > 0 and c.budget.left() > 150.0:   # reported baseline
            feats = c.feats
            if c.host_table is not None:   # the table already is host memory: view it, no copy
                import ctypes
                feats = np.ctypeslib.as_array(ctypes.cast(c.host_table, ctypes.POINTER(ctypes.c_float)), shape=(c.spec.V, c.spec.F))
            try:
                line["cpu_baseline"] = run_cpu_baseline(args, c.spec, c.indptr, c.indices, feats, c.mine, c.my_labels, c.B, c.fan, c.steps_avail)
            except Exception as ex:   # noqa: BLE001 -- reported baseline only: never lose the headline line over it
                line["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(ex)[:200]}

        for name in extra_leg_names(c):      # BASELINE configs 4 and 5, same processes, same line (after everything that needs the headline graph)
            line["extra_legs"][name] = budgeted(name, args.extra_timeout, lambda name=name: extra_leg(c, name))

    except Exception as ex:   # noqa: BLE001
        if rank == 0:
            import traceback
            line["worker_error"] = {"error": repr(ex)[:300], "where": traceback.format_exc()[-1200:]}
            line["time_budget"] = {"budget_s": args.time_budget, "used_s": round(time.time() - c.budget.t0, 1)}
            emit_line(line)
        raise
    if rank == 0:
        sv = line["extra_legs"].get("served") or {}
        if isinstance(sv, dict) and sv.get("value"):      # the headline batches as a trainer sees them, next to `value` (serial) and `value_overlap` (two streams, in-process):
            # value_served = a consumer that READS every served row and edge before it hands the pipe back; value_served_null = the hand-off alone
            line["value_served"], line["ms_per_step_served"] = sv["value"], sv["ms_per_step"]
            nul = sv.get("null_consumer") or {}
            if nul.get("value"):
                line["value_served_null"], line["ms_per_step_served_null"] = nul["value"], nul["ms_per_step"]
        line["time_budget"] = {"budget_s": args.time_budget, "used_s": round(time.time() - c.budget.t0, 1)}
        if own_audit_counts(L) is not None:     # $LEGION_DEVICE_AUDIT=1 (a rehearsal, never a measurement): what rank 0's library saw
            line["device_audit"] = own_audit_counts(L)
        emit_line(line)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def load_workload(c, workload):
    """(Re)build the synthetic graph of `workload` on this rank's GPU and this rank's seed list; frees the previous one first."""
    import torch
    args, K, L, S = c.args, c.K, c.L, c.S
    for name in ("indptr", "indices", "feats", "mine", "my_labels"):
        setattr(c, name, None)
    if getattr(c, "host_table", None) is not None:
        L.host_free_space(c.host_table)
    c.host_table = None
    torch.cuda.empty_cache()
    c.spec = spec = S.spec_for(workload, scale=args.scale)
    V, F = spec.V, spec.F
    t0 = time.time()
    c.pitch = L.legion_row_pitch(F) if (args.table == "device" and args.row_pitch != "dense") else F
    c.indptr, c.indices, c.feats, c.E = build_graph_on_gpu(K, spec, c.dev, args.skew, c.pitch)
    c.feat_ptr, c.feat_loc = c.feats.data_ptr(), K.LOC_DEVICE
    if args.table == "host":   # move the table to pinned, device-mapped host memory; misses then cross PCIe
        nbytes = V * F * 4
        c.host_table = L.host_alloc_space64(nbytes)
        L.d_copy_d_2_h(c.host_table, c.feats.data_ptr(), nbytes)
        K.check()
        c.feats = None
        torch.cuda.empty_cache()
        c.feat_ptr, c.feat_loc = c.host_table, K.LOC_HOST_PINNED
    make_seeds(c)
    c.gen_s = time.time() - t0


class LegGuard:
    """Watchdog of the legs that run after the headline leg.  A leg that raises is reported inside the line ("legs_failed" at
    the top level + an "error" object in its place) and the run goes on.  A leg that does not come back within its timeout
    (a collective or an IPC import that never returns) cannot be recovered from inside the process: rank 0 prints the line it
    has -- headline included, the leg named in "legs_failed" with the Python stack it was stuck in -- and EVERY rank leaves with
    exit code LEG_HUNG_EXIT.  The timer of a rank stays armed through the agreement all-gather and the barrier behind the leg,
    so a rank that finished cannot be left waiting in a collective for one that hangs: its own timer ends it too.
    A leg that raises on SOME ranks only leaves the others inside the leg's collectives; the failing rank says so in the job's
    key-value store, and a rank that is still inside the leg a few seconds after a peer reported a failure ends the same way
    (named error, exit code LEG_HUNG_EXIT) without waiting for the whole timeout."""

    PEER_GRACE_S = 5.0

    def __init__(self, c, line):
        self.c, self.line = c, line
        self.partial = None        # what a leg has measured so far (printed if a later part of it hangs)
        self.store = None
        if c.world > 1:
            try:
                import torch.distributed.distributed_c10d as c10d
                self.store = c10d._get_default_store()
            except Exception:   # noqa: BLE001 -- the plain timeout still applies
                self.store = None

    def announce(self, text):
        """A PART of the running leg failed on this rank and was caught further down (run_leg keeps going with an "error" object): tell
        the peers, which may be waiting in that part's collectives.  The part ends with an agreement collective (part_agreed): when every
        rank got there -- the failure was symmetric, or the others finished the part -- the announcement is settled and the leg goes on
        with all ranks; only a rank still stuck INSIDE the part PEER_GRACE_S after the announcement ends the run."""
        if self.store is not None and getattr(self, "key", None):
            try:
                self.store.set(self.key + "/part", "rank %d: %s" % (self.c.rank, text[:300]))
            except Exception:   # noqa: BLE001
                pass

    def part_agreed(self):
        """Every rank has left the announced part (called behind that part's agreement collective)."""
        self._part_agreed = True

    def run(self, name, timeout, fn):
        import faulthandler
        import traceback
        c, line = self.c, self.line
        main_thread = threading.main_thread().ident
        self.partial = None

        in_leg = [True]
        self._part_agreed = False
        fired = threading.Lock()       # timer thread and watcher thread may both get here: the line is printed once
        hung_exit = LEG_HUNG_EXIT     # at every world size: a watchdog that fires on a process that has touched the GPU never exits 0

        def fire(msg=f"did not finish within {timeout:.0f} s"):
            if not fired.acquire(blocking=False):
                time.sleep(3600)       # the other thread is printing and will end the process
            try:
                stack = "".join(traceback.format_stack(sys._current_frames().get(main_thread)))[-1500:]
                if c.rank == 0:
                    line["legs_failed"].append({"leg": name, "error": msg, "hung": True, "stuck_at": stack})
                    part = self.partial
                    err = dict(part, error=msg) if isinstance(part, dict) else {"error": msg}
                    if name in ("unified_cache",):
                        line[name] = err
                    else:
                        line["extra_legs"][name] = err
                    emit_line(line)
                faulthandler.dump_traceback(file=sys.stderr)
                for ch in getattr(c, "children", []):      # exactly the processes a leg of this run started
                    try:
                        if ch.poll() is None:
                            ch.kill()
                    except Exception:   # noqa: BLE001
                        pass
                unlink_served_namespaces(c)                # a killed server never ran IPCEnv_Finalize: its slab and semaphores would stay in /dev/shm
            finally:
                os._exit(hung_exit)
        timer = threading.Timer(timeout, fire)
        timer.daemon = True
        timer.start()
        key = self.key = "legion_leg_failed/" + name

        def watch_peers():       # a peer that raised inside the leg will never join this rank's collectives
            seen, seen_part, who, who_part = None, None, "", ""
            mine = "rank %d:" % c.rank
            while in_leg[0] and self.store is not None:
                try:
                    if seen is None and self.store.check([key]):
                        who = self.store.get(key).decode(errors="replace")
                        seen = time.time() if not who.startswith(mine) else float("inf")     # never act on this rank's own announcement
                    if seen_part is None and self.store.check([key + "/part"]):
                        who_part = self.store.get(key + "/part").decode(errors="replace")
                        seen_part = time.time()      # own or a peer's: what counts is whether THIS rank gets out of the part
                    if seen is not None and time.time() - seen > self.PEER_GRACE_S and in_leg[0]:
                        fire("a peer failed inside the leg while this rank was still in it: " + who)
                    if seen_part is not None and not self._part_agreed and time.time() - seen_part > self.PEER_GRACE_S and in_leg[0]:
                        fire("a part of the leg failed on a rank and not every rank left that part: " + who_part)
                except Exception:   # noqa: BLE001
                    return
                time.sleep(0.5)
        if self.store is not None:
            threading.Thread(target=watch_peers, daemon=True).start()
        try:
            ok, res = True, None
            try:
                # test hook (tests/test_gpu_bench_legs.py): LEGION_BENCH_INJECT_HANG=<leg>[:<rank>] makes that rank never
                # come back from the leg, LEGION_BENCH_INJECT_ERROR=<leg>[:<rank>] makes it raise
                for kind in ("HANG", "ERROR"):
                    inj = os.environ.get("LEGION_BENCH_INJECT_" + kind, "").split(":")
                    if inj[0] == name and (len(inj) < 2 or int(inj[1]) == c.rank):
                        if kind == "ERROR":
                            raise RuntimeError("injected failure of leg %s on rank %d" % (name, c.rank))
                        time.sleep(10 ** 6)
                res = fn()
            except Exception as ex:  # noqa: BLE001 -- reported inside the line, never fatal
                ok, res = False, {"error": repr(ex)[:300]}
                if self.store is not None:
                    try:
                        self.store.set(key, "rank %d: %s" % (c.rank, repr(ex)[:300]))
                    except Exception:   # noqa: BLE001
                        pass
            in_leg[0] = False
            # every rank must agree that the leg worked before its numbers are believed
            try:
                flags = c.D.allgather_object((ok, None if ok else res["error"]), c.world)
                bad = [(i, e) for i, (f, e) in enumerate(flags) if not f]
                if bad:
                    if ok:
                        res = {"error": "failed on rank(s) %s" % [i for i, _ in bad]}
                    line["legs_failed"].append({"leg": name, "error": "; ".join("rank %d: %s" % (i, e) for i, e in bad)[:600], "hung": False})
                c.D.barrier(c.world)
            except Exception as ex:   # noqa: BLE001 -- the peers left (they were stuck inside the leg this rank failed in): same ending
                fire("the agreement after the leg failed (%s); this rank's own result: %s" % (repr(ex)[:200], res if not ok else "ok"))
            return res
        finally:
            timer.cancel()


N1_LEGS = ["served", "lp", "cached_gather", "products_2hop", "products_3hop", "partitioned_csr", "partitioned_csr_host_spill"]   # run order: graph re-use first
PCIE_PEAK_GBPS = 64.0       # PCIe Gen5 x16, one direction (the MI355X host link): what pinned-host rows can arrive at
NN_LEGS = ["lp", "uk_union", "served_all"]     # served_all last: it starts other processes on every GPU of the job


def extra_leg_names(c):
    """auto: at N = 1 every single-GPU-measurable BASELINE path (N1_LEGS), at N > 1 configs 5 and 4 (NN_LEGS) -- for the default
    workload only.  An explicit comma list is run in the canonical order (legs that re-use the headline graph first)."""
    a = c.args
    if a.extra_legs == "none":
        return []
    if a.extra_legs == "auto":
        default = (a.workload == "papers100M" and a.task == "node" and a.cache == "replicated" and a.table == "device" and not a.headline_only)
        if not default:
            return []
        return list(N1_LEGS) if c.world == 1 else list(NN_LEGS)
    names = [x for x in a.extra_legs.split(",") if x]
    known = N1_LEGS + ["uk_union", "served_all"]      # (partitioned_csr_host_spill: N = 1 only -- 137 GB of pinned host memory per process)
    bad = [x for x in names if x not in known]
    if bad:
        raise SystemExit("bench.py: unknown --extra-legs %s" % bad)
    return [x for x in known if x in names]


def extra_leg(c, name):
    """The other BASELINE.json configurations in the processes of this run, one object each in `extra_legs`:
      lp               config 5: [src | pos | neg] seed batches on the headline graph
      cached_gather    config 3's gather path on one GPU: --cache-frac of the feature rows in a hotness-ranked shard (Kg = world),
                       FindFeat per row (k_row_ptrs) in front of the gather, misses from the HBM table (GPUCache.cu:387-400, Kernels.cu:687-698)
      products_2hop    config 1's workload (the CPU-sampler baseline's) on the GPU, with its CPU legs beside it
      products_3hop    config 2
      partitioned_csr  config 4 on the GPUs of this run (= uk_union at N > 1): uk-union shape, {25,10}, the hottest 30 % of the adjacency
                       rows as partitioned CSR fragments (k_sample<partitioned>), 10 % of the feature rows cached
      partitioned_csr_host_spill   config 4 as BASELINE.json states it ("CSR sharded across GPUs + pinned-host spillover"): the same leg with
                       the 137 GB feature table in pinned, device-mapped HOST memory behind the 10 % HBM cache (GPUGraphStore.cu:264-265,315;
                       misses read over PCIe in-kernel, Kernels.cu:692-699): ms / batch, hit rate, miss rows, PCIe GB/s against the link peak"""
    import copy
    a = copy.copy(c.args)
    c2 = copy.copy(c)
    c2.args = a
    a.min_time = c.args.extra_min_time

    def adopt_graph(workload):
        for k in ("indptr", "indices", "feats", "mine", "my_labels"):     # drop the resident graph FIRST: c2 is a shallow copy, and
            setattr(c, k, None)                                          # 64 GB + 160 GB would otherwise be resident together
            setattr(c2, k, None)
        load_workload(c2, workload)
        for k in ("spec", "pitch", "indptr", "indices", "feats", "E", "feat_ptr", "feat_loc", "host_table", "mine", "my_labels", "n_mine", "steps_avail", "gen_s"):
            setattr(c, k, getattr(c2, k))      # the previous graph is gone: later legs see this one

    if name == "served":
        # the headline batches as a trainer sees them.  Two consumers, same server, same schedule: one that READS what it is served before it hands
        # the pipe back (the figure that counts: a working trainer reads every row from the same HBM the next batch is produced in) and the null
        # consumer (the hand-off alone: the ceiling of this formulation)
        out = served_leg(c, c.args.workload, c.fan, c.head, consumer="reading")
        if c.budget.left() > 60.0:
            try:
                nul = served_leg(c, c.args.workload, c.fan, c.head, consumer="null")
                out["null_consumer"] = {k: nul[k] for k in ("value", "ms_per_step", "windows", "window_ms_min_median_max", "ratio_to_alt_schedule_levels",
                                                            "ratio_to_alt_schedule", "served_batches_equal_the_timed_ones", "pipeline_frac")}
                out["ratio_to_null_consumer"] = round(out["ms_per_step"] / nul["ms_per_step"], 4)
            except Exception as ex:   # noqa: BLE001 -- the reading consumer's figure stays valid
                out["null_consumer"] = {"error": repr(ex)[:300]}
        return out
    if name == "served_all":
        # the last leg: every rank gives its graph back first (uk-union: 160 GB per GPU), the server generates its own replica on every GPU
        import torch
        for k in ("indptr", "indices", "feats", "mine", "my_labels"):
            setattr(c, k, None)
        torch.cuda.empty_cache()
        return served_all_leg(c) if c.rank == 0 else None
    if name == "lp":
        # lp_sage.py:87-90: [src | pos | neg] seed thirds; triples dealt to the ranks by src % N; the graph is the headline's
        a.task, a.batch = "lp", (c.B // 3) * 3
        c2.B = a.batch
        make_seeds(c2)
        leg = run_leg(c2, unified=False, headline=False, min_time=a.min_time)
        out = leg_summary(c2, leg, "link-prediction seed batches [src | pos | neg] (lp_sage.py:87-90) on the headline graph, replicated tables")
        if c.world == 1 and c.budget.left() > 90.0:
            try:      # config 5 as a trainer sees it: the server generates the same seed lists (synth: source + meta flag 2)
                out["served"] = served_leg(c2, c.args.workload, c2.fan, leg, lp=True)
            except Exception as ex:   # noqa: BLE001 -- the leg's own numbers stay valid
                out["served"] = {"error": repr(ex)[:300]}
        return out
    if name == "cached_gather":
        if not c.spec.name.startswith(c.args.workload):     # an earlier leg replaced the headline graph
            adopt_graph(c.args.workload)
        a.no_exchange_leg, a.topo_frac = True, 0.0
        leg = run_leg(c2, unified=True, headline=False, min_time=a.min_time)
        out = leg_summary(c2, leg, f"headline graph and batches; the hottest {a.cache_frac:.0%} of the feature rows in a cache shard (Kg = {c.world}), every row "
                                   "resolved by FindFeat (k_row_ptrs: id -> slot -> shard row / table row) in front of the gather, misses from the HBM table")
        out["gather_launch_includes"] = "k_row_ptrs + k_gather (HIP events around get_feature_kernel)"
        if c.world == 1 and c.budget.left() > 90.0:
            try:      # the cached path as a trainer sees it: the server builds its cache with a budget of cache_frac of the feature table ($LEGION_SYNTH_CACHE=1;
                      # the cost model splits it between adjacency rows and feature rows: sampler over CSR fragments + cached gather)
                out["served"] = served_leg(c2, c.args.workload, c2.fan, leg, cache_bytes=int(c.spec.V * c.spec.F * 4 * a.cache_frac))
            except Exception as ex:   # noqa: BLE001 -- the leg's own numbers stay valid
                out["served"] = {"error": repr(ex)[:300]}
        return out
    if name in ("products_2hop", "products_3hop"):
        if not c.spec.name.startswith("products"):
            adopt_graph("products")
        a.workload, a.task = "products", "node"
        a.fanout = "25,10" if name == "products_2hop" else "25,10,5"
        c2.fan = [int(x) for x in a.fanout.split(",")]
        c2.H = len(c2.fan)
        leg = run_leg(c2, unified=False, headline=False, min_time=a.min_time, with_alt=True)
        out = leg_summary(c2, leg, f"ogbn-products-shape graph, {c2.H}-hop fan-out {c2.fan}, CSR + features resident in HBM (row pitch {c2.pitch} floats)")
        if c.world == 1 and c.budget.left() > 90.0:
            try:      # the same batches as a trainer sees them: the server binary on the products shape
                out["served"] = served_leg(c2, "products", c2.fan, leg)
            except Exception as ex:   # noqa: BLE001 -- the leg's own numbers stay valid
                out["served"] = {"error": repr(ex)[:300]}
        if name == "products_2hop" and c.rank == 0 and c.world == 1 and c.args.cpu_baseline_seconds > 0:
            # BASELINE config 1 is this workload on the CPU (DGL's NeighborSampler): the CPU legs on the same graph, bounded
            a.cpu_baseline_seconds = min(4.0, c.args.cpu_baseline_seconds)
            try:
                out["cpu_baseline"] = run_cpu_baseline(a, c2.spec, c2.indptr, c2.indices, c2.feats, c2.mine, c2.my_labels, c2.B, c2.fan, c2.steps_avail)
            except Exception as ex:   # noqa: BLE001 -- reported baseline only
                out["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(ex)[:200]}
        return out
    if name == "partitioned_csr_host_spill":
        a.workload, a.fanout, a.task, a.topo_frac, a.cache_frac, a.no_exchange_leg, a.table = "uk-union", "25,10", "node", 0.3, 0.10, True, "host"
        c2.fan, c2.H = [25, 10], 2
        t_pin = time.time()
        adopt_graph("uk-union")          # generates the graph in HBM, then moves the feature table into pinned host memory (load_workload, --table host)
        pin_s = time.time() - t_pin
        leg = run_leg(c2, unified=True, headline=False, min_time=a.min_time)
        out_hs = leg_summary(c2, leg, f"uk-union-shape graph, CSR sharded over the {c.world}-GPU clique (30 % of the adjacency rows in partitioned fragments), "
                                      f"the {c2.spec.V * c2.spec.F * 4 / 1e9:.0f} GB feature table in PINNED HOST memory behind a unified HBM cache of 10 % of its rows: "
                                      "hits from the HBM shard, misses read in-kernel over PCIe (BASELINE config 4 as stated)")
        rl = out_hs.get("rows_last_batch") or {}
        rows = sum(rl.get(k, 0) for k in ("own_shard", "peer_shards", "backing_table"))
        g = np.asarray(leg["g_ms"], np.float64)
        # the gather launches of the LAST timed batch of every window: the batch whose rows were classified (same rows, same time)
        g_last = float(g.reshape(-1, a.steps)[:, -1].mean()) if len(g) and len(g) % a.steps == 0 else (float(g.mean()) if len(g) else None)
        if rows and g_last:
            miss = rl.get("backing_table", 0)
            gbps = miss * c2.spec.F * 4 / (g_last * 1e-3) / 1e9
            out_hs["host_spill"] = {"rows_last_batch": rows, "miss_rows_last_batch": miss, "hit_rate": round(1.0 - miss / rows, 4),
                                    "gather_ms_last_batch": round(g_last, 3),
                                    "pcie_read_GBps": round(gbps, 2), "pcie_peak_GBps": PCIE_PEAK_GBPS, "pcie_frac_of_peak": round(gbps / PCIE_PEAK_GBPS, 4),
                                    "how": "miss rows of the last timed batch x 4F bytes / the HIP-event time of that batch's gather launches (lookups and the hits' HBM "
                                           "reads included); the peak is PCIe Gen5 x16 one way, nominal -- the batch is bound by the host link, not by HBM"}
        out_hs["table_generate_and_pin_s"] = round(pin_s, 1)
        return out_hs
    # uk_union / partitioned_csr: legion_server.py:23-37 shape, 2-hop GCN fan-out; the hottest 30 % of the adjacency rows as partitioned CSR
    # fragments over the N-GPU clique (GPU_Memory_Graph_Storage.cu:98-133), 10 % of the feature rows in the unified cache, the rest from the
    # HBM replica (a pinned-host backing table of 137 GB per process is not attempted here; tests/test_gpu_full_shape.py covers it)
    a.workload, a.fanout, a.task, a.topo_frac, a.cache_frac, a.no_exchange_leg = "uk-union", "25,10", "node", 0.3, 0.10, True
    c2.fan, c2.H = [25, 10], 2
    served = None
    if c.world == 1 and name == "partitioned_csr" and c.budget.left() > 120.0:
        # config 4's SHAPE as a trainer sees it, before this process generates its own 160 GB copy (the two do not fit one GPU together): the server
        # replicates the uk-union tables into its HBM -- on a 288 GB part nothing has to be partitioned or spilled -- and serves {25,10} batches
        for k in ("indptr", "indices", "feats", "mine", "my_labels"):
            setattr(c, k, None)
            setattr(c2, k, None)
        import torch
        torch.cuda.empty_cache()
        try:
            served = served_leg(c2, "uk-union", c2.fan, None)
            served["what"] = "REPLICATED configuration of the uk-union shape (the server holds the whole CSR + 137 GB of features in HBM; no fragments, no cache): " + served["what"]
        except Exception as ex:   # noqa: BLE001
            served = {"error": repr(ex)[:300]}
    adopt_graph("uk-union")
    leg = run_leg(c2, unified=True, headline=False, min_time=a.min_time)
    out_uk = leg_summary(c2, leg, f"uk-union-shape graph, CSR sharded over the {c.world}-GPU clique (30 % of the adjacency rows in partitioned "
                                "fragments) + 10 % of the feature rows in the unified cache, misses from the HBM replica")
    if served is not None:
        out_uk["served_replicated"] = served
    return out_uk


def served_consumer(argv):
    """Child process of the `served` legs: a trainer-side consumer on the C-ABI IPC client (legion_ipc_client_*: what ipc_service binds).
    argv: hops epochs [gpu [mode F workload scale]].
      mode null     wait for the batch (sem_w), read its counters, hand the pipe back (sem_r): the reference's get_next / synchronize with
                    nothing in between (ipc_cuda_kernel.cu:98-107,178-230) -- the hand-off alone.
      mode reading  ... and, before the pipe goes back, READ what was served the way a trainer does (legion_graphsage.py:72-89 touches every
                    feature row and both COO arrays of the batch before torch.cuda.synchronize()): the whole [nc9, F] feature view and both
                    COO arrays are summed word by word on a stream of this process (legion_sum_words: one 16-byte-per-lane read kernel,
                    ~1 GB per papers100M batch) -- the memory traffic of a training step without the model, competing with the server's
                    next batch for the same HBM.  Batch 1's sums are checked: the COO sums against a host copy of the arrays, the feature
                    sum against the generator's closed form of the rows the batch names (legion-1_amd/synth.py) -- the read is real.
    No torch.  One JSON object on stdout: the arrival time, edges and rows of every batch (+ the checksums)."""
    import ctypes as C
    import legion1_amd.capi as K
    hops, epochs = int(argv[0]), int(argv[1])
    gpu = int(argv[2]) if len(argv) > 2 and int(argv[2]) >= 0 else 0     # logical GPU of the server this consumer is the trainer of (served_all: one per GPU)
    mode = argv[3] if len(argv) > 3 else "null"
    lib = K.lib()
    lib.SetGPUDevice(gpu)                             # physical device gpu % visible devices, as the server maps it
    if len(argv) > 2 and int(argv[2]) >= 0:
        os.environ["LEGION_IPC_DEVICE"] = str(gpu)    # row of the handle table (differs from the physical device on a shared GPU)
    c = C.c_void_p(lib.legion_ipc_client_open(-1))
    K.check()
    steps = (C.c_int32 * 3)()
    lib.legion_ipc_client_steps(c, steps)
    total = (steps[0] + steps[1]) * epochs + steps[2]
    nc, ec = (C.c_int32 * 16)(), (C.c_int32 * 16)()
    t, edges, nodes = [], [], []
    reading = mode == "reading"
    check = None
    if reading:
        F = int(argv[4])
        stream = lib.d_stream_create()
        acc = K.DevBuf(64)
        lib.d_memset_async(acc.ptr, 0, 64, stream)
        rows_cap = lib.legion_ipc_client_feature_rows(c)

        def read_acc():
            lib.d_stream_sync(stream)
            return acc.to_numpy(np.uint64, 3)
    for b in range(total):
        lib.legion_ipc_client_wait(c)
        lib.legion_ipc_client_read_counters(c, nc, ec)
        t.append(time.perf_counter())
        n, e = nc[5 + 2 * hops], ec[2 + hops]
        edges.append(e)
        nodes.append(n)
        if reading and n > 0:
            rows = min(n, rows_cap) if rows_cap > 0 else n
            before = read_acc() if b == 1 else None
            lib.legion_sum_words(stream, lib.legion_ipc_client_buffer(c, 1), rows * F * 4, acc.ptr)
            lib.legion_sum_words(stream, lib.legion_ipc_client_buffer(c, 3), e * 4, acc.ptr + 8)
            lib.legion_sum_words(stream, lib.legion_ipc_client_buffer(c, 4), e * 4, acc.ptr + 16)
            if b == 1:      # a batch of the warm-up part of epoch 0 (outside every timed window): keep what is needed to check the sums afterwards
                check = {"batch": b, "rows": int(rows), "edges": int(e), "got": (read_acc() - before).tolist(),
                         "ids": K.read_dev(lib.legion_ipc_client_buffer(c, 0), np.int32, rows),
                         "src": K.read_dev(lib.legion_ipc_client_buffer(c, 3), np.int32, e), "dst": K.read_dev(lib.legion_ipc_client_buffer(c, 4), np.int32, e)}
        lib.legion_ipc_client_post(c)                 # hipDeviceSynchronize (the reads above are done) + sem_post
    K.check()
    lib.legion_ipc_client_close(c)
    out = {"steps": list(steps), "hops": hops, "t0": t[0], "t": [round(x - t[0], 7) for x in t], "edges": edges, "nodes": nodes, "consumer": mode}
    if reading:
        import legion1_amd.synth as S
        words = lambda a: int(np.ascontiguousarray(a).view(np.uint32).astype(np.uint64).sum(dtype=np.uint64))   # noqa: E731
        ok = None
        if check is not None:
            spec = S.spec_for(argv[5], scale=float(argv[6]))
            want_f = 0
            for i in range(0, check["rows"], 1 << 16):      # the generator's closed form of the rows the batch names, 65536 rows at a time
                want_f = (want_f + words(S.features(spec, check["ids"][i:i + (1 << 16)]))) & 0xFFFFFFFFFFFFFFFF
            want = [want_f, words(check["src"]), words(check["dst"])]
            ok = [int(g) == int(w) for g, w in zip(check["got"], want)]
            out["checksum"] = {"batch": check["batch"], "rows": check["rows"], "edges": check["edges"], "features_equal_the_generators_rows": ok[0],
                               "coo_src_equal_host_copy": ok[1], "coo_dst_equal_host_copy": ok[2], "sum_of_feature_words": str(check["got"][0])}
        out["read_bytes"] = [int(min(n, rows_cap) if rows_cap > 0 else n) * F * 4 + 8 * int(e) for n, e in zip(nodes, edges)]
        lib.d_stream_destroy(stream)
        acc.free()
    sys.stdout.write(json.dumps(out) + "\n")


def server_audit_line(log_text):
    """The summary a `legion` server leaves under $LEGION_DEVICE_AUDIT=1 (csrc/audit.h), parsed; None when the audit was off."""
    import re
    m = re.search(r"Device audit: (\d+) checks, (\d+) violations, (\d+) unattributed, (\d+) launches with peer arguments", log_text)
    return dict(zip(("checks", "violations", "unattributed", "peer_launches"), (int(x) for x in m.groups()))) if m else None


def own_audit_counts(L):
    """The logical-device audit of THIS process's library ($LEGION_DEVICE_AUDIT=1), or None."""
    import ctypes
    if not L.legion_audit_enabled():
        return None
    cnt = (ctypes.c_int64 * 4)()
    L.legion_audit_counts(cnt)
    return {"checks": cnt[0], "violations": cnt[1], "unattributed": cnt[2], "peer_launches": cnt[3],
            "first_violations": [L.legion_audit_message(i).decode()[:300] for i in range(min(3, L.legion_audit_message_count()))]}


def unlink_served_namespaces(c, only=None):
    """A `legion` server that was KILLED (a failed or hung served leg) leaves its shm slab, the extension object and 2 x depth named semaphores
    per GPU in /dev/shm -- one set per failed leg per run.  Remove them: every namespace a served leg of this run used, or `only`."""
    for ns in ([only] if only else list(getattr(c, "served_namespaces", []))):
        try:
            c.L.legion_ipc_unlink_namespace(ns.encode(), 8)
        except Exception:   # noqa: BLE001 -- cleaning up must never cost the line
            pass


def served_deadline(c):
    """Seconds a served leg gives ITS child processes before it stops them and raises: well inside what the watchdog granted the leg, so that a
    server or consumer that never comes back is a leg that FAILED (reported, exit code 0, the ranks stay in step) and not a leg that hung (exit code 3)."""
    a = c.args
    return time.time() + max(20.0, min(a.extra_timeout, c.budget.left() - Budget.RESERVE_S) - 25.0)


def served_schedule_windows(t, train_step, valid_step, epochs, K_steps, warm):
    """Windows of K consecutive TRAINING batches inside one epoch of the served schedule (CUDA_IPC_Service.cu:219-259: every epoch is
    train_step training batches, then valid_step validation batches), the first `warm` training batches of every epoch left out: seconds per
    window, measured arrival to arrival (t[i + K - 1] - t[i - 1]), and the global batch numbers each one covers."""
    per = train_step + valid_step
    out = []
    for e in range(epochs):
        first = e * per + warm          # every epoch: the pipeline restarts behind the epoch's validation batches
        i = max(first, 1)
        while i + K_steps <= e * per + train_step:
            out.append((t[i + K_steps - 1] - t[i - 1], i))
            i += K_steps
    return out


def served_leg(c, workload, fan, ref_leg, lp=False, cache_bytes=0, consumer="null"):
    """The whole path through the reference's surface, as a trainer sees it: the `legion` server binary -- started as a
    FRESH child process, dataset source `synth:<workload>` (the tables generated in its own HBM by the legion_synth_* calls this file uses),
    pre-sampling epoch, then its default software-pipelined RunOnce loop (runner.cpp; Server.cu:301-328) -- hands every batch of its schedule
    to a second child, a null consumer on legion_ipc_client_* (CUDA_IPC_Service.cu:289-297 / ipc_cuda_kernel.cu:98-107).  Reported: the
    consumer-side ms per training batch (median window of K batches, arrival to arrival), edges/s as the consumer sees them, the schedule,
    and the ratio to `alt_schedule_levels` -- the same two-stream schedule run by this process without the hand-off.
    consumer = "reading": the consumer reads every served feature row and both COO arrays on its own stream before it hands the pipe back
    (served_consumer; the memory traffic of legion_graphsage.py:72-89 without the model) and checks one batch's sums."""
    import shutil
    import tempfile
    args = c.args
    if c.world != 1:
        raise RuntimeError("the served leg runs at N = 1 (one server + one consumer process beside this one)")
    spec, H, B = c.S.spec_for(workload, scale=args.scale), len(fan), c.B
    server = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")
    if not os.path.exists(server):
        raise RuntimeError("the server binary is missing: make -C legion-1_amd/csrc legion")
    n_seeds = spec.n_train
    if lp:      # meta flag 2 on a synth: source: the server generates the [src | pos | neg] seed list itself (one triple per training id)
        if B % 3:
            raise RuntimeError("link-prediction batches need a batch size divisible by 3")
        n_seeds = -(-spec.n_train // (B // 3)) * B
    train_step = (n_seeds - 1) // B
    # windows of K consecutive training batches inside one epoch: a shape with few batches per epoch (products: 24) gets shorter windows
    warm = min(args.warmup, train_step // 4)
    K_win = min(args.steps, train_step - warm - 1)
    if K_win < 4:
        raise RuntimeError("the shape has %d training batches per epoch; at least %d needed" % (train_step, warm + 5))
    ref_leg = ref_leg or {}          # None: no in-process leg of this configuration to compare with (uk-union replicated: its tables and the server's do not fit together)
    ref_ms = ref_leg["elapsed"] / args.steps * 1e3 if ref_leg else 0.6
    epochs = args.served_epochs or int(max(2, min(50, -(-2000.0 // (train_step * ref_ms)))))
    n_eval = min(512, spec.n_valid, spec.n_test)         # one validation / test batch per epoch: the schedule stays training batches
    tmp = tempfile.mkdtemp(prefix="legion_served_")
    src = "synth:%s" % workload + ("" if (args.scale == 1.0 and args.skew == 205) else ":%r" % args.scale) + ("" if args.skew == 205 else ":%d" % args.skew)
    meta = os.path.join(tmp, "meta_config")
    with open(meta, "w") as f:
        f.write("%s %d %d 0 %d %d %d %d %d %d %d" % (src, B, spec.V, spec.F, spec.n_train, n_eval, n_eval, int(cache_bytes), epochs, 2 if lp else 0))
    env = dict(os.environ, LEGION_IPC_NAMESPACE="bs%d_%d_" % (os.getpid(), len(c.children)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    c.served_namespaces = getattr(c, "served_namespaces", []) + [env["LEGION_IPC_NAMESPACE"]]
    if cache_bytes > 0:
        env["LEGION_SYNTH_CACHE"] = "1"      # build the hotness cache on top of the generated tables: cost model, FillUp, cached gather / partitioned sampler
    env.pop("LEGION_LOG", None)          # the server's log is its stdout (a file here), as with the reference
    if c.local_rank != 0:
        env["HIP_VISIBLE_DEVICES"] = str(c.local_rank)
    log_path = os.path.join(tmp, "server.log")
    t0 = time.time()
    deadline = served_deadline(c)
    srv = cons = None
    try:
        with open(log_path, "w") as lf:
            srv = subprocess.Popen([server, "1", "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=tmp)
        c.children.append(srv)
        while "System is ready for serving" not in open(log_path, errors="ignore").read():
            if srv.poll() is not None:
                raise RuntimeError("the server exited before serving: " + open(log_path, errors="ignore").read()[-600:])
            if time.time() > deadline:
                raise RuntimeError("the server was not ready in time: " + open(log_path, errors="ignore").read()[-400:])
            time.sleep(0.05)
        ready_s = time.time() - t0
        cons = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--served-consumer", str(H), str(epochs), "-1", consumer, str(spec.F), workload,
                                 repr(args.scale)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        c.children.append(cons)
        try:
            out, err = cons.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            raise RuntimeError("the consumer did not finish in time (server log: %s)" % open(log_path, errors="ignore").read()[-300:])
        if cons.returncode != 0:
            raise RuntimeError("the consumer failed (%d): %s" % (cons.returncode, (out + err)[-600:]))
        srv.wait(timeout=max(1.0, min(60.0, deadline + 10.0 - time.time())))
        if srv.returncode != 0:
            raise RuntimeError("the server failed (%d): %s" % (srv.returncode, open(log_path, errors="ignore").read()[-600:]))
        log_text = open(log_path, errors="ignore").read()
    finally:
        killed = False
        for p in (cons, srv):
            if p is not None and p.poll() is None:
                p.kill()
                p.wait()
                killed = True
        if killed or (srv is not None and srv.returncode != 0):
            unlink_served_namespaces(c, env["LEGION_IPC_NAMESPACE"])
        shutil.rmtree(tmp, ignore_errors=True)
    got = json.loads(out.strip().splitlines()[-1])
    ts, vs, es = got["steps"]
    if ts != train_step or len(got["t"]) != (ts + vs) * epochs + es:
        raise RuntimeError("served schedule %s x %d epochs does not match the shape (train_step %d)" % (got["steps"], epochs, train_step))
    reading = None
    if consumer == "reading":
        ck = got.get("checksum") or {}
        if not (ck.get("features_equal_the_generators_rows") and ck.get("coo_src_equal_host_copy") and ck.get("coo_dst_equal_host_copy")):
            raise RuntimeError("the reading consumer's checksums of batch 1 do not match what was served: %s" % ck)
        reading = {"checksum": ck}
    t, edges, nodes = got["t"], np.asarray(got["edges"], np.int64), np.asarray(got["nodes"], np.int64)
    wins = served_schedule_windows(t, ts, vs, epochs, K_win, warm)
    secs = np.array([w[0] for w in wins])
    med = float(np.median(secs))
    is_train = np.array([(b % (ts + vs)) < ts for b in range((ts + vs) * epochs)] + [False] * es)
    e_mean, n_mean = float(edges[is_train].mean()), float(nodes[is_train].mean())
    ms = med / K_win * 1e3
    # the batches this process timed (W .. W + K - 1 of the seed list, modulo the epoch) as the trainer received them in epoch 0: same edges, same rows
    same = None
    if ref_leg.get("edges_per_step") is not None:
        idx = (args.warmup + np.arange(args.steps)) % ts
        same = bool(np.array_equal(edges[idx], ref_leg["edges_per_step"]) and np.array_equal(nodes[idx], ref_leg["nodes_per_step"]))
    lv = (ref_leg.get("alt_levels") or {}).get("ms_per_step")
    ov = (ref_leg.get("alt") or {}).get("ms_per_step") if (ref_leg.get("alt") or {}).get("pipeline") == "overlap" else None
    train_t = float(sum(t[e * (ts + vs) + ts - 1] - t[max(e * (ts + vs) - 1, 0)] for e in range(epochs)))
    if reading is not None:
        rb = np.asarray(got["read_bytes"], np.int64)[is_train]
        reading.update(read_GB_per_batch=round(float(rb.mean()) / 1e9, 4), consumer_read_GB_per_s_of_wall_clock=round(float(rb.mean()) / (ms * 1e-3) / 1e9, 1),
                       reads="every served feature row ([nc9, F] view) and both COO arrays, summed word by word on the consumer's own stream before the pipe "
                             "goes back (legion_sum_words): the memory traffic of legion_graphsage.py:72-89 without the model")
    return {"what": "the `legion` server binary (fresh child process; dataset source %s: tables generated in its HBM; pre-sampling epoch; default "
                    "software-pipelined RunOnce, 2 streams, depth-2 pipes) serving a %s consumer process over shm + semaphores + HIP-IPC buffers "
                    "(legion_ipc_client_*: wait, read counters, %spost); times are the CONSUMER's clock, arrival to arrival"
                    % (src, consumer, "read the whole batch, " if reading is not None else ""),
            "consumer": consumer, "reading_consumer": reading,
            "value": round(e_mean / (ms * 1e-3), 1), "unit": "edges/s", "ms_per_step": round(ms, 4),
            "feature_GBps": round(n_mean * 4 * spec.F / (ms * 1e-3) / 1e9, 2), "batch": B, "fanout": list(fan), "V": spec.V, "F": spec.F,
            "schedule": {"train_steps": ts, "valid_steps": vs, "test_steps": es, "epochs": epochs, "batches_served": len(t),
                         "eval_batch_seeds": n_eval, "seeds": "link-prediction [src | pos | neg] lists generated by the server (meta flag 2)" if lp else "training ids"},
            "windows": len(wins), "steps_per_window": K_win,
            "window_ms_min_median_max": [round(float(secs.min()) * 1e3, 4), round(med * 1e3, 4), round(float(secs.max()) * 1e3, 4)],
            "all_training_batches_ms_per_step": round(train_t / (ts * epochs) * 1e3, 4),
            "edges_per_batch": round(e_mean, 1), "unique_nodes_per_batch": round(n_mean, 1),
            "served_batches_equal_the_timed_ones": same,
            "ms_per_step_same_schedule_in_process": lv, "ratio_to_alt_schedule_levels": round(ms / lv, 4) if lv else None,
            "ms_per_step_overlap_in_process": ov, "ratio_to_alt_schedule": round(ms / ov, 4) if ov else None,
            "ms_per_step_serial_in_process": round(ref_ms, 4) if ref_leg else None,
            # which gather formulation the server's runner chose after its pre-sampling epoch (LEGION_RUNNER_GATHER=auto): "per level" is the
            # schedule of alt_schedule_levels, "one launch over all rows" that of alt_schedule (overlap)
            "server_gather": next((ln.split("Runner gather:")[1].strip() for ln in log_text.splitlines() if "Runner gather:" in ln),
                                  os.environ.get("LEGION_RUNNER_GATHER")),
            # (sampler + gather algorithmic bytes of the K timed batches, per batch) / served time per batch / 8 TB/s
            # served rows x (8F + 8) + the sampler's algorithmic bytes of the timed batches (in-process census; absent: gather bytes only)
            "pipeline_frac": round(float(ref_leg["job_bytes"]) / args.steps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ref_leg
                             else round(n_mean * (8 * spec.F + 8) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "server_ready_s": round(ready_s, 2), "server_tables": "generated in HBM" if "Tables generated in HBM" in log_text else "?",
            "server_cache": next((ln.strip() for ln in log_text.splitlines() if ln.startswith("Feat capacity")), None) if cache_bytes > 0 else None,
            "server_first_epoch_s": next((float(ln.split(":")[1].split()[0]) for ln in log_text.splitlines() if ln.startswith("First epoch cost")), None),
            "server_device_audit": server_audit_line(log_text),
            "processes": "bench.py (idle) + legion + consumer"}


def served_all_leg(c):
    """N > 1: the reference's own deployment on the N GPUs of the node -- ONE `legion` server process driving all of them (a runner thread per GPU,
    Server.cu:116-135; dataset source synth:<workload>, a replica of the tables generated in every GPU's HBM, seeds split tid % N) and N trainer-side
    consumer processes, one per GPU (legion_ipc_client_*).  Started by rank 0 as fresh child processes while the bench's own ranks wait in the leg's
    agreement collective; consumer-clock windows per GPU, the job's rate = sum of the GPUs' edges per batch / the slowest GPU's time per batch.
    Never run on more than one physical GPU in any round (two logical GPUs on one device in tests/test_gpu_bench_legs.py): on a node this is a first run."""
    import shutil
    import tempfile
    args, N = c.args, c.world
    if c.shared_device and 2 * N + 1 > 6:
        return {"skipped": "a shared-device rehearsal may hold at most 6 GPU processes; this leg needs %d (N ranks + server + N consumers)" % (2 * N + 1)}
    workload = args.workload
    spec, fan, H, B = c.S.spec_for(workload, scale=args.scale), c.head_fan, len(c.head_fan), args.batch
    server = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")
    per_gpu = spec.n_train // N
    train_step = (per_gpu - 1) // B - 1          # lower bound (the split is by tid % N: shards differ by a few ids)
    warm = min(args.warmup, max(train_step, 0) // 4)
    K_win = min(args.steps, train_step - warm - 1)
    if K_win < 4:
        raise RuntimeError("the shape has about %d training batches per GPU and epoch; more needed" % train_step)
    epochs = args.served_epochs or int(max(2, min(50, -(-2000.0 // (train_step * 0.5)))))
    n_eval = min(512 * N, spec.n_valid, spec.n_test)
    tmp = tempfile.mkdtemp(prefix="legion_served_all_")
    src = "synth:%s" % workload + ("" if (args.scale == 1.0 and args.skew == 205) else ":%r" % args.scale) + ("" if args.skew == 205 else ":%d" % args.skew)
    meta = os.path.join(tmp, "meta_config")
    with open(meta, "w") as f:
        f.write("%s %d %d 0 %d %d %d %d 0 %d 0" % (src, B, spec.V, spec.F, spec.n_train, n_eval, n_eval, epochs))
    env = dict(os.environ, LEGION_IPC_NAMESPACE="ba%d_%d_" % (os.getpid(), len(c.children)), HSA_ENABLE_IPC_MODE_LEGACY="0")
    c.served_namespaces = getattr(c, "served_namespaces", []) + [env["LEGION_IPC_NAMESPACE"]]
    for k in ("LEGION_LOG", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LEGION_IPC_DEVICE"):
        env.pop(k, None)
    log_path = os.path.join(tmp, "server.log")
    t0 = time.time()
    deadline = served_deadline(c)
    srv, cons = None, []
    try:
        with open(log_path, "w") as lf:
            srv = subprocess.Popen([server, str(N), "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=tmp)
        c.children.append(srv)
        while "System is ready for serving" not in open(log_path, errors="ignore").read():
            if srv.poll() is not None:
                raise RuntimeError("the server exited before serving: " + open(log_path, errors="ignore").read()[-600:])
            if time.time() > deadline:
                raise RuntimeError("the server was not ready in time: " + open(log_path, errors="ignore").read()[-400:])
            time.sleep(0.05)
        ready_s = time.time() - t0
        for g in range(N):
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--served-consumer", str(H), str(epochs), str(g)], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            cons.append(p)
            c.children.append(p)
        outs = []
        for g, p in enumerate(cons):
            try:
                out, err = p.communicate(timeout=max(1.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                raise RuntimeError("consumer %d did not finish in time (server log: %s)" % (g, open(log_path, errors="ignore").read()[-300:]))
            if p.returncode != 0:
                raise RuntimeError("consumer %d failed (%d): %s" % (g, p.returncode, (out + err)[-500:]))
            outs.append(json.loads(out.strip().splitlines()[-1]))
        srv.wait(timeout=max(1.0, min(60.0, deadline + 10.0 - time.time())))
        if srv.returncode != 0:
            raise RuntimeError("the server failed (%d): %s" % (srv.returncode, open(log_path, errors="ignore").read()[-600:]))
        log_text = open(log_path, errors="ignore").read()
    finally:
        killed = False
        for p in cons + [srv]:
            if p is not None and p.poll() is None:
                p.kill()
                p.wait()
                killed = True
        if killed or (srv is not None and srv.returncode != 0):
            unlink_served_namespaces(c, env["LEGION_IPC_NAMESPACE"])
        shutil.rmtree(tmp, ignore_errors=True)
    ts, vs, es = outs[0]["steps"]
    per_gpu_ms, per_gpu_edges, per_gpu_nodes, n_windows = [], [], [], []
    for got in outs:
        if got["steps"] != [ts, vs, es] or len(got["t"]) != (ts + vs) * epochs + es:
            raise RuntimeError("the consumers disagree about the schedule: %s vs %s" % (got["steps"], [ts, vs, es]))
        wins = served_schedule_windows(got["t"], ts, vs, epochs, min(K_win, ts - warm - 1), warm)
        is_train = np.array([(b % (ts + vs)) < ts for b in range((ts + vs) * epochs)] + [False] * es)
        per_gpu_ms.append(float(np.median([w[0] for w in wins])) / min(K_win, ts - warm - 1) * 1e3)
        per_gpu_edges.append(float(np.asarray(got["edges"], np.int64)[is_train].mean()))
        per_gpu_nodes.append(float(np.asarray(got["nodes"], np.int64)[is_train].mean()))
        n_windows.append(len(wins))
    slow = max(per_gpu_ms)
    same = None
    if c.head.get("edges_per_step") is not None:      # rank 0's own census of its shard's batches W .. W + K - 1 against what GPU 0's trainer received
        idx = (args.warmup + np.arange(args.steps)) % ts
        same = bool(np.array_equal(np.asarray(outs[0]["edges"], np.int64)[idx], c.head["edges_per_step"]))
    span = [o["t0"] for o in outs]
    return {"what": "ONE `legion` server process over the %d GPUs (a runner thread and a replica of the %s tables per GPU, seeds split tid %% %d) + %d consumer "
                    "processes on legion_ipc_client_* -- the reference's deployment (Server.cu:116-135, legion_graphsage.py:186-190); consumer-clock windows per GPU; "
                    "value = sum of the GPUs' edges per batch / the slowest GPU's time per batch" % (N, src, N, N)
                    + ("; REHEARSAL: every logical GPU sits on ONE physical device" if c.shared_device else ""),
            "value": round(sum(per_gpu_edges) / (slow * 1e-3), 1), "unit": "edges/s", "ms_per_step": round(slow, 4), "n_gpus": N,
            "ms_per_step_per_gpu": [round(x, 4) for x in per_gpu_ms], "edges_per_batch_per_gpu": [round(x, 1) for x in per_gpu_edges],
            "feature_GBps": round(sum(per_gpu_nodes) * 4 * spec.F / (slow * 1e-3) / 1e9, 2),
            "schedule": {"train_steps": ts, "valid_steps": vs, "test_steps": es, "epochs": epochs, "batches_served_per_gpu": len(outs[0]["t"])},
            "windows_per_gpu": n_windows, "steps_per_window": min(K_win, ts - warm - 1),
            "first_batch_spread_ms": round((max(span) - min(span)) * 1e3, 3),
            "gpu0_batches_equal_rank0s_timed_ones": same,
            "server_ready_s": round(ready_s, 2),
            "server_gather": [ln.split("Runner gather:")[1].strip()[:60] for ln in log_text.splitlines() if "Runner gather:" in ln][:N],
            "server_device_audit": server_audit_line(log_text),
            "shared_device": bool(c.shared_device), "processes": "%d bench ranks (idle) + legion + %d consumers" % (N, N)}


def leg_summary(c, leg, what):
    args, F = c.args, c.spec.F
    el = leg["elapsed"]
    g = leg["g_ms"]
    reps = max(1, len(g) // args.steps)
    ach = float(leg["gather_bytes"].sum()) * reps / (g.sum() * 1e-3) / 1e9 if len(g) else None
    samp_us = (el / args.steps * 1e6 - float(g.mean()) * 1e3) if len(g) else None     # serial schedule: the batch minus its gather launch
    alt = leg.get("alt") or {}
    return {"what": what, "value": round(leg["job_edges"] / el, 1), "unit": "edges/s", "ms_per_step": round(el / args.steps * 1e3, 4),
            "feature_GBps": round(leg["job_nodes"] * 4 * F / el / 1e9, 2), "batch": c.B, "fanout": c.fan, "V": c.spec.V, "E": c.E, "F": F,
            "edges_per_batch": round(leg["job_edges"] / (args.steps * c.world), 1),
            "unique_nodes_per_batch": round(leg["job_nodes"] / (args.steps * c.world), 1),
            "gather_avg_launch_us": round(float(g.mean()) * 1e3, 2) if len(g) else None,
            "gather_frac_of_hbm_peak": round(ach / HBM_PEAK_GBPS, 4) if ach else None,
            "sampler_us_per_batch": round(samp_us, 1) if samp_us is not None else None,
            "sampler_algorithmic_bytes_per_batch": int(leg["samp_bytes"].mean()),
            "gather_algorithmic_bytes_per_batch": int(leg["gather_bytes"].mean()),
            # (sampler + gather algorithmic bytes) / time / (8 TB/s x N): the whole batch against the HBM roofline
            "pipeline_frac": round(leg["job_bytes"] / el / 1e9 / (HBM_PEAK_GBPS * c.world), 4),
            "value_overlap": alt.get("value"), "ms_per_step_overlap": alt.get("ms_per_step"), "pipeline_frac_overlap": alt.get("pipeline_frac"),
            "ms_per_step_levels": (leg.get("alt_levels") or {}).get("ms_per_step"), "pipeline_frac_levels": (leg.get("alt_levels") or {}).get("pipeline_frac"),
            "windows": len(leg["windows"]), "graph_gen_s": round(c.gen_s, 2), **(leg["cache_info"] or {}), **(leg["xgmi"] or {})}


def make_seeds(c):
    """This rank's seed list + labels on the device.  node: train ids with tid % world == rank
    (GPUGraphStore.cu:332-346).  lp: [src | pos | neg] batches over the triples whose src % world == rank."""
    import torch
    args, L, spec, dev = c.args, c.L, c.spec, c.dev
    V = spec.V
    all_train = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, all_train.data_ptr(), 0, spec.n_train, V, spec.M2, spec.C2, 1, 0)
    torch.cuda.synchronize()
    mask = (all_train % c.world) == c.rank
    mine = all_train[mask].contiguous()              # == dist.shard_seeds(all_train, rank, world)
    if args.task == "lp":
        if c.B % 3:
            raise SystemExit("--task lp needs a batch size divisible by 3 ([src | pos | neg] thirds, lp_sage.py:87-90)")
        k = c.B // 3
        # one triple per training id; triple t of the global list belongs to rank src_t % N and keeps its global
        # number for the pos / neg draws, so the lists of all ranks together are the 1-rank list re-dealt
        triple_no = torch.nonzero(mask).reshape(-1).contiguous()
        n_tr = int(mine.numel())
        seeds = torch.empty((n_tr + k - 1) // k * c.B, dtype=torch.int32, device=dev)
        L.legion_synth_lp_seeds(None, seeds.data_ptr(), mine.data_ptr(), triple_no.data_ptr(), n_tr, c.B, c.indptr.data_ptr(),
                                c.indices.data_ptr(), V, 1)
        torch.cuda.synchronize()
        c.K.check()
        mine = seeds
        del triple_no
    del all_train, mask
    labels_all = torch.empty(V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, labels_all.data_ptr(), 0, V, spec.classes)
    torch.cuda.synchronize()
    c.my_labels = labels_all[mine.long()].contiguous()
    del labels_all
    c.mine = mine
    c.n_mine = int(mine.numel())
    c.steps_avail = max(1, (c.n_mine - 1) // c.B)  # train_step = (n-1)/B, CUDA_IPC_Service.cu:89


def run_leg(c, unified, headline, min_time=None, with_alt=False):
    """W warm-up steps, then R windows of exactly K timed steps (barrier + synchronize on both sides of every
    window, max over ranks per window, median over windows).  Returns the raw numbers of the leg."""
    import torch
    args, K, D, L = c.args, c.K, c.D, c.L
    min_time = args.min_time if min_time is None else min_time
    rank, world, dev = c.rank, c.world, c.dev
    V, F, B, fan, H = c.spec.V, c.spec.F, c.B, c.fan, c.H
    G = world if unified else 1          # logical GPUs of the clique this process knows about
    me = rank if unified else 0          # the one this process drives
    for g in range(G):
        L.legion_set_device_map(g, c.local_rank)
    if me != 0:
        L.legion_audit_alias(0, me)     # one process per GPU: this process's device is GPU 0 of its replicated engine and GPU <rank> of the clique
    L.SetGPUDevice(me)
    n_mine = c.n_mine
    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32))
    seeds = dict(train=[((c.mine.data_ptr(), n_mine), (c.my_labels.data_ptr(), n_mine)) if g == me else empty for g in range(G)])
    overlap = args.pipeline == "overlap" and args.gather == "all" and headline
    intra = args.pipeline == "intra" and headline
    depth = 2      # the reference's PIPELINE_DEPTH; the serial schedule only uses pipe 0
    eng = K.Engine(c.indptr.data_ptr(), c.indices.data_ptr(), c.feat_ptr, V, F, seeds, B, fan, G=G,
                   csr_location=K.LOC_DEVICE, features_location=c.feat_loc, E=c.E, pipeline_depth=depth,
                   local_devs=[me], train_step=max(1, args.presc_steps), features_pitch=c.pitch if c.feat_loc == K.LOC_DEVICE else 0)
    eng.alloc_features()
    cache_info = None
    if unified:
        cache_info = build_unified_cache(args, K, D, L, eng, me, world, V, F, B, fan, dev)
    else:
        L.GPUCache_SetPreSc(eng.cache, 0)  # steady state: no pre-sampling epoch in the all-resident configuration
    if args.cu_split > 0 and headline:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (n_cu + 31) // 32
        owner = [(i % 8 if args.cu_pattern == "mod" else (i // 32) % 8) < args.cu_split for i in range(words * 32)]
        m_s = np.array([sum(1 << b for b in range(32) if owner[w * 32 + b] and w * 32 + b < n_cu) for w in range(words)], dtype=np.uint32)
        m_g = np.array([sum(1 << b for b in range(32) if not owner[w * 32 + b] and w * 32 + b < n_cu) for w in range(words)], dtype=np.uint32)
        stream = L.d_stream_create_cu_mask(m_s.ctypes.data, words)
        gstream2 = L.d_stream_create_cu_mask(m_g.ctypes.data, words)
    elif args.stream_priority != "none" and headline:
        stream = L.d_stream_create_priority(1 if args.stream_priority == "sampler" else 0)
        gstream2 = L.d_stream_create_priority(1 if args.stream_priority == "gather" else 0)
    else:
        # ONE pair of streams per process, shared by every leg.  HIP maps streams onto a few hardware queues; a leg that created its own pair
        # could land both streams on one queue, depending on how many streams earlier legs had created -- its two-stream schedules then ran
        # SLOWER than the serial one (products {25,10} directly behind the headline: overlap 0.283 / levels 0.317 ms against 0.233 / 0.245 ms
        # behind another leg, profiles/r05_stream_pair.md).  The first two streams of a process are what the `legion` runner uses, too.
        if getattr(c, "stream_pair", None) is None:
            c.stream_pair = (L.d_stream_create(), L.d_stream_create())
        stream, gstream2 = c.stream_pair   # sampler stream; gather stream of the overlapped schedule (reference: streams_[1], Server.cu:178-181)
    steps_avail = c.steps_avail
    K_steps, W = args.steps, args.warmup

    per_level = args.gather == "level" and not intra and headline
    ev_hop = [L.d_event_create() for _ in range(H + 1)]   # intra: hop h of the running batch is complete
    ev_gdone = L.d_event_create()                          # intra: every gather of the running batch is complete
    intra_started = [False]
    log = K.DevBuf((K_steps + W) * 128)  # nc/ec of every step, copied on-stream
    ev = [(L.d_event_create(), L.d_event_create()) for _ in range(K_steps)]
    pool = eng.pools[me]
    exchange = None
    if unified:
        # The owner-computes exchange variant runs FIRST: it never touches a peer's memory, so its numbers exist (and are what the
        # watchdog prints) even if the HIP-IPC import of the peers' shards -- needed by the in-kernel variant below -- never returns
        # (that import hung in rounds 1 and 2 for single allocations of 2 GiB and more: profiles/r02_ipc_limit.md).
        if not headline and world > 1 and not args.no_exchange_leg:
            try:
                inj = os.environ.get("LEGION_BENCH_INJECT_ERROR", "").split(":")     # test hook (tests/test_gpu_bench_legs.py)
                if inj[0] == "exchange" and (len(inj) < 2 or int(inj[1]) == rank):
                    raise RuntimeError("injected failure of the exchange variant on rank %d" % rank)
                exchange = exchange_leg(c, eng, me, pool, stream, steps_avail)
            except Exception as ex:  # noqa: BLE001 -- reported inside the line, never fatal
                exchange = {"error": repr(ex)[:300]}
                if getattr(c, "guard", None) is not None:
                    c.guard.announce("exchange variant: " + repr(ex))
            # agree right here (under the leg's armed timer): a symmetric or recoverable failure must not cost the in-kernel
            # numbers and the legs behind -- all ranks go on together; a rank stuck inside the exchange never gets here and
            # the watcher ends the run (LegGuard.announce)
            ok_x = not (isinstance(exchange, dict) and "error" in exchange)
            flags = D.allgather_object(ok_x, world)
            if getattr(c, "guard", None) is not None:
                c.guard.part_agreed()
            if ok_x and not all(flags):
                exchange = {"error": "failed on rank(s) %s" % [i for i, f in enumerate(flags) if not f]}
            if getattr(c, "guard", None) is not None:
                c.guard.partial = dict(cache_info, what="only the exchange variant ran", exchange_variant=exchange)
        cache_info["shard_import_s"] = import_peer_shards(D, eng, me, world)
    ev_sampled = [L.d_event_create() for _ in range(depth)]   # sampling of the batch in pipe q is complete
    ev_gathered = [L.d_event_create() for _ in range(depth)]  # gather of the batch in pipe q is complete
    used = [False] * depth

    census = [True]   # copy nc/ec of the batch into the log (the untimed census pass only: two 64-byte copies cost ~14 us of stream time)

    def step_levels(i, it):
        """The `legion` server's schedule (runner.cpp, Server.cu:301-328 + its depth-2 pipes): the rows of level l are gathered on the
        second stream behind the event of the op that produced them, while hop l + 1 -- and then the next batch, which uses the other
        pipe -- is sampled on the first; a pipe is reused only after its last gather finished.  Untimed per kernel (whole windows only)."""
        q = i % depth
        L.GPUMemoryPool_SetCurrentPipe(pool, q)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        if used[q]:
            L.d_stream_wait_event(stream, ev_gathered[q])
        L.batch_generator_kernel(stream, eng.noder, eng.cache, pool, B, it, me, me, K.TRAINMODE)
        L.d_event_record(ev_hop[0], stream)
        L.d_stream_wait_event(gstream2, ev_hop[0])
        L.get_feature_kernel(gstream2, eng.cache, eng.noder, pool, me, 1, 1)
        for h in range(H):
            L.GPU_Random_Sampling(stream, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 0)
            L.d_event_record(ev_hop[h + 1], stream)
            L.d_stream_wait_event(gstream2, ev_hop[h + 1])
            L.get_feature_kernel(gstream2, eng.cache, eng.noder, pool, me, 2 * h + 3, 1)
        L.d_event_record(ev_gathered[q], gstream2)
        used[q] = True
        L.make_update_plan(stream, eng.graph, eng.cache, pool, me, K.TRAINMODE)
        L.update_cache(stream, eng.cache, eng.noder, pool, me, K.TRAINMODE)

    def step(i, timed_idx=None, overlap=overlap, levels=False):
        """One mini-batch.  overlap: depth-2 pipes (the reference's PIPELINE_DEPTH), the sampler of batch
        i+1 runs on `stream` while the gather of batch i runs on `gstream`; a pipe's buffers are reused
        only after its gather finished.  levels: step_levels."""
        it = i % steps_avail
        if levels:
            return step_levels(i, it)
        q = i % depth if overlap else 0
        gstream = gstream2 if overlap else stream
        if intra and not overlap:
            return step_intra(i, it, timed_idx)
        o = eng.out[me][q]
        L.GPUMemoryPool_SetCurrentPipe(pool, q)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        if overlap and used[q]:
            L.d_stream_wait_event(stream, ev_gathered[q])
        L.batch_generator_kernel(stream, eng.noder, eng.cache, pool, B, it, me, me, K.TRAINMODE)
        if per_level:
            L.get_feature_kernel(stream, eng.cache, eng.noder, pool, me, 1, 1)
        for h in range(H):
            L.GPU_Random_Sampling(stream, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 0)
            if per_level:
                L.get_feature_kernel(stream, eng.cache, eng.noder, pool, me, 2 * h + 3, 1)
        if census[0]:
            L.d_copy_async(log.ptr + i * 128, o["nc"].ptr, 64, stream)
            L.d_copy_async(log.ptr + i * 128 + 64, o["ec"].ptr, 64, stream)
        if not per_level:
            if overlap:
                L.d_event_record(ev_sampled[q], stream)
                L.d_stream_wait_event(gstream, ev_sampled[q])
            if timed_idx is not None:
                L.d_event_record(ev[timed_idx][0], gstream)
            L.get_feature_kernel_all(gstream, eng.cache, eng.noder, pool, me, 1)
            if timed_idx is not None:
                L.d_event_record(ev[timed_idx][1], gstream)
            if overlap:
                L.d_event_record(ev_gathered[q], gstream)
                used[q] = True
        L.make_update_plan(stream, eng.graph, eng.cache, pool, me, K.TRAINMODE)
        L.update_cache(stream, eng.cache, eng.noder, pool, me, K.TRAINMODE)

    def step_intra(i, it, timed_idx):
        """The reference's schedule inside one batch: FeatureExtractor ops on the second stream behind the event of
        the op that produced their rows (Server.cu:309-316); the next batch starts when the last gather is done."""
        o = eng.out[me][0]
        L.GPUMemoryPool_SetCurrentPipe(pool, 0)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        if intra_started[0]:
            L.d_stream_wait_event(stream, ev_gdone)
        L.batch_generator_kernel(stream, eng.noder, eng.cache, pool, B, it, me, me, K.TRAINMODE)
        L.d_event_record(ev_hop[0], stream)
        L.d_stream_wait_event(gstream2, ev_hop[0])
        L.get_feature_kernel(gstream2, eng.cache, eng.noder, pool, me, 1, 1)
        for h in range(H):
            L.GPU_Random_Sampling(stream, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 0)
            L.d_event_record(ev_hop[h + 1], stream)
            L.d_stream_wait_event(gstream2, ev_hop[h + 1])
            last = h == H - 1
            if last and timed_idx is not None:
                L.d_event_record(ev[timed_idx][0], gstream2)
            L.get_feature_kernel(gstream2, eng.cache, eng.noder, pool, me, 2 * h + 3, 1)
            if last and timed_idx is not None:
                L.d_event_record(ev[timed_idx][1], gstream2)
        L.d_event_record(ev_gdone, gstream2)
        intra_started[0] = True
        if census[0]:
            L.d_copy_async(log.ptr + i * 128, o["nc"].ptr, 64, stream)
            L.d_copy_async(log.ptr + i * 128 + 64, o["ec"].ptr, 64, stream)
        L.make_update_plan(stream, eng.graph, eng.cache, pool, me, K.TRAINMODE)
        L.update_cache(stream, eng.cache, eng.noder, pool, me, K.TRAINMODE)

    def drain():
        L.d_stream_sync(stream)
        L.d_stream_sync(gstream2)
        torch.cuda.synchronize()

    def window(timed=True, **kw):
        """EXACTLY K steps between (synchronize + barrier) and (synchronize + barrier)."""
        drain()
        D.barrier(world)
        t_start = time.perf_counter()
        for i in range(K_steps):
            step(W + i, timed_idx=i if timed else None, **kw)
        drain()
        el = time.perf_counter() - t_start
        D.barrier(world)
        return el

    xc = None
    if unified and rank == 0 and world > 1:
        import legion1_amd.xgmi_counters as xc
    for i in range(W):
        step(i)
    # census (untimed): the K batches of the timed windows once with their counters logged -- what the windows produce
    # (edges, rows, algorithmic bytes).  The windows replay exactly these batches without the instrumentation.
    for i in range(K_steps):
        step(W + i)
    drain()
    census[0] = False
    g_ms = []                      # HIP-event time of every timed gather launch (its own stream)
    x0, t_x0 = (xc.read(), time.perf_counter()) if xc else (None, 0.0)
    windows = [window()]
    if not per_level:
        g_ms += [L.d_event_elapsed_ms(a, b) for a, b in ev]
    # every rank runs the same number of windows: R from the slowest rank's first window
    first_max, _ = D.aggregate(windows[0], [0.0], world, device=dev)
    reps = int(min(args.max_reps, max(1, -(-min_time // max(first_max, 1e-6)))))
    for _ in range(reps - 1):
        windows.append(window())
        if not per_level:
            g_ms += [L.d_event_elapsed_ms(a, b) for a, b in ev]
    K.check()
    xgmi_hw = xc.rate(x0, xc.read(), time.perf_counter() - t_x0, world) if xc else None

    counters = log.to_numpy(np.int32, (K_steps + W) * 32).reshape(K_steps + W, 2, 16)[W:]
    edges = counters[:, 1, 2 + H].astype(np.int64)          # ec[2+H]: cumulative edges of the batch
    nodes = counters[:, 0, 5 + 2 * H].astype(np.int64)      # nc[5+2H]: unique nodes (rows gathered)
    cum = [np.zeros(K_steps, dtype=np.int64)] + [counters[:, 1, 2 + h].astype(np.int64) for h in range(1, H + 1)]
    e_h = [cum[h] - cum[h - 1] for h in range(1, H + 1)]                       # edges sampled in hop h
    n_in = [counters[:, 0, 4].astype(np.int64)] + e_h[:-1]                     # input slots of hop h
    u_h = [counters[:, 0, 4 + 2 * h].astype(np.int64) for h in range(1, H + 1)]  # new unique nodes of hop h
    # algorithmic bytes: sampler 20*N_h + 28*E_h + 8*U_h per hop, gather (8F+8) per row
    samp_bytes = sum(20 * n_in[h] + 28 * e_h[h] + 8 * u_h[h] for h in range(H))
    gather_bytes = nodes * (8 * F + 8)
    tot_edges, tot_nodes = int(edges.sum()), int(nodes.sum())

    # per window: max over ranks; then the median window.  Totals: sum over ranks (every window runs the same K batches)
    win_max = D.aggregate_max_vec(windows, world, device=dev)
    _, (job_edges, job_nodes, job_bytes) = D.aggregate(
        0.0, [tot_edges, tot_nodes, float(samp_bytes.sum() + gather_bytes.sum())], world, device=dev)
    elapsed_max = float(np.median(win_max))

    leg = dict(unified=unified, cache_info=cache_info, elapsed=elapsed_max, windows=[round(w * 1e3, 4) for w in win_max],
               job_edges=job_edges, job_nodes=job_nodes, job_bytes=job_bytes, samp_bytes=samp_bytes, gather_bytes=gather_bytes,
               edges_per_step=edges, nodes_per_step=nodes, u_h=u_h, g_ms=np.array(g_ms, dtype=np.float64), per_level=per_level, intra=intra, overlap=overlap, alt=None, alt_levels=None, graph=None,
               xgmi=None, exchange=exchange, xgmi_hw=xgmi_hw)

    # the other schedule on the very same K batches, in windows like the headline (median window): with --pipeline serial this
    # is the two-stream schedule the `legion` server runs (gather of batch i on stream 1 while batch i+1 is sampled)
    if (headline or with_alt) and not per_level and not intra and not args.headline_only:
        alt_w = [window(timed=False, overlap=not overlap)]
        a_first, _ = D.aggregate(alt_w[0], [0.0], world, device=dev)
        a_reps = int(min(args.max_reps, max(1, -(-(min_time / 2) // max(a_first, 1e-6)))))
        for _ in range(a_reps - 1):
            alt_w.append(window(timed=False, overlap=not overlap))
        alt_max = float(np.median(D.aggregate_max_vec(alt_w, world, device=dev)))
        leg["alt"] = {"pipeline": "serial" if overlap else "overlap", "ms_per_step": round(alt_max / K_steps * 1e3, 4),
                      "value": round(job_edges / alt_max, 1), "unit": "edges/s", "windows": len(alt_w),
                      "feature_GBps": round(job_nodes * 4 * F / alt_max / 1e9, 2),
                      "pipeline_frac": round(job_bytes / alt_max / 1e9 / (HBM_PEAK_GBPS * world), 4)}
        if not overlap:
            # ... and the server's own two-stream schedule: per-level gathers behind the hop events + depth-2 pipes
            drain()
            used[:] = [False] * depth
            lv_w = [window(timed=False, levels=True) for _ in range(max(1, a_reps))]
            lv_max = float(np.median(D.aggregate_max_vec(lv_w, world, device=dev)))
            leg["alt_levels"] = {"pipeline": "levels: per-level gathers on stream 1 behind the hop events, next batch on the other pipe (the `legion` server's loop)",
                                 "ms_per_step": round(lv_max / K_steps * 1e3, 4), "value": round(job_edges / lv_max, 1), "unit": "edges/s",
                                 "windows": len(lv_w), "feature_GBps": round(job_nodes * 4 * F / lv_max / 1e9, 2),
                                 "pipeline_frac": round(job_bytes / lv_max / 1e9 / (HBM_PEAK_GBPS * world), 4)}
            drain()
            used[:] = [False] * depth

    # the serial schedule again, recorded once as a hipGraph and replayed with one launch per batch (same K batches)
    if headline and not per_level and not intra and not args.headline_only and world == 1:   # informational leg: N = 1 only, never fatal
        try:
            L.legion_set_error_mode(K.ERR_RETURN)     # a HIP error in this leg raises (K.check) instead of exit(1)
            L.GPUMemoryPool_SetCurrentPipe(pool, 0)
            hgraph = eng.capture_batch(me, pipe=0, per_level=False, plan=True, stream=stream)
            eng.run_graph(hgraph, W % steps_avail, sync=True)
            t_g = time.perf_counter()
            for i in range(K_steps):
                eng.run_graph(hgraph, (W + i) % steps_avail, sync=False)
            drain()
            g_max = time.perf_counter() - t_g
            leg["graph"] = {"pipeline": "serial, one hipGraph launch per batch", "ms_per_step": round(g_max / K_steps * 1e3, 4),
                            "value": round(job_edges / g_max, 1), "unit": "edges/s"}
        except Exception as ex:   # noqa: BLE001 -- the headline line must still be printed
            leg["graph"] = {"error": repr(ex)[:200]}
        finally:
            L.legion_set_error_mode(K.ERR_EXIT)

    if unified and not per_level:
        leg["xgmi"] = unified_cache_traffic(c, eng, me, cache_info, float(leg["g_ms"].mean()))
    D.barrier(world)   # nobody unmaps a cache shard while a peer may still read it
    drain()
    eng.close()
    log.free()
    return leg


def exchange_leg(c, eng, me, pool, stream, steps_avail):
    """The same K batches with the owner-computes exchange gather (legion-1_amd/exchange.py) instead of in-kernel peer loads:
    per batch one all-to-all of request lists and one of rows over RCCL, pre-allocated buffers, ONE host synchronisation per
    batch (the split sizes), nothing waited for at the end of a batch.  Timed in windows of K steps like every other leg."""
    import torch
    from legion1_amd.exchange import ExchangeGather
    args, K, D, L = c.args, c.K, c.D, c.L
    world, dev, B, fan, H, F = c.world, c.dev, c.B, c.fan, c.H, c.spec.F
    xg = ExchangeGather(K, eng, me, world, F, dev, eng.num_ids)
    K_steps, W = args.steps, args.warmup
    ev = [(L.d_event_create(), L.d_event_create()) for _ in range(K_steps)]
    info = [None]

    def step(i, timed_idx=None):
        L.GPUMemoryPool_SetCurrentPipe(pool, 0)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        L.batch_generator_kernel(stream, eng.noder, eng.cache, pool, B, i % steps_avail, me, me, K.TRAINMODE)
        for h in range(H):
            L.GPU_Random_Sampling(stream, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 0)
        if timed_idx is not None:
            L.d_event_record(ev[timed_idx][0], stream)
        info[0] = xg.run(stream, pool)          # leaves `stream` waiting for the scatter
        if timed_idx is not None:
            L.d_event_record(ev[timed_idx][1], stream)

    def drain():
        L.d_stream_sync(stream)
        xg.wait()
        torch.cuda.synchronize()

    def window():
        drain()
        D.barrier(world)
        t0 = time.perf_counter()
        for i in range(K_steps):
            step(W + i, timed_idx=i)
        drain()
        el = time.perf_counter() - t0
        D.barrier(world)
        return el

    for i in range(max(W, 2)):      # also grows the row buffers to their steady-state size
        step(i)
    drain()
    # census (untimed): edges and rows of the K batches the windows replay
    o = eng.out[me][0]
    tot_e = tot_n = 0
    for i in range(K_steps):
        step(W + i)
        drain()
        tot_e += int(o["ec"].to_numpy(np.int32, 16)[2 + H])
        tot_n += int(o["nc"].to_numpy(np.int32, 16)[5 + 2 * H])
    _, (job_edges, job_nodes) = D.aggregate(0.0, [tot_e, tot_n], world, device=dev)
    allocs_before = xg.allocations
    xc = None
    if c.rank == 0:
        import legion1_amd.xgmi_counters as xc
    x0, t_x0 = (xc.read(), time.perf_counter()) if xc else (None, 0.0)
    wins = [window()]
    x_ms = [L.d_event_elapsed_ms(a, b) for a, b in ev]
    first, _ = D.aggregate(wins[0], [0.0], world, device=dev)
    reps = int(min(args.max_reps, max(1, -(-args.extra_min_time // max(first, 1e-6)))))
    for _ in range(reps - 1):
        wins.append(window())
        x_ms += [L.d_event_elapsed_ms(a, b) for a, b in ev]
    hw = xc.rate(x0, xc.read(), time.perf_counter() - t_x0, world) if xc else None
    el_max = float(np.median(D.aggregate_max_vec(wins, world, device=dev)))
    x_s = float(np.mean(x_ms)) * 1e-3                                   # HIP-event time of one exchange gather on this rank
    _, (rows_req, x_sum) = D.aggregate(0.0, [info[0]["rows_requested"], x_s], world, device=dev)
    out = {"what": "owner-computes exchange: request lists and rows over one RCCL all-to-all each per batch, owners gather from their own HBM; "
                   "buffers allocated once, one host synchronisation per batch (the split sizes), windows of K steps",
           "ms_per_step": round(el_max / K_steps * 1e3, 4), "value": round(job_edges / el_max, 1), "unit": "edges/s",
           "feature_GBps": round(job_nodes * 4 * F / el_max / 1e9, 2), "windows": len(wins),
           "exchange_gather_ms_per_step": round(x_sum / world * 1e3, 4),
           "host_syncs_per_batch": round(xg.host_syncs_per_batch, 3), "gloo_staging_syncs_per_batch": round(xg.staging_syncs / max(1, xg.batches), 3),
           "allocations_in_timed_windows": xg.allocations - allocs_before,
           "rows_requested_last_batch_per_gpu": round(rows_req / world, 1), "xgmi_hw_counters": hw}
    rate = rows_req / world * 4 * F / max(x_sum / world, 1e-9) / 1e9     # rows received per GPU / HIP-event time of the exchange gather
    if c.shared_device:
        # one-GPU rehearsal (LEGION_BENCH_FORCE_DEVICE): the all-to-alls run over gloo, i.e. every list and every row is staged through
        # host memory -- this checks the orchestration, it is NOT a measurement: no `value`, no rate
        out["staged_through_host"] = True
        out["rehearsal_ms_per_step_not_a_result"] = out.pop("ms_per_step")
        for k in ("value", "unit", "feature_GBps", "exchange_gather_ms_per_step"):
            out.pop(k, None)
    else:
        out["staged_through_host"] = False
        out.update({"xgmi_recv_GBps_per_gpu": round(rate, 1), "xgmi_frac_of_peak": round(rate / XGMI_PEAK_GBPS, 4)})
    xg.close()
    for a, b in ev:
        L.d_event_destroy(a)
        L.d_event_destroy(b)
    return out


def roofline_of(c, leg):
    """Dominant kernel (k_gather: it moves ~94 % of the batch's algorithmic bytes), HIP events on its stream."""
    args, F, H = c.args, c.spec.F, c.H
    if leg["per_level"]:
        return None
    g_ms = leg["g_ms"]
    reps = len(g_ms) // args.steps
    if leg["intra"]:   # the timed launches are the last level's gather: its rows are the new nodes of hop H
        launch_bytes = leg["u_h"][H - 1] * (8 * F + 8)
    else:
        launch_bytes = leg["gather_bytes"]
    ach = float(launch_bytes.sum()) * reps / (g_ms.sum() * 1e-3) / 1e9
    traffic, traffic_src = None, None
    import glob
    pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))   # newest round last
    pmc = pmcs[-1] if pmcs else ""
    default_shape = (args.workload == "papers100M" and args.scale == 1.0 and args.batch == 8000 and c.fan == [25, 10, 5] and args.task == "node"
                     and not leg["unified"] and not leg["intra"] and args.table == "device" and args.skew == 205)
    if default_shape and os.path.exists(pmc):
        # HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, collected by
        # profiles/make_pmc_traffic.py (counters cannot be read from inside the process); gfx950 FETCH_SIZE x2 correction applied
        with open(pmc) as f:
            traffic = json.load(f)["k_gather"]["traffic_bytes_per_launch"]
        traffic_src = "profiles/" + os.path.basename(pmc)
    # the sampler group (k_seed + 3 kernels per hop): everything of the serial batch that is not the gather
    sampler = None
    if not leg["intra"] and not leg["overlap"]:
        samp_us = leg["elapsed"] / args.steps * 1e6 - float(g_ms.mean()) * 1e3
        samp_alg = float(leg["samp_bytes"].mean())
        raw = None
        if traffic_src:
            with open(pmc) as f:
                raw = json.load(f).get("sampler", {}).get("raw_over_algorithmic")
        sampler = {"us_per_batch": round(samp_us, 1), "algorithmic_bytes_per_batch": int(samp_alg),
                   "frac_of_peak_algorithmic": round(samp_alg / (samp_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                   "counter_bytes_over_algorithmic": raw,
                   "frac_of_peak_counter_bytes": round(raw * samp_alg / (samp_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if raw else None,
                   "bound": "random-access rate of the memory system (~50 G loads/s, ~26 G atomics/s), not bandwidth: profiles/r01_probe_rate.md"}
    return dict(bound="hbm", kernel="k_gather<float4, non-temporal>", achieved=round(ach, 1),
                peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4), traffic=traffic,
                traffic_measured_in_run=False if traffic is not None else None,
                traffic_source=(traffic_src + " (separate rocprofv3 --pmc passes of this command; not measured by the run that printed this line)") if traffic_src else None,
                avg_launch_us=round(float(g_ms.mean()) * 1e3, 2), timed_launches=int(len(g_ms)),
                algorithmic_bytes_per_launch=int(launch_bytes.mean()),
                launch="last level (hop %d rows) of the per-level gathers" % H if leg["intra"] else "all rows of the batch",
                pipeline_frac=round(leg["job_bytes"] / leg["elapsed"] / 1e9 / (HBM_PEAK_GBPS * c.world), 4), sampler=sampler)


def headline_line(c, leg):
    args, spec, H, F, B, fan, world = c.args, c.spec, c.H, c.spec.F, c.B, c.fan, c.world
    K_steps = args.steps
    el = leg["elapsed"]
    unified = leg["unified"]
    task = "" if args.task == "node" else ", link-prediction [src|pos|neg] seed batches"
    return {
        # BASELINE.json's metric on its own workload (value = sampled edges/s, the feature GB/s is "feature_GBps")
        "metric": "sampled edges/s + feature GB/s, 3-hop GraphSAGE ogbn-papers100M at 1/2/4/8 GPU"
                  if (H == 3 and args.workload == "papers100M" and args.scale == 1.0 and args.task == "node")
                  else f"sampled edges/s + feature GB/s, {H}-hop GraphSAGE mini-batch pipeline ({spec.name} shape{task})",
        "value": round(leg["job_edges"] / el, 1),
        "unit": "edges/s",
        "n_gpus": world,
        "steps": K_steps,
        "warmup": args.warmup,
        "ms_per_step": round(el / K_steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32 ids / f32 rows (verbatim copy)",
        "data": "synthetic",
        "config": {"workload": f"{spec.name}-shape synthetic graph{task}, {H}-hop fan-out {fan}, batch {B}, CSR "
                               + ("+ features resident in HBM" if args.table == "device" else "in HBM, features in pinned host memory (PCIe zero-copy)")
                               + (" (Kg=1 replicas)" if not unified else f", unified feature cache over the {world}-GPU clique")
                               + ("" if args.skew == 205 else f", neighbour skew {args.skew}/256"),
                   "V": spec.V, "E": c.E, "F": F, "row_pitch_floats": c.pitch, "batch": B, "fanout": fan, "gather": args.gather,
                   "pipeline": args.pipeline if args.gather == "all" else "serial", "seeds_per_rank": c.n_mine, "task": args.task,
                   "parallelism": f"dp{world} (seed shards tid % {world}, no data-path collective)"},
        "timing": {"windows": len(leg["windows"]), "steps_per_window": K_steps,
                   "window_ms_min_median_max": [min(leg["windows"]), round(float(np.median(leg["windows"])), 4), max(leg["windows"])],
                   "window_ms_first_32": leg["windows"][:32],
                   "reported": "median window, max over ranks per window"},
        "feature_GBps": round(leg["job_nodes"] * 4 * F / el / 1e9, 2),
        "batches_per_s": round(K_steps * world / el, 2),
        "edges_per_batch": round(leg["job_edges"] / (K_steps * world), 1),
        "unique_nodes_per_batch": round(leg["job_nodes"] / (K_steps * world), 1),
        "sampler_algorithmic_bytes_per_batch": int(leg["samp_bytes"].mean()),   # 20 N_h + 28 E_h + 8 U_h summed over the hops
        "gather_algorithmic_bytes_per_batch": int(leg["gather_bytes"].mean()),
        "graph_gen_s": round(c.gen_s, 2),
        # the two-stream schedule the `legion` server runs (gather of batch i on stream 1 while batch i+1 is sampled), same
        # batches, median of its own windows: the best number the pipeline produces; `value` stays the serial schedule, whose
        # kernels run alone and give the clean per-kernel roofline
        "value_served": None, "ms_per_step_served": None,      # filled from extra_legs.served (N = 1): the `legion` server binary + a consumer process
        "value_overlap": (leg["alt"] or {}).get("value") if (leg["alt"] or {}).get("pipeline") == "overlap" else None,
        "ms_per_step_overlap": (leg["alt"] or {}).get("ms_per_step") if (leg["alt"] or {}).get("pipeline") == "overlap" else None,
        "alt_schedule": leg["alt"],
        "alt_schedule_levels": leg.get("alt_levels"),
        "graph_replay": leg["graph"],
        "legs_failed": [],          # legs after the headline that raised or hung: [] = every leg in this line is valid
        "legs_skipped": [],         # legs the time budget did not admit (--time-budget)
        "extra_legs": {},
        "cache": {"mode": "unified" if unified else "replicated", **(leg["cache_info"] or {}), **(leg["xgmi"] or {})},
        "roofline": roofline_of(c, leg),
        "cpu_baseline": None,
    }


def unified_summary(c, leg):
    """The unified-cache leg as one object of the headline line (N > 1)."""
    args, F = c.args, c.spec.F
    el = leg["elapsed"]
    g = leg["g_ms"]
    reps = max(1, len(g) // args.steps)
    ach = float(leg["gather_bytes"].sum()) * reps / (g.sum() * 1e-3) / 1e9 if len(g) else None
    return {"what": f"same seeds and batches, unified feature cache over the {c.world}-GPU clique (Kg={c.world}: rank-t hot row on GPU t % {c.world}), "
                    "peer shards read in-kernel over xGMI, misses from the local HBM replica",
            "value": round(leg["job_edges"] / el, 1), "unit": "edges/s", "ms_per_step": round(el / args.steps * 1e3, 4),
            "feature_GBps": round(leg["job_nodes"] * 4 * F / el / 1e9, 2),
            "gather_avg_launch_us": round(float(g.mean()) * 1e3, 2) if len(g) else None,
            "gather_frac_of_hbm_peak": round(ach / HBM_PEAK_GBPS, 4) if ach else None,
            "windows": len(leg["windows"]), **(leg["cache_info"] or {}), **(leg["xgmi"] or {}), "xgmi_hw_counters": leg["xgmi_hw"],
            "exchange_variant": leg["exchange"]}


def build_unified_cache(args, K, D, L, eng, me, world, V, F, B, fan, dev):
    """Server::PreSc (Server.cu:83-114) for one process per GPU: pre-sampling epoch on the own seed shard, the
    clique-wide hotness sum as an RCCL all-reduce (the reference reads its peers' arrays, GPUCache.cu:624-627),
    ranking + fill-up of the OWN shard (rank-t row on GPU t % N), then a HIP-IPC exchange of the shards."""
    H = len(fan)
    pool = eng.pools[me]
    t_presc = time.time()
    for it in range(args.presc_steps):
        L.GPUMemoryPool_SetCurrentPipe(pool, 0)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        L.batch_generator_kernel(None, eng.noder, eng.cache, pool, B, it, me, me, K.TRAINMODE)
        for h in range(H):
            L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 1)
        L.make_update_plan(None, eng.graph, eng.cache, pool, me, K.TRAINMODE)
    L.d_stream_sync(None)
    K.check()
    t_presc, t_build = time.time() - t_presc, time.time()
    D.allreduce_device_u64(K, L.GPUCache_GetNodeAccessedMap(eng.cache, me), V, world, device=dev)
    D.allreduce_device_u64(K, L.GPUCache_GetEdgeAccessedMap(eng.cache, me), V, world, device=dev)
    rows = int(V * args.cache_frac) // world + 1
    mode = {1: 0, 2: 1, 4: 2, 8: 3}[world]
    topo_rows = (int(V * args.topo_frac) // world + 1) if args.topo_frac > 0 else 0
    eng.build_cache(cache_agg_mode=mode, node_capacity=rows, edge_capacity=topo_rows, train_step=args.presc_steps)
    L.d_stream_sync(None)
    info = {"Kg": world, "rows_per_gpu": rows, "cached_fraction_of_V": round(rows * world / V, 4), "presc_steps": args.presc_steps,
            # one-off costs (S7 / S8 / S9): the pre-sampling epoch, then hotness all-reduce + ranking (two radix sorts of V keys) + maps + fill-up
            "presc_s": round(t_presc, 3), "cache_build_s": round(time.time() - t_build, 3),
            "topology": "replicated (4-byte peer probes are latency bound; SURVEY 5)" if topo_rows == 0 else
                        f"hottest {topo_rows} adjacency rows per GPU in partitioned CSR fragments (owner/row lookup fused into the sampler), rest from the replica",
            "topo_rows_per_gpu": topo_rows}
    return info


def import_peer_shards(D, eng, me, world):
    """The clique's shards / CSR fragments over HIP IPC: every rank exports its own, one importer at a time.  Needed by the in-kernel
    peer gather (and by a partitioned sampler) only -- the owner-computes exchange never touches a peer's memory."""
    everyone = D.allgather_object(eng.export_shards(me), world)
    t_imp = time.time()
    for turn in range(world):          # one importer at a time
        if turn == me:
            for g in range(world):
                if g != me:
                    eng.import_shards(g, everyone[g])
        D.barrier(world)
    return round(time.time() - t_imp, 2)


def unified_cache_traffic(c, eng, me, cache_info, gather_ms):
    """Where the rows of this rank's last batch came from (own shard / peer shards / backing table), summed over the
    ranks, and the peer-read rate per GPU that follows from the gather's HIP-event time."""
    import torch
    K, L, D, V, F, world, dev = c.K, c.L, c.D, c.spec.V, c.spec.F, c.world, c.dev
    o = eng.out[me][0]
    nc = o["nc"].to_numpy(np.int32, 16)
    n = int(nc[0])
    ids = torch.from_numpy(o["ids"].to_numpy(np.int32, n).astype(np.int64)).to(dev)
    fmap = torch.empty(V, dtype=torch.int32, device=dev)
    L.d_copy_async(fmap.data_ptr(), L.GPUCache_GetFeatureMap(eng.cache, me), V * 4, None)
    L.d_stream_sync(None)
    slot = fmap[ids]
    owner = torch.div(slot, cache_info["rows_per_gpu"], rounding_mode="floor")
    local = int(((slot >= 0) & (owner == me)).sum().item())
    peer = int(((slot >= 0) & (owner != me)).sum().item())
    miss = int((slot < 0).sum().item())
    del fmap, slot, owner, ids
    rate = peer * 4 * F / (gather_ms * 1e-3) / 1e9                      # this rank's peer-shard read rate
    _, (s_local, s_peer, s_miss, s_rate) = D.aggregate(0.0, [local, peer, miss, rate], world, device=dev)
    rmin = D.aggregate_max_vec([-rate], world, device=dev)[0] * -1.0
    out = {"rows_last_batch": {"own_shard": int(s_local), "peer_shards": int(s_peer), "backing_table": int(s_miss), "summed_over_ranks": world}}
    if c.shared_device or world == 1:
        # rehearsal: every rank sits on ONE GPU, a "peer" shard is the same device's HBM -- this is not an xGMI number
        out.update({"xgmi_read_GBps_per_gpu": None, "xgmi_frac_of_peak": None,
                    "peer_shard_read_GBps_same_device": round(s_rate / world, 1) if world > 1 else 0.0})
    else:
        out.update({"xgmi_read_GBps_per_gpu": round(s_rate / world, 1), "xgmi_read_GBps_min_rank": round(rmin, 1),
                    "xgmi_peak_GBps_per_gpu": XGMI_PEAK_GBPS, "xgmi_frac_of_peak": round(s_rate / world / XGMI_PEAK_GBPS, 4),
                    "xgmi_note": "peer rows x 4F bytes / the gather's HIP-event time (the gather also reads own-shard and replica rows in that time)"})
    return out


def measure_traffic_in_run(args):
    """HBM bytes per launch of the dominant kernel (k_gather), from the PMC counters, as MI355X_MICROARCH.md prescribes: one
    `rocprofv3 --pmc` pass per counter (FETCH_SIZE, WRITE_SIZE; no trace domains), the program itself after `--`; both counters
    come in KiB, and on gfx950 FETCH_SIZE counts half of the bytes of 16 B/lane reads (x2).  Counters cannot be read from inside
    a process, so each pass is a short child run of this same workload (10 steps); the parent's graph stays resident meanwhile.
    Returns the roofline fields to overwrite, or None (not on PATH, already under a profiler, a pass failed: the committed file stays)."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3")
    under_profiler = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("LEGION_BENCH_PMC_CHILD")
    if not rocprof or under_profiler:
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "2", "--min-time", "0", "--headline-only",
             "--cpu-baseline-seconds", "0", "--measure-traffic", "off", "--workload", args.workload, "--fanout", args.fanout,
             "--batch", str(args.batch), "--scale", str(args.scale), "--task", args.task, "--skew", str(args.skew), "--row-pitch", args.row_pitch]
    env = dict(os.environ, TMPDIR="/tmp", LEGION_BENCH_PMC_CHILD="1")
    kib, launches, t0 = {}, 0, time.time()
    tmp = tempfile.mkdtemp(prefix="legion_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            # own session: if a pass does not come back, the profiler AND the run under it are stopped (exactly the group started here)
            pr = subprocess.Popen([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=90)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None
            files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
            if rc != 0 or not files:
                return None
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if "legion::k_gather" in row["Kernel_Name"] and row["Counter_Name"] == counter]
            if not vals:
                return None
            kib[counter], launches = sum(vals) / len(vals), len(vals)
    except Exception:   # noqa: BLE001 -- evidence only: never lose the line over it
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    read, write = kib["FETCH_SIZE"] * 1024 * 2, kib["WRITE_SIZE"] * 1024
    return {"traffic": int(read + write), "traffic_measured_in_run": True,
            "traffic_source": "two child runs of this workload started by this run, after its timed legs: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                              "(one pass per counter, no trace domains; KiB; FETCH_SIZE x2 for the 16 B/lane reads on gfx950); average of %d k_gather "
                              "launches (batches 0-11 of the same seed list; the timed windows run batches %d-%d)" % (launches, args.warmup, args.warmup + args.steps - 1),
            "traffic_hbm_read_bytes": int(read), "traffic_hbm_write_bytes": int(write), "traffic_passes_s": round(time.time() - t0, 1)}


def measure_copy(c):
    """Streaming float4 copy sized like the gather (non-temporal, one 16-byte chunk per lane and iteration, 1 GiB in,
    1 GiB out per launch + a 4 GiB pair): what a pure HBM stream reaches on this box.  Printed beside the vendor peak;
    the gather of skewed ids may exceed it because re-touched hot rows are served by the Infinity Cache."""
    import torch
    L, dev = c.L, c.dev
    best = 0.0
    for nbytes in (1 << 30, 4 << 30):
        a_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        e0, e1 = L.d_event_create(), L.d_event_create()
        L.legion_copy_f4(None, b_buf.data_ptr(), a_buf.data_ptr(), nbytes)
        L.d_event_record(e0, None)
        for _ in range(5):
            L.legion_copy_f4(None, b_buf.data_ptr(), a_buf.data_ptr(), nbytes)
        L.d_event_record(e1, None)
        L.d_stream_sync(None)
        best = max(best, 5 * 2 * nbytes / (L.d_event_elapsed_ms(e0, e1) * 1e-3) / 1e9)
        del a_buf, b_buf
    return round(best, 1)


def run_cpu_baseline(args, spec, indptr, indices, feats, mine, my_labels, B, fan, steps_avail):
    """The CPU side of the same workload on a bounded sample (the first batches of rank 0's seed list), --cpu-baseline-seconds
    of CPU time in total: the oracle (reference semantics; scalar C, 1 thread = `value`, and its OpenMP run on all cores the
    process may use) and DGL's CPU NeighborSampler (BASELINE.json config 1) if importable, else our own OpenMP implementation of
    the same semantics -- labelled as such, never as DGL."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import oracle as O
    t0 = time.time()
    h_indptr = indptr.cpu().numpy()
    h_indices = indices.cpu().numpy()
    with_feat = not args.no_cpu_features
    h_feats = (feats if isinstance(feats, np.ndarray) else feats.cpu().numpy()) if with_feat else None
    if h_feats is not None and h_feats.shape[1] != spec.F:      # padded HBM rows: the CPU side reads the dense layout
        h_feats = np.ascontiguousarray(h_feats[:, :spec.F])
    h_ids = mine.cpu().numpy()
    h_lab = my_labels.cpu().numpy()
    copy_s = time.time() - t0
    total = float(args.cpu_baseline_seconds)
    budget = {"serial": 0.4 * total, "omp": 0.2 * total, "dgl": 0.4 * total}
    H = len(fan)
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    runner = O.OracleRunner(h_indptr, h_indices, h_feats, spec.V, spec.F, B, fan, with_features=with_feat)

    def timed(fn, seconds):
        edges, n, used = 0, 0, 0.0
        while used < seconds and n < min(steps_avail, 64):
            t1 = time.perf_counter()
            edges += fn(n)
            used += time.perf_counter() - t1
            n += 1
        return edges, n, used

    edges, n, t_used = timed(lambda i: int(runner.run_batch(h_ids, h_lab, i)["ec"][2 + H]), budget["serial"])
    smp = O.DglSemanticsSampler(h_indptr, h_indices, h_feats, spec.V, spec.F, B, fan)      # also tells the OpenMP thread count
    omp = None
    try:
        e1, n1, t1 = timed(lambda i: int(runner.run_batch(h_ids, h_lab, i, omp=True)["ec"][2 + H]), budget["omp"])
        omp = {"value": round(e1 / t1, 1), "unit": "edges/s", "cores": smp.threads, "kind": "port",
               "sample": f"{n1} batches, oracle/legion_oracle.c lo_run_batch_omp: draws, COO offsets and row copies parallel, "
                         "the order-defining compaction serial (byte-identical to the 1-thread run)", "seconds": round(t1, 2)}
    except Exception as ex:   # noqa: BLE001
        omp = {"error": repr(ex)[:200]}
    # DGL's CPU NeighborSampler if importable, else our own OpenMP implementation of the same semantics
    dgl_like = None
    try:
        try:
            import dgl  # noqa: F401
            dgl_like = run_dgl_baseline(dgl, torch, h_indptr, h_indices, h_feats, h_ids, B, fan, budget["dgl"])
        except ImportError:
            e2, n2, t2 = timed(lambda i: int(smp.run_batch(h_ids[i * B:(i + 1) * B], rng_seed=i + 1, gather=with_feat)[1]), budget["dgl"])
            dgl_like = {"value": round(e2 / t2, 1), "unit": "edges/s", "cores": smp.threads,
                        "kind": "DGL-semantics CPU sampler (own OpenMP implementation; dgl not installed)",
                        "sample": f"{n2} batches, uniform w/o replacement + to_block per layer + index_select"
                                  + ("" if with_feat else " (no feature gather)"), "seconds": round(t2, 2)}
    except Exception as ex:  # the headline CPU number must not depend on this leg
        dgl_like = {"error": repr(ex)}
    return {"value": round(edges / t_used, 1), "unit": "edges/s", "cores": 1, "kind": "port",
            "openmp": omp, "dgl_semantics": dgl_like,
            "sample": f"{n} batches (batch {B}, fan-out {fan}) of the same workload, oracle/legion_oracle.c single thread"
                      + ("" if with_feat else ", sampler+COO only (no feature gather)"),
            "seconds": round(t_used, 2), "host_copy_s": round(copy_s, 1), "cpu_model": cpu_model,
            "host_cores_available": os.cpu_count(), "host_cores_in_affinity_mask": affinity,
            "threads_note": "the OpenMP legs run with omp_get_max_threads() threads.  bench.py imports torch first, and torch sizes the shared "
                            "OpenMP pool to the PHYSICAL core count (SMT siblings unused): e.g. 128 threads on a 2 x 64-core host with 256 "
                            "hardware threads.  `cores` is what was really used."}


def run_dgl_baseline(dgl, torch, indptr, indices, feats, ids, B, fan, budget):
    """DGL's own CPU NeighborSampler + feature index_select on the same graph (only if dgl is installed)."""
    g = dgl.graph(("csc", (torch.from_numpy(indptr), torch.from_numpy(indices), torch.tensor([], dtype=torch.int64))))
    sampler = dgl.dataloading.NeighborSampler(list(reversed(fan)))   # DGL lists fan-outs input layer first
    ft = torch.from_numpy(feats) if feats is not None else None
    e, n, t = 0, 0, 0.0
    while t < budget and (n + 1) * B <= len(ids):
        t1 = time.perf_counter()
        _, _, blocks = sampler.sample(g, torch.from_numpy(ids[n * B:(n + 1) * B].astype(np.int64)))
        if ft is not None:
            _ = ft[blocks[0].srcdata[dgl.NID]]
        t += time.perf_counter() - t1
        e += sum(b.num_edges() for b in blocks)
        n += 1
    return {"value": round(e / t, 1), "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "dgl.dataloading.NeighborSampler (CPU)",
            "sample": f"{n} batches", "seconds": round(t, 2)}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if argv and argv[0] == "--served-consumer":      # internal: the consumer child of the `served` leg
        return served_consumer(argv[1:])
    args = parse(argv)
    plan = launch_plan(args, os.environ)
    if plan.startswith("error"):
        raise SystemExit("bench.py: " + plan[7:])
    if plan == "spawn":
        raise SystemExit(launch_children(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
