#!/usr/bin/env python3
"""bench.py -- throughput of the mini-batch hot path (sampler + compaction + feature gather) on a
synthetic graph of the ogbn-papers100M shape (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one mini-batch (B seeds, H-hop sampling, COO construction, gather of every unique
node's feature row) with CSR and features already resident in HBM.  Each rank owns the seeds
`tid % N == rank` (GPUGraphStore.cu:332-346) and a full replica of the graph (Kg = 1), so there is
no data-path collective: scaling is weak.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E vendor peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="papers100M", choices=["products", "papers100M", "uk-union"])
    ap.add_argument("--scale", type=float, default=1.0, help="shrink V and seed sets (debug only)")
    ap.add_argument("--batch", type=int, default=8000)
    ap.add_argument("--fanout", default="25,10,5")
    ap.add_argument("--gather", default="all", choices=["all", "level"], help="one gather per batch or one per level")
    ap.add_argument("--pipeline", default="serial", choices=["overlap", "serial", "intra"],
                    help="serial: one stream.  intra: the reference's two-stream schedule inside a batch (Server.cu:301-328): "
                         "the rows of level l are gathered on a second stream while hop l+1 is sampled; the last level runs alone.  "
                         "overlap: whole-batch gather of batch i on a second stream while batch i+1 is sampled (depth-2 pipes)")
    ap.add_argument("--cache", default="replicated", choices=["replicated", "unified"],
                    help="replicated: every GPU holds all features (Kg=1).  unified: the clique-wide hotness-partitioned "
                         "feature cache of the reference (rank-t row on GPU t %% N), peer shards read in-kernel over xGMI")
    ap.add_argument("--table", default="device", choices=["device", "host"],
                    help="where the V x F feature table lives: HBM (default) or pinned host memory read over PCIe -- the "
                         "reference's UVA configuration (GPUGraphStore.cu:315); combine with --cache unified for an HBM cache")
    ap.add_argument("--cache-frac", type=float, default=0.25, help="unified: fraction of the V feature rows cached per clique")
    ap.add_argument("--topo-frac", type=float, default=0.0, help="unified: fraction of the V adjacency rows cached as partitioned CSR "
                    "fragments per clique (0: topology stays replicated)")
    ap.add_argument("--presc-steps", type=int, default=8, help="unified: batches of the pre-sampling (hotness) epoch")
    ap.add_argument("--headline-only", action="store_true", help="skip the alt_schedule and graph_replay legs (clean kernel profiles)")
    ap.add_argument("--stream-priority", default="none", choices=["none", "sampler", "gather"],
                    help="overlap schedule: which of the two streams gets the high stream priority (the other the low one)")
    ap.add_argument("--cu-split", type=int, default=0, help="experiment (overlap schedule): of every 8 compute units, this many run "
                    "the sampler stream and the rest the gather stream (hipExtStreamCreateWithCUMask); 0 = unrestricted streams")
    ap.add_argument("--cu-pattern", default="mod", choices=["mod", "block"], help="--cu-split: bit i belongs to the sampler if "
                    "i %% 8 < S (mod) or (i // 32) %% 8 < S (block)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--no-cpu-features", action="store_true", help="CPU baseline: sampler only (skip the 57 GB host copy)")
    return ap.parse_args()


def build_graph_on_gpu(K, spec, dev):
    """Synthetic dataset generated on the GPU by csrc/synth.hip (spec: legion-1_amd/synth.py)."""
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
    L.legion_synth_neighbors(None, indices.data_ptr(), 0, E, V, spec.M, spec.C)
    feats = torch.empty((V, F), dtype=torch.float32, device=dev)
    L.legion_synth_features(None, feats.data_ptr(), 0, V, F)
    torch.cuda.synchronize()
    K.check()
    return indptr, indices, feats, E


def main():
    args = parse()
    if os.environ.get("LEGION_BENCH_WATCHDOG"):  # debugging aid: dump all Python stacks and exit if stuck
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LEGION_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    # Rehearsal on a one-GPU box: LEGION_BENCH_FORCE_DEVICE=0 puts every rank on that device and uses gloo
    # (RCCL refuses two ranks on one GPU).  Never set on a real multi-GPU run.
    forced = os.environ.get("LEGION_BENCH_FORCE_DEVICE")
    if forced is not None:
        local_rank = int(forced)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced is not None:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import legion1_amd.capi as K
    import legion1_amd.synth as S
    L = K.lib()
    unified = args.cache == "unified"
    G = world if unified else 1          # logical GPUs of the clique this process knows about
    me = rank if unified else 0          # the one this process drives
    for g in range(G):
        L.legion_set_device_map(g, local_rank)
    L.SetGPUDevice(me)

    fan = [int(x) for x in args.fanout.split(",")]
    H = len(fan)
    B = args.batch
    spec = S.spec_for(args.workload, scale=args.scale)
    V, F = spec.V, spec.F
    t0 = time.time()
    indptr, indices, feats, E = build_graph_on_gpu(K, spec, dev)
    feat_ptr, feat_loc, host_table = feats.data_ptr(), K.LOC_DEVICE, None
    if args.table == "host":   # move the table to pinned, device-mapped host memory; misses then cross PCIe
        nbytes = V * F * 4
        host_table = L.host_alloc_space64(nbytes)
        L.d_copy_d_2_h(host_table, feats.data_ptr(), nbytes)
        K.check()
        del feats
        feats = None
        torch.cuda.empty_cache()
        feat_ptr, feat_loc = host_table, K.LOC_HOST_PINNED
    # seeds of this rank: train ids with tid % world == rank, labels from the generator
    all_train = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, all_train.data_ptr(), 0, spec.n_train, V, spec.M2, spec.C2, 1, 0)
    torch.cuda.synchronize()
    import legion1_amd.dist as D
    mine = D.shard_seeds(all_train, rank, world).contiguous()
    del all_train
    labels_all = torch.empty(V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, labels_all.data_ptr(), 0, V, spec.classes)
    torch.cuda.synchronize()
    my_labels = labels_all[mine.long()].contiguous()
    del labels_all
    n_mine = int(mine.numel())
    gen_s = time.time() - t0

    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32))
    seeds = dict(train=[((mine.data_ptr(), n_mine), (my_labels.data_ptr(), n_mine)) if g == me else empty for g in range(G)])
    overlap = args.pipeline == "overlap" and args.gather == "all"
    intra = args.pipeline == "intra"
    depth = 2      # the reference's PIPELINE_DEPTH; the serial schedule only uses pipe 0
    eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feat_ptr, V, F, seeds, B, fan, G=G,
                   csr_location=K.LOC_DEVICE, features_location=feat_loc, E=E, pipeline_depth=depth,
                   local_devs=[me], train_step=max(1, args.presc_steps))
    eng.alloc_features()
    cache_info = None
    if unified:
        cache_info = build_unified_cache(args, K, D, L, eng, me, world, V, F, B, fan, dev)
    else:
        L.GPUCache_SetPreSc(eng.cache, 0)  # steady state: no pre-sampling epoch in the all-resident configuration
    if args.cu_split > 0:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (n_cu + 31) // 32
        owner = [(i % 8 if args.cu_pattern == "mod" else (i // 32) % 8) < args.cu_split for i in range(words * 32)]
        m_s = np.array([sum(1 << b for b in range(32) if owner[w * 32 + b] and w * 32 + b < n_cu) for w in range(words)], dtype=np.uint32)
        m_g = np.array([sum(1 << b for b in range(32) if not owner[w * 32 + b] and w * 32 + b < n_cu) for w in range(words)], dtype=np.uint32)
        stream = L.d_stream_create_cu_mask(m_s.ctypes.data, words)
        gstream2 = L.d_stream_create_cu_mask(m_g.ctypes.data, words)
    elif args.stream_priority != "none":
        stream = L.d_stream_create_priority(1 if args.stream_priority == "sampler" else 0)
        gstream2 = L.d_stream_create_priority(1 if args.stream_priority == "gather" else 0)
    else:
        stream = L.d_stream_create()       # sampler stream
        gstream2 = L.d_stream_create()     # gather stream of the overlapped schedule (reference: streams_[1], Server.cu:178-181)
    steps_avail = max(1, (n_mine - 1) // B)  # train_step = (n-1)/B, CUDA_IPC_Service.cu:89
    K_steps, W = args.steps, args.warmup

    per_level = args.gather == "level" and not intra
    ev_hop = [L.d_event_create() for _ in range(H + 1)]   # intra: hop h of the running batch is complete
    ev_gdone = L.d_event_create()                          # intra: every gather of the running batch is complete
    intra_started = [False]
    log = K.DevBuf((K_steps + W) * 128)  # nc/ec of every step, copied on-stream
    ev = [(L.d_event_create(), L.d_event_create()) for _ in range(K_steps)]
    pool = eng.pools[me]
    ev_sampled = [L.d_event_create() for _ in range(depth)]   # sampling of the batch in pipe q is complete
    ev_gathered = [L.d_event_create() for _ in range(depth)]  # gather of the batch in pipe q is complete
    used = [False] * depth

    def step(i, timed_idx=None, overlap=overlap):
        """One mini-batch.  overlap: depth-2 pipes (the reference's PIPELINE_DEPTH), the sampler of batch
        i+1 runs on `stream` while the gather of batch i runs on `gstream`; a pipe's buffers are reused
        only after its gather finished."""
        it = i % steps_avail
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
        L.d_copy_async(log.ptr + i * 128, o["nc"].ptr, 64, stream)
        L.d_copy_async(log.ptr + i * 128 + 64, o["ec"].ptr, 64, stream)
        L.make_update_plan(stream, eng.graph, eng.cache, pool, me, K.TRAINMODE)
        L.update_cache(stream, eng.cache, eng.noder, pool, me, K.TRAINMODE)

    def drain():
        L.d_stream_sync(stream)
        L.d_stream_sync(gstream2)
        torch.cuda.synchronize()

    for i in range(W):
        step(i)
    drain()
    if world > 1:
        torch.distributed.barrier()
    t_start = time.perf_counter()
    for i in range(K_steps):
        step(W + i, timed_idx=i)
    drain()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        torch.distributed.barrier()
    K.check()

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

    import legion1_amd.dist as D
    elapsed_max, (job_edges, job_nodes, job_bytes) = D.aggregate(
        elapsed, [tot_edges, tot_nodes, float(samp_bytes.sum() + gather_bytes.sum())], world, device=dev)

    # the other schedule on the very same K batches (reported beside the headline, never instead of it)
    alt = None
    if not per_level and not intra and not args.headline_only:
        if world > 1:
            torch.distributed.barrier()
        t_alt = time.perf_counter()
        for i in range(K_steps):
            step(W + i, overlap=not overlap)
        drain()
        alt_elapsed = time.perf_counter() - t_alt
        if world > 1:
            torch.distributed.barrier()
        alt_max, _ = D.aggregate(alt_elapsed, [0.0], world, device=dev)
        alt = {"pipeline": "serial" if overlap else "overlap", "ms_per_step": round(alt_max / K_steps * 1e3, 4),
               "value": round(job_edges / alt_max, 1), "unit": "edges/s",
               "pipeline_frac": round(job_bytes / alt_max / 1e9 / (HBM_PEAK_GBPS * world), 4)}

    # the serial schedule again, recorded once as a hipGraph and replayed with one launch per batch (same K batches)
    graph_leg = None
    if not per_level and not intra and not args.headline_only and world == 1:   # informational leg: N = 1 only, never fatal
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
            graph_leg = {"pipeline": "serial, one hipGraph launch per batch", "ms_per_step": round(g_max / K_steps * 1e3, 4),
                         "value": round(job_edges / g_max, 1), "unit": "edges/s"}
        except Exception as ex:   # noqa: BLE001 -- the headline line must still be printed
            graph_leg = {"error": repr(ex)[:200]}
        finally:
            L.legion_set_error_mode(K.ERR_EXIT)

    # dominant kernel (k_gather: it moves ~94 % of the batch's algorithmic bytes), HIP events on its stream
    roofline = None
    if not per_level:
        g_ms = np.array([L.d_event_elapsed_ms(a, b) for a, b in ev], dtype=np.float64)
        if intra:   # the timed launches are the last level's gather: its rows are the new nodes of hop H
            gather_launch_bytes = u_h[H - 1] * (8 * F + 8)
        else:
            gather_launch_bytes = gather_bytes
        ach = float(gather_launch_bytes.sum()) / (g_ms.sum() * 1e-3) / 1e9
        traffic, traffic_src = None, None
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))   # newest round last
        pmc = pmcs[-1] if pmcs else ""
        if args.workload == "papers100M" and args.scale == 1.0 and args.batch == 8000 and fan == [25, 10, 5] and not unified and not intra and args.table == "device" and os.path.exists(pmc):
            # HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
            # command (counters cannot be read from inside the process); gfx950 FETCH_SIZE x2 correction applied
            with open(pmc) as f:
                traffic = json.load(f)["k_gather"]["traffic_bytes_per_launch"]
            traffic_src = "profiles/" + os.path.basename(pmc)
        roofline = dict(bound="hbm", kernel="k_gather<float4, non-temporal>", achieved=round(ach, 1),
                        peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(ach / HBM_PEAK_GBPS, 4), traffic=traffic,
                        traffic_source=traffic_src, avg_launch_us=round(float(g_ms.mean()) * 1e3, 2),
                        algorithmic_bytes_per_launch=int(gather_launch_bytes.mean()),
                        launch="last level (hop %d rows) of the per-level gathers" % H if intra else "all rows of the batch",
                        pipeline_frac=round(job_bytes / elapsed_max / 1e9 / (HBM_PEAK_GBPS * world), 4))

    xgmi = None
    if unified and roofline is not None:
        xgmi = unified_cache_traffic(K, L, eng, me, world, V, F, cache_info, float(np.mean([L.d_event_elapsed_ms(a, b) for a, b in ev])), dev)
    if world > 1:
        torch.distributed.barrier()   # nobody unmaps a cache shard while a peer may still read it

    # measured streaming-copy rate of this box (float4 copy kernel, read + write bytes), printed beside the vendor peak
    copy_gbps = None
    if rank == 0:
        nbytes = 4 << 30
        a_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        e0, e1 = L.d_event_create(), L.d_event_create()
        L.legion_copy_f4(stream, b_buf.data_ptr(), a_buf.data_ptr(), nbytes)
        L.d_event_record(e0, stream)
        for _ in range(5):
            L.legion_copy_f4(stream, b_buf.data_ptr(), a_buf.data_ptr(), nbytes)
        L.d_event_record(e1, stream)
        copy_gbps = round(5 * 2 * nbytes / (L.d_event_elapsed_ms(e0, e1) * 1e-3) / 1e9, 1)
        del a_buf, b_buf
        if roofline is not None:
            roofline["measured_copy_GBps"] = copy_gbps
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbps, 4)

    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_baseline_seconds > 0:   # reported baseline: N = 1 only
        if host_table is not None:   # the table already is host memory: view it, no copy
            import ctypes
            feats = np.ctypeslib.as_array(ctypes.cast(host_table, ctypes.POINTER(ctypes.c_float)), shape=(V, F))
        try:
            cpu_baseline = run_cpu_baseline(args, spec, indptr, indices, feats, mine, my_labels, B, fan, steps_avail)
        except Exception as ex:   # noqa: BLE001 -- reported baseline only: never lose the headline line over it
            cpu_baseline = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(ex)[:200]}

    if rank == 0:
        out = {
            # BASELINE.json's metric on its own workload (value = sampled edges/s, the feature GB/s is "feature_GBps")
            "metric": "sampled edges/s + feature GB/s, 3-hop GraphSAGE ogbn-papers100M at 1/2/4/8 GPU"
                      if (H == 3 and args.workload == "papers100M" and args.scale == 1.0)
                      else f"sampled edges/s + feature GB/s, {H}-hop GraphSAGE mini-batch pipeline ({spec.name} shape)",
            "value": round(job_edges / elapsed_max, 1),
            "unit": "edges/s",
            "n_gpus": world,
            "steps": K_steps,
            "warmup": W,
            "ms_per_step": round(elapsed_max / K_steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 ids / f32 rows (verbatim copy)",
            "data": "synthetic",
            "config": {"workload": f"{spec.name}-shape synthetic graph, {H}-hop fan-out {fan}, batch {B}, CSR " + ("+ features resident in HBM" if args.table == "device" else "in HBM, features in pinned host memory (PCIe zero-copy)") + (" (Kg=1 replicas)" if not unified else f", unified feature cache over the {world}-GPU clique"),
                       "V": V, "E": E, "F": F, "batch": B, "fanout": fan, "gather": args.gather, "pipeline": args.pipeline if args.gather == "all" else "serial", "seeds_per_rank": n_mine,
                       "parallelism": f"dp{world} (seed shards tid % {world}, no data-path collective)"},
            "feature_GBps": round(job_nodes * 4 * F / elapsed_max / 1e9, 2),
            "batches_per_s": round(K_steps * world / elapsed_max, 2),
            "edges_per_batch": round(job_edges / (K_steps * world), 1),
            "unique_nodes_per_batch": round(job_nodes / (K_steps * world), 1),
            "sampler_algorithmic_bytes_per_batch": int(samp_bytes.mean()),   # 20 N_h + 28 E_h + 8 U_h summed over the hops
            "gather_algorithmic_bytes_per_batch": int(gather_bytes.mean()),
            "graph_gen_s": round(gen_s, 2),
            "alt_schedule": alt,
            "graph_replay": graph_leg,
            "cache": {"mode": args.cache, **(cache_info or {}), **(xgmi or {})},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def build_unified_cache(args, K, D, L, eng, me, world, V, F, B, fan, dev):
    """Server::PreSc (Server.cu:83-114) for one process per GPU: pre-sampling epoch on the own seed shard, the
    clique-wide hotness sum as an RCCL all-reduce (the reference reads its peers' arrays, GPUCache.cu:624-627),
    ranking + fill-up of the OWN shard (rank-t row on GPU t % N), then a HIP-IPC exchange of the shards."""
    H = len(fan)
    pool = eng.pools[me]
    for it in range(args.presc_steps):
        L.GPUMemoryPool_SetCurrentPipe(pool, 0)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        L.batch_generator_kernel(None, eng.noder, eng.cache, pool, B, it, me, me, K.TRAINMODE)
        for h in range(H):
            L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 1)
        L.make_update_plan(None, eng.graph, eng.cache, pool, me, K.TRAINMODE)
    L.d_stream_sync(None)
    K.check()
    D.allreduce_device_u64(K, L.GPUCache_GetNodeAccessedMap(eng.cache, me), V, world, device=dev)
    D.allreduce_device_u64(K, L.GPUCache_GetEdgeAccessedMap(eng.cache, me), V, world, device=dev)
    rows = int(V * args.cache_frac) // world + 1
    mode = {1: 0, 2: 1, 4: 2, 8: 3}[world]
    topo_rows = (int(V * args.topo_frac) // world + 1) if args.topo_frac > 0 else 0
    eng.build_cache(cache_agg_mode=mode, node_capacity=rows, edge_capacity=topo_rows, train_step=args.presc_steps)
    everyone = D.allgather_object(eng.export_shards(me), world)
    for turn in range(world):          # one importer at a time
        if turn == me:
            t_imp = time.time()
            for g in range(world):
                if g != me:
                    eng.import_shards(g, everyone[g])
            if os.environ.get("LEGION_BENCH_WATCHDOG"):
                print(f"[rank {me}] imported {world - 1} shard(s) of {rows * F * 4 / 1e9:.1f} GB in {time.time() - t_imp:.2f} s", flush=True)
        D.barrier(world)
    return {"Kg": world, "rows_per_gpu": rows, "cached_fraction_of_V": round(rows * world / V, 4), "presc_steps": args.presc_steps,
            "topology": "replicated (4-byte peer probes are latency bound; SURVEY 5)" if topo_rows == 0 else
                        f"hottest {topo_rows} adjacency rows per GPU in partitioned CSR fragments (owner/row lookup fused into the sampler), rest from the replica",
            "topo_rows_per_gpu": topo_rows}


def unified_cache_traffic(K, L, eng, me, world, V, F, cache_info, gather_ms, dev):
    """Where the rows of the last batch came from: own shard / peer shards (xGMI) / backing table."""
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
    return {"rows_last_batch": {"own_shard": local, "peer_shards": peer, "backing_table": miss},
            "xgmi_read_GBps_per_gpu": round(peer * 4 * F / (gather_ms * 1e-3) / 1e9, 1) if world > 1 else 0.0,
            "xgmi_peak_GBps_per_gpu": 7 * 153}


def run_cpu_baseline(args, spec, indptr, indices, feats, mine, my_labels, B, fan, steps_avail):
    """The CPU oracle (reference semantics, scalar C, 1 thread) timed on the host cores on a bounded
    sample of the SAME workload: the first few batches of rank 0's seed list."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    t0 = time.time()
    h_indptr = indptr.cpu().numpy()
    h_indices = indices.cpu().numpy()
    with_feat = not args.no_cpu_features
    h_feats = (feats if isinstance(feats, np.ndarray) else feats.cpu().numpy()) if with_feat else None
    h_ids = mine.cpu().numpy()
    h_lab = my_labels.cpu().numpy()
    copy_s = time.time() - t0
    runner = O.OracleRunner(h_indptr, h_indices, h_feats, spec.V, spec.F, B, fan, with_features=with_feat)
    edges, n, t_used = 0, 0, 0.0
    H = len(fan)
    while t_used < args.cpu_baseline_seconds and n < min(steps_avail, 64):
        t1 = time.perf_counter()
        res = runner.run_batch(h_ids, h_lab, n)
        t_used += time.perf_counter() - t1
        edges += int(res["ec"][2 + H])
        n += 1
    # second reported baseline: DGL's CPU NeighborSampler (BASELINE.json config 0) if importable, else our
    # own OpenMP implementation of the same semantics -- labelled as such, never as DGL
    dgl_like = None
    try:
        budget = min(10.0, args.cpu_baseline_seconds)
        try:
            import dgl  # noqa: F401
            dgl_like = run_dgl_baseline(dgl, h_indptr, h_indices, h_feats, h_ids, B, fan, budget)
        except ImportError:
            smp = O.DglSemanticsSampler(h_indptr, h_indices, h_feats, spec.V, spec.F, B, fan)
            e2, n2, t2 = 0, 0, 0.0
            while t2 < budget and n2 < min(steps_avail, 64):
                t1 = time.perf_counter()
                _, e = smp.run_batch(h_ids[n2 * B:(n2 + 1) * B], rng_seed=n2 + 1, gather=with_feat)
                t2 += time.perf_counter() - t1
                e2 += e
                n2 += 1
            dgl_like = {"value": round(e2 / t2, 1), "unit": "edges/s", "cores": smp.threads,
                        "kind": "DGL-semantics CPU sampler (own OpenMP implementation; dgl not installed)",
                        "sample": f"{n2} batches, uniform w/o replacement + to_block per layer + index_select"
                                  + ("" if with_feat else " (no feature gather)"), "seconds": round(t2, 2)}
    except Exception as ex:  # the headline CPU number must not depend on this leg
        dgl_like = {"error": repr(ex)}
    return {"value": round(edges / t_used, 1), "unit": "edges/s", "cores": 1, "kind": "port",
            "dgl_semantics": dgl_like,
            "sample": f"{n} batches (batch {B}, fan-out {fan}) of the same workload, oracle/legion_oracle.c single thread"
                      + ("" if with_feat else ", sampler+COO only (no feature gather)"),
            "seconds": round(t_used, 2), "host_copy_s": round(copy_s, 1), "host_cores_available": os.cpu_count()}


def run_dgl_baseline(dgl, indptr, indices, feats, ids, B, fan, budget):
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


if __name__ == "__main__":
    main()
