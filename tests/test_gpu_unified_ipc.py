"""One process per GPU with a clique-wide unified cache (BASELINE config "hotness-partitioned unified
cache over xGMI"): every rank drives one logical GPU of a Kg = 2 clique, fills ITS shard of the
feature cache / CSR fragments (rank-t item on clique GPU t % Kg, GPUCache.cu:88-108), exports it over
HIP IPC and reads the peer's shard in-kernel.  The only collective is the build-time hotness sum
(CandidateSelection, GPUCache.cu:624-627) -- gloo here, RCCL in bench.py.  Both ranks share the one
GPU of the test box; on a real node the same loads cross xGMI."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_SHARD_CHUNK_BYTES="200000")   # 4 chunks per shard
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import legion1_amd.capi as K
        import legion1_amd.dist as D
        import legion1_amd.synth as S
        import oracle as O
        from conftest import assert_batch_equal
        L = K.lib()
        L.legion_set_error_mode(K.ERR_RETURN)
        for g in range(world):
            L.legion_set_device_map(g, 0)          # the test box has one GPU: both clique members live on it
        ds = S.generate(S.spec_for("products", scale=0.01))
        V, F = ds.spec.V, ds.spec.F
        B, fan, cap = 250, [10, 5], 2000
        parts = O.split_seeds(ds.train, world)
        steps = min((len(p) - 1) // B for p in parts)
        seeds = dict(train=[(p, ds.labels[p]) for p in parts])
        eng = K.Engine(ds.indptr, ds.indices, ds.features, V, F, seeds, B, fan, G=world, local_devs=[rank], train_step=steps)
        eng.alloc_features()
        peers = [g for g in range(world) if g != rank]
        assert all(L.legion_is_remote_device(g) == 1 for g in peers) and L.legion_is_remote_device(rank) == 0
        # pre-sampling epoch on the own partition, then the clique-wide hotness sum as a collective
        for it in range(steps):
            eng.run_batch(rank, it, is_presc=True)
        D.allreduce_device_u64(K, L.GPUCache_GetNodeAccessedMap(eng.cache, rank), V, world)
        D.allreduce_device_u64(K, L.GPUCache_GetEdgeAccessedMap(eng.cache, rank), V, world)
        eng.build_cache(cache_agg_mode={2: 1, 4: 2}[world], node_capacity=cap, edge_capacity=cap, train_step=steps)
        # oracle: hotness of BOTH partitions -> same ranking on every rank
        orcs = [O.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan, partition_count=world) for _ in range(world)]
        for g in range(world):
            for it in range(steps):
                orcs[g].run_batch(parts[g], ds.labels[parts[g]], it, is_presc=True)
        _, QF = O.candidate_selection([o.node_access_time for o in orcs], V)
        _, QT = O.candidate_selection([o.edge_access_time for o in orcs], V)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQF(eng.cache, 0), np.int32, V), QF)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQT(eng.cache, 0), np.int32, V), QT)
        # exchange the shards over HIP IPC
        mine = eng.export_shards(rank)
        assert mine[0] is not None and len(mine[0]) == L.GPUCache_ShardChunkCount(eng.cache, rank) > 1 and mine[1] is not None
        assert len(mine[2]) == L.GPUGraphStorage_FragmentChunkCount(eng.graph, rank, 1) > 1      # CSR fragment in several chunks too
        everyone = D.allgather_object(mine, world)
        for g in range(world):
            if g != rank:
                eng.import_shards(g, everyone[g])
        for g in peers:
            assert L.GPUCache_Float_Feature_Cache(eng.cache, g)
            assert L.GPUGraphStorage_GetFragmentIndex(eng.graph, rank, g)                          # peer fragment visible from here
            assert L.GPUGraphStorage_FragmentChunkCount(eng.graph, g, 1) == len(everyone[g][2])
            assert L.GPUGraphStorage_FragmentEdges(eng.graph, g) == everyone[g][3][1]
        # steady state through the unified cache: own shard, peer shard (IPC) and backing-table misses
        me = orcs[rank]
        me.set_feature_cache(QF, cap, world)
        me.set_topo_cache(QT, cap, world, 0)
        peer_rows = 0
        for it in range(min(3, (len(parts[rank]) + B - 1) // B)):       # never past the end of the seed shard
            ref = me.run_batch(parts[rank], ds.labels[parts[rank]], it)
            eng.run_batch(rank, it)
            got = eng.result(rank)
            assert_batch_equal(ref, got)
            slot = me.node_map[got["ids"]]
            peer_rows += int(((slot >= 0) & (slot // cap != rank)).sum())
            assert ((slot >= 0) & (slot // cap == rank)).any() and (slot < 0).any()
        assert peer_rows > 0
        # the owner-computes exchange variant of the same gather (legion1_amd/exchange.py: plan -> all-to-all of the request
        # lists -> every owner gathers from its own shard -> all-to-all of the rows -> scatter): bit-identical batches
        import torch
        from legion1_amd.exchange import ExchangeGather
        xg = ExchangeGather(K, eng, rank, world, F, torch.device("cuda", 0), eng.num_ids)
        exchanged = 0
        for it in range(min(2, (len(parts[rank]) + B - 1) // B)):
            ref = me.run_batch(parts[rank], ds.labels[parts[rank]], it)
            eng.run_batch(rank, it, gather=False, plan=False)                        # sampler only
            feat = eng.out[rank][0]["feat"]
            L.d_memset_async(feat.ptr, 0xFF, feat.nbytes, None)                      # poison: every row must be rewritten
            L.d_stream_sync(None)
            info = xg.run(None, eng.pools[rank])
            got = eng.result(rank)
            assert_batch_equal(ref, got)
            slot = me.node_map[got["ids"]]
            assert info["rows_requested"] == int(((slot >= 0) & (slot // cap != rank)).sum()) > 0
            assert info["per_owner"][rank] == 0 and sum(info["per_owner"]) == info["rows_requested"]
            exchanged += info["rows_served"]
        assert exchanged > 0
        xg.close()
        dist.barrier()        # nobody unmaps a shard while the peer may still read it
        eng.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", peer_rows))
    except Exception as ex:  # surface the failure in the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(ex)))


@pytest.mark.parametrize("world", [2, 4])
def test_process_per_gpu_clique_unified_cache_over_ipc(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, status, info in out:
        assert status == "ok", info
    assert all(info > 0 for _, _, info in out)
