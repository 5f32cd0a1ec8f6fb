"""One process per GPU with a clique-wide unified cache (BASELINE config "hotness-partitioned unified
cache over xGMI"): every rank drives one logical GPU of a Kg = 2 / 4 clique, fills ITS shard of the
feature cache / CSR fragments (rank-t item on clique GPU t % Kg, GPUCache.cu:88-108), exports it over
HIP IPC and reads the peers' shards in-kernel.  The only collective is the build-time hotness sum
(CandidateSelection, GPUCache.cu:624-627) -- gloo here, RCCL in bench.py.  All ranks share the one
GPU of the test box; on a real node the same loads cross xGMI.

Kg = 8 (cache_agg_mode 3, GPUCache.cu:593-607 -- the configuration BASELINE.json states): the GPU box allows at most 6
processes on its card, so the eight-member clique runs as 4 processes x 2 logical GPUs each: every process fills and
exports two shards and imports the other six (eight-way shard tables, 6 x 8 feature chunks + CSR fragment chunks over
HIP IPC per process)."""
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


def _worker(rank, world, per, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_SHARD_CHUNK_BYTES="200000")   # several chunks per shard
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import legion1_amd.capi as K
        import legion1_amd.dist as D
        import legion1_amd.synth as S
        import oracle as O
        from conftest import assert_batch_equal
        L = K.lib()
        L.legion_set_error_mode(K.ERR_RETURN)
        G = world * per                            # clique size Kg; this process drives `mine`
        mine = [rank * per + k for k in range(per)]
        for g in range(G):
            L.legion_set_device_map(g, 0)          # the test box has one GPU: every clique member lives on it
        ds = S.generate(S.spec_for("products", scale=0.01))
        V, F = ds.spec.V, ds.spec.F
        B, fan, cap = (250 if G < 8 else 100), [10, 5], 2000      # 8 partitions of the 1966 seeds hold 245 each
        parts = O.split_seeds(ds.train, G)
        steps = min((len(p) - 1) // B for p in parts)
        seeds = dict(train=[(p, ds.labels[p]) for p in parts])
        eng = K.Engine(ds.indptr, ds.indices, ds.features, V, F, seeds, B, fan, G=G, local_devs=mine, train_step=steps)
        eng.alloc_features()
        peers = [g for g in range(G) if g not in mine]
        assert all(L.legion_is_remote_device(g) == 1 for g in peers) and all(L.legion_is_remote_device(g) == 0 for g in mine)
        # pre-sampling epoch on the own partitions, then the clique-wide hotness sum as a collective: member k of every
        # process is reduced with member k of the others, CandidateSelection adds up the local members itself
        for g in mine:
            for it in range(steps):
                eng.run_batch(g, it, is_presc=True)
        for g in mine:
            D.allreduce_device_u64(K, L.GPUCache_GetNodeAccessedMap(eng.cache, g), V, world)
            D.allreduce_device_u64(K, L.GPUCache_GetEdgeAccessedMap(eng.cache, g), V, world)
        eng.build_cache(cache_agg_mode={2: 1, 4: 2, 8: 3}[G], node_capacity=cap, edge_capacity=cap, train_step=steps)
        assert L.GPUCache_Kg(eng.cache) == G and L.GPUCache_Kc(eng.cache) == 1
        # oracle: hotness of ALL partitions -> same ranking on every rank
        orcs = [O.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan, partition_count=G) for _ in range(G)]
        for g in range(G):
            for it in range(steps):
                orcs[g].run_batch(parts[g], ds.labels[parts[g]], it, is_presc=True)
        _, QF = O.candidate_selection([o.node_access_time for o in orcs], V)
        _, QT = O.candidate_selection([o.edge_access_time for o in orcs], V)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQF(eng.cache, 0), np.int32, V), QF)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQT(eng.cache, 0), np.int32, V), QT)
        # exchange the shards over HIP IPC
        exported = {g: eng.export_shards(g) for g in mine}
        for g in mine:
            e = exported[g]
            assert e[0] is not None and len(e[0]) == L.GPUCache_ShardChunkCount(eng.cache, g) > 1 and e[1] is not None
            assert len(e[2]) == L.GPUGraphStorage_FragmentChunkCount(eng.graph, g, 1) > 1      # CSR fragment in several chunks too
        everyone = {}
        for d in D.allgather_object(exported, world):
            everyone.update(d)
        assert sorted(everyone) == list(range(G))
        for g in peers:
            eng.import_shards(g, everyone[g])
        for g in peers:
            assert L.GPUCache_Float_Feature_Cache(eng.cache, g)
            for v in mine:
                assert L.GPUGraphStorage_GetFragmentIndex(eng.graph, v, g)                         # peer fragment visible from every local member
            assert L.GPUGraphStorage_FragmentChunkCount(eng.graph, g, 1) == len(everyone[g][2])
            assert L.GPUGraphStorage_FragmentEdges(eng.graph, g) == everyone[g][3][1]
        # steady state through the unified cache: own shard, peer shards (IPC) and backing-table misses
        peer_rows, owners_seen = 0, set()
        for g in mine:
            me = orcs[g]
            me.set_feature_cache(QF, cap, G)
            me.set_topo_cache(QT, cap, G, 0)
            for it in range(min(3, (len(parts[g]) + B - 1) // B)):       # never past the end of the seed shard
                ref = me.run_batch(parts[g], ds.labels[parts[g]], it)
                eng.run_batch(g, it)
                got = eng.result(g)
                assert_batch_equal(ref, got)
                slot = me.node_map[got["ids"]]
                remote = (slot >= 0) & ~np.isin(slot // cap, mine)       # rows read from another PROCESS' shard
                peer_rows += int(remote.sum())
                owners_seen |= set(np.unique(slot[slot >= 0] // cap).tolist())
                assert ((slot >= 0) & (slot // cap == g)).any() and (slot < 0).any()
        assert peer_rows > 0 and owners_seen == set(range(G)), owners_seen   # every member's shard served rows
        # the owner-computes exchange variant of the same gather (legion1_amd/exchange.py: plan -> all-to-all of the request
        # lists -> every owner gathers from its own shard -> all-to-all of the rows -> scatter): bit-identical batches
        if per == 1:
            import torch
            from legion1_amd.exchange import ExchangeGather
            me = orcs[rank]
            xg = ExchangeGather(K, eng, rank, world, F, torch.device("cuda", 0), eng.num_ids)
            exchanged = 0
            for it in range(min(2, (len(parts[rank]) + B - 1) // B)):
                ref = me.run_batch(parts[rank], ds.labels[parts[rank]], it)
                eng.run_batch(rank, it, gather=False, plan=False)                        # sampler only
                feat = eng.out[rank][0]["feat"]
                L.d_memset_async(feat.ptr, 0xFF, feat.nbytes, None)                      # poison: every row must be rewritten
                L.d_stream_sync(None)
                info = xg.run(None, eng.pools[rank])
                xg.wait()
                got = eng.result(rank)
                assert_batch_equal(ref, got)
                slot = me.node_map[got["ids"]]
                assert info["rows_requested"] == int(((slot >= 0) & (slot // cap != rank)).sum()) > 0
                assert info["per_owner"][rank] == 0 and sum(info["per_owner"]) == info["rows_requested"]
                exchanged += info["rows_served"]
            assert exchanged > 0
            assert xg.host_syncs_per_batch <= 1, xg.host_syncs_per_batch
            xg.close()
        dist.barrier()        # nobody unmaps a shard while a peer may still read it
        eng.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", peer_rows))
    except Exception as ex:  # surface the failure in the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc() + repr(ex)))


@pytest.mark.parametrize("world,per", [(2, 1), (4, 1), (4, 2)])
def test_process_per_gpu_clique_unified_cache_over_ipc(world, per):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, per, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, status, info in out:
        assert status == "ok", info
    assert all(info > 0 for _, _, info in out)
