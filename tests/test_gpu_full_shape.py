"""BASELINE.json configs 3, 4 and 5 in their stated form and at their full shape, on the one device of the test
box (the clique's logical GPUs are mapped onto it with legion_set_device_map; on a node the same loads cross xGMI):

  3  papers100M 3-hop, hotness-partitioned unified feature cache over a Kg = 2 clique (default 1 GiB shard chunks)
  4  uk-union 2-hop, CSR sharded across the clique (partitioned fragments, E = 5.5e9 > 2^32) + features in pinned
     host memory with a capped HBM cache (hits from both shards, misses spill to the host table over PCIe)
  5  link prediction on the papers100M shape ([src | pos | neg] seed thirds with duplicates)

Checks: the size-independent properties of tests/props.py, byte equality of every gathered row with the generator's closed form, split by
where the row came from, and -- since round 4 -- every batch word for word against the (OpenMP) oracle run on a host copy of the CSR."""
import numpy as np
import pytest

from conftest import assert_batch_equal
from props import check_batch, device_graph, device_seeds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import legion1_amd.capi as K
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    L.SetGPUDevice(0)
    return K


def _clique_engine(K, spec, indptr, indices, feat_ptr, feat_loc, E, parts, B, fan, presc_steps):
    G = len(parts)
    L = K.lib()
    for g in range(G):
        L.legion_set_device_map(g, 0)
    seeds = dict(train=[((ids.data_ptr(), int(ids.numel())), (lab.data_ptr(), int(lab.numel()))) for ids, lab in parts])
    eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feat_ptr, spec.V, spec.F, seeds, B, fan, G=G, csr_location=K.LOC_DEVICE,
                   features_location=feat_loc, E=E, train_step=presc_steps)
    eng.alloc_features()
    for g in range(G):                      # pre-sampling epoch (Server.cu:83-95): hotness of nodes and adjacency rows
        for it in range(presc_steps):
            eng.run_batch(g, it, is_presc=True)
    return eng


def _host_oracle(oracle, spec, indptr, indices, B, fan):
    """The resident-form oracle on a host copy of the CSR.  The sampler's output does not depend on WHERE a row is read (whole CSR, own or
    peer fragment: same values, Kernels.cu:392-409) and neither do the gathered rows (a cached row is a copy of the table row), so the
    resident batch is the expected result of the partitioned / cached configurations too -- bit for bit."""
    return oracle.OracleRunner(indptr.cpu().numpy(), indices.cpu().numpy(), None, spec.V, spec.F, B, fan, with_features=False)


ORACLE_KEYS = ("nc", "ec", "ids", "labels", "src_off", "dst_off")


def _row_sources(K, eng, g, ids, cap):
    """Where FindFeat sends each row of a batch of logical GPU g: own shard / peer shard / backing table."""
    L = K.lib()
    L.SetGPUDevice(g)
    fmap = K.read_dev(L.GPUCache_GetFeatureMap(eng.cache, g), np.int32, eng.V)
    slot = fmap[ids]
    own = (slot >= 0) & (slot // cap == g)
    peer = (slot >= 0) & (slot // cap != g)
    return own, peer, slot < 0


@pytest.mark.parametrize("G,mode,presc", [(2, 1, 6), (8, 3, 1)])     # G x presc pre-sampled batches must fit the cache (see below)
def test_papers100m_unified_cache_full_shape(K, oracle, synth, G, mode, presc):
    """Config 3: 25 % of the V feature rows cached by hotness over the clique (rank-t row on GPU t % Kg, GPUCache.cu:88-108).
    Kg = 2: 7 x 1 GiB chunks per shard.  Kg = 8 (cache_agg_mode 3, GPUCache.cu:593-607 -- the clique BASELINE.json states):
    eight shards of 2 x 1 GiB chunks, eight-way shard tables; every owner's rows are checked byte for byte."""
    L = K.lib()
    spec, indptr, indices, feats, E = device_graph(K, synth, "papers100M")
    B, fan = 8000, [25, 10, 5]
    parts = device_seeds(K, spec, G)
    eng = _clique_engine(K, spec, indptr, indices, feats.data_ptr(), K.LOC_DEVICE, E, parts, B, fan, presc)
    cap = int(spec.V * 0.25) // G + 1
    eng.build_cache(cache_agg_mode=mode, node_capacity=cap, edge_capacity=0, train_step=presc)
    assert L.GPUCache_Kg(eng.cache) == G and L.GPUCache_Kc(eng.cache) == 1 and L.GPUCache_NodeCapacity(eng.cache, 0) == cap
    nch = -(-cap * spec.F * 4 // (1 << 30))                          # 7.1 GB / 1.8 GB per shard in 1 GiB chunks
    assert all(L.GPUCache_ShardChunkCount(eng.cache, g) == nch for g in range(G)) and nch == {2: 7, 8: 2}[G]
    rs = np.random.RandomState(3)
    orc = _host_oracle(oracle, spec, indptr, indices, B, fan)
    for g in (range(G) if G == 2 else (0, 3, 7)):
        seeds_g, labs_g = parts[g][0].cpu().numpy(), parts[g][1].cpu().numpy()
        for it in (0, presc + 1):           # a batch of the pre-sampling epoch and one the cache has never seen
            eng.run_batch(g, it, per_level=(it == 0))
            res = eng.result(g)
            check_batch(res, spec, synth, B, fan, indptr, indices, seeds_g[it * B:(it + 1) * B], rs)
            # the cached configuration against the oracle at full shape, word for word
            assert_batch_equal(orc.run_batch(seeds_g, labs_g, it, gather=False, omp=True), res, keys=ORACLE_KEYS)
            own, peer, miss = _row_sources(K, eng, g, res["ids"], cap)
            # every node of a pre-sampled batch has hotness > 0 and the cache holds more rows than were ever seen, so
            # batch 0 is served from the shards alone; a batch the cache has never seen also misses
            kinds = (own.sum(), peer.sum(), miss.sum())
            assert own.sum() > 0.2 / G * len(res["ids"]) and peer.sum() > 0.05 * len(res["ids"]), kinds
            assert (kinds[2] > 0.05 * len(res["ids"]) if it else kinds[2] == 0), kinds
            for name, m in (("own", own), ("peer", peer), ("miss", miss)):       # every source serves the right bytes
                if m.sum() == 0:
                    continue
                rows = rs.choice(np.flatnonzero(m), size=1500, replace=False)
                assert np.array_equal(res["features"][rows], synth.features(spec, res["ids"][rows])), name
            # per owner: the rows every clique member's shard served
            fmap = K.read_dev(L.GPUCache_GetFeatureMap(eng.cache, g), np.int32, eng.V)
            slot = fmap[res["ids"]]
            for owner in range(G):
                m = np.flatnonzero((slot >= 0) & (slot // cap == owner))
                assert len(m) > 0.1 / G * len(slot), (g, owner, len(m))      # all eight shards serve rows
                rows = rs.choice(m, size=min(400, len(m)), replace=False)
                assert np.array_equal(res["features"][rows], synth.features(spec, res["ids"][rows])), ("owner", owner)
    # the cached rows are the hottest: hit rate well above the cached fraction
    assert (own.sum() + peer.sum()) / len(res["ids"]) > 0.3
    eng.close()


@pytest.mark.parametrize("G,mode,host_spill", [(2, 1, True), (8, 3, False)])
def test_uk_union_sharded_csr_full_shape(K, oracle, synth, G, mode, host_spill):
    """Config 4: uk-union 2-hop {25,10}; the hottest adjacency rows as partitioned CSR fragments over the clique (several 1 GiB index
    chunks per fragment at Kg = 2: the row_shift / edge_shift chunk tables are exercised at E > 2^32) behind a capped feature cache.
    Kg = 2: the feature table in pinned host memory (137 GB), misses spill over PCIe.  Kg = 8 (cache_agg_mode 3, the clique
    BASELINE.json states): eight fragments / shards, rank-t row on GPU t % 8, the backing table stays in HBM."""
    import torch
    L = K.lib()
    spec, indptr, indices, feats, E = device_graph(K, synth, "uk-union")
    assert E > 2 ** 32
    V, F = spec.V, spec.F
    host = None
    if host_spill:
        nbytes = V * F * 4
        host = L.host_alloc_space64(nbytes)
        if not host:
            L.legion_clear_error()
            pytest.skip("no %d GB of pinned host memory on this box" % (nbytes >> 30))
        L.d_copy_d_2_h(host, feats.data_ptr(), nbytes)
        K.check()
        del feats
        torch.cuda.empty_cache()
    B, fan, presc = 8000, [25, 10], (6 if G == 2 else 2)
    parts = device_seeds(K, spec, G)
    eng = _clique_engine(K, spec, indptr, indices, host if host_spill else feats.data_ptr(), K.LOC_HOST_PINNED if host_spill else K.LOC_DEVICE,
                         E, parts, B, fan, presc)
    cap_n, cap_e = int(V * 0.10) // G + 1, int(V * 0.30) // G + 1
    eng.build_cache(cache_agg_mode=mode, node_capacity=cap_n, edge_capacity=cap_e, train_step=presc)
    assert L.GPUCache_Kg(eng.cache) == G
    for g in range(G):
        assert L.GPUGraphStorage_FragmentRows(eng.graph, g) == cap_e
        edges = L.GPUGraphStorage_FragmentEdges(eng.graph, g)
        nix = L.GPUGraphStorage_FragmentChunkCount(eng.graph, g, 1)
        assert nix == (edges - 1) // L.GPUGraphStorage_FragmentChunkSpan(eng.graph, 1) + 1 and (nix > 1 or G == 8), (edges, nix)
    rs = np.random.RandomState(4)
    L.SetGPUDevice(0)
    d_probe = K.DevBuf.from_numpy(np.arange(V, dtype=np.int32))
    d_pi, d_po = K.DevBuf(V), K.DevBuf(V * 4)
    L.GPUCache_FindTopo(eng.cache, d_probe.ptr, d_pi.ptr, d_po.ptr, V, 2, None, 0)
    L.d_stream_sync(None)
    owner = d_pi.to_numpy(np.int8, V)
    for b in (d_probe, d_pi, d_po):
        b.free()
    assert all((owner == g).sum() == cap_e for g in range(G))     # rank-t row on GPU t % Kg
    orc = _host_oracle(oracle, spec, indptr, indices, B, fan)
    for g in (range(G) if G == 2 else (0, 5)):
        seeds_g, labs_g = parts[g][0].cpu().numpy(), parts[g][1].cpu().numpy()
        for it in (1, presc + 2):
            eng.run_batch(g, it)
            res = eng.result(g)
            levels = check_batch(res, spec, synth, B, fan, indptr, indices, seeds_g[it * B:(it + 1) * B], rs)
            # the sampler over partitioned CSR fragments (own, peer, whole-CSR rows mixed; E > 2^32) against the oracle, word for word
            assert_batch_equal(orc.run_batch(seeds_g, labs_g, it, gather=False, omp=True), res, keys=ORACLE_KEYS)
            ids = res["ids"]
            # the sampler expanded rows of all three kinds: own fragment, peer fragments, the whole-CSR replica
            srcs = ids[:levels[0] + levels[1]]
            kinds = [(owner[srcs] == g).sum(), ((owner[srcs] >= 0) & (owner[srcs] != g)).sum(), (owner[srcs] < 0).sum()]
            assert min(kinds[:2]) > 0 and (kinds[2] > 0 or it < presc), kinds    # rows of a pre-sampled batch are all cached
            if G == 8:
                assert len(np.unique(owner[srcs][owner[srcs] >= 0])) == 8         # every member's fragment was read
            own, peer, miss = _row_sources(K, eng, g, ids, cap_n)
            assert own.sum() > 0.03 / G * len(ids) and peer.sum() > 0.02 * len(ids) and (miss.sum() > 0.02 * len(ids) or it < presc), (own.sum(), peer.sum(), miss.sum())
            for name, m in (("own", own), ("peer", peer), ("backing", miss)):
                if m.sum() == 0:
                    continue
                rows = rs.choice(np.flatnonzero(m), size=min(1000, int(m.sum())), replace=False)
                assert np.array_equal(res["features"][rows], synth.features(spec, ids[rows])), name
    # the same batch again: bit-identical (fragments, cache and backing rows are all deterministic sources)
    eng.run_batch(0, 1)
    again = eng.result(0)
    eng.run_batch(0, 1)
    assert_batch_equal(again, eng.result(0))
    eng.close()
    if host:
        L.host_free_space(host)


@pytest.mark.parametrize("world,rank", [(2, 1), (8, 5)])
def test_papers100m_link_prediction_full_shape(K, oracle, synth, world, rank):
    """Config 5: [src | pos | neg] seed batches at the papers100M shape (11.1 M triples generated on the GPU by
    legion_synth_lp_seeds; the list of logical GPU `rank` of `world`: triples dealt by src % world -- 2 GPUs, and the 8 GPUs
    BASELINE.json states), 3-hop."""
    import torch
    L = K.lib()
    spec, indptr, indices, feats, E = device_graph(K, synth, "papers100M")
    dev = indptr.device
    B, fan = 7998, [25, 10, 5]
    k = B // 3
    tr = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, tr.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
    torch.cuda.synchronize()
    mask = (tr % world) == rank
    srcs, triple_no = tr[mask].contiguous(), torch.nonzero(mask).reshape(-1).contiguous()
    n_tr = int(srcs.numel())
    seeds = torch.empty((n_tr + k - 1) // k * B, dtype=torch.int32, device=dev)
    L.legion_synth_lp_seeds(None, seeds.data_ptr(), srcs.data_ptr(), triple_no.data_ptr(), n_tr, B, indptr.data_ptr(), indices.data_ptr(), spec.V, 1)
    torch.cuda.synchronize()
    K.check()
    assert n_tr > 10_000_000 // world
    lab = torch.empty(spec.V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, lab.data_ptr(), 0, spec.V, spec.classes)
    my_lab = lab[seeds.long()].contiguous()
    n = int(seeds.numel())
    eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feats.data_ptr(), spec.V, spec.F,
                   dict(train=[((seeds.data_ptr(), n), (my_lab.data_ptr(), n))]), B, fan, E=E)
    eng.alloc_features()
    rs = np.random.RandomState(5)
    h_seeds = seeds.cpu().numpy()
    # the same rule in numpy on the first and the last batch (closed form of the generator: no graph needed on the host)
    m31 = 2147483647
    for b in (0, n // B - 1):
        j = np.arange(b * k, min((b + 1) * k, n_tr))
        t = triple_no[torch.from_numpy(j).to(dev)].cpu().numpy()
        s = srcs[torch.from_numpy(j).to(dev)].cpu().numpy().astype(np.int64)
        x1 = synth.minstd_pow(np.uint64(1) + np.uint64(2) * t.astype(np.uint64) + np.uint64(1))
        x2 = (x1 * np.uint64(48271)) % np.uint64(m31)
        st = torch.from_numpy(s).to(dev)
        lo = indptr[st].cpu().numpy()
        deg = indptr[st + 1].cpu().numpy() - lo
        pick = lo + ((x1 - np.uint64(1)) % np.maximum(deg, 1).astype(np.uint64)).astype(np.int64)
        pos = np.where(deg > 0, indices[torch.from_numpy(pick).to(dev)].cpu().numpy(), s)
        neg = ((x2 - np.uint64(1)) % np.uint64(spec.V)).astype(np.int64)
        batch = h_seeds[b * B:(b + 1) * B]
        m = len(j)
        assert np.array_equal(batch[:m], s) and np.array_equal(batch[k:k + m], pos) and np.array_equal(batch[2 * k:2 * k + m], neg)
        assert (s % world == rank).all()
    dup_batches = 0
    orc = _host_oracle(oracle, spec, indptr, indices, B, fan)
    h_lab = my_lab.cpu().numpy()
    for it in (0, 3, n // B - 1):
        eng.run_batch(0, it)
        res = eng.result(0)
        batch = h_seeds[it * B:(it + 1) * B]
        dup_batches += int(len(np.unique(batch)) < B)
        check_batch(res, spec, synth, B, fan, indptr, indices, batch, rs, distinct_seeds=False)
        assert_batch_equal(orc.run_batch(h_seeds, h_lab, it, gather=False, omp=True), res, keys=ORACLE_KEYS)   # the per-rank list of a 2 / 8-GPU job
    assert dup_batches > 0          # hot positives repeat inside a batch: the duplicate-seed path really ran
    eng.close()


# ---- bit-exact parity with the oracle at the BASELINE shapes -----------------------------------------------------------
def _full_shape_cases(workload):
    from conftest import load_golden
    gold = load_golden("full_shape_digests")
    return gold["fields"], {n: c for n, c in gold["cases"].items() if c["workload"] == workload}


@pytest.mark.parametrize("workload", ["products", "papers100M", "uk-union"])
def test_full_shape_matches_oracle(K, oracle, synth, workload):
    """Every BASELINE.json configuration at its full shape, bit for bit against the oracle (kernel_random_sampler_2 +
    construct_graph + update_counter, Kernels.cu:342-463, :112-150; batch_generator incl. the short last batch, :68-96,224):
    the CSR is generated on the GPU, copied to the host once, and the OpenMP oracle (byte-identical to the serial one,
    tests/test_oracle_batches.py) runs batch 0, a middle batch and the last batch there; nc, ec, ids, labels, src_off and
    dst_off must be equal word for word, AND their SHA-256 must be the digests the SERIAL oracle produced on a CSR generated
    by oracle/synth_gen.c in the build container (tests/golden/full_shape_digests.json, `oracle/make_golden.py full`).
    What only these sizes reach: slot indices up to 12.2 M in the pow-table RNG, positions >= 2^21, k_write's grouped LDS
    prefix (gshift > 0) on real data, edge offsets > 2^32 (uk-union), loser -> winner chains at real contention.
    Feature rows: every gathered row against the table row on the device (the table against the closed form on a sample)."""
    import torch
    from conftest import sha
    L = K.lib()
    fields, cases = _full_shape_cases(workload)
    assert cases
    spec, indptr, indices, feats, E = device_graph(K, synth, workload)
    dev = indptr.device
    h_indptr, h_indices = indptr.cpu().numpy(), indices.cpu().numpy()
    rs = np.random.RandomState(11)
    probe = rs.randint(0, spec.V, 512)
    assert np.array_equal(feats[torch.from_numpy(probe).to(dev)].cpu().numpy(), synth.features(spec, probe))
    tr = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, tr.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
    lab_all = torch.empty(spec.V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, lab_all.data_ptr(), 0, spec.V, spec.classes)
    torch.cuda.synchronize()
    for name, case in cases.items():
        B, fan, H = case["batch"], case["fanout"], len(case["fanout"])
        assert case["V"] == spec.V and case["E"] == E and sha(h_indptr) == case["indptr_sha256"], name
        assert sha(h_indices[:1 << 24]) == case["indices_head_sha256"], name
        seeds = tr
        if case["task"] == "lp":
            k = B // 3
            triple_no = torch.arange(spec.n_train, dtype=torch.int64, device=dev)
            seeds = torch.empty((spec.n_train + k - 1) // k * B, dtype=torch.int32, device=dev)
            L.legion_synth_lp_seeds(None, seeds.data_ptr(), tr.data_ptr(), triple_no.data_ptr(), spec.n_train, B, indptr.data_ptr(),
                                    indices.data_ptr(), spec.V, 1)
            torch.cuda.synchronize()
            K.check()
        my_lab = lab_all[seeds.long()].contiguous()
        n = int(seeds.numel())
        h_seeds, h_lab = seeds.cpu().numpy(), my_lab.cpu().numpy()
        assert n == case["n_seeds"] and sha(h_seeds) == case["seeds_sha256"], name
        eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feats.data_ptr(), spec.V, spec.F,
                       dict(train=[((seeds.data_ptr(), n), (my_lab.data_ptr(), n))]), B, fan, E=E)
        eng.alloc_features()
        orc = oracle.OracleRunner(h_indptr, h_indices, None, spec.V, spec.F, B, fan, with_features=False)
        for want in case["batches"]:
            counter = want["counter"]
            eng.run_batch(0, counter)
            res = eng.result(0)
            ref = orc.run_batch(h_seeds, h_lab, counter, gather=False, omp=True)
            assert_batch_equal(ref, res, keys=tuple(fields))
            assert int(res["nc"][4]) == want["size"] and len(res["ids"]) == want["n_nodes"] and len(res["src_off"]) == want["n_edges"], (name, counter)
            for f in fields:
                assert sha(res[f]) == want[f + "_sha256"], (name, counter, f)
            # S5: every row of the batch is the table row of its id
            ids_d = torch.from_numpy(res["ids"].astype(np.int64)).to(dev)
            step = 1 << 19
            for r0 in range(0, len(res["ids"]), step):
                assert np.array_equal(res["features"][r0:r0 + step], feats[ids_d[r0:r0 + step]].cpu().numpy()), (name, counter, r0)
        eng.close()
        del orc
    del feats, indices, indptr
    torch.cuda.empty_cache()


def test_synth_server_serves_the_headline_shape(K, oracle, synth, tmp_path):
    """The `legion` server binary with the dataset source `synth:papers100M` (tables generated in HBM, no files), {25,10,5}, B = 8000 --
    what bench.py's `served` leg times -- handing batches to a trainer-side process over shm + semaphores + IPC handles (Server.cu:301-328,
    CUDA_IPC_Service.cu:289-297, ipc_cuda_kernel.cu:98-107,178-230).  The first, the middle and the last served training batch, the
    validation and the test batch are compared word for word with the OpenMP oracle on a host copy of the same CSR; batches 0 and 694
    also with the SERIAL oracle's digests from the build container (tests/golden/full_shape_digests.json); every served feature row of those
    batches with the generator's closed form (SHA-256 over all rows)."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    import time
    from conftest import ROOT, sha
    server_bin = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")
    fields, cases = _full_shape_cases("papers100M")
    case = cases["papers100M-25,10,5"]
    B, fan, H = case["batch"], case["fanout"], 3
    spec = synth.spec_for("papers100M")
    n_eval = 512
    # soak (profiles/r05_runs_soak.sh): LEGION_TEST_EPOCHS=3 LEGION_TEST_SERVED_EVERY=97 also checks every 97th served batch of three epochs
    epochs, every = int(os.environ.get("LEGION_TEST_EPOCHS", "1")), int(os.environ.get("LEGION_TEST_SERVED_EVERY", "0"))
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write("synth:papers100M %d %d %d %d %d %d %d 0 %d 0" % (B, spec.V, case["E"], spec.F, spec.n_train, n_eval, n_eval, epochs))
    env = dict(os.environ, LEGION_IPC_NAMESPACE="fs%d_" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_CLIENT_DUMP_IDS="1")
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([server_bin, "1", "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=str(tmp_path))
    out = str(tmp_path / "client.json")
    train_step = (spec.n_train - 1) // B
    record = [0, train_step // 2, train_step - 1, train_step, train_step + 1]
    assert record[1] == 694
    total = (train_step + 1) * epochs + 1
    if epochs > 1:
        record = [0, train_step // 2, train_step - 1, train_step, total - 1]        # the test batch comes behind the last epoch
    soak = sorted(set(range(0, total, every)) - set(record)) if every > 0 else []
    record = sorted(record + soak)
    try:
        t0 = time.time()
        while "System is ready for serving" not in open(log, errors="ignore").read():
            assert server.poll() is None, open(log).read()[-3000:]
            assert time.time() - t0 < 300, open(log).read()[-3000:]
            time.sleep(0.2)
        client = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ipc_client_plain.py"), str(spec.F), str(epochs), out, ",".join(map(str, record))],
                                env=env, capture_output=True, text=True, timeout=1000)
        assert client.returncode == 0, client.stdout[-2000:] + client.stderr[-3000:]
        server.wait(timeout=120)
        assert server.returncode == 0, open(log).read()[-3000:]
    finally:
        if server.poll() is None:
            server.kill()
    text = open(log).read()
    assert "Graph generated in HBM: %d edges" % case["E"] in text and "Train Steps: %d" % train_step in text
    got = json.load(open(out))
    assert got["steps"] == [train_step, 1, 1] and got["hops"] == H and [r["b"] for r in got["batches"]] == record
    per = train_step + 1
    # the oracle side: the CSR from the independent C generator on the host (oracle/synth_gen.c, OpenMP)
    h_indptr, h_indices = oracle.synth_csr(spec)
    assert int(h_indptr[-1]) == case["E"] and sha(h_indptr) == case["indptr_sha256"]
    orc = oracle.OracleRunner(h_indptr, h_indices, None, spec.V, spec.F, B, fan, with_features=False)
    n1, n2 = spec.n_train, spec.n_train + spec.n_valid
    sets = {0: synth.seed_ids(spec, 0, n1), 1: synth.seed_ids(spec, n1, n1 + n_eval), 2: synth.seed_ids(spec, n2, n2 + n_eval)}
    assert sha(sets[0]) == case["seeds_sha256"]
    labs = {m: synth.labels(spec, ids) for m, ids in sets.items()}
    golden = {w["counter"]: w for w in case["batches"]}
    for rec in got["batches"]:
        if rec["b"] >= per * epochs:
            mode, local = 2, 0
        else:
            mode, local = (0, rec["b"] % per) if rec["b"] % per < train_step else (1, 0)
        ids = sets[mode]
        ref = orc.run_batch(ids, labs[mode], local, mode=mode, batch_size=B if mode == 0 else n_eval, gather=False, omp=(mode == 0))
        assert rec["nc"] == ref["nc"].tolist() and rec["ec"] == ref["ec"].tolist(), rec["b"]
        assert rec["ids"] == sha(ref["ids"]) and rec["labels"] == sha(ref["labels"]), rec["b"]
        assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"]), rec["b"]
        if mode == 0 and local in golden:          # ... and the serial oracle's digests of the build container
            w = golden[local]
            assert (rec["n"], rec["e"]) == (w["n_nodes"], w["n_edges"])
            assert rec["ids"] == w["ids_sha256"] and rec["src"] == w["src_off_sha256"] and rec["dst"] == w["dst_off_sha256"] and rec["labels"] == w["labels_sha256"]
        served_ids = np.load(out + ".ids%d.npy" % rec["b"])
        assert np.array_equal(served_ids, ref["ids"])
        if rec["b"] in soak and soak.index(rec["b"]) % 4:      # soak batches: the rows of every fourth one against the closed form
            continue
        h = hashlib.sha256()
        for r0 in range(0, len(served_ids), 1 << 16):
            h.update(np.ascontiguousarray(synth.features(spec, served_ids[r0:r0 + (1 << 16)])).tobytes())
        assert rec["features"] == h.hexdigest(), rec["b"]


def test_cost_model_without_pcm_counters_matches_the_golden_plan(K, synth):
    """The product's cache plan at the full products shape with counters == NULL (the `legion` server's only mode: no Intel PCM) against
    tests/golden/cost_model_pcm_free.json, which the oracle produced in the build container: pre-sampling hotness (SHA-256 of both u64[V]
    arrays), max_ids, and per budget alpha and both capacities.  An edit of the transaction estimate's weight changes these numbers."""
    from conftest import load_golden, sha
    L = K.lib()
    g = load_golden("cost_model_pcm_free")
    spec, indptr, indices, feats, E = device_graph(K, synth, g["workload"])
    assert (spec.V, E, spec.F) == (g["V"], g["E"], g["F"])
    (ids, lab), = device_seeds(K, spec, 1)
    n = int(ids.numel())
    B, fan, steps = g["batch"], g["fanout"], g["presc_steps"]
    for plan in g["plans"]:
        eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feats.data_ptr(), spec.V, spec.F, dict(train=[((ids.data_ptr(), n), (lab.data_ptr(), n))]),
                       B, fan, E=E, cache_memory=plan["cache_memory"], train_step=steps)
        eng.alloc_features()
        for it in range(steps):
            eng.run_batch(0, it, is_presc=True)
        assert L.GPUCache_MaxIdNum(eng.cache, 0) == g["max_ids"]
        assert sha(K.read_dev(L.GPUCache_GetNodeAccessedMap(eng.cache, 0), np.uint64, spec.V)) == g["node_hotness_sha256"]
        assert sha(K.read_dev(L.GPUCache_GetEdgeAccessedMap(eng.cache, 0), np.uint64, spec.V)) == g["edge_hotness_sha256"]
        eng.build_cache(cache_agg_mode=0, counters=None, train_step=steps)
        assert sha(K.read_dev(L.GPUCache_GetQF(eng.cache, 0), np.int32, spec.V)) == g["QF_sha256"]
        assert sha(K.read_dev(L.GPUCache_GetQT(eng.cache, 0), np.int32, spec.V)) == g["QT_sha256"]
        got = (L.GPUCache_NodeCapacity(eng.cache, 0), L.GPUCache_EdgeCapacity(eng.cache, 0), round(L.GPUCache_Alpha(eng.cache, 0) * 100))
        assert got == (plan["node_capacity"], plan["edge_capacity"], plan["alpha_idx"]), (plan["budget_frac"], got)
        eng.close()
