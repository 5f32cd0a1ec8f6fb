"""Parity tests proper: the HIP path, called through the C ABI (ctypes over liblegion_amd.so),
against the CPU oracle on the same seeded inputs -- bit exact for every id / index / counter and
for the (verbatim copied) f32 feature rows.  Run on the GPU box with `pytest -m gpu`."""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest

from conftest import KEYS_NO_FEATURES, ROOT, assert_batch_equal, load_golden, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import legion1_amd.capi as K
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    L.SetGPUDevice(0)
    return K


def make_engine(K, ds_or_arrays, B, fan, G=1, seeds=None, **kw):
    if hasattr(ds_or_arrays, "spec"):
        ds = ds_or_arrays
        V, F, indptr, indices, feats = ds.spec.V, ds.spec.F, ds.indptr, ds.indices, ds.features
        if seeds is None:
            import oracle as O
            parts = O.split_seeds(ds.train, G)
            seeds = dict(train=[(p, ds.labels[p]) for p in parts])
    else:
        V, F, indptr, indices, feats = ds_or_arrays
    eng = K.Engine(indptr, indices, feats, V, F, seeds, B, fan, G=G, **kw)
    eng.alloc_features()
    return eng


# ---------------------------------------------------------------------------------------------------
def test_rng_on_gpu_matches_thrust_vectors(K):
    kat = load_golden("rng_kat")
    rows = np.array(kat["rows"], dtype=np.int64)
    idx, deg = rows[:, 0].astype(np.int32), rows[:, 1].astype(np.int32)
    di, dd, dk = K.DevBuf.from_numpy(idx), K.DevBuf.from_numpy(deg), K.DevBuf(idx.nbytes)
    K.lib().legion_rng_probe(None, di.ptr, dd.ptr, dk.ptr, len(idx))
    K.lib().d_stream_sync(None)
    K.check()
    assert np.array_equal(dk.to_numpy(np.int32, len(idx)), rows[:, 2].astype(np.int32))
    for b in (di, dd, dk):
        b.free()


def test_toy_golden(K, oracle):
    g = load_golden("toy_batches")
    V, F = g["V"], g["F"]
    indptr, indices = np.array(g["indptr"], np.int64), np.array(g["indices"], np.int32)
    feats = np.array(g["features"], np.float32).reshape(V, F)
    labels, seeds = np.array(g["labels"], np.int32), np.array(g["seeds"], np.int32)
    for case in g["cases"]:
        eng = make_engine(K, (V, F, indptr, indices, feats), case["batch"], case["fanout"],
                          seeds=dict(train=[(seeds, labels[seeds])]))
        eng.run_batch(0, case["counter"])
        got = eng.result(0)
        eng.close()
        for k in ("nc", "ec", "ids", "labels", "src_off", "dst_off"):
            assert got[k].tolist() == case[k], (case["fanout"], k)
        assert sha(got["features"]) == case["features_sha256"]


def test_medium_digests(K, synth):
    g = load_golden("medium_digests")
    ds = synth.generate(synth.spec_for("products", scale=0.04))
    engines = {}
    for case in g["cases"]:
        key = (case["batch"], tuple(case["fanout"]))
        if key not in engines:
            engines[key] = make_engine(K, ds, case["batch"], case["fanout"])
        eng = engines[key]
        eng.run_batch(0, case["counter"])
        got = eng.result(0)
        assert got["nc"].tolist() == case["nc"] and got["ec"].tolist() == case["ec"], key
        for k in ("ids", "labels", "src_off", "dst_off", "features"):
            assert sha(got[k]) == case[k + "_sha256"], (key, k)
    for e in engines.values():
        e.close()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_graphs_with_holes(K, oracle, seed):
    """degree-0 rows, -1 neighbour entries, degree < fan-out, short last batch, batch of one."""
    rng = np.random.RandomState(seed)
    V, F = 500, 7 + seed           # odd F -> scalar gather path; F=8 (seed 1) -> float4 path
    deg = rng.randint(0, 12, size=V)
    deg[rng.randint(0, V, 5)] = 300
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(-1, V, size=int(indptr[-1])).astype(np.int32)
    feats = rng.rand(V, F).astype(np.float32)
    labels = rng.randint(0, 9, size=V).astype(np.int32)
    seeds = rng.permutation(V)[:203].astype(np.int32)
    for fan, B in (([3, 2], 50), ([5, 4, 3], 64), ([25, 10], 203), ([1, 1, 1, 1], 7), ([2], 1)):
        orc = oracle.OracleRunner(indptr, indices, feats, V, F, B, fan)
        eng = make_engine(K, (V, F, indptr, indices, feats), B, fan, seeds=dict(train=[(seeds, labels[seeds])]))
        for counter in range((len(seeds) + B - 1) // B):
            ref = orc.run_batch(seeds, labels[seeds], counter)
            eng.run_batch(0, counter)
            assert_batch_equal(ref, eng.result(0))
        eng.close()


def test_modes_and_padding(K, oracle, small_ds):
    """valid / test seed sets, the size*counter offset quirk of a short last batch (Kernels.cu:224-227)."""
    ds = small_ds
    V, F = ds.spec.V, ds.spec.F
    seeds = dict(train=[(ds.train, ds.labels[ds.train])], valid=[(ds.valid, ds.labels[ds.valid])],
                 test=[(ds.test[:1000], ds.labels[ds.test[:1000]])])
    B, fan = 300, [10, 5]
    eng = make_engine(K, ds, B, fan, seeds=seeds)
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan)
    for mode, (ids, labs) in ((0, seeds["train"][0]), (1, seeds["valid"][0]), (2, seeds["test"][0])):
        n_batches = (len(ids) + B - 1) // B
        for counter in (0, n_batches - 1):
            ref = orc.run_batch(ids, labs, counter, mode=mode)
            eng.run_batch(0, counter, mode=mode)
            assert_batch_equal(ref, eng.result(0))
    eng.close()


def test_host_pinned_tables_equal_device_tables(K, oracle, small_ds):
    """CSR + features in pinned host memory read through the GPU's host mapping (the reference's UVA
    configuration, GPUGraphStore.cu:264-265,315) give the same bytes as HBM-resident tables."""
    ds = small_ds
    B, fan = 500, [25, 10]
    ref = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan).run_batch(ds.train, ds.labels[ds.train], 0)
    for loc in (K.LOC_HOST_PINNED, K.LOC_HOST_PAGEABLE, K.LOC_DEVICE):
        eng = make_engine(K, ds, B, fan, csr_location=loc, features_location=loc)
        eng.run_batch(0, 0)
        assert_batch_equal(ref, eng.result(0))
        eng.close()


def test_per_level_and_single_gather_agree(K, oracle, small_ds):
    ds = small_ds
    B, fan = 400, [10, 5, 3]
    ref = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan).run_batch(ds.train, ds.labels[ds.train], 2)
    eng = make_engine(K, ds, B, fan)
    for per_level in (True, False):
        eng.run_batch(0, 2, per_level=per_level)
        assert_batch_equal(ref, eng.result(0))
    eng.close()


def test_back_to_back_batches_without_planner(K, oracle, small_ds):
    """The position table is wiped lazily when the planner op did not run; two pipes alternate."""
    ds = small_ds
    B, fan = 200, [10, 5]
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan, pipeline_depth=2)
    for it in range(5):
        ref = orc.run_batch(ds.train, ds.labels[ds.train], it)
        eng.run_batch(0, it, plan=(it % 2 == 0), pipe=it % 2)
        assert_batch_equal(ref, eng.result(0, pipe=it % 2))
    eng.close()


def test_batch_graph_replay_matches_oracle(K, oracle, small_ds):
    """One mini-batch recorded as a hipGraph per pipe (device-resident batch cursor + table epoch) and replayed:
    consecutive batches, a jump in the cursor, a host-driven batch in between, the short last batch of the seed
    list (the size clamp of Kernels.cu:224 computed on the device) and the -1 padded batch after it."""
    ds = small_ds
    B, fan = 200, [10, 5]
    L = K.lib()
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan, pipeline_depth=2)
    L.GPUCache_SetPreSc(eng.cache, 0)
    graphs = [eng.capture_batch(0, pipe=q) for q in (0, 1)]
    last = (len(ds.train) - 1) // B               # B * (last + 1) >= n_train: short batch
    assert 0 < len(ds.train) - B * last <= B
    plan = [("g", 0), ("g", 1), ("g", 2), ("h", 3), ("g", 4), ("g", 7), ("g", 8), ("g", last), ("g", 0), ("h", 1), ("g", 2)]
    serial0 = L.GPUMemoryPool_GetBatchSerial(eng.pools[0])
    for n, (how, it) in enumerate(plan):
        q = n % 2
        ref = orc.run_batch(ds.train, ds.labels[ds.train], it)
        if how == "g":
            eng.run_graph(graphs[q], it)
        else:
            eng.run_batch(0, it, pipe=q, stream=graphs[q][1])
        assert_batch_equal(ref, eng.result(0, pipe=q))
    assert L.GPUMemoryPool_GetBatchSerial(eng.pools[0]) == serial0 + len(plan)   # one table epoch per batch, either way
    eng.close()


def test_batch_graph_one_gather_and_three_hops(K, oracle, small_ds):
    """Graph of the bench's schedule (3 hops, one gather over all levels, no planner), back to back without syncs."""
    ds = small_ds
    B, fan = 300, [10, 5, 3]
    L = K.lib()
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan, pipeline_depth=2)
    L.GPUCache_SetPreSc(eng.cache, 0)
    st = L.d_stream_create()
    graphs = [eng.capture_batch(0, pipe=q, per_level=False, plan=False, stream=st) for q in (0, 1)]
    for it in range(0, 6, 2):                      # two batches in flight on one stream, then compare both pipes
        eng.run_graph(graphs[0], it, sync=False)
        eng.run_graph(graphs[1], it + 1, sync=True)
        for q in (0, 1):
            ref = orc.run_batch(ds.train, ds.labels[ds.train], it + q)
            assert_batch_equal(ref, eng.result(0, pipe=q))
    eng.close()
    L.d_stream_destroy(st)


def test_operator_plugin_api(K, oracle, small_ds):
    """The reference's Operator objects (Operator.h:4-27) driven the way GPURunner::RunOnce does."""
    ds = small_ds
    B, fan = 300, [10, 5]
    L = K.lib()
    eng = make_engine(K, ds, B, fan)
    info = eng.info
    env = L.NewIPCEnv(1)
    ns_steps = L.IPCEnv_Coordinate(env, C.byref(info))
    ops = [L.NewBatchGenerator(0), L.NewFeatureExtractor(1), L.NewRandomSampler(2), L.NewFeatureExtractor(3),
           L.NewRandomSampler(4), L.NewFeatureExtractor(5), L.NewCachePlanner(6), L.NewCacheUpdater(7)]
    L.GPUCache_SetPreSc(eng.cache, 0)
    params = []
    for i in range(8):
        p = K.OpParams()
        p.device_id, p.stream, p.event = 0, None, None
        p.memorypool, p.cache, p.graph, p.noder, p.env = eng.pools[0], eng.cache, eng.graph, eng.noder, env
        p.neighbor_count = fan[(i - 2) // 2] if i in (2, 4) else 0
        p.is_presc, p.in_memory = 0, 1
        params.append(p)
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    for it in (0, 3):
        L.GPUMemoryPool_SetCurrentMode(eng.pools[0], 0)
        L.GPUMemoryPool_SetIter(eng.pools[0], it)
        for op, p in zip(ops, params):
            L.Operator_run(op, C.byref(p))
        L.d_stream_sync(None)
        K.check()
        assert_batch_equal(orc.run_batch(ds.train, ds.labels[ds.train], it), eng.result(0))
    for op in ops:
        L.Operator_Delete(op)
    L.IPCEnv_Finalize(env)
    eng.close()


def test_degree_boundary_rows_parity(K, oracle, synth):
    """Rows at every degree the sampler distinguishes -- 0, 1, below / at / above the fan-out, one 64-byte line of neighbours exactly,
    one more, long rows -- and -1 neighbours (the reference skips them, Kernels.cu:411): bit-identical to the oracle."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    V = spec.V
    rs = np.random.RandomState(11)
    deg = rs.choice([0, 1, 2, 7, 14, 15, 16, 17, 30, 31, 32, 33, 200], size=V, p=[.04, .08, .1, .2, .1, .1, .1, .08, .05, .05, .04, .03, .03]).astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = rs.randint(0, V, size=int(indptr[-1])).astype(np.int32)
    indices[rs.rand(len(indices)) < 0.002] = -1
    B, fan = 300, [10, 5, 3]
    seeds = dict(train=[(ds.train, ds.labels[ds.train])])
    eng = K.Engine(indptr, indices, ds.features, V, spec.F, seeds, B, fan)
    eng.alloc_features()
    orc = oracle.OracleRunner(indptr, indices, ds.features, V, spec.F, B, fan)
    for it in (0, 2):
        ref = orc.run_batch(ds.train, ds.labels[ds.train], it)
        eng.run_batch(0, it)
        assert_batch_equal(ref, eng.result(0))
    eng.close()


@pytest.mark.parametrize("F", [100, 36, 7, 52])
def test_padded_row_pitch_gathers_the_same_bytes(K, oracle, synth, F):
    """an HBM feature table whose rows are padded to a 128-byte-aligned pitch (legion_row_pitch; F = 100 ->
    128 floats) must deliver exactly the dense rows -- the trainer-facing buffer stays [n, F].  F = 7: scalar path with a pitch;
    F = 36 / 52: float4 path, C = 9 / 13 lanes per row."""
    L = K.lib()
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    V = spec.V
    rs = np.random.RandomState(F)
    feats = rs.standard_normal((V, F)).astype(np.float32)
    pitch = L.legion_row_pitch(F)
    assert pitch >= F and (pitch * 4) % 128 == 0 and (pitch == F) == ((F * 4) % 128 == 0)
    padded = np.full((V, pitch), np.float32(-777.0))       # poison in the pad floats: must never reach the output
    padded[:, :F] = feats
    B, fan = 300, [10, 5]
    ref = oracle.OracleRunner(ds.indptr, ds.indices, feats, V, F, B, fan).run_batch(ds.train, ds.labels[ds.train], 1)
    seeds = dict(train=[(ds.train, ds.labels[ds.train])])
    eng = K.Engine(ds.indptr, ds.indices, padded.reshape(-1), V, F, seeds, B, fan, features_pitch=pitch)
    eng.alloc_features()
    for per_level in (True, False):
        eng.run_batch(0, 1, per_level=per_level)
        assert_batch_equal(ref, eng.result(0))
    eng.close()


def test_shard_geometry_travels_with_the_handles_and_a_mismatch_is_refused(K, small_ds):
    """every process derives the shard pitch / chunk geometry from its own alpha and capacity, and an importer
    addresses a peer's rows with ITS OWN.  The geometry travels next to the exported handles (Engine.export_shards) and
    GPUCache_CheckShardGeometry refuses a shard that was built differently -- never a silent read at the wrong stride."""
    import ctypes as C
    ds = small_ds
    L = K.lib()
    eng = make_engine(K, ds, 300, [10, 5], G=2, train_step=2)
    for g in range(2):
        for it in range(2):
            eng.run_batch(g, it, is_presc=True)
    cap = ds.spec.V // 8
    eng.build_cache(cache_agg_mode=1, node_capacity=cap, edge_capacity=0, train_step=2)
    g4 = (C.c_int32 * 4)()
    assert L.GPUCache_ShardGeometry(eng.cache, 0, g4) == 0
    pitch, rows_per_chunk, nchunks, rows = list(g4)
    assert pitch == L.GPUCache_ShardPitch(eng.cache) == L.legion_row_pitch(ds.spec.F) and rows == cap
    assert rows_per_chunk == L.GPUCache_ShardChunkRows(eng.cache, 0) and nchunks == L.GPUCache_ShardChunkCount(eng.cache, 0)
    assert eng.export_shards(0)[4] == (pitch, rows_per_chunk, nchunks, rows) == eng.export_shards(1)[4]
    assert L.GPUCache_CheckShardGeometry(eng.cache, 1, g4) == 0 and not L.legion_last_error()
    for k, bad in ((0, ds.spec.F), (1, rows_per_chunk // 2), (2, nchunks + 1), (3, rows + 1)):
        wrong = (C.c_int32 * 4)(*g4)
        wrong[k] = bad
        assert L.GPUCache_CheckShardGeometry(eng.cache, 1, wrong) == -1
        msg = L.legion_last_error().decode()
        assert "refusing to import" in msg and "wrong stride" in msg and str(bad) in msg, msg
        L.legion_clear_error()
        with pytest.raises(RuntimeError, match="refusing to import"):      # ... and Engine.import_shards stops before it opens a handle
            eng.import_shards(1, ([b"\0" * 64], None, None, (0, 0), tuple(wrong)))
    eng.close()


def test_launchers_refuse_bad_arguments_and_stay_usable(K, oracle, small_ds):
    """Argument errors of the boundary (sticky string, nothing launched, no exit) -- and the engine still produces the oracle's
    batch afterwards: a refusal must not leave half-updated host-side bounds behind."""
    ds = small_ds
    B, fan = 300, [10, 5]
    L = K.lib()
    eng = make_engine(K, ds, B, fan)
    pool = eng.pools[0]
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)

    def refused(text, fn):
        L.legion_clear_error()
        fn()
        msg = (L.legion_last_error() or b"").decode()
        assert text in msg, (text, msg)
        L.legion_clear_error()

    def good_batch(it):
        eng.run_batch(0, it)
        assert_batch_equal(orc.run_batch(ds.train, ds.labels[ds.train], it), eng.result(0))

    good_batch(0)
    refused("batch larger than the pool", lambda: L.batch_generator_kernel(None, eng.noder, eng.cache, pool, B + 1, 0, 0, 0, K.TRAINMODE))
    good_batch(1)
    L.batch_generator_kernel(None, eng.noder, eng.cache, pool, B, 2, 0, 0, K.TRAINMODE)
    refused("op_id must be 2,4", lambda: L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, fan[0], 3, 0))
    refused("op_id must be 2,4", lambda: L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, fan[0], 2 * len(fan) + 2, 0))
    refused("fan-out exceeds what the pool was sized for", lambda: L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, 10 ** 6, 2, 0))
    refused("op_id must be 1,3", lambda: L.get_feature_kernel(None, eng.cache, eng.noder, pool, 0, 2, 1))
    refused("op_id must be 1,3", lambda: L.get_feature_kernel(None, eng.cache, eng.noder, pool, 0, 2 * len(fan) + 3, 1))
    refused("needs a filled unified cache", lambda: L.legion_exchange_local(None, eng.cache, eng.noder, pool, 0))
    L.d_stream_sync(None)
    good_batch(2)
    good_batch(0)
    # a pool without scratch refuses every launcher
    bare = L.NewGPUMemoryPool(1)
    refused("AllocateScratch was not called", lambda: L.batch_generator_kernel(None, eng.noder, eng.cache, bare, B, 0, 0, 0, K.TRAINMODE))
    refused("AllocateScratch was not called", lambda: L.GPU_Random_Sampling(None, eng.graph, eng.cache, bare, fan[0], 2, 0))
    refused("AllocateScratch was not called", lambda: L.get_feature_kernel_all(None, eng.cache, eng.noder, bare, 0, 1))
    L.GPUMemoryPool_Delete(bare)
    eng.close()


def test_runner_posts_a_poisoned_pipe_when_an_operator_refuses(K, small_ds):
    """LEGION_ERR_RETURN (the tests' mode): a batch an operator refused must not leave a consumer blocked on sem_w --
    Runner_RunOnce posts the pipe with every node-counter word = -1 and keeps the error.  Here the
    feature buffers were never registered (Runner_InitializeFeaturesBuffer skipped), so every FeatureExtractor refuses."""
    import ctypes.util
    ds = small_ds
    B, fan = 300, [10, 5]
    L = K.lib()
    ns = "lgn_t_poison_%d_" % os.getpid()
    L.legion_ipc_set_namespace(ns.encode())
    eng = make_engine(K, ds, B, fan)
    env = L.NewIPCEnv(1)
    L.IPCEnv_Coordinate(env, C.byref(eng.info))
    fan_arr = np.asarray(fan, dtype=np.int32)
    rp = K.RunnerParams()
    rp.device_id, rp.fanout, rp.hops = 0, fan_arr.ctypes.data, len(fan)
    rp.cache, rp.graph, rp.noder, rp.env, rp.global_batch_id, rp.in_memory = eng.cache, eng.graph, eng.noder, env, 0, 1
    runner = L.NewGPURunner()
    L.Runner_Initialize(runner, C.byref(rp))
    L.GPUCache_SetPreSc(eng.cache, 0)
    K.check()
    libc = C.CDLL(ctypes.util.find_library("c") or "libc.so.6", use_errno=True)
    libc.sem_open.restype = C.c_void_p
    libc.sem_open.argtypes = [C.c_char_p, C.c_int]
    libc.sem_post.argtypes = libc.sem_trywait.argtypes = libc.sem_close.argtypes = [C.c_void_p]
    sem_r = libc.sem_open(("/%ssem_r_0_0" % ns).encode(), 0)
    sem_w = libc.sem_open(("/%ssem_w_0_0" % ns).encode(), 0)
    assert sem_r and sem_w
    libc.sem_post(sem_r)                       # the trainer's "pipe 0 is free"
    L.Runner_RunOnce(runner, C.byref(rp))      # must return, not block and not exit
    msg = L.legion_last_error()
    assert msg and b"feature buffer of the current pipe is not set" in msg, msg
    assert libc.sem_trywait(sem_w) == 0, "the failed pipe was not posted: a consumer would block forever"
    nc = K.read_dev(L.IPCEnv_GetNodeCounter(env, 0, 0), np.int32, 16)
    assert (nc == -1).all(), nc
    L.legion_clear_error()
    libc.sem_close(sem_r)
    libc.sem_close(sem_w)
    L.Runner_Delete(runner)
    L.IPCEnv_Finalize(env)
    eng.close()
    L.legion_ipc_set_namespace(b"")


def test_poisoned_pipe_reaches_the_trainer_as_an_error(K, small_ds, tmp_path):
    """the only shipped consumer must act on the poisoned pipe.  A real `ipc_service` trainer process attaches to
    an in-process runner; the runner's next batch fails (its feature buffer is taken away after the hand-shake), the pipe is
    posted with nc[*] = -1 and zeroed edge counters, and the trainer's get_next raises "sampling server failed" instead of
    building tensors with a negative dimension or the previous batch's edge counts."""
    import subprocess
    ds = small_ds
    B, fan = 300, [10, 5]
    L = K.lib()
    ns = "lgn_t_poison2_%d_" % os.getpid()
    L.legion_ipc_set_namespace(ns.encode())
    eng = make_engine(K, ds, B, fan)
    env = L.NewIPCEnv(1)
    L.IPCEnv_Coordinate(env, C.byref(eng.info))
    fan_arr = np.asarray(fan, dtype=np.int32)
    rp = K.RunnerParams()
    rp.device_id, rp.fanout, rp.hops = 0, fan_arr.ctypes.data, len(fan)
    rp.cache, rp.graph, rp.noder, rp.env, rp.global_batch_id, rp.in_memory = eng.cache, eng.graph, eng.noder, env, 0, 1
    runner = L.NewGPURunner()
    L.Runner_Initialize(runner, C.byref(rp))
    L.Runner_InitializeFeaturesBuffer(runner, C.byref(rp))
    L.GPUCache_SetPreSc(eng.cache, 0)
    K.check()
    log = open(str(tmp_path / "client.log"), "w+")
    client = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_client_poison.py"), str(ds.spec.F)], stdout=log, stderr=subprocess.STDOUT,
                              env=dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        t0 = time.time()
        while "ATTACHED" not in open(log.name).read():
            assert client.poll() is None and time.time() - t0 < 240, open(log.name).read()[-3000:]
            time.sleep(0.1)
        # a good batch first: the pipe's edge counters now hold real counts a poisoned post must not leave behind
        L.Runner_RunOnce(runner, C.byref(rp))
        K.check()
        pool = L.Runner_GetMemoryPool(runner)
        for q in range(2):
            L.GPUMemoryPool_SetFloatFeatures(pool, None, q)
        rp.global_batch_id = 1
        L.Runner_RunOnce(runner, C.byref(rp))      # fails: posts batch 0 (in flight) and then the poisoned pipe
        assert b"feature buffer of the current pipe is not set" in (L.legion_last_error() or b"")
        L.legion_clear_error()
        ec = K.read_dev(L.IPCEnv_GetEdgeCounter(env, 0, 1), np.int32, 16)
        assert (ec == 0).all(), ec
        # the trainer consumes batch 0 (good) and raises on the poisoned pipe: exit code 0 only for exactly that
        rc = client.wait(timeout=120)
    except BaseException:
        client.kill()
        raise
    text = open(log.name).read()
    L.Runner_Delete(runner)
    L.IPCEnv_Finalize(env)
    eng.close()
    L.legion_ipc_set_namespace(b"")
    assert rc == 0 and "RAISED after 1 good batches" in text and "sampling server failed" in text, text[-3000:]


def test_a_batch_larger_than_the_feature_buffer_is_refused_by_the_trainer_and_named_by_the_server(K, small_ds, tmp_path, capfd):
    """VERDICT r05 next 4.  The feature buffers hold a bounded number of rows (1.2 x the largest batch of the pre-sampling epoch,
    Server.cu:275); the reference's trainer views [nc9, F] of them unchecked (ipc_cuda_kernel.cu:200: a read past the allocation).  Here
    (a) `ipc_service.get_next` must RAISE for a batch with more nodes than rows -- a real trainer process attached to an in-process runner
    whose published row capacity is shrunk after the first batch -- and (b) the server must not drop rows silently: the runner names the
    first short batch in its log and counts them all (Runner_ShortBatches)."""
    import subprocess
    ds = small_ds
    B, fan = 300, [10, 5]
    L = K.lib()
    ns = "lgn_t_short_%d_" % os.getpid()
    L.legion_ipc_set_namespace(ns.encode())
    eng = make_engine(K, ds, B, fan)
    env = L.NewIPCEnv(1)
    L.IPCEnv_Coordinate(env, C.byref(eng.info))
    fan_arr = np.asarray(fan, dtype=np.int32)
    rp = K.RunnerParams()
    rp.device_id, rp.fanout, rp.hops = 0, fan_arr.ctypes.data, len(fan)
    rp.cache, rp.graph, rp.noder, rp.env, rp.global_batch_id, rp.in_memory = eng.cache, eng.graph, eng.noder, env, 0, 1
    runner = L.NewGPURunner()
    L.Runner_Initialize(runner, C.byref(rp))
    L.Runner_InitializeFeaturesBuffer(runner, C.byref(rp))
    L.GPUCache_SetPreSc(eng.cache, 0)
    K.check()
    small_rows = 100
    log = open(str(tmp_path / "client.log"), "w+")
    client = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_client_poison.py"), str(ds.spec.F), "feature buffer holds %d rows" % small_rows],
                              stdout=log, stderr=subprocess.STDOUT, env=dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        t0 = time.time()
        while "ATTACHED" not in open(log.name).read():
            assert client.poll() is None and time.time() - t0 < 240, open(log.name).read()[-3000:]
            time.sleep(0.1)
        L.Runner_RunOnce(runner, C.byref(rp))          # batch 0 queued
        rp.global_batch_id = 1
        L.Runner_RunOnce(runner, C.byref(rp))          # batch 1 queued, batch 0 handed over: complete, the trainer takes it
        K.check()
        while "BATCH" not in open(log.name).read():
            assert client.poll() is None and time.time() - t0 < 240, open(log.name).read()[-3000:]
            time.sleep(0.05)
        assert L.Runner_ShortBatches(runner) == 0
        # from here on the buffers "hold" 100 rows: the gather of batch 2 stops there, and batch 1 (thousands of nodes) no longer fits
        pool = L.Runner_GetMemoryPool(runner)
        L.GPUMemoryPool_SetFeatureRows(pool, small_rows)
        L.IPCEnv_SetFeatureRows(env, 0, small_rows)
        rp.global_batch_id = 2
        L.Runner_RunOnce(runner, C.byref(rp))          # hands batch 1 over
        K.check()
        rc = client.wait(timeout=120)
    except BaseException:
        client.kill()
        raise
    text = open(log.name).read()
    short = L.Runner_ShortBatches(runner)
    L.d_stream_sync(None)
    L.Runner_Delete(runner)
    L.IPCEnv_Finalize(env)
    eng.close()
    L.legion_ipc_set_namespace(b"")
    assert rc == 0 and "RAISED after 1 good batches" in text and "feature buffer holds %d rows" % small_rows in text, text[-3000:]
    assert short == 1, short
    server_said = capfd.readouterr().out
    assert "Feature buffer too small: a batch has" in server_said and "the buffer holds %d rows" % small_rows in server_said, server_said[-1500:]


# ---------------------------------------------------------------------------------------------------
# cache: pre-sampling, ranking, cost model, fill-up, unified cache with Kg logical GPUs on one device
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("G,mode,chunk_bytes,pcm", [(1, 0, None, True), (2, 1, None, True), (4, 2, 65536, True), (4, 1, 40000, True), (8, 3, None, True),
                                                      (8, 2, 50000, True), (1, 0, None, False), (4, 1, None, False)])
def test_presampling_cache_pipeline(K, oracle, small_ds, G, mode, chunk_bytes, pcm, monkeypatch):
    # chunk_bytes: force the cache shards to be split into several chunk allocations (default chunk: 1 GiB)
    # pcm=False: CostModel without the two Intel-PCM counters (what the `legion` server binary does: runner.cpp passes NULL) --
    # the PCM-free transaction estimate of SURVEY section 5, computed on the device from AT / QT / indptr, against the oracle's
    if chunk_bytes is not None:
        monkeypatch.setenv("LEGION_SHARD_CHUNK_BYTES", str(chunk_bytes))
    else:
        monkeypatch.delenv("LEGION_SHARD_CHUNK_BYTES", raising=False)
    ds = small_ds
    V, F = ds.spec.V, ds.spec.F
    L = K.lib()
    B, fan = (250 if G < 8 else 100), [10, 5]      # 8 partitions of the 1966 seeds hold 245 each
    Kg = {0: 1, 1: 2, 2: 4, 3: 8}[mode]
    parts = oracle.split_seeds(ds.train, G)
    steps = min((len(p) - 1) // B for p in parts)
    counters = [500000, 250000] if pcm else None   # stand-ins for the two Intel-PCM PCIe counters (Server.cu:100)
    budget = int(V * F * 4 * 0.15)
    eng = make_engine(K, ds, B, fan, G=G, cache_memory=budget, train_step=steps)
    orcs = [oracle.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan, partition_count=G) for _ in range(G)]
    max_ids = []
    for g in range(G):      # pre-sampling epoch (Server.cu:83-95): even ops only, hotness collected
        seen = 0
        for it in range(steps):
            eng.run_batch(g, it, is_presc=True)
            ref = orcs[g].run_batch(parts[g], ds.labels[parts[g]], it, is_presc=True)
            assert_batch_equal(ref, eng.result(g, with_features=False), keys=KEYS_NO_FEATURES)    # pre-sampling gathers nothing
            seen = max(seen, int(ref["nc"][5 + 2 * len(fan)]))
        L.SetGPUDevice(g)
        assert np.array_equal(K.read_dev(L.GPUCache_GetNodeAccessedMap(eng.cache, g), np.uint64, V), orcs[g].node_access_time)
        assert np.array_equal(K.read_dev(L.GPUCache_GetEdgeAccessedMap(eng.cache, g), np.uint64, V), orcs[g].edge_access_time)
        assert L.GPUCache_MaxIdNum(eng.cache, g) == seen      # max_ids_ (GPUCache.cu:294-296)
        max_ids.append(seen)
    # ranking + cost model with explicit counters + fill-up
    eng.build_cache(cache_agg_mode=mode, counters=counters, train_step=steps)
    assert L.GPUCache_Kg(eng.cache) == Kg and L.GPUCache_Kc(eng.cache) == G // Kg
    pinned_by_indexing = []
    for Ki in range(G // Kg):
        members = list(range(Ki * Kg, (Ki + 1) * Kg))
        AF, QF = oracle.candidate_selection([orcs[m].node_access_time for m in members], V)
        AT, QT = oracle.candidate_selection([orcs[m].edge_access_time for m in members], V)
        L.SetGPUDevice(Ki * Kg)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQF(eng.cache, Ki), np.int32, V), QF)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQT(eng.cache, Ki), np.int32, V), QT)
        # the reference feeds MaxIdNum of GPUs 0..Kg-1 (the FIRST clique) to every clique (GPUCache.cu:677-680); the
        # G=4/Kg=2 and G=8/Kg=4 cases have two cliques with different max_ids, so this pins the indexing
        cm = oracle.cost_model(AF, AT, QT, ds.indptr, V, F, budget, Kg, counters, [max_ids[j] for j in range(Kg)], steps)
        if Ki > 0 and [max_ids[m] for m in members] != [max_ids[j] for j in range(Kg)]:
            own = oracle.cost_model(AF, AT, QT, ds.indptr, V, F, budget, Kg, counters, [max_ids[m] for m in members], steps)
            pinned_by_indexing.append((own["node_capacity"], own["edge_capacity"]) != (cm["node_capacity"], cm["edge_capacity"]))
        assert L.GPUCache_NodeCapacity(eng.cache, Ki * Kg) == cm["node_capacity"]
        assert L.GPUCache_EdgeCapacity(eng.cache, Ki * Kg) == cm["edge_capacity"]
        assert abs(L.GPUCache_Alpha(eng.cache, Ki) - cm["alpha_idx"] * 0.01) < 1e-9
        for m in members:
            orcs[m].set_feature_cache(QF, cm["node_capacity"], Kg)
            orcs[m].set_topo_cache(QT, cm["edge_capacity"], Kg, Ki)
        # FindFeat / FindTopo against the oracle's maps, cache rows and CSR fragments
        probe = np.concatenate([QF[:50], QT[:50], np.arange(0, V, 97, dtype=np.int32), np.array([-1], np.int32)])
        for m in members:
            L.SetGPUDevice(m)
            d_in = K.DevBuf.from_numpy(probe)
            d_nc = K.DevBuf.from_numpy(np.array([0, 0, 0, 0, len(probe)] + [0] * 11, np.int32))
            d_out, d_pi, d_po = K.DevBuf(len(probe) * 4), K.DevBuf(len(probe)), K.DevBuf(len(probe) * 4)
            L.GPUCache_FindFeat(eng.cache, d_in.ptr, d_out.ptr, d_nc.ptr, 1, None, m)
            L.GPUCache_FindTopo(eng.cache, d_in.ptr, d_pi.ptr, d_po.ptr, len(probe), 2, None, m)
            L.d_stream_sync(None)
            safe = np.where(probe >= 0, probe, 0)
            assert np.array_equal(d_out.to_numpy(np.int32, len(probe)), np.where(probe >= 0, orcs[m].node_map[safe], -1))
            assert np.array_equal(d_pi.to_numpy(np.int8, len(probe)), np.where(probe >= 0, orcs[m].part_index_map[safe], -1))
            assert np.array_equal(d_po.to_numpy(np.int32, len(probe)), np.where(probe >= 0, orcs[m].part_offset_map[safe], -1))
            j = m - Ki * Kg
            rpc, nch = L.GPUCache_ShardChunkRows(eng.cache, m), L.GPUCache_ShardChunkCount(eng.cache, m)
            assert nch == (cm["node_capacity"] + rpc - 1) // rpc and (nch > 1) == (chunk_bytes is not None)
            # rows of a shard start on a 128-byte line (F = 100 -> 128 floats) only if the padded shard fits the feature share
            # of the budget -- here (15 % of the table, planned in dense rows) it does not: cache_memory is a contract
            pitch = L.GPUCache_ShardPitch(eng.cache)
            assert pitch == L.legion_shard_pitch(F, cm["node_capacity"], int((1.0 - cm["alpha_idx"] * 0.01) * budget)) or G // Kg > 1
            assert pitch in (F, L.legion_row_pitch(F)) and cm["node_capacity"] * pitch * 4 <= budget
            cache_rows = np.concatenate([K.read_dev(L.GPUCache_GetShardChunk(eng.cache, m, q), np.float32,
                                                    min(rpc, cm["node_capacity"] - q * rpc) * pitch).reshape(-1, pitch)[:, :F] for q in range(nch)])
            n_valid = len(range(j, min(V, cm["node_capacity"] * Kg), Kg))
            assert np.array_equal(cache_rows[:n_valid], orcs[m].caches[j][:n_valid])
            # CSR fragment, reassembled from its chunk allocations (one chunk unless chunk_bytes is small)
            ecap = cm["edge_capacity"]
            assert L.GPUGraphStorage_FragmentRows(eng.graph, m) == ecap
            rspan, espan = (L.GPUGraphStorage_FragmentChunkSpan(eng.graph, w) for w in (0, 1))
            nip, nix = (L.GPUGraphStorage_FragmentChunkCount(eng.graph, m, w) for w in (0, 1))
            assert nip == (ecap - 1) // rspan + 1
            fi = np.empty(ecap + 1, np.int64)
            for q in range(nip):
                n = min(rspan, ecap - q * rspan) + 1
                fi[q * rspan:q * rspan + n] = K.read_dev(L.GPUGraphStorage_GetFragmentChunk(eng.graph, m, 0, q), np.int64, n)
            assert np.array_equal(fi, orcs[m].frag_indptr[m])
            total = int(fi[-1])
            assert L.GPUGraphStorage_FragmentEdges(eng.graph, m) == total and nix == max(1, (total - 1) // espan + 1)
            assert (nip == nix == 1) if chunk_bytes is None else (nix > 1)
            fx = np.full(total, -7, np.int32)
            for q in range(nix):
                first = np.flatnonzero((fi[:-1] >> int(np.log2(espan))) == q)      # rows that start in chunk q
                if len(first) == 0:
                    continue
                lo, end = int(fi[first[0]]), int(fi[first[-1] + 1])   # [q*espan, lo) belongs to a row of the previous chunk
                fx[lo:end] = K.read_dev(L.GPUGraphStorage_GetFragmentChunk(eng.graph, m, 1, q), np.int32, end - q * espan)[lo - q * espan:]
            assert np.array_equal(fx, orcs[m].frag_indices[m][:total])
            if nip == 1 and nix == 1:
                assert L.GPUGraphStorage_GetFragmentIndex(eng.graph, m, m) == L.GPUGraphStorage_GetFragmentChunk(eng.graph, m, 0, 0)
                assert L.GPUGraphStorage_GetFragmentMatrix(eng.graph, m, m) == L.GPUGraphStorage_GetFragmentChunk(eng.graph, m, 1, 0)
            for b in (d_in, d_nc, d_out, d_pi, d_po):
                b.free()
    # steady state through the unified cache (local + peer shards + backing-table misses)
    monkeypatch.setenv("LEGION_CACHE_HIT_PERIOD", "1")        # the reference samples every 500th batch (GPUCache.cu:414)
    for g in range(G):
        for it in (0, 1):
            for m_mode, ids in ((0, parts[g]),):
                ref = orcs[g].run_batch(ids, ds.labels[ids], it, mode=m_mode)
                eng.run_batch(g, it, mode=m_mode, per_level=(it == 0))
                got = eng.result(g)
                assert_batch_equal(ref, got)
                hit = orcs[g].node_map[got["ids"]] >= 0
                # "Feature Cache Hit" counter of the lookup pass (feature_cache_hit, GPUCache.cu:130-147)
                n_hit, n_rows = C.c_int32(0), C.c_int32(0)
                rate = L.GPUCache_FeatureCacheHitRate(eng.cache, g, C.byref(n_hit), C.byref(n_rows))
                if L.GPUCache_NodeCapacity(eng.cache, g) > 0:
                    assert (n_hit.value, n_rows.value) == (int(hit.sum()), len(hit)) and abs(rate - hit.mean()) < 1e-12
                # hits and misses both occur, unless the clique-wide cache (capacity x Kg) already holds nearly every node
                assert 0 < hit.sum() and (hit.sum() < len(hit) or L.GPUCache_NodeCapacity(eng.cache, g) * Kg > 0.9 * V)
    # the bulk-copy peer path of the one-process server (peer_exchange.cpp, $LEGION_PEER_GATHER=exchange): the peers' rows arrive
    # through hipMemcpyPeerAsync instead of in-kernel loads -- same bytes, one host synchronisation per batch
    if Kg > 1 and all(L.GPUCache_NodeCapacity(eng.cache, g) > 0 for g in range(G)):
        monkeypatch.setenv("LEGION_PEER_GATHER", "exchange")
        for g in range(G):
            for it, per_level in ((0, True), (1, False)):
                ref = orcs[g].run_batch(parts[g], ds.labels[parts[g]], it)
                feat = eng.out[g][0]["feat"]
                L.SetGPUDevice(g)
                L.d_memset_async(feat.ptr, 0xFF, feat.nbytes, None)       # poison: every row must be rewritten
                L.d_stream_sync(None)
                eng.run_batch(g, it, per_level=per_level)
                assert_batch_equal(ref, eng.result(g))
            st = (C.c_int64 * 3)()
            L.legion_peer_exchange_stats(eng.pools[g], st)
            slot = orcs[g].node_map[eng.result(g)["ids"]]
            K0 = (g // Kg) * Kg
            assert st[0] == 2 and st[2] == 2 and st[1] > 0, list(st)       # 2 batches, 2 host syncs, peers were asked for rows
            assert int(((slot >= 0) & (slot // cm_cap(eng, L, g) != g - K0)).sum()) > 0
        monkeypatch.delenv("LEGION_PEER_GATHER")
    eng.close()


def cm_cap(eng, L, g):
    return max(1, L.GPUCache_NodeCapacity(eng.cache, g))


def test_cost_model_everything_fits(K, small_ds):
    """Budget >= all features + all adjacency: cache everything (the reference degenerates here,
    GPUCache.cu:744-751 -- documented extension)."""
    ds = small_ds
    eng = make_engine(K, ds, 100, [5, 5], G=2, cache_memory=1 << 40)
    for g in range(2):
        eng.run_batch(g, 0, is_presc=True)
    eng.build_cache(cache_agg_mode=1)
    L = K.lib()
    assert L.GPUCache_NodeCapacity(eng.cache, 0) == ds.spec.V // 2 + 1 == L.GPUCache_EdgeCapacity(eng.cache, 1)
    eng.close()


# ---------------------------------------------------------------------------------------------------
# generator + full-size runs (BASELINE.json shapes): size-independent properties
# ---------------------------------------------------------------------------------------------------
def test_gpu_generator_matches_numpy(K, synth):
    import torch
    L = K.lib()
    spec = synth.spec_for("papers100M", scale=0.002)
    ds = synth.generate(spec)
    dev = torch.device("cuda", 0)
    deg = torch.empty(spec.V, dtype=torch.int64, device=dev)
    lad = np.ascontiguousarray(spec.ladder, dtype=np.int32)
    L.legion_synth_degrees(None, deg.data_ptr(), 0, spec.V, lad.ctypes.data)
    ind = torch.empty(ds.E, dtype=torch.int32, device=dev)
    L.legion_synth_neighbors(None, ind.data_ptr(), 0, ds.E, spec.V, spec.M, spec.C)
    ft = torch.empty((spec.V, spec.F), dtype=torch.float32, device=dev)
    L.legion_synth_features(None, ft.data_ptr(), 0, spec.V, spec.F)
    lb = torch.empty(spec.V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, lb.data_ptr(), 0, spec.V, spec.classes)
    tr = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, tr.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
    torch.cuda.synchronize()
    K.check()
    assert np.array_equal(deg.cpu().numpy(), np.diff(ds.indptr))
    assert np.array_equal(ind.cpu().numpy(), ds.indices)
    assert np.array_equal(ft.cpu().numpy(), ds.features)
    assert np.array_equal(lb.cpu().numpy(), ds.labels)
    assert np.array_equal(tr.cpu().numpy(), ds.train)


@pytest.mark.parametrize("workload,fan,padded", [("products", [25, 10], True),      # BASELINE config 1's shape (the CPU-sampler baseline's workload)
                                                 ("products", [25, 10, 5], True), ("products", [25, 10, 5], False),
                                                 ("papers100M", [25, 10, 5], False), ("papers100M", [25, 10], False),
                                                 ("uk-union", [25, 10, 5], False)])     # uk-union: E = 5.5e9 > 2^32 edge offsets, F = 256
def test_full_size_properties(K, oracle, synth, workload, fan, padded):
    """BASELINE.json sizes (V up to 133 M, up to 160 GB resident): the size-independent properties every reference execution satisfies
     and, since round 4, the oracle itself on a host copy of the CSR.
    padded: the HBM feature table with legion_row_pitch(F) floats per row (F = 100 -> 128), as bench.py builds it."""
    import torch
    sys_bench = __import__("bench")
    L = K.lib()
    spec = synth.spec_for(workload)
    dev = torch.device("cuda", 0)
    pitch = L.legion_row_pitch(spec.F) if padded else 0
    assert not padded or pitch > spec.F
    indptr, indices, feats, E = sys_bench.build_graph_on_gpu(K, spec, dev, pitch=pitch)
    B, H = 8000, len(fan)
    tr = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, tr.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
    lab = torch.empty(spec.V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, lab.data_ptr(), 0, spec.V, spec.classes)
    torch.cuda.synchronize()
    my_lab = lab[tr.long()].contiguous()
    seeds = dict(train=[((tr.data_ptr(), spec.n_train), (my_lab.data_ptr(), spec.n_train))])
    eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feats.data_ptr(), spec.V, spec.F, seeds, B, fan, E=E, features_pitch=pitch)
    eng.alloc_features()
    # round 4: besides the properties below, every batch word for word against the (OpenMP) oracle on a host copy of the CSR
    orc = oracle.OracleRunner(indptr.cpu().numpy(), indices.cpu().numpy(), None, spec.V, spec.F, B, fan, with_features=False)
    h_tr, h_lab = tr.cpu().numpy(), my_lab.cpu().numpy()
    for it in (0, 7):
        eng.run_batch(0, it)
        res = eng.result(0)
        assert_batch_equal(orc.run_batch(h_tr, h_lab, it, gather=False, omp=True), res, keys=("nc", "ec", "ids", "labels", "src_off", "dst_off"))
        nc, ec, ids = res["nc"], res["ec"], res["ids"]
        levels = [int(nc[4 + 2 * l]) for l in range(H + 1)]
        assert nc[5 + 2 * H] == sum(levels) == len(ids) and levels[0] == B
        assert len(np.unique(ids)) == len(ids) and ids.min() >= 0 and ids.max() < spec.V       # dedup
        assert np.array_equal(ids[:B], synth.seed_ids(spec, it * B, (it + 1) * B))
        assert np.array_equal(res["labels"], synth.labels(spec, ids[:B]))
        src, dst = res["src_off"], res["dst_off"]
        cum_nodes, e0 = np.cumsum(levels), 0
        for h in range(1, H + 1):
            e1 = int(ec[2 + h])
            assert (dst[e0:e1] < cum_nodes[h - 1]).all() and (src[e0:e1] < cum_nodes[h]).all() and (src[e0:e1] >= 0).all()
            if h > 1:   # hop h expands every endpoint of hop h-1 in order: dst offsets are non-decreasing runs
                assert (np.diff(dst[e0:e1].astype(np.int64)) != 0).sum() <= e1 - e0
            e0 = e1
        # first-seen order: a new node's position is increasing with the edge that discovered it
        first_edge = np.full(len(ids), -1, np.int64)
        order = np.arange(len(src) - 1, -1, -1)
        first_edge[src[order]] = order
        new_nodes = np.arange(B, len(ids))
        fe = first_edge[new_nodes]
        assert (fe >= 0).all() and (np.diff(fe) > 0).all()
        # hop-1 edge multiset == closed form (RNG stream depends only on the slot index)
        ip = indptr[torch.from_numpy(ids[:B].astype(np.int64)).to(dev)].cpu().numpy()
        ip1 = indptr[torch.from_numpy(ids[:B].astype(np.int64) + 1).to(dev)].cpu().numpy()
        deg = (ip1 - ip)
        assert int(np.minimum(deg, fan[0]).sum()) == int(ec[3])
        import oracle as O
        e = 0
        for i in list(range(0, B, 997)):
            base = int(np.minimum(deg[:i], fan[0]).sum())
            for j in range(min(int(deg[i]), fan[0])):
                k = O.sample_index(i * fan[0] + j, int(deg[i]))
                want = int(indices[int(ip[i]) + k].item())
                assert ids[src[base + j]] == want and dst[base + j] == i
                e += 1
        assert e > 0
        # every sampled edge is an edge of the graph (sample)
        rs = np.random.RandomState(it)
        for eidx in rs.choice(len(src), size=300, replace=False):
            d_id, s_id = int(ids[dst[eidx]]), int(ids[src[eidx]])
            row = indices[int(indptr[d_id].item()):int(indptr[d_id + 1].item())].cpu().numpy()
            assert s_id in row
        # gathered rows are the table rows, byte for byte (generator closed form)
        rows = rs.choice(len(ids), size=2000, replace=False)
        assert np.array_equal(res["features"][rows], synth.features(spec, ids[rows]))
        assert sha(res["features"][:B]) == sha(synth.features(spec, ids[:B]))
    # idempotence: the same batch index gives the same bytes again
    eng.run_batch(0, 0)
    again = eng.result(0)
    eng.run_batch(0, 0)
    assert_batch_equal(again, eng.result(0))
    eng.close()


def test_link_prediction_seed_batches(K, oracle, synth, small_ds):
    """[src | pos | neg] seed thirds with duplicates inside a batch (lp_sage.py:87-90)."""
    ds = small_ds
    B, fan = 96, [5, 3]
    seeds = synth.lp_trainingset(ds, 300, B, seed=3)
    seeds[5] = seeds[40]
    seeds[B + 7] = seeds[B + 8]
    seeds[2 * B + 1] = seeds[2 * B + 90]
    lab = ds.labels[seeds]
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan, seeds=dict(train=[(seeds, lab)]))
    for counter in range(len(seeds) // B):
        ref = orc.run_batch(seeds, lab, counter)
        eng.run_batch(0, counter)
        assert_batch_equal(ref, eng.result(0))
    eng.close()


def test_five_hops_and_empty_hops(K, oracle):
    """H = 5 fills the int32[16] counter layout to its last slot (nc[15]); a frontier of degree-0 nodes gives
    hops with zero slots (only the counters move)."""
    rng = np.random.RandomState(11)
    V, F = 300, 4
    deg = rng.randint(0, 5, size=V)
    deg[:20] = 0                                   # seeds 0..19: isolated -> every hop of that batch is empty
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(0, V, size=int(indptr[-1])).astype(np.int32)
    feats = rng.rand(V, F).astype(np.float32)
    labels = rng.randint(0, 3, size=V).astype(np.int32)
    seeds = np.concatenate([np.arange(20), rng.permutation(np.arange(20, V))[:40]]).astype(np.int32)
    fan, B = [2, 2, 2, 2, 2], 20
    orc = oracle.OracleRunner(indptr, indices, feats, V, F, B, fan)
    eng = make_engine(K, (V, F, indptr, indices, feats), B, fan, seeds=dict(train=[(seeds, labels[seeds])]))
    for counter in range(3):
        ref = orc.run_batch(seeds, labels[seeds], counter)
        eng.run_batch(0, counter)
        got = eng.result(0)
        assert_batch_equal(ref, got)
        if counter == 0:
            assert got["ec"][7] == 0 and got["nc"][15] == B          # nothing sampled, total == seeds
        else:
            assert got["nc"][15] > B
    eng.close()


def test_position_table_epoch_wraparound(K, oracle, small_ds):
    """The position table is never cleared: entries carry the batch epoch.  Force the epoch counter to its
    end: the pool wipes the table once and starts over, results stay bit exact across the wrap."""
    ds = small_ds
    B, fan = 200, [10, 5]
    L = K.lib()
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan)
    eng.run_batch(0, 0)
    L.GPUMemoryPool_SetBatchSerial(eng.pools[0], 0xFFFFFFF0 - 3)
    serials = []
    for it in range(6):
        ref = orc.run_batch(ds.train, ds.labels[ds.train], it % 3)
        eng.run_batch(0, it % 3)
        assert_batch_equal(ref, eng.result(0))
        serials.append(L.GPUMemoryPool_GetBatchSerial(eng.pools[0]))
    assert 1 in serials and max(serials) < 0xFFFFFFF0       # wrapped exactly once
    eng.close()


def test_overlapped_two_stream_schedule(K, oracle, small_ds):
    """bench.py's --pipeline overlap: sampler of batch i+1 on one stream while the gather of batch i runs on a
    second stream, depth-2 pipes, a pipe reused only after its gather finished.  Every batch must still be
    bit exact (the schedules differ in timing only)."""
    ds = small_ds
    B, fan = 300, [10, 5, 3]
    H = len(fan)
    L = K.lib()
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan, pipeline_depth=2)
    L.GPUCache_SetPreSc(eng.cache, 0)
    pool = eng.pools[0]
    s_samp, s_gath = L.d_stream_create(), L.d_stream_create()
    ev_sampled = [L.d_event_create(), L.d_event_create()]
    ev_gathered = [L.d_event_create(), L.d_event_create()]
    used = [False, False]

    def enqueue(i):
        q = i % 2
        L.GPUMemoryPool_SetCurrentPipe(pool, q)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        if used[q]:
            L.d_stream_wait_event(s_samp, ev_gathered[q])
        L.batch_generator_kernel(s_samp, eng.noder, eng.cache, pool, B, i, 0, 0, K.TRAINMODE)
        for h in range(H):
            L.GPU_Random_Sampling(s_samp, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 0)
        L.d_event_record(ev_sampled[q], s_samp)
        L.d_stream_wait_event(s_gath, ev_sampled[q])
        L.get_feature_kernel_all(s_gath, eng.cache, eng.noder, pool, 0, 1)
        L.d_event_record(ev_gathered[q], s_gath)
        used[q] = True
        L.make_update_plan(s_samp, eng.graph, eng.cache, pool, 0, K.TRAINMODE)

    n = 6
    enqueue(0)
    for i in range(n):
        if i + 1 < n:
            enqueue(i + 1)          # batch i+1 is sampled while batch i is still being gathered
        L.d_stream_sync(s_samp)
        L.d_stream_sync(s_gath)
        K.check()
        assert_batch_equal(orc.run_batch(ds.train, ds.labels[ds.train], i), eng.result(0, pipe=i % 2))
    eng.close()


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("LEGION_STRESS_N", "24")))))
def test_randomised_differential(K, oracle, seed):
    """Random graphs (hubs, isolated nodes, -1 entries, self loops), random fan-outs / batch sizes / hop counts,
    duplicate and repeated seeds, several consecutive batches on one engine: HIP == oracle, bit for bit."""
    rng = np.random.RandomState(1000 + seed)
    V = int(rng.choice([33, 200, 1500, 6000]))
    F = int(rng.choice([1, 3, 4, 8, 20, 64]))
    deg = rng.geometric(0.25, size=V) - 1
    hubs = rng.randint(0, V, size=max(1, V // 100))
    deg[hubs] = rng.randint(50, 400, size=len(hubs))
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    # skewed neighbours: hubs are hit often -> many duplicate claims inside a hop
    nbr = np.where(rng.rand(int(indptr[-1])) < 0.5, rng.choice(hubs, size=int(indptr[-1])), rng.randint(0, V, size=int(indptr[-1])))
    nbr[rng.rand(len(nbr)) < 0.02] = -1
    indices = nbr.astype(np.int32)
    feats = rng.rand(V, F).astype(np.float32)
    labels = rng.randint(0, 7, size=V).astype(np.int32)
    n_seeds = int(rng.randint(5, max(6, min(V, 900))))
    seeds = rng.randint(0, V, size=n_seeds).astype(np.int32)          # with repetitions
    hops = int(rng.randint(1, 5))
    fan = [int(rng.randint(1, 12)) for _ in range(hops)]
    B = int(rng.randint(1, n_seeds + 1))
    orc = oracle.OracleRunner(indptr, indices, feats, V, F, B, fan)
    eng = make_engine(K, (V, F, indptr, indices, feats), B, fan, seeds=dict(train=[(seeds, labels[seeds])]))
    n_batches = (n_seeds + B - 1) // B
    for counter in list(range(min(n_batches, 4))) + [0]:
        ref = orc.run_batch(seeds, labels[seeds], counter)
        eng.run_batch(0, counter, per_level=bool(rng.randint(2)), plan=bool(rng.randint(2)))
        assert_batch_equal(ref, eng.result(0))
    eng.close()


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("LEGION_STRESS_CACHE_N", "16")))))
def test_randomised_clique_cache_differential(K, oracle, seed, monkeypatch):
    """The cached path under random configurations (S5 / S6 / S8 / S9): random graph, feature width (odd, not a multiple of 4, not a
    whole number of 128-byte lines), hop count and fan-outs, clique size Kg in {1, 2, 4, 8} with one or two cliques, forced node / edge
    capacities from 0 to "more than V", shard / fragment chunk sizes, in-kernel or bulk-copy peer gather, FindFeat as a lookup pass
    (every batch hit-sampled: LEGION_CACHE_HIT_PERIOD=1) or fused into the gather (one probe per row and wave) --
    pre-sampling hotness, ranking, id -> slot maps and every steady-state batch of every GPU bit-identical to the oracle."""
    rng = np.random.RandomState(7000 + seed)
    L = K.lib()
    V = int(rng.choice([700, 2500, 9000]))
    F = int(rng.choice([1, 4, 7, 20, 36, 100, 128]))
    deg = rng.geometric(0.12, size=V) - 1
    hubs = rng.randint(0, V, size=max(1, V // 80))
    deg[hubs] = rng.randint(40, 300, size=len(hubs))
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    nbr = np.where(rng.rand(int(indptr[-1])) < 0.4, rng.choice(hubs, size=int(indptr[-1])), rng.randint(0, V, size=int(indptr[-1])))
    nbr[rng.rand(len(nbr)) < 0.01] = -1
    indices = nbr.astype(np.int32)
    feats = rng.standard_normal((V, F)).astype(np.float32)
    labels = rng.randint(0, 9, size=V).astype(np.int32)
    Kg, mode = [(1, 0), (2, 1), (4, 2), (8, 3)][int(rng.randint(4))]
    G = Kg * (2 if (Kg <= 4 and rng.rand() < 0.4) else 1)
    hops = int(rng.randint(1, 4))
    fan = [int(rng.randint(2, 9)) for _ in range(hops)]
    train = rng.permutation(V)[:max(G * 40, V // 3)].astype(np.int32)
    parts = oracle.split_seeds(train, G)
    B = int(rng.randint(8, min(len(p) for p in parts)))
    steps = max(1, min((len(p) - 1) // B for p in parts))
    for name, val in (("LEGION_SHARD_CHUNK_BYTES", rng.choice(["20000", "150000", None])), ("LEGION_PEER_GATHER", rng.choice(["exchange", None])),
                      ("LEGION_CACHE_HIT_PERIOD", ["1", None][seed % 2])):     # FindFeat as a pass (sampled batches) / inside the gather
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(val))
    cap_n = int(rng.choice([0, 1, V // (3 * Kg), V // Kg + 1, V]))
    cap_e = int(rng.choice([0, 1, V // (4 * Kg), V // Kg + 1]))
    seeds = dict(train=[(p, labels[p]) for p in parts])
    eng = make_engine(K, (V, F, indptr, indices, feats), B, fan, G=G, seeds=seeds, train_step=steps)
    orcs = [oracle.OracleRunner(indptr, indices, feats, V, F, B, fan, partition_count=G) for _ in range(G)]
    for g in range(G):
        for it in range(steps):
            eng.run_batch(g, it, is_presc=True)
            assert_batch_equal(orcs[g].run_batch(parts[g], labels[parts[g]], it, is_presc=True), eng.result(g, with_features=False), keys=KEYS_NO_FEATURES)
    eng.build_cache(cache_agg_mode=mode, node_capacity=cap_n, edge_capacity=cap_e, train_step=steps)
    for Ki in range(G // Kg):
        members = list(range(Ki * Kg, (Ki + 1) * Kg))
        _, QF = oracle.candidate_selection([orcs[m].node_access_time for m in members], V)
        _, QT = oracle.candidate_selection([orcs[m].edge_access_time for m in members], V)
        L.SetGPUDevice(Ki * Kg)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQF(eng.cache, Ki), np.int32, V), QF)
        assert np.array_equal(K.read_dev(L.GPUCache_GetQT(eng.cache, Ki), np.int32, V), QT)
        for m in members:
            if cap_n > 0:
                orcs[m].set_feature_cache(QF, cap_n, Kg)
            if cap_e > 0:
                orcs[m].set_topo_cache(QT, cap_e, Kg, Ki)
    for g in range(G):
        n_b = (len(parts[g]) + B - 1) // B
        for it in [0, n_b - 1, int(rng.randint(n_b))]:          # incl. the short last batch of the shard
            ref = orcs[g].run_batch(parts[g], labels[parts[g]], it)
            eng.run_batch(g, it, per_level=bool(rng.randint(2)))
            assert_batch_equal(ref, eng.result(g))
    eng.close()


@pytest.mark.parametrize("V,B,fan", [(3000, 1000, [25, 10, 5]), (400, 400, [12, 12, 6, 3])])
def test_heavy_duplicate_contention(K, oracle, V, B, fan):
    """Far more slots than nodes (1.25 M slots over 3000 nodes: every node is claimed hundreds of times per hop from
    every XCD): the who-beat-whom notifications, replaced claims and loser -> winner chains of k_sample / k_resolve are
    exercised across thousands of concurrently running tiles.  HIP == oracle, bit for bit, on consecutive batches."""
    rng = np.random.RandomState(V)
    F = 8
    deg = rng.randint(20, 60, size=V)
    deg[rng.randint(0, V, size=V // 50)] = 0
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(0, V, size=int(indptr[-1])).astype(np.int32)
    indices[rng.rand(len(indices)) < 0.01] = -1
    feats = rng.rand(V, F).astype(np.float32)
    labels = rng.randint(0, 5, size=V).astype(np.int32)
    seeds = rng.permutation(V)[:min(V, 3 * B)].astype(np.int32)
    orc = oracle.OracleRunner(indptr, indices, feats, V, F, B, fan)
    eng = make_engine(K, (V, F, indptr, indices, feats), B, fan, seeds=dict(train=[(seeds, labels[seeds])]))
    for counter in range(len(seeds) // B):
        ref = orc.run_batch(seeds, labels[seeds], counter)
        eng.run_batch(0, counter)
        assert_batch_equal(ref, eng.result(0))
        assert ref["ec"][2 + len(fan)] > 20 * ref["nc"][5 + 2 * len(fan)]      # > 20 sampled edges per unique node
    eng.close()


def test_fanout_other_than_the_pool_was_prepared_for(K, oracle, small_ds):
    """GPU_Random_Sampling takes the fan-out per call (Operator.cu:48 passes neighbor_count): a hop launched with a
    count the previous launch did not prepare the slot states for falls back to an explicit fill (k_fill_aux)."""
    ds = small_ds
    B, pool_fan, fan = 200, [10, 5], [7, 3]
    L = K.lib()
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, pool_fan)
    L.GPUCache_SetPreSc(eng.cache, 0)
    pool = eng.pools[0]
    for it in range(3):
        L.GPUMemoryPool_SetCurrentPipe(pool, 0)
        L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
        L.batch_generator_kernel(None, eng.noder, eng.cache, pool, B, it, 0, 0, K.TRAINMODE)
        for h in range(2):
            L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, fan[h], 2 * h + 2, 0)
        L.get_feature_kernel_all(None, eng.cache, eng.noder, pool, 0, 1)
        L.d_stream_sync(None)
        K.check()
        assert_batch_equal(orc.run_batch(ds.train, ds.labels[ds.train], it), eng.result(0))
    eng.close()


def test_repeatability_under_contention(K, oracle):
    """The same heavily contended batch over and over ($LEGION_SOAK_N times, default 30): the result never depends on
    which slot reached the table first -- every repetition is bit-identical to the oracle's canonical schedule."""
    import os
    V, B, fan, F = 5000, 1000, [25, 10, 5], 4
    rng = np.random.RandomState(7)
    deg = rng.randint(10, 80, size=V)
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    hot = rng.randint(0, V, size=50)
    indices = np.where(rng.rand(int(indptr[-1])) < 0.6, rng.choice(hot, size=int(indptr[-1])), rng.randint(0, V, size=int(indptr[-1]))).astype(np.int32)
    feats = rng.rand(V, F).astype(np.float32)
    labels = rng.randint(0, 5, size=V).astype(np.int32)
    seeds = rng.permutation(V)[:2 * B].astype(np.int32)
    orc = oracle.OracleRunner(indptr, indices, feats, V, F, B, fan)
    refs = [orc.run_batch(seeds, labels[seeds], c) for c in (0, 1)]
    eng = make_engine(K, (V, F, indptr, indices, feats), B, fan, seeds=dict(train=[(seeds, labels[seeds])]))
    for rep in range(int(os.environ.get("LEGION_SOAK_N", "30"))):
        c = rep & 1
        eng.run_batch(0, c, per_level=False)
        assert_batch_equal(refs[c], eng.result(0))
    eng.close()


# ---------------------------------------------------------------------------------------------------
# HIP-IPC size limit (include/legion_amd.h: LEGION_IPC_MAX_BYTES) and the link-prediction seed generator
# ---------------------------------------------------------------------------------------------------
def test_ipc_exports_above_the_limit_are_refused(K, small_ds, monkeypatch):
    """A single allocation above $LEGION_IPC_MAX_BYTES is never exported / imported (larger ones never finished
    importing on this driver, profiles/r01_unified_ipc_notes.md): sticky error instead of a stalled trainer."""
    L = K.lib()
    ds = small_ds
    eng = make_engine(K, ds, 100, [5, 3], G=2, train_step=2)
    for g in range(2):
        eng.run_batch(g, 0, is_presc=True)
    eng.build_cache(cache_agg_mode=1, node_capacity=1000, edge_capacity=500, train_step=2)   # 1000 rows x 400 B = 400 kB shards
    h = C.create_string_buffer(64)
    assert L.GPUCache_ExportFeatureShardChunk(eng.cache, 0, 0, h) == 0
    assert L.GPUGraphStorage_ExportFragmentChunk(eng.graph, 0, 1, 0, h) == 0
    K.check()
    monkeypatch.setenv("LEGION_IPC_MAX_BYTES", "100000")
    assert L.GPUCache_ExportFeatureShardChunk(eng.cache, 0, 0, h) == -1
    with pytest.raises(RuntimeError, match="HIP-IPC limit"):
        K.check()
    assert L.GPUCache_ExportFeatureShard(eng.cache, 0, h) == -1
    with pytest.raises(RuntimeError, match="HIP-IPC limit"):
        K.check()
    # the trainer hand-off buffers: a sample buffer above the limit is refused, nothing registered ...
    env = L.NewIPCEnv(1)
    L.IPCEnv_InitializeSamplesBuffer(env, 100, 50000, ds.spec.F, 0, 1)        # 50 000 ids x 4 bytes = 200 kB > limit
    with pytest.raises(RuntimeError, match="HIP-IPC limit"):
        K.check()
    # ... while a FEATURE buffer above it is built from chunks and mapped contiguously (end to end: test_gpu_ipc.py)
    monkeypatch.setenv("LEGION_SHARD_CHUNK_BYTES", "131072")
    L.IPCEnv_InitializeFeaturesBuffer(env, 0, 1000, ds.spec.F, 0, 1)          # 400 kB = 4 chunks
    K.check()
    p = L.IPCEnv_GetFloatFeatures(env, 0, 0)
    assert p
    pattern = np.arange(1000 * ds.spec.F, dtype=np.float32)
    L.d_copy_h_2_d(p, pattern.ctypes.data, pattern.nbytes)                    # one copy across the chunk seams
    assert np.array_equal(K.read_dev(p, np.float32, pattern.size), pattern)
    monkeypatch.delenv("LEGION_IPC_MAX_BYTES")
    L.IPCEnv_Finalize(env)
    eng.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_lp_seed_generator_matches_numpy(K, synth, small_ds, world):
    """legion_synth_lp_seeds (what bench.py --task lp uses at the papers100M shape) == synth.lp_trainingset, per rank."""
    L = K.lib()
    ds = small_ds
    B = 96
    d_ip, d_ix = K.DevBuf.from_numpy(ds.indptr), K.DevBuf.from_numpy(ds.indices)
    n = len(ds.train) - 5                      # a ragged tail: the last batch is padded
    for rank in range(world):
        ref = synth.lp_trainingset(ds, n, B, seed=1, rank=rank, world=world)
        srcs = ds.train[:n]
        sel = np.flatnonzero(srcs % world == rank)
        d_s, d_t = K.DevBuf.from_numpy(srcs[sel].astype(np.int32)), K.DevBuf.from_numpy(sel.astype(np.int64))
        d_o = K.DevBuf(len(ref) * 4)
        L.legion_synth_lp_seeds(None, d_o.ptr, d_s.ptr, d_t.ptr, len(sel), B, d_ip.ptr, d_ix.ptr, ds.spec.V, 1)
        L.d_stream_sync(None)
        K.check()
        assert np.array_equal(d_o.to_numpy(np.int32, len(ref)), ref)
        for b in (d_s, d_t, d_o):
            b.free()
    d_ip.free(); d_ix.free()


def test_full_batch_bound_on_a_small_graph(K, oracle, synth):
    """B = 8000 with fan-out {25,10,5} sizes hop 3 for 10 M slots = 9 766 tiles: k_write's LDS tile prefix then holds one
    entry per PAIR of tiles (kWriteEntries = 6144, gshift = 1) -- the layout every full-size run uses.  On a 98 k-node graph
    the oracle still finishes in seconds, so that path is compared bit for bit (the other oracle tests all have gshift = 0)."""
    ds = synth.generate(synth.spec_for("products", scale=0.04))
    B, fan = 8000, [25, 10, 5]
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan)
    assert len(ds.train) < B                       # one short batch: 7 864 seeds, -1 padding never sampled
    ref = orc.run_batch(ds.train, ds.labels[ds.train], 0)
    assert ref["ec"][5] > 2_000_000                # several thousand tiles in hop 3
    for _ in range(2):
        eng.run_batch(0, 0, per_level=False)
        assert_batch_equal(ref, eng.result(0))
    eng.close()


def test_four_hops_at_the_full_batch_bound(K, oracle, synth):
    """B = 8000, fan-out {25,10,5,2}: hop 4 is sized for 20 M slots = 19 532 tiles, so k_write's LDS prefix holds one entry per
    FOUR tiles in blocks of 16 groups (gshift = 2, the 16-bit in-block prefixes then span 64 tiles) -- the coarsest layout a
    realistic configuration reaches.  14 M edges on a 49 k-node graph: nearly every edge loses its claim and is resolved through
    the grouped prefix."""
    ds = synth.generate(synth.spec_for("products", scale=0.02))
    B, fan = 8000, [25, 10, 5, 2]
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
    eng = make_engine(K, ds, B, fan)
    ref = orc.run_batch(ds.train, ds.labels[ds.train], 0)
    assert ref["ec"][6] > 10_000_000
    eng.run_batch(0, 0, per_level=False)
    assert_batch_equal(ref, eng.result(0))
    eng.close()
