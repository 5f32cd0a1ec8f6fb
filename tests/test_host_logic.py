"""Host-side logic on CPU: step schedule (S11), seed split (S10), cache placement / ranking /
cost model (S6, S8), fragments (S9), the synthetic generator and the on-disk layout."""
import os

import numpy as np

from conftest import load_golden, sha


def test_schedule_table(oracle, synth):
    for row in load_golden("schedule_table"):
        steps, tb, vb, sb = oracle.coordinate(row["train_num"], row["valid_num"], row["test_num"], 8000)
        assert steps.tolist() == row["steps"] and vb.tolist() == row["valid_bs"] and sb.tolist() == row["test_bs"]
        assert (tb == 8000).all()
        # train_step floors and drops the tail; valid/test ceil to 512-sized steps (CUDA_IPC_Service.cu:89-117)
        assert steps[0] == (min(row["train_num"]) - 1) // 8000
        assert steps[1] == (max(row["valid_num"]) - 1) // 512 + 1
        assert oracle.max_step(steps, row["epoch"]) == row["max_step"] == (steps[0] + steps[1]) * row["epoch"] + steps[2]
        for b, (mode, local) in zip(row["probe"], row["mode_local"]):
            assert list(oracle.schedule(steps, row["epoch"], b)) == [mode, local]


def test_schedule_modes(oracle):
    steps = np.array([3, 2, 4], np.int32)
    seq = [oracle.schedule(steps, 2, b) for b in range(oracle.max_step(steps, 2))]
    assert seq == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1)] * 2 + [(2, 0), (2, 1), (2, 2), (2, 3)]


def test_split_seeds(oracle):
    ids = np.array([5, 2, 9, 4, 7, 12, 1], np.int32)
    parts = oracle.split_seeds(ids, 2)
    assert parts[0].tolist() == [2, 4, 12] and parts[1].tolist() == [5, 9, 7, 1]
    part_idx = np.zeros(16, np.int32)
    part_idx[[5, 9]] = 1
    part_idx[7] = 3                       # >= G: dropped (GPUGraphStore.cu:343)
    parts = oracle.split_seeds(ids, 2, part_idx, 1)
    assert parts[0].tolist() == [2, 4, 12, 1] and parts[1].tolist() == [5, 9]
    import legion1_amd.dist as D
    for r in range(3):
        assert D.shard_seeds(ids, r, 3).tolist() == oracle.split_seeds(ids, 3)[r].tolist()


def test_cache_fixture(oracle, synth):
    g = load_golden("cache_fixture")
    spec = synth.spec_for("products", scale=0.002)
    ds = synth.generate(spec)
    V = spec.V
    assert V == g["V"]
    rng = np.random.RandomState(7)
    i = 0
    for Kg in (1, 2, 4):
        acc_n = [rng.zipf(1.6, V).astype(np.uint64) % 50 for _ in range(Kg)]
        acc_e = [rng.zipf(1.5, V).astype(np.uint64) % 40 for _ in range(Kg)]
        AF, QF = oracle.candidate_selection(acc_n, V)
        AT, QT = oracle.candidate_selection(acc_e, V)
        # ranking: hotness descending, ties by ascending id (documented tie rule)
        tot = sum(a.astype(np.int64) for a in acc_n)
        order = np.lexsort((np.arange(V), -tot))
        assert np.array_equal(QF, order.astype(np.int32)) and np.array_equal(AF, tot[order].astype(np.uint64))
        for budget in (200_000, 1_000_000, 3_000_000):
            case = g["cases"][i]; i += 1
            cm = oracle.cost_model(AF, AT, QT, ds.indptr, V, spec.F, budget, Kg, [123456, 654321], [5000] * Kg, 24)
            assert case["Kg"] == Kg and case["budget"] == budget
            for k in ("node_capacity", "edge_capacity", "alpha_idx"):
                assert cm[k] == case[k], (Kg, budget, k)
            assert abs(cm["best_trans"] - case["best_trans"]) <= 1e-6 * max(1.0, abs(case["best_trans"]))
            assert sha(QF) == case["QF_sha256"] and sha(QT) == case["QT_sha256"]
            # capacities respect the budget split: alpha*budget on topology, the rest on features
            assert (cm["node_capacity"] - 1) * Kg * spec.F * 4 <= budget * Kg


def test_placement_rule(oracle):
    """(rank t, Kg) -> owner t%Kg + Ki*Kg, row t/Kg, global slot (t%Kg)*cap + t/Kg  (GPUCache.cu:88-108)."""
    import ctypes as C
    L = oracle.lib()
    V = 64
    QF = np.random.RandomState(3).permutation(V).astype(np.int32)
    for p in load_golden("cache_fixture")["placement"]:
        t, Kg, Ki, cap = p["t"], p["Kg"], p["Ki"], p["capacity"]
        assert p["owner"] == t % Kg + Ki * Kg and p["row"] == t // Kg and p["slot"] == (t % Kg) * cap + t // Kg
    for Kg, cap in ((1, 10), (2, 7), (4, 5), (8, 8)):
        nm = np.empty(V, np.int32)
        L.lo_build_feat_map(nm.ctypes.data_as(C.c_void_p), C.c_int32(V), QF.ctypes.data_as(C.c_void_p), C.c_int32(cap), C.c_int32(Kg))
        own, row = np.empty(V, np.int8), np.empty(V, np.int32)
        L.lo_build_topo_map(own.ctypes.data_as(C.c_void_p), row.ctypes.data_as(C.c_void_p), C.c_int32(V),
                            QF.ctypes.data_as(C.c_void_p), C.c_int32(cap), C.c_int32(Kg), C.c_int32(1))
        n = min(cap * Kg, V)
        for t in range(V):
            if t < n:
                assert nm[QF[t]] == (t % Kg) * cap + t // Kg and own[QF[t]] == t % Kg + Kg and row[QF[t]] == t // Kg
            else:
                assert nm[QF[t]] == -1 and own[QF[t]] == -1 and row[QF[t]] == -1


def test_cached_batch_equals_uncached(oracle, small_ds):
    """Cache and backing table hold byte-identical rows / adjacency, so a batch does not depend on the
    cache state (Kernels.cu:692-699, :391-410)."""
    ds = small_ds
    V, F = ds.spec.V, ds.spec.F
    lab = ds.labels[ds.train]
    B, fan = 400, [10, 5]
    base = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan).run_batch(ds.train, lab, 0)
    hot = np.bincount(ds.indices, minlength=V).astype(np.uint64)
    _, Q = oracle.candidate_selection([hot], V)
    for Kg, cap in ((1, 900), (2, 700), (4, 300)):
        r = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan, partition_count=Kg)
        r.set_feature_cache(Q, cap, Kg)
        r.set_topo_cache(Q, cap, Kg)
        res = r.run_batch(ds.train, lab, 0)
        for k in base:
            assert np.array_equal(base[k], res[k]), (Kg, k)
        hits = (r.node_map[res["ids"]] >= 0).mean()
        assert 0.0 < hits < 1.0          # both the hit and the miss path were exercised


def test_presampling_hotness(oracle, small_ds):
    ds = small_ds
    V, F = ds.spec.V, ds.spec.F
    lab = ds.labels[ds.train]
    B, fan = 300, [10, 5]
    r = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan)
    tot_nodes = tot_edges = 0
    for it in range(3):
        res = r.run_batch(ds.train, lab, it, is_presc=True)
        tot_nodes += int(res["nc"][5 + 2 * len(fan)])
        tot_edges += int(res["ec"][2 + len(fan)])
    assert int(r.node_access_time.sum()) == tot_nodes      # HotnessMeasure: +1 per unique node per batch
    assert int(r.edge_access_time.sum()) == tot_edges      # kernel_pre_sampler: +1 on the source per sampled edge
    assert (r.position_map[res["ids"]] == 0).all()          # ClearPosMap ran (train mode)


def test_synth_generator(synth):
    spec = synth.spec_for("papers100M", scale=0.0005)
    ds = synth.generate(spec)
    assert ds.indptr[0] == 0 and (np.diff(ds.indptr) >= 1).all()
    assert ds.indices.min() >= 0 and ds.indices.max() < spec.V
    assert abs(ds.E / spec.V - spec.mean_degree) / spec.mean_degree < 0.15
    assert len(np.unique(np.concatenate([ds.train, ds.valid, ds.test]))) == len(ds.train) + len(ds.valid) + len(ds.test)
    assert np.array_equal(ds.features[:7], synth.features(spec, np.arange(7)))
    assert np.abs(ds.features).max() <= 0.5
    ind = np.bincount(ds.indices, minlength=spec.V)
    assert ind.max() > 50 * ind.mean()                      # skewed in-degree: hot nodes exist
    # determinism
    assert sha(synth.generate(spec).indices) == sha(ds.indices)


def test_legion_file_layout(synth, tmp_path):
    spec = synth.spec_for("products", scale=0.001)
    ds = synth.generate(spec)
    path = str(tmp_path / "ds")
    synth.write_legion_files(ds, path, partition_count=2)
    assert os.path.getsize(os.path.join(path, "edge_src")) == (spec.V + 1) * 8
    assert os.path.getsize(os.path.join(path, "edge_dst")) == ds.E * 4
    assert os.path.getsize(os.path.join(path, "features")) == spec.V * spec.F * 4
    assert np.array_equal(np.fromfile(os.path.join(path, "trainingset"), dtype="<i4"), ds.train)
    line = synth.meta_config_line(ds, path, 8000, 1 << 30, 10, 0).split()
    assert len(line) == 11 and int(line[2]) == spec.V and int(line[3]) == ds.E and line[0].endswith("/")


def test_launcher_meta_config_line():
    """launch_server.py writes the launcher contract of legion_server.py:58-59; the expected line is the sample
    `meta_config` the reference ships at its root (uk-union, 32 GB cache, 10 epochs, clique mode)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("launch_server", os.path.join(ROOT, "legion-1_amd", "launch_server.py"))
    ls = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ls)

    class A:
        dataset_path, dataset, train_batch_size, cache_memory, epoch, usenvlink = "/home/atc-artifacts-user/datasets", "UKS", 8000, 32000000000, 10, 1
        seed_lists = False
    assert ls.meta_line(A) == "/home/atc-artifacts-user/datasets/ukunion/ 8000 133633040 5507679822 256 13363304 100000 100000 32000000000 10 0"
    assert ls.cache_agg_mode(8, 1) == 1 and ls.cache_agg_mode(1, 1) == 0 and ls.cache_agg_mode(8, 0) == 0
    assert len(ls.meta_line(A).split()) == 11
    A.usenvlink = 0
    assert ls.meta_line(A).split()[-1] == "1"          # partition file, as the reference writes it (legion_server.py:59)
    A.seed_lists = True                                # extension: per-GPU training lists (link prediction on G > 1 GPUs)
    assert ls.meta_line(A).split()[-1] == "2"


def test_lp_seed_lists_golden(synth):
    """synth.lp_trainingset against tests/golden/lp_seed_lists.json (the layout `trainingset` / `trainingset_<G>_<g>` files have):
    [src | pos | neg] thirds per batch, positives are neighbours of their source, the 2-GPU lists re-deal the 1-GPU triples by src % 2."""
    from conftest import load_golden
    import numpy as np
    g = load_golden("lp_seed_lists")
    ds = synth.generate(synth.spec_for("products", scale=0.002), with_features=False)
    n, B = g["n_triples"], g["batch"]
    k = B // 3
    assert synth.lp_trainingset(ds, n, B).tolist() == g["lists"]["1of1"]
    for r in (0, 1):
        assert synth.lp_trainingset(ds, n, B, rank=r, world=2).tolist() == g["lists"]["%dof2" % r]
    one = np.array(g["lists"]["1of1"]).reshape(-1, 3, k)
    for b in range(one.shape[0]):
        for s_, p_ in zip(one[b, 0], one[b, 1]):
            assert p_ == s_ or p_ in ds.indices[ds.indptr[s_]:ds.indptr[s_ + 1]]
    triples = lambda lst: {tuple(t) for t in np.array(lst).reshape(-1, 3, k).transpose(0, 2, 1).reshape(-1, 3).tolist()}   # noqa: E731
    assert triples(g["lists"]["0of2"]) | triples(g["lists"]["1of2"]) == triples(g["lists"]["1of1"])


def _client_open_without_listener(tmp_ns):
    """a feature-buffer slot that holds a chunk descriptor (LGNVMM01) whose server socket is not there must
    make legion_ipc_client_open fail as a whole -- null client, sticky error, no semaphore posted, nothing left mapped --
    instead of handing the trainer a null feature tensor."""
    import struct
    import legion1_amd.capi as K
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    L.legion_clear_error()
    L.legion_ipc_set_namespace(tmp_ns.encode())
    size = 12 + 8 * 2 * 7 * 64            # exactly the reference's struct: nothing is appended to the shared slab
    path = "/dev/shm/" + tmp_ns + "simpleIPCshm"
    desc = b"LGNVMM01" + struct.pack("<QQII", 1 << 21, 1 << 21, 1, 0)
    slab = bytearray(size)
    for pipe in range(2):
        off = 12 + ((0 * 2 + pipe) * 7 + 1) * 64
        slab[off:off + len(desc)] = desc
    with open(path, "wb") as f:
        f.write(slab)
    os.environ["LEGION_VMM_ATTACH_TIMEOUT_MS"] = "300"
    try:
        client = L.legion_ipc_client_open(0)
        msg = L.legion_last_error()
        assert not client, "client must be refused"
        assert msg and b"attaching the chunked feature buffer failed" in msg, msg
        assert not os.path.exists("/dev/shm/sem." + tmp_ns + "sem_r_0_0"), "no pipe may be posted as free"
    finally:
        os.environ.pop("LEGION_VMM_ATTACH_TIMEOUT_MS", None)
        L.legion_clear_error()
        L.legion_ipc_set_namespace(b"")
        os.unlink(path)


def test_client_open_refuses_a_chunk_descriptor_without_listener():
    _client_open_without_listener("lgn_t_nolisten_%d_" % os.getpid())


def test_client_open_refuses_a_server_that_has_not_registered_its_buffers():
    """a trainer that attaches before the server has registered its hand-off buffers (the feature buffers only exist
    after the pre-sampling epoch, Server.cu:33,273-282) found all-zero handle slots, skipped them, and went on with null buffers -- its first
    kernel then faulted on the GPU.  The reference fails in cudaIpcOpenMemHandle there; here the client is refused with the start order named
    (only the device-free test mode, $LEGION_IPC_NO_DEVICE=1, may skip empty slots)."""
    import legion1_amd.capi as K
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    ns = "lgn_t_early_%d_" % os.getpid()
    L.legion_ipc_set_namespace(ns.encode())
    path = "/dev/shm/" + ns + "simpleIPCshm"
    with open(path, "wb") as f:
        f.write(bytes(12 + 8 * 2 * 7 * 64))          # a slab whose server has registered nothing yet
    try:
        L.legion_clear_error()
        client = L.legion_ipc_client_open(0)
        msg = L.legion_last_error()
        assert not client, "a client on null buffers must be refused"
        assert msg and b"has not registered buffer 0 of pipe 0 of GPU 0 yet" in msg and b"System is ready for serving" in msg, msg
        assert not os.path.exists("/dev/shm/sem." + ns + "sem_r_0_0"), "no pipe may be posted as free"
    finally:
        L.legion_clear_error()
        L.legion_ipc_set_namespace(b"")
        os.unlink(path)


def test_shard_pitch_respects_the_cache_budget():
    """the cost model plans the feature cache in DENSE rows (capacity = budget / (F * 4), GPUCache.cu:727),
    so a shard may only take the line-aligned pitch (F = 100 -> 128 floats, +28 %) when the padded shard still fits the budget;
    the cache_memory contract: node_capacity * pitch * 4 <= cache_memory."""
    import legion1_amd.capi as K
    L = K.lib()
    assert L.legion_row_pitch(100) == 128 and L.legion_row_pitch(128) == 128 and L.legion_row_pitch(256) == 256
    for F in (100, 36, 7, 52, 128, 256):
        aligned = L.legion_row_pitch(F)
        for cache_memory in (1 << 20, 3_000_000, 10 << 30):
            cap = cache_memory // (F * 4)                     # what the reference's sweep hands out at alpha = 0
            pitch = L.legion_shard_pitch(F, cap, cache_memory)
            assert pitch in (F, aligned) and cap * pitch * 4 <= cache_memory
            assert pitch == (aligned if cap * aligned * 4 <= cache_memory else F)
            assert L.legion_shard_pitch(F, cache_memory // (aligned * 4), cache_memory) == aligned   # fewer rows: the padded shard fits
        assert L.legion_shard_pitch(F, 1 << 20, 0) == aligned                # no budget known: the caller set the capacity
    assert L.legion_shard_pitch(100, 1000, 1000 * 400) == 100 and L.legion_shard_pitch(100, 1000, 1000 * 512) == 128


def test_cost_model_without_pcm_counters(oracle, synth):
    """counters=None: the PCM-free transaction estimate: sum over the ranked rows of
    edge hotness x ceil((8 + 4 * min(deg, 16)) / 64).  Equals the explicit-counter model fed with that sum."""
    spec = synth.spec_for("products", scale=0.002)
    ds = synth.generate(spec, with_features=False)
    V = spec.V
    rng = np.random.RandomState(11)
    AF, QF = oracle.candidate_selection([rng.zipf(1.6, V).astype(np.uint64) % 50], V)
    AT, QT = oracle.candidate_selection([rng.zipf(1.5, V).astype(np.uint64) % 40], V)
    deg = np.diff(ds.indptr)[QT]
    w = (8 + 4 * np.minimum(deg, 16) + 63) // 64
    assert set(np.unique(w)) <= {1, 2} and (w[deg <= 14] == 1).all() and (w[deg >= 15] == 2).all()
    est = int((AT.astype(np.int64) * w).sum())
    assert int(AT.sum()) < est <= 2 * int(AT.sum())
    for budget in (200_000, 1_000_000, 3_000_000):
        a = oracle.cost_model(AF, AT, QT, ds.indptr, V, spec.F, budget, 1, None, [5000], 24)
        b = oracle.cost_model(AF, AT, QT, ds.indptr, V, spec.F, budget, 1, [est, 0], [5000], 24)
        assert a == b


def test_synth_spec_c_equals_python(synth):
    """`legion_synth_spec` (csrc/synth.hip, host code) is what a `synth:` dataset source of the server generates from: every
    field must equal synth.py's spec_for() for the three shapes at any scale -- ladder (80 bisection steps in IEEE doubles, Python's
    round-half-even), the coprime multipliers, the shrunk seed sets."""
    import ctypes as C
    import legion1_amd.capi as K
    L = K.lib()
    for name in ("products", "papers100M", "uk-union"):
        for scale in (1.0, 0.5, 0.3, 0.2, 0.1, 0.037, 0.01, 0.004, 0.002, 1e-5, 1e-7):
            want = synth.spec_for(name, scale=scale)
            got = K.LegionSynthSpec()
            assert L.legion_synth_spec(name.encode(), scale, C.byref(got)) == 0
            K.check()
            assert (got.V, got.F, got.classes) == (want.V, want.F, want.classes), (name, scale)
            assert (got.n_train, got.n_valid, got.n_test) == (want.n_train, want.n_valid, want.n_test), (name, scale)
            assert (got.M, got.C, got.M2, got.C2) == (want.M, want.C, want.M2, want.C2), (name, scale)
            assert list(got.ladder) == want.ladder.tolist(), (name, scale)
            assert got.mean_degree == want.mean_degree
            # the two host-side closed forms the server builds its seed lists from
            ids = synth.seed_ids(want, 0, min(50, want.n_train))
            assert [L.legion_synth_seed_id_host(i, want.V, want.M2, want.C2) for i in range(len(ids))] == ids.tolist()
            lab = synth.labels(want, ids)
            assert [L.legion_synth_label_host(int(v), want.classes) for v in ids] == lab.tolist()
    bad = K.LegionSynthSpec()
    for name, scale in ((b"nope", 1.0), (b"products", 0.0), (b"products", 1.5)):
        assert L.legion_synth_spec(name, scale, C.byref(bad)) == -1
        assert L.legion_last_error()
        L.legion_clear_error()


def test_cost_model_pcm_free_golden_at_the_products_shape(oracle):
    """the default cost-model input of this repository's server (counters == NULL: no Intel PCM) is an ESTIMATE --
    sum over the ranked rows of edge hotness x ceil((8 + 4 min(deg, 16)) / 64) -- and every caller's alpha / capacities follow from it.
    The plan at the full products shape (8 pre-sampling batches of {25,10}, three budgets) is pinned in tests/golden/cost_model_pcm_free.json
    (`python oracle/make_golden.py`); the GPU suite checks the product against the same numbers (tests/test_gpu_full_shape.py)."""
    import make_golden as M
    g = load_golden("cost_model_pcm_free")
    d = M.cost_model_pcm_free()
    for k in ("V", "E", "F", "max_ids", "node_hotness_sha256", "edge_hotness_sha256", "QF_sha256", "QT_sha256"):
        assert d[k] == g[k], k
    assert len(d["plans"]) == len(g["plans"]) == 3
    for a, b in zip(d["plans"], g["plans"]):
        for k in ("cache_memory", "node_capacity", "edge_capacity", "alpha_idx"):
            assert a[k] == b[k], (b["budget_frac"], k, a[k], b[k])
        assert abs(a["best_trans"] - b["best_trans"]) <= 1e-6 * b["best_trans"]
    # the plan moves with the budget the way a cost model should: more budget -> more feature rows, a smaller topology share
    caps = [p["node_capacity"] for p in g["plans"]]
    assert caps == sorted(caps) and [p["alpha_idx"] for p in g["plans"]] == sorted((p["alpha_idx"] for p in g["plans"]), reverse=True)


def test_server_binary_refuses_a_malformed_meta_config(tmp_path):
    """The `legion` binary against meta_config lines the reference would read unchecked (GPUGraphStore.cu:190-223: a short line leaves zeros
    behind; the first division by the batch size ends the process without a message).  Refused by name, exit code 1, before any device call --
    so this runs in the build container."""
    import subprocess
    server = os.environ.get("LEGION_SERVER_BIN") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "legion-1_amd", "csrc", "legion")
    assert os.path.exists(server), "build the server: make -C legion-1_amd/csrc legion"      # LEGION_SERVER_BIN: the sanitizer build of the binary
    cases = [("garbage", "fewer than eleven fields"), ("/d/ 8000 100 1000 16 10 10 10 0 1", "fewer than eleven fields"),
             ("/d/ 0 100 1000 16 10 10 10 0 1 0", "batch size < 1"), ("/d/ 8 0 1000 16 10 10 10 0 1 0", "node count < 1"),
             ("/d/ 8 100 -5 16 10 10 10 0 1 0", "negative edge count"), ("/d/ 8 100 1000 0 10 10 10 0 1 0", "feature dim < 1"),
             ("/d/ 8 100 1000 16 -1 10 10 0 1 0", "negative seed-set size"), ("/d/ 8 100 1000 16 101 10 10 0 1 0", "larger than the node count"),
             ("/d/ 8 100 1000 16 10 10 10 -1 1 0", "negative cache budget"), ("/d/ 8 100 1000 16 10 10 10 0 -2 0", "negative epoch count"),
             ("/d/ 8 100 1000 16 10 10 10 0 1 3", "partition flag outside 0..2"), ("synth:nothing 8 100 0 16 10 10 10 0 1 0", "names no known workload"),
             # a synth: source whose meta line is not the generator's (products at 0.4 %: V = 9796, F = 100, 786 training ids) -- refused before any device call
             ("synth:products:0.004 8 9797 0 100 10 10 10 0 1 0", "differ from the synth: generator's"), ("synth:products:0.004 8 9796 0 64 10 10 10 0 1 0", "differ from the synth: generator's"),
             ("synth:products:0.004:300 8 9796 0 100 10 10 10 0 1 0", "differ from the synth: generator's"), ("synth:products:0.004 8 9796 0 100 787 10 10 0 1 0", "larger than the synth: generator's"),
             ("synth:products:7 8 9796 0 100 10 10 10 0 1 0", "names no known workload")]
    for line, want in cases:
        meta = tmp_path / "meta_config"
        meta.write_text(line)
        r = subprocess.run([server, "1", "0", "5,4", str(meta)], capture_output=True, text=True, timeout=60, cwd=str(tmp_path),
                           env=dict(os.environ, LEGION_IPC_NAMESPACE="badmeta%d_" % os.getpid()))
        assert r.returncode == 1 and want in r.stderr and "ready for serving" not in r.stdout, (line, r.returncode, r.stderr[-400:])
    r = subprocess.run([server, "1", "0", "5,4", str(tmp_path / "missing")], capture_output=True, text=True, timeout=60, cwd=str(tmp_path))
    assert r.returncode == 1 and "meta_config missing" in r.stderr
    r = subprocess.run([server], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage:" in r.stderr


def test_runner_gather_choice_at_the_baseline_shapes():
    """$LEGION_RUNNER_GATHER=auto: the estimate the runner makes once after the pre-sampling epoch, at the measured batch statistics of the BASELINE
    shapes (profiles/r05_runner_gather.md: `all` -2.8 % at papers100M {25,10,5}, -4 % at products {25,10}, +5.7 % at products {25,10,5})."""
    import ctypes as C
    import legion1_amd.capi as K
    L = K.lib()
    g, s_ = C.c_double(), C.c_double()
    cases = [("papers100M {25,10,5}", 128, 1.94e6, 8000 * 25 + 0.196e6 * 10 + 0.9e6 * 5, 1),
             ("products {25,10}", 100, 0.89e6, 8000 * 25 + 0.196e6 * 10, 1),
             ("products {25,10,5}", 100, 2.13e6, 8000 * 25 + 0.196e6 * 10 + 1.96e6 * 5, 0),
             ("uk-union {25,10}", 256, 1.24e6, 8000 * 25 + 0.2e6 * 10, 1)]
    for name, F, rows, slots, want in cases:
        assert L.legion_runner_gather_estimate(F, rows, slots, C.byref(g), C.byref(s_)) == want, (name, g.value, s_.value)
        assert g.value > 0 and s_.value > 0
    assert L.legion_runner_gather_estimate(128, 0.0, 0.0, None, None) == 0          # no pre-sampled batch: the reference's list


def test_launcher_against_the_references_own_launcher(tmp_path, monkeypatch):
    """tests/golden/launcher_lines.json was produced by the REFERENCE's legion_server.py (imported in the build container, os.system recorded instead of
    run: `python oracle/make_golden_launcher.py`): the `meta_config` line and the server command for 6 datasets x 4 GPU counts x both interconnect
    switches x 2 (batch, cache, epoch) settings.  launch_server.py must write the same line and start its server with the same <gpu_number>
    <cache_agg_mode>, for the same command line."""
    import importlib.util
    g = load_golden("launcher_lines")
    assert len(g["cases"]) == 96
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("launch_server", os.path.join(root, "legion-1_amd", "launch_server.py"))
    ls = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ls)
    monkeypatch.chdir(tmp_path)
    printed = []
    monkeypatch.setattr("builtins.print", lambda *a, **k: printed.append(" ".join(map(str, a))))
    for c in g["cases"]:
        del printed[:]
        rc = ls.main(["--dataset_path", c["dataset_path"], "--dataset", c["dataset"], "--train_batch_size", str(c["train_batch_size"]), "--gpu_number", str(c["gpu_number"]),
                      "--epoch", str(c["epoch"]), "--cache_memory", str(c["cache_memory"]), "--usenvlink", str(c["usenvlink"]), "--dry_run"])
        assert rc == 0
        assert open("meta_config").read() == c["meta_config"], c
        ours, theirs = printed[-1].split(), c["command"].split()
        assert ours[0].endswith("legion") and theirs[0].endswith("legion") and ours[1:3] == theirs[1:3], (ours, theirs)
        assert ours[3] == "25,10" and ours[4] == os.path.abspath("meta_config")     # the two extras: fan-outs (the reference hard-codes 25,10), the file it just wrote
