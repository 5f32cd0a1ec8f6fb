"""End-to-end drop-in boundary on the GPU: the `legion` server binary (meta_config + raw dataset
files, pre-sampling epoch, cache build, run loop) hands batches to a trainer-side process that uses
the `ipc_service` Python module -- POSIX shm + named semaphores + HIP IPC handles.
Every batch the trainer sees is compared with the oracle."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, sha

pytestmark = pytest.mark.gpu

SERVER = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")


def _audit_clean(log_text, gpus=1):
    """$LEGION_DEVICE_AUDIT=1 (tests/conftest.py): the server's summary line -- checks ran, none failed, and the server attributed every stream,
    allocation and launch to a logical GPU (returns the parsed counts; None when the audit is off)."""
    import re
    if os.environ.get("LEGION_DEVICE_AUDIT") != "1":
        return None
    m = re.search(r"Device audit: (\d+) checks, (\d+) violations, (\d+) unattributed, (\d+) launches with peer arguments", log_text)
    assert m, log_text[-1500:]
    checks, bad, unattributed, peer = (int(x) for x in m.groups())
    assert checks > 100 * gpus and bad == 0 and unattributed == 0, (m.group(0), log_text[-1500:])
    from conftest import note_server_audit
    note_server_audit(dict(checks=checks, violations=bad, unattributed=unattributed, peer_launches=peer))
    return dict(checks=checks, violations=bad, unattributed=unattributed, peer_launches=peer)


def _wait_ready(proc, log_path, timeout=240):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if os.path.exists(log_path) and "System is ready for serving" in open(log_path, errors="ignore").read():
            return
        if proc.poll() is not None:
            raise AssertionError("server exited early:\n" + open(log_path, errors="ignore").read()[-3000:])
        time.sleep(0.2)
    proc.kill()
    raise AssertionError("server not ready:\n" + open(log_path, errors="ignore").read()[-3000:])


@pytest.mark.parametrize("fan,budget_frac,tables,graph", [([25, 10], 0.2, "host", "0"), ([5, 4, 3], 10.0, "auto", "0"),
                                                         ([10, 5], 0.1, "device", "0"), ([25, 10], 0.2, "host", "1"),
                                                         ([5, 4, 3], 10.0, "auto", "1")])
def test_server_binary_to_ipc_service(tmp_path, synth, oracle, fan, budget_frac, tables, graph):
    assert os.path.exists(SERVER), "build the server: make -C legion-1_amd/csrc legion"
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    B, epochs = 512, int(os.environ.get("LEGION_TEST_EPOCHS", "2"))     # soak: LEGION_TEST_EPOCHS=50 checks thousands of served batches
    budget = int(spec.V * spec.F * 4 * budget_frac)
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write(synth.meta_config_line(ds, data, B, budget, epochs, 0))
    ns = "t%d_%d_" % (os.getpid(), len(fan))
    # LEGION_TABLES: host = the reference's pinned-host tables read over PCIe, device/auto = replicated into HBM
    # LEGION_BATCH_GRAPH=1: the runner replays the sampler side of a batch as one recorded hipGraph per (pipe, mode), one plain gather on stream 1 behind it
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_TABLES=tables, LEGION_BATCH_GRAPH=graph)
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, "1", "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT,
                                  env=env, cwd=str(tmp_path))
    try:
        _wait_ready(server, log)
        out = str(tmp_path / "client.json")
        client = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ipc_client.py"), str(spec.F), str(epochs), out],
                                env=env, capture_output=True, text=True, timeout=300)
        assert client.returncode == 0, client.stdout[-2000:] + client.stderr[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
        _audit_clean(open(log).read())
    finally:
        if server.poll() is None:
            server.kill()
    got = json.load(open(out))
    H = len(fan)
    assert got["hops"] == H
    if H == 2:      # what the reference's trainers unpack (tests/golden/trainer_api.json, extracted from their source): 7 tensors, 4 block sizes, 3 step counts
        api = json.load(open(os.path.join(ROOT, "tests", "golden", "trainer_api.json")))["surface"]
        rec0 = got["batches"][0]
        assert 3 + 2 * len(rec0["edges"]) == api["get_next"][0][1] == 7 and len(rec0["sizes"]) == api["get_block_size"][0][1] == 4
        assert len(got["steps"]) == api["get_steps"][0][1] == 3
    sets = {0: ds.train, 1: ds.valid, 2: ds.test}
    steps, tb, vb, sb = oracle.coordinate([len(ds.train)], [len(ds.valid)], [len(ds.test)], B)
    assert got["steps"] == steps.tolist()
    bs = {0: int(tb[0]), 1: int(vb[0]), 2: int(sb[0])}
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
    assert len(got["batches"]) == oracle.max_step(steps, epochs)
    for rec in got["batches"]:
        mode, local = oracle.schedule(steps, epochs, rec["b"])
        ids = sets[mode]
        ref = orc.run_batch(ids, ds.labels[ids], local, mode=mode, batch_size=bs[mode])
        nc, ec = ref["nc"], ref["ec"]
        assert rec["n"] == nc[5 + 2 * H]
        assert rec["sizes"] == [int(x) for k in range(1, H + 1) for x in (nc[5 + 2 * (H - k + 1)], nc[5 + 2 * (H - k)])]
        assert rec["edges"] == [int(ec[2 + (H - k + 1)]) for k in range(1, H + 1)]
        assert rec["ids"] == sha(ref["ids"]) and rec["features"] == sha(ref["features"]) and rec["labels"] == sha(ref["labels"])
        assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"])
    text = open(log).read()
    assert "Train Steps: %d" % steps[0] in text and "Server Stopped" in text
    assert ("Tables stay in pinned host memory" if tables == "host" else "Tables replicated into HBM") in text


@pytest.mark.parametrize("workload,scale,fan,B,G,gather", [("products", 0.004, [5, 4, 3], 512, 1, "auto"), ("papers100M", 0.0008, [25, 10], 1000, 1, "level"),
                                                          ("uk-union", 0.0005, [10, 5], 512, 2, "all"), ("products", 0.004, [25, 10, 5], 512, 1, "auto"),
                                                          ("products", 0.004, [10, 5], 512, 1, "cache-all"), ("papers100M", 0.0008, [10, 5], 512, 2, "cache-level")])
def test_server_synth_dataset_source(tmp_path, synth, oracle, workload, scale, fan, B, G, gather):
    """meta_config dataset path `synth:<workload>:<scale>`: the server generates CSR + features in HBM with the legion_synth_* calls
    (no files) -- what bench.py's `served` leg starts at the papers100M shape.  Every served batch (train, valid, test) must equal the
    oracle run on the numpy statement of the same generator; G = 2: two logical GPUs, one trainer each, tid % G seed split."""
    spec = synth.spec_for(workload, scale=scale)
    ds = synth.generate(spec)
    epochs = int(os.environ.get("LEGION_TEST_EPOCHS", "2"))                          # soak: LEGION_TEST_EPOCHS=60
    n_valid, n_test = min(700, spec.n_valid), min(300, spec.n_test)                   # the meta line takes the first n ids of each range
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        # cache-*: $LEGION_SYNTH_CACHE=1 with a budget of 20 % of the feature table -- cost model, FillUp, cached gather and partitioned sampler behind the server
        budget = int(spec.V * spec.F * 4 * 0.2) if gather.startswith("cache") else 1 << 40
        f.write("synth:%s:%r %d %d %d %d %d %d %d %d %d 0" % (workload, scale, B, spec.V, ds.E, spec.F, spec.n_train, n_valid, n_test, budget, epochs))
    ns = "sy%d_%s%d%s_" % (os.getpid(), workload[:2], len(fan) + fan[0], gather[:2])
    # LEGION_RUNNER_GATHER: one FeatureExtractor op per level (the reference's op list) / one gather over all rows behind the last hop /
    # auto = decided once after the pre-sampling epoch from its counters -- the served batches are the same bytes either way
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_RUNNER_GATHER=gather.replace("cache-", ""))
    if gather.startswith("cache"):
        env["LEGION_SYNTH_CACHE"] = "1"
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, str(G), "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=str(tmp_path))
    clients = []
    try:
        _wait_ready(server, log)
        for g in range(G):
            out = str(tmp_path / ("client%d.json" % g))
            clients.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_client.py"), str(spec.F), str(epochs), out],
                                                  env=dict(env, LEGION_IPC_DEVICE=str(g)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for out, c in clients:
            stdout, _ = c.communicate(timeout=300)
            assert c.returncode == 0, stdout[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
        _audit_clean(open(log).read())
    finally:
        for _, c in clients:
            if c.poll() is None:
                c.kill()
        if server.poll() is None:
            server.kill()
    text = open(log).read()
    assert "Graph generated in HBM: %d edges" % ds.E in text and "Tables generated in HBM" in text and "Server Stopped" in text
    assert ("Runner gather:" in text) == (gather == "auto")
    assert ("cache built on top" in text and "Feat capacity" in text) == gather.startswith("cache")        # the cost model sized a real cache
    H = len(fan)
    parts = {0: oracle.split_seeds(ds.train, G), 1: oracle.split_seeds(ds.valid[:n_valid], G), 2: oracle.split_seeds(ds.test[:n_test], G)}
    steps, tb, vb, sb = oracle.coordinate([len(p) for p in parts[0]], [len(p) for p in parts[1]], [len(p) for p in parts[2]], B)
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
    for g in range(G):
        got = json.load(open(clients[g][0]))
        assert got["hops"] == H and got["steps"] == steps.tolist() and len(got["batches"]) == oracle.max_step(steps, epochs)
        bs = {0: int(tb[g]), 1: int(vb[g]), 2: int(sb[g])}
        for rec in got["batches"]:
            mode, local = oracle.schedule(steps, epochs, rec["b"])
            ids = parts[mode][g]
            ref = orc.run_batch(ids, ds.labels[ids], local, mode=mode, batch_size=bs[mode])
            assert rec["n"] == ref["nc"][5 + 2 * H] and rec["edges"] == [int(ref["ec"][2 + (H - k + 1)]) for k in range(1, H + 1)]
            assert rec["ids"] == sha(ref["ids"]) and rec["features"] == sha(ref["features"]) and rec["labels"] == sha(ref["labels"])
            assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"])


@pytest.mark.parametrize("G", [1, 2])
def test_server_synth_link_prediction_lists(tmp_path, synth, oracle, G):
    """`synth:` source + meta flag 2: the per-GPU link-prediction seed lists are generated by the server (legion_synth_lp_seeds: one triple per
    training id, dealt by src % G with its global number, batches laid out as [src | pos | neg] thirds, lp_sage.py:87-90) -- what bench.py's `lp` leg
    serves at the papers100M shape.  Every served batch against the oracle run on synth.lp_trainingset (the numpy statement of the rule): the thirds
    survive, duplicate seeds keep the last-occurrence position rule."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    B, fan, epochs = 510, [10, 5], 1
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write("synth:products:0.004 %d %d %d %d %d 100 60 0 %d 2" % (B, spec.V, ds.E, spec.F, spec.n_train, epochs))
    env = dict(os.environ, LEGION_IPC_NAMESPACE="lp%d_%d_" % (os.getpid(), G), HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_CLIENT_DUMP_SEEDS="1")
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, str(G), "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=str(tmp_path))
    clients = []
    try:
        _wait_ready(server, log)
        for g in range(G):
            out = str(tmp_path / ("client%d.json" % g))
            clients.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_client.py"), str(spec.F), str(epochs), out],
                                                  env=dict(env, LEGION_IPC_DEVICE=str(g)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for out, c in clients:
            stdout, _ = c.communicate(timeout=300)
            assert c.returncode == 0, stdout[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
        _audit_clean(open(log).read())
    finally:
        for _, c in clients:
            if c.poll() is None:
                c.kill()
        if server.poll() is None:
            server.kill()
    assert "Link-prediction seed lists generated" in open(log).read()
    lists = [synth.lp_trainingset(ds, len(ds.train), B, rank=g, world=G) for g in range(G)]
    va, te = oracle.split_seeds(ds.valid[:100], G), oracle.split_seeds(ds.test[:60], G)
    steps, tb, vb, sb = oracle.coordinate([len(x) for x in lists], [len(p) for p in va], [len(p) for p in te], B)
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
    k, dup = B // 3, 0
    for g in range(G):
        got = json.load(open(clients[g][0]))
        assert got["steps"] == steps.tolist() and len(got["batches"]) == oracle.max_step(steps, epochs)
        sets = {0: lists[g], 1: va[g], 2: te[g]}
        bs = {0: int(tb[g]), 1: int(vb[g]), 2: int(sb[g])}
        for rec in got["batches"]:
            mode, local = oracle.schedule(steps, epochs, rec["b"])
            ids = sets[mode]
            ref = orc.run_batch(ids, ds.labels[ids], local, mode=mode, batch_size=bs[mode])
            assert rec["ids"] == sha(ref["ids"]) and rec["features"] == sha(ref["features"]) and rec["labels"] == sha(ref["labels"]), (g, rec["b"])
            assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"]), (g, rec["b"])
            if mode == 0:       # the thirds: sources of this GPU, a neighbour (or the source itself) each, any node id
                seeds = np.array(rec["seeds"])
                assert len(seeds) == B and np.array_equal(seeds, lists[g][local * B:(local + 1) * B])
                assert (seeds[:k] % G == g).all()
                dup += int(len(np.unique(seeds)) < B)
    assert dup > 0          # hot positives repeat inside a batch: the duplicate-seed path really ran


def test_server_log_goes_to_stderr_on_request(tmp_path, synth):
    """$LEGION_LOG=stderr: every progress print of the library (the reference prints to stdout: "Train Steps", "Storage Initialized", "System is
    ready for serving" ...) goes to stderr, for a host that owns stdout (bench.py's one JSON line); unset: stdout, like the reference."""
    spec = synth.spec_for("products", scale=0.004)
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write("synth:products:0.004 128 %d 0 %d %d 10 10 0 1 0" % (spec.V, spec.F, spec.n_train))
    for log_env, where in (("stderr", "stderr"), (None, "stdout")):
        env = dict(os.environ, LEGION_IPC_NAMESPACE="lg%d_%s_" % (os.getpid(), where), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("LEGION_LOG", None)
        if log_env:
            env["LEGION_LOG"] = log_env
        fo, fe = str(tmp_path / ("out_" + where)), str(tmp_path / ("err_" + where))
        with open(fo, "w") as o, open(fe, "w") as e:
            server = subprocess.Popen([SERVER, "1", "0", "5,4", meta], stdout=o, stderr=e, env=env, cwd=str(tmp_path))
        try:
            _wait_ready(server, fe if where == "stderr" else fo)
            client = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ipc_client_plain.py"), str(spec.F), "1", str(tmp_path / "c.json"), "0"],
                                    env=env, capture_output=True, text=True, timeout=300)
            assert client.returncode == 0, client.stdout[-1500:] + client.stderr[-1500:]
            server.wait(timeout=120)
        finally:
            if server.poll() is None:
                server.kill()
        out, err = open(fo).read(), open(fe).read()
        text, other = (err, out) if where == "stderr" else (out, err)
        assert "Train Steps:" in text and "System is ready for serving" in text and "Server Stopped" in text, (where, out[-800:], err[-800:])
        assert "Train Steps:" not in other and "ready for serving" not in other, (where, other[-800:])


def test_server_synth_dataset_source_refuses_a_wrong_meta_line(tmp_path, synth):
    """V / F / E / seed-set sizes of the meta line that are not the generator's: the server stops with an error, no trainer is ever posted."""
    spec = synth.spec_for("products", scale=0.004)
    for bad in ("synth:products:0.004 512 %d 0 %d 10 10 10 0 1 0" % (spec.V + 1, spec.F),
                "synth:products:0.004 512 %d 12345 %d 10 10 10 0 1 0" % (spec.V, spec.F),
                "synth:products:0.004 512 %d 0 %d %d 10 10 0 1 0" % (spec.V, spec.F, spec.n_train + 1),
                "synth:nothing 512 %d 0 %d 10 10 10 0 1 0" % (spec.V, spec.F)):
        meta = str(tmp_path / "meta_config")
        with open(meta, "w") as f:
            f.write(bad)
        env = dict(os.environ, LEGION_IPC_NAMESPACE="sybad%d_" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([SERVER, "1", "0", "5,4", meta], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "Server_Initialize" in r.stderr and "ready for serving" not in r.stdout, (bad, r.stdout[-500:], r.stderr[-500:])


def test_synchronize_with_queued_trainer_work(tmp_path, synth, oracle):
    """get_next reads the host mirror of the counters and no longer synchronises the trainer's device (the
    reference's blocking counter copy did, ipc_cuda_kernel.cu:195-196), and the reference trainers post with work still queued
    (legion_graphsage.py:93-116).  ipc_service.synchronize() must therefore wait for the device BEFORE it posts: a trainer that queues a
    long kernel in front of its reads of the batch and posts at once must still see every batch intact."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    B, epochs, fan = 128, 3, [10, 5]
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write("synth:products:0.004 %d %d %d %d %d %d %d 0 %d 0" % (B, spec.V, ds.E, spec.F, spec.n_train, spec.n_valid, 600, epochs))
    env = dict(os.environ, LEGION_IPC_NAMESPACE="qw%d_" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_CLIENT_QUEUED_WORK="1")
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, "1", "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT, env=env, cwd=str(tmp_path))
    try:
        _wait_ready(server, log)
        out = str(tmp_path / "client.json")
        client = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ipc_client.py"), str(spec.F), str(epochs), out],
                                env=env, capture_output=True, text=True, timeout=300)
        assert client.returncode == 0, client.stdout[-2000:] + client.stderr[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
        _audit_clean(open(log).read())
    finally:
        if server.poll() is None:
            server.kill()
    got = json.load(open(out))
    sets = {0: ds.train, 1: ds.valid, 2: ds.test[:600]}
    steps, tb, vb, sb = oracle.coordinate([len(sets[0])], [len(sets[1])], [len(sets[2])], B)
    bs = {0: int(tb[0]), 1: int(vb[0]), 2: int(sb[0])}
    assert bs[2] > B and bs[1] > B          # evaluation batches LARGER than the training batch: the feature buffer is sized for them too
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, max(bs.values()), fan)
    assert len(got["batches"]) == oracle.max_step(steps, epochs) > 8
    for rec in got["batches"]:
        mode, local = oracle.schedule(steps, epochs, rec["b"])
        ref = orc.run_batch(sets[mode], ds.labels[sets[mode]], local, mode=mode, batch_size=bs[mode])
        assert rec["ids"] == sha(ref["ids"]) and rec["features"] == sha(ref["features"]) and rec["labels"] == sha(ref["labels"]), rec["b"]
        assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"]), rec["b"]


@pytest.mark.parametrize("peer_gather", ["in-kernel", "exchange", "graph3"])
def test_two_gpu_server_unified_cache_two_trainers(tmp_path, synth, oracle, peer_gather):
    """`legion 2 1`: two logical GPUs in ONE server process (a thread per GPU, Server.cu:119-127), Kg = 2 unified
    cache with in-kernel peer reads (both logical GPUs map onto the box's single device), one trainer process
    per GPU.  Each trainer must see exactly its partition's batches (tid % 2 split, GPUGraphStore.cu:332-346).
    peer_gather = exchange: the same server with $LEGION_PEER_GATHER=exchange -- the peers' rows arrive as hipMemcpyPeerAsync
    bulk copies (peer_exchange.cpp), driven concurrently by the two runner threads, each launching on the other's device.
    graph3: in-kernel peer reads under $LEGION_BATCH_GRAPH=1 -- every runner thread replays its sampler graphs and launches one
    cached gather over all rows on its second stream."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    B, epochs, fan, G = 512, 1, [10, 5], 2    # >= 512: valid/test batches are up to 512 seeds (CUDA_IPC_Service.cu:101-117)
    budget = int(spec.V * spec.F * 4 * 0.1)
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write(synth.meta_config_line(ds, data, B, budget, epochs, 0))
    ns = "g2_%d_%s_" % (os.getpid(), peer_gather[:2])
    # host tables: with HBM replicas the server would (rightly) not build a cache at all
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_TABLES="host")
    if peer_gather == "exchange":
        env["LEGION_PEER_GATHER"] = "exchange"
    if peer_gather == "graph3":
        env["LEGION_BATCH_GRAPH"] = "1"
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, str(G), "1", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT,
                                  env=env, cwd=str(tmp_path))
    clients = []
    try:
        _wait_ready(server, log)
        for g in range(G):
            out = str(tmp_path / ("client%d.json" % g))
            clients.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ipc_client.py"), str(spec.F), str(epochs), out],
                                                  env=dict(env, LEGION_IPC_DEVICE=str(g)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for out, c in clients:
            stdout, _ = c.communicate(timeout=300)
            assert c.returncode == 0, stdout[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
    finally:
        for _, c in clients:
            if c.poll() is None:
                c.kill()
        if server.poll() is None:
            server.kill()
    parts = {0: oracle.split_seeds(ds.train, G), 1: oracle.split_seeds(ds.valid, G), 2: oracle.split_seeds(ds.test, G)}
    steps, tb, vb, sb = oracle.coordinate([len(p) for p in parts[0]], [len(p) for p in parts[1]], [len(p) for p in parts[2]], B)
    text = open(log).read()
    assert "xGMI Clique: 1 GPU Per Clique: 2" in text and "Feat capacity" in text   # the cost model sized a real cache
    audit = _audit_clean(text, G)     # two runner threads, shards + fragments on their owners, peer reads / bulk copies between them: clean
    assert audit is None or audit["peer_launches"] > 0, audit     # ... and the cached gathers / fill-ups really read a peer's memory
    assert ("peer exchange gather:" in text) == (peer_gather == "exchange"), text[-1500:]
    H = len(fan)
    for g in range(G):
        got = json.load(open(clients[g][0]))
        assert got["steps"] == steps.tolist() and len(got["batches"]) == oracle.max_step(steps, epochs)
        bs = {0: int(tb[g]), 1: int(vb[g]), 2: int(sb[g])}
        orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
        for rec in got["batches"]:
            mode, local = oracle.schedule(steps, epochs, rec["b"])
            ids = parts[mode][g]
            ref = orc.run_batch(ids, ds.labels[ids], local, mode=mode, batch_size=bs[mode])
            assert rec["n"] == ref["nc"][5 + 2 * H]
            assert rec["ids"] == sha(ref["ids"]) and rec["features"] == sha(ref["features"]) and rec["labels"] == sha(ref["labels"])
            assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"])


@pytest.mark.parametrize("model", ["sage", "gcn"])
def test_torch_trainer_learns_from_served_batches(tmp_path, synth, model):
    """examples/legion_sage_torch.py (the reference trainer's loop, legion_graphsage.py:72-172, with a DGL-free mean
    aggregator) against the server binary: labels are made recoverable from the features, so a few dozen steps on
    the served blocks must beat chance (1/47) by a wide margin -- ids, features, labels and COO blocks are consistent."""
    spec = synth.spec_for("products", scale=0.02)
    ds = synth.generate(spec)
    ds.features[np.arange(spec.V), ds.labels % spec.F] += 3.0
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    B, epochs, fan = 512, 6, [10, 5]
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write(synth.meta_config_line(ds, data, B, 1 << 40, epochs, 0))
    ns = "tr%s_%d_" % (model, os.getpid())
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", LEGION_BATCH_GRAPH="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "legion-1_amd", "ipc_service"), os.environ.get("PYTHONPATH", "")]))
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, "1", "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT,
                                  env=env, cwd=str(tmp_path))
    try:
        _wait_ready(server, log)
        tr = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "legion_sage_torch.py"), "--features_num", str(spec.F),
                             "--class_num", str(spec.classes), "--hidden_dim", "64", "--learning_rate", "0.01", "--drop_rate", "0.1",
                             "--epoch", str(epochs), "--model", model, "--seed", "0"], env=env, capture_output=True, text=True, timeout=600)
        assert tr.returncode == 0, tr.stdout[-2000:] + tr.stderr[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
        _audit_clean(open(log).read())
    finally:
        if server.poll() is None:
            server.kill()
    lines = [l for l in tr.stdout.splitlines() if l.startswith("Epoch:")]
    assert len(lines) == epochs, tr.stdout
    acc = float(tr.stdout.split("Accuracy on test data:")[1].split()[0])
    losses = [float(l.split("Train Loss:")[1].split(",")[0]) for l in lines]
    if model == "sage":
        assert acc > 0.2, tr.stdout      # chance: 1/47
    else:   # GraphConv has no self term: a node's own (label-bearing) features never reach its output; the loss still falls
        assert np.isfinite(losses).all() and losses[-1] < losses[0], tr.stdout


def test_torch_link_prediction_trainer_on_triple_seeds(tmp_path, synth):
    """lp_sage.py's loop (three thirds [src | pos | neg] per batch, lp_sage.py:87-90) on a `trainingset` written by
    synth.lp_trainingset.  Positives are neighbours (skewed towards hot = small ids), negatives uniform ids; one
    feature channel tells how hot a node is, so the pairwise loss must fall below its untrained value 2 ln 2."""
    spec = synth.spec_for("products", scale=0.02)
    ds = synth.generate(spec)
    ds.features[:, 0] = 3.0 * (1.0 - (np.arange(spec.V) / spec.V) ** (1.0 / 3.0))
    B, epochs, fan = 513, 4, [10, 5]
    ds.train = synth.lp_trainingset(ds, 171 * 12, B)          # 12 full batches of 171 triples
    ds.spec = __import__("dataclasses").replace(ds.spec, n_train=len(ds.train))
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write(synth.meta_config_line(ds, data, B, 1 << 40, epochs, 0))
    ns = "lp_%d_" % os.getpid()
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "legion-1_amd", "ipc_service"), os.environ.get("PYTHONPATH", "")]))
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, "1", "0", ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT,
                                  env=env, cwd=str(tmp_path))
    try:
        _wait_ready(server, log)
        tr = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "legion_sage_torch.py"), "--task", "lp", "--features_num",
                             str(spec.F), "--class_num", "32", "--hidden_dim", "64", "--learning_rate", "0.01", "--drop_rate", "0.0",
                             "--epoch", str(epochs), "--seed", "0"], env=env, capture_output=True, text=True, timeout=600)
        assert tr.returncode == 0, tr.stdout[-2000:] + tr.stderr[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
        _audit_clean(open(log).read())
    finally:
        if server.poll() is None:
            server.kill()
    losses = [float(l.split("Train Loss:")[1].split(",")[0]) for l in tr.stdout.splitlines() if l.startswith("Epoch:")]
    assert len(losses) == epochs and losses[-1] < 1.2 < 2 * np.log(2) + 0.2, tr.stdout


def _serve(tmp_path, spec, meta_line, fan, G, agg_mode, epochs, extra_env=None, client_env=None, client="ipc_client.py"):
    """Start `legion G agg_mode fan meta`, one ipc_client per GPU; returns ([client json per GPU], server log text)."""
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write(meta_line)
    ns = "s%d_%d_" % (os.getpid(), abs(hash(str(tmp_path))) % 100000)
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([SERVER, str(G), str(agg_mode), ",".join(map(str, fan)), meta], stdout=lf, stderr=subprocess.STDOUT,
                                  env=env, cwd=str(tmp_path))
    clients = []
    try:
        _wait_ready(server, log)
        for g in range(G):
            out = str(tmp_path / ("client%d.json" % g))
            clients.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", client), str(spec.F), str(epochs), out],
                                                  env=dict(env, LEGION_IPC_DEVICE=str(g), **(client_env or {})), stdout=subprocess.PIPE,
                                                  stderr=subprocess.STDOUT, text=True)))
        for out, c in clients:
            stdout, _ = c.communicate(timeout=300)
            assert c.returncode == 0, stdout[-3000:]
        server.wait(timeout=60)
        assert server.returncode == 0, open(log).read()[-3000:]
    finally:
        for _, c in clients:
            if c.poll() is None:
                c.kill()
        if server.poll() is None:
            server.kill()
    _audit_clean(open(log).read(), G)
    return [json.load(open(out)) for out, _ in clients], open(log).read()


def test_two_gpu_server_partition_file_split(tmp_path, synth, oracle):
    """meta_config flag 1 + partition_2_bn: training seeds go to the GPU the partition file names, validation / test
    still by tid % G (GPUGraphStore.cu:332-346).  The file is NOT tid % 2, so a server that ignored it would fail."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    G, B, epochs, fan = 2, 512, 1, [10, 5]
    part = ((np.arange(spec.V, dtype=np.int64) // 7) % G).astype("<i4")
    part.tofile(os.path.join(data, "partition_%d_bn" % G))
    got, text = _serve(tmp_path, spec, synth.meta_config_line(ds, data, B, 1 << 40, epochs, 1), fan, G, 0, epochs)
    assert "Partition?:         1" in text
    tr = oracle.split_seeds(ds.train, G, part, 1)
    assert not all(np.array_equal(a, b) for a, b in zip(tr, oracle.split_seeds(ds.train, G)))
    parts = {0: tr, 1: oracle.split_seeds(ds.valid, G), 2: oracle.split_seeds(ds.test, G)}
    steps, tb, vb, sb = oracle.coordinate([len(p) for p in parts[0]], [len(p) for p in parts[1]], [len(p) for p in parts[2]], B)
    for g in range(G):
        assert got[g]["steps"] == steps.tolist() and len(got[g]["batches"]) == oracle.max_step(steps, epochs)
        bs = {0: int(tb[g]), 1: int(vb[g]), 2: int(sb[g])}
        orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
        for rec in got[g]["batches"]:
            mode, local = oracle.schedule(steps, epochs, rec["b"])
            ids = parts[mode][g]
            ref = orc.run_batch(ids, ds.labels[ids], local, mode=mode, batch_size=bs[mode])
            assert rec["ids"] == sha(ref["ids"]) and rec["labels"] == sha(ref["labels"]) and rec["features"] == sha(ref["features"])
            assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"])


def test_two_gpu_server_link_prediction_lists_keep_their_thirds(tmp_path, synth, oracle):
    """Link prediction on G = 2 GPUs: the reference's `tid % G` split of one seed file shreds the [src | pos | neg]
    thirds lp_sage.py:87-90 relies on, so the lists are written per GPU (triples dealt by src % G, synth.lp_trainingset)
    and served verbatim (meta flag 2).  Every train batch a trainer receives must be its list's batch: src third on
    its own GPU, pos third neighbours of the src third, and the batch bit-identical to the oracle on that list."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    G, B, epochs, fan = 2, 513, 1, [10, 5]
    k, n_triples = B // 3, len(ds.train)
    lists = [synth.lp_trainingset(ds, n_triples, B, rank=g, world=G) for g in range(G)]
    for g in range(G):
        lists[g].astype("<i4").tofile(os.path.join(data, "trainingset_%d_%d" % (G, g)))
    got, text = _serve(tmp_path, spec, synth.meta_config_line(ds, data, B, 1 << 40, epochs, 2), fan, G, 0, epochs,
                       client_env={"LEGION_CLIENT_DUMP_SEEDS": "1"})
    assert "Partition?:         2" in text
    parts = {0: lists, 1: oracle.split_seeds(ds.valid, G), 2: oracle.split_seeds(ds.test, G)}
    steps, tb, vb, sb = oracle.coordinate([len(p) for p in lists], [len(p) for p in parts[1]], [len(p) for p in parts[2]], B)
    assert steps[0] >= 2
    for g in range(G):
        bs = {0: int(tb[g]), 1: int(vb[g]), 2: int(sb[g])}
        orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
        n_train = 0
        for rec in got[g]["batches"]:
            mode, local = oracle.schedule(steps, epochs, rec["b"])
            ids = parts[mode][g]
            ref = orc.run_batch(ids, ds.labels[ids], local, mode=mode, batch_size=bs[mode])
            assert rec["ids"] == sha(ref["ids"]) and rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"])
            if mode != 0:
                continue
            n_train += 1
            seeds = np.array(rec["seeds"], dtype=np.int64)
            assert np.array_equal(seeds, lists[g][local * B:(local + 1) * B])
            src, pos = seeds[:k], seeds[k:2 * k]
            assert (src % G == g).all()
            for s_, p_ in zip(src[:64], pos[:64]):
                row = ds.indices[ds.indptr[s_]:ds.indptr[s_ + 1]]
                assert p_ == s_ or p_ in row
        assert n_train == steps[0]


@pytest.mark.parametrize("client", ["ipc_client.py", "ipc_client_plain.py"])
def test_chunked_feature_handoff_buffer(tmp_path, synth, oracle, client):
    """A feature hand-off buffer above the HIP-IPC size limit (2^31 bytes under the PyTorch-bundled runtime) is built from
    chunks, exported as file descriptors and mapped contiguously by the trainer (ipc_env.cpp, VmmDesc).  Forced here with a
    tiny limit and 1 MiB chunks (a ~3 MB buffer = 3 chunks per pipe); the PyTorch client imports through the bundled ROCm 7.0
    runtime (descriptor by pointer), the plain ctypes client through the system runtime (by value)."""
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    B, epochs, fan = 512, 1, [10, 5]
    got, text = _serve(tmp_path, spec, synth.meta_config_line(ds, data, B, 1 << 40, epochs, 0), fan, 1, 0, epochs,
                       extra_env={"LEGION_IPC_MAX_BYTES": "1000000", "LEGION_SHARD_CHUNK_BYTES": "1048576"}, client=client)
    steps, tb, vb, sb = oracle.coordinate([len(ds.train)], [len(ds.valid)], [len(ds.test)], B)
    bs = {0: int(tb[0]), 1: int(vb[0]), 2: int(sb[0])}
    sets = {0: ds.train, 1: ds.valid, 2: ds.test}
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
    assert len(got[0]["batches"]) == oracle.max_step(steps, epochs)
    for rec in got[0]["batches"]:
        mode, local = oracle.schedule(steps, epochs, rec["b"])
        ref = orc.run_batch(sets[mode], ds.labels[sets[mode]], local, mode=mode, batch_size=bs[mode])
        assert rec["n"] == ref["nc"][9] and rec["n"] * spec.F * 4 > 1048576        # the rows span several chunks
        assert rec["ids"] == sha(ref["ids"]) and rec["features"] == sha(ref["features"]) and rec["labels"] == sha(ref["labels"])
        assert rec["src"] == sha(ref["src_off"]) and rec["dst"] == sha(ref["dst_off"])


def test_three_gigabyte_feature_buffer_reaches_a_pytorch_process(tmp_path):
    """6.5 M rows x 128 floats = 3.1 GiB per pipe: above the 2^31-byte limit at which the PyTorch-bundled runtime's
    hipIpcOpenMemHandle hangs (profiles/r02_ipc_limit.md).  The server side (no PyTorch: system runtime) builds it from
    1 GiB chunks; the PyTorch process attaches through legion_ipc_client_open and finds the right rows at the start, the
    end and on both sides of every chunk seam."""
    rows, F = 6_500_000, 128
    assert rows * F * 4 > 2 ** 31
    ns = "big%d_" % os.getpid()
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = os.path.join(ROOT, "tests", "handoff_big.py")
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:      # a file, polled with a deadline: never block on the server's pipe
        server = subprocess.Popen([sys.executable, script, "server", str(rows), str(F)], env=env, stdout=lf, stderr=subprocess.STDOUT)
    try:
        t0 = time.time()
        while "ready" not in open(log, errors="ignore").read():
            assert server.poll() is None and time.time() - t0 < 120, open(log, errors="ignore").read()[-2000:]
            time.sleep(0.2)
        client = subprocess.run([sys.executable, script, "client", str(rows), str(F)], env=env, capture_output=True, text=True, timeout=180)
        assert client.returncode == 0 and "0 mismatches" in client.stdout, client.stdout[-2000:] + client.stderr[-2000:]
        server.wait(timeout=60)
        assert server.returncode == 0 and "server: done" in open(log).read(), open(log).read()[-2000:]
    finally:
        if server.poll() is None:
            server.kill()


@pytest.mark.parametrize("pinned", [True, False])
def test_counter_mirror_serves_every_kind_of_producer(tmp_path, pinned):
    """The slab's host mirror of the counters (what `ipc_service.get_next` reads instead of the reference's blocking device copy): a producer
    that only calls IPCEnv_IPCPost (a reference-style RunOnce), one that queues IPCEnv_MirrorCounters on its stream (the runner), and a
    host-decided mirror (poisoned pipe) -- the client gets the right words each time, and the IPC device buffers 5 / 6 still hold the
    counters for a trainer that reads them the reference's way."""
    ns = "mir%d_%d_" % (os.getpid(), pinned)
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not pinned:      # a runtime that refuses to page-lock the slab: queued copies go through pinned staging words instead
        env["LEGION_IPC_NO_PIN"] = "1"
    script = os.path.join(ROOT, "tests", "ipc_mirror.py")
    log = str(tmp_path / "server.log")
    with open(log, "w") as lf:
        server = subprocess.Popen([sys.executable, script, "server"], env=env, stdout=lf, stderr=subprocess.STDOUT)
    try:
        t0 = time.time()
        while "ready" not in open(log, errors="ignore").read():
            assert server.poll() is None and time.time() - t0 < 120, open(log, errors="ignore").read()[-2000:]
            time.sleep(0.2)
        client = subprocess.run([sys.executable, script, "client"], env=env, capture_output=True, text=True, timeout=120)
        assert client.returncode == 0 and client.stdout.count("mirror ok, device buffers ok") == 3, client.stdout[-2000:] + client.stderr[-2000:]
        server.wait(timeout=60)
        assert server.returncode == 0 and "server: done" in open(log).read(), open(log).read()[-2000:]
        assert ("slab pinned = %d" % pinned) in open(log).read()   # pinned: the queued copies are real asynchronous DMA into the slab on this runtime
    finally:
        if server.poll() is None:
            server.kill()


def test_client_open_refuses_a_chunk_descriptor_without_listener_gpu():
    """Same refusal as the CPU test, with a real device behind it: no mapping may survive (the reserve / import / map steps
    that ran are undone) and the process can still allocate afterwards.  Runs in a child: the trainer half is per process."""
    code = ("import sys, os; sys.path[:0] = [%r, %r]\n"
            "from test_host_logic import _client_open_without_listener\n"
            "import legion1_amd.capi as K\n"
            "K.lib().SetGPUDevice(0)\n"
            "_client_open_without_listener('lgn_t_nolisten_gpu_%%d_' %% os.getpid())\n"
            "b = K.DevBuf(1 << 20); b.free(); print('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
