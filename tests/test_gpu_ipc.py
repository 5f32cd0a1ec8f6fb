"""End-to-end drop-in boundary on the GPU: the `legion` server binary (meta_config + raw dataset
files, pre-sampling epoch, cache build, run loop) hands batches to a trainer-side process that uses
the `ipc_service` Python module -- POSIX shm + named semaphores + HIP IPC handles.
Every batch the trainer sees is compared with the oracle."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT, sha

pytestmark = pytest.mark.gpu

SERVER = os.path.join(ROOT, "legion-1_amd", "csrc", "legion")


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


@pytest.mark.parametrize("fan,budget_frac", [([25, 10], 0.2), ([5, 4, 3], 10.0)])
def test_server_binary_to_ipc_service(tmp_path, synth, oracle, fan, budget_frac):
    assert os.path.exists(SERVER), "build the server: make -C legion-1_amd/csrc legion"
    spec = synth.spec_for("products", scale=0.004)
    ds = synth.generate(spec)
    data = str(tmp_path / "ds") + "/"
    synth.write_legion_files(ds, data)
    B, epochs = 512, 2
    budget = int(spec.V * spec.F * 4 * budget_frac)
    meta = str(tmp_path / "meta_config")
    with open(meta, "w") as f:
        f.write(synth.meta_config_line(ds, data, B, budget, epochs, 0))
    ns = "t%d_%d_" % (os.getpid(), len(fan))
    env = dict(os.environ, LEGION_IPC_NAMESPACE=ns, HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    finally:
        if server.poll() is None:
            server.kill()
    got = json.load(open(out))
    H = len(fan)
    assert got["hops"] == H
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
