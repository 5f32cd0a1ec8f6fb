"""A trainer-side consumer WITHOUT PyTorch: the C ABI client of liblegion_amd.so through ctypes ("a non-PyTorch
consumer").  Such a process runs on the system HIP runtime (ROCm 7.2), a PyTorch trainer on the runtime bundled with the torch
wheel (7.0) -- the two differ in how a chunked hand-off buffer is imported.
usage: ipc_client_plain.py <feature_dim> <epochs> <out.json> [<b0,b1,...>]   (the global batch numbers to record; default: all --
the others are consumed like a null trainer: wait, read the counters, post)"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import legion1_amd.capi as K  # noqa: E402

assert "torch" not in sys.modules


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    F, epochs, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    only = set(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else None
    L = K.lib()
    L.SetGPUDevice(0)
    c = C.c_void_p(L.legion_ipc_client_open(-1))
    K.check()
    steps = (C.c_int32 * 3)()
    L.legion_ipc_client_steps(c, steps)
    H = L.legion_ipc_client_hops(c)
    total = (steps[0] + steps[1]) * epochs + steps[2]
    nc, ec = (C.c_int32 * 16)(), (C.c_int32 * 16)()
    recs = []
    for b in range(total):
        L.legion_ipc_client_wait(c)
        L.legion_ipc_client_read_counters(c, nc, ec)
        n, e = nc[5 + 2 * H], ec[2 + H]
        if only is not None and b not in only:
            L.legion_ipc_client_post(c)
            continue
        ids = K.read_dev(L.legion_ipc_client_buffer(c, 0), np.int32, n)
        feats = K.read_dev(L.legion_ipc_client_buffer(c, 1), np.float32, n * F).reshape(n, F)
        labels = K.read_dev(L.legion_ipc_client_buffer(c, 2), np.int32, nc[5])
        src = K.read_dev(L.legion_ipc_client_buffer(c, 3), np.int32, e)
        dst = K.read_dev(L.legion_ipc_client_buffer(c, 4), np.int32, e)
        recs.append(dict(b=b, n=int(n), e=int(e), nc=list(nc), ec=list(ec), ids=sha(ids), features=sha(feats), labels=sha(labels), src=sha(src), dst=sha(dst)))
        if os.environ.get("LEGION_CLIENT_DUMP_IDS"):     # the test recomputes the expected rows from the ids (generator closed form)
            np.save(out + ".ids%d.npy" % b, ids)
        L.legion_ipc_client_post(c)
    L.legion_ipc_client_close(c)
    assert "torch" not in sys.modules
    with open(out, "w") as f:
        json.dump(dict(steps=list(steps), hops=H, batches=recs), f)


if __name__ == "__main__":
    main()
