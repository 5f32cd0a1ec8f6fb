"""A feature hand-off buffer ABOVE 2^31 bytes from a standalone server-side process (system HIP runtime) to a PyTorch process
(runtime bundled with the torch wheel, whose hipIpcOpenMemHandle hangs at that size): profiles/r02_ipc_limit.md.
  handoff_big.py server <rows> <F>      allocates both pipes, fills pipe 0 with the synthetic feature rows 0..rows-1, posts it
  handoff_big.py client <rows> <F>      (imports torch first) attaches through legion_ipc_client_open and checks the rows"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
role, rows, F = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if role == "client":
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
import legion1_amd.capi as K  # noqa: E402
import legion1_amd.synth as S  # noqa: E402

L = K.lib()
L.SetGPUDevice(0)
if role == "server":
    assert "torch" not in sys.modules
    env = L.NewIPCEnv(1)
    L.IPCEnv_InitializeSamplesBuffer(env, 16, 1024, F, 0, 2)
    L.IPCEnv_InitializeFeaturesBuffer(env, 0, rows, F, 0, 2)
    K.check()
    p = L.IPCEnv_GetFloatFeatures(env, 0, 0)
    L.legion_synth_features(None, p, 0, rows, F)
    L.d_stream_sync(None)
    K.check()
    print("server: %d rows x %d = %.2f GiB per pipe ready" % (rows, F, rows * F * 4 / 2 ** 30), flush=True)
    L.IPCEnv_IPCWait(env, 0, 0)          # the client posts sem_r for both pipes when it attaches
    L.IPCEnv_IPCPost(env, 0, 0)          # pipe 0 is filled
    L.IPCEnv_IPCWait(env, 0, 0)          # the client is done with it
    L.IPCEnv_Finalize(env)
    print("server: done", flush=True)
else:
    c = C.c_void_p(L.legion_ipc_client_open(0))
    K.check()
    assert c.value
    L.legion_ipc_client_wait(c)
    ptr = L.legion_ipc_client_buffer(c, 1)
    spec = S.spec_for("papers100M", F=F)
    chunk_rows = (1 << 30) // (F * 4)
    probes = [0, 1, rows - 2, rows - 1] + [k * chunk_rows + d for k in range(1, rows // chunk_rows + 1) for d in (-1, 0) if k * chunk_rows + d < rows]
    bad = 0
    for r in probes:                                         # first / last rows and the rows at every 1 GiB chunk seam
        got = K.read_dev(ptr + r * F * 4, np.float32, F)
        bad += int(not np.array_equal(got, S.features(spec, np.array([r]))[0]))
    # single words at a fixed stride from the start of the buffer
    n = rows * F
    stride = 1 << 18
    idx = np.arange(0, n, stride, dtype=np.int64)
    vals = np.array([K.read_dev(ptr + int(i) * 4, np.float32, 1)[0] for i in idx[:64]])
    ref = S.features(spec, idx[:64] // F)[np.arange(64), idx[:64] % F]
    bad += int(not np.array_equal(vals, ref))
    print("client: %.2f GiB attached, %d probes, %d mismatches" % (n * 4 / 2 ** 30, len(probes) + 1, bad), flush=True)
    L.legion_ipc_client_post(c)
    L.legion_ipc_client_close(c)
    sys.exit(1 if bad else 0)
