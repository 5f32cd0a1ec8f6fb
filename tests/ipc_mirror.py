"""The host mirror of the hand-off counters (extension behind the reference's shm struct), with a hand-made producer:
  ipc_mirror.py server    batch A on pipe 0 is posted WITHOUT IPCEnv_MirrorCounters (a reference-style producer: IPCEnv_IPCPost copies
                          synchronously), batch B on pipe 1 WITH the queued copy on a stream, batch C on pipe 0 via IPCEnv_SetMirror
  ipc_mirror.py client    reads each batch's counters through legion_ipc_client_read_counters (the mirror) AND, like the reference's
                          trainer, straight from the IPC device buffers 5 / 6: both must be the producer's values"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import legion1_amd.capi as K  # noqa: E402

L = K.lib()
L.SetGPUDevice(0)

def counters(tag):
    return (np.arange(16, dtype=np.int32) + 100 * tag), (np.arange(16, dtype=np.int32) * 3 + 1000 * tag)


if sys.argv[1] == "server":
    env = L.NewIPCEnv(1)
    L.IPCEnv_InitializeSamplesBuffer(env, 16, 1024, 4, 0, 2)
    L.IPCEnv_InitializeFeaturesBuffer(env, 0, 64, 4, 0, 2)
    K.check()
    stream = L.d_stream_create()
    print("server: ready, slab pinned = %d" % L.IPCEnv_SlabPinned(env), flush=True)
    for tag, pipe, how in ((1, 0, "post"), (2, 1, "queued"), (3, 0, "host")):
        L.IPCEnv_IPCWait(env, 0, pipe)                     # the client frees both pipes when it attaches, then one per batch
        nc, ec = counters(tag)
        L.d_copy_h_2_d(L.IPCEnv_GetNodeCounter(env, 0, pipe), nc.ctypes.data, 64)
        L.d_copy_h_2_d(L.IPCEnv_GetEdgeCounter(env, 0, pipe), ec.ctypes.data, 64)
        if how == "queued":
            L.IPCEnv_MirrorCounters(env, 0, pipe, stream)
            L.d_stream_sync(stream)
        elif how == "host":                                 # what a poisoned pipe does: the host decides the mirror
            L.IPCEnv_SetMirror(env, 0, pipe, -1, 0)
        K.check()
        L.IPCEnv_IPCPost(env, 0, pipe)
    L.IPCEnv_IPCWait(env, 0, 1)
    L.IPCEnv_Finalize(env)
    print("server: done", flush=True)
else:
    c = C.c_void_p(L.legion_ipc_client_open(0))
    K.check()
    bad = 0
    for tag, how in ((1, "post"), (2, "queued"), (3, "host")):
        L.legion_ipc_client_wait(c)
        nc, ec = (C.c_int32 * 16)(), (C.c_int32 * 16)()
        L.legion_ipc_client_read_counters(c, nc, ec)
        want_nc, want_ec = counters(tag)
        dev_nc = K.read_dev(L.legion_ipc_client_buffer(c, 5), np.int32, 16)    # the reference trainer's way (ipc_cuda_kernel.cu:195-196)
        dev_ec = K.read_dev(L.legion_ipc_client_buffer(c, 6), np.int32, 16)
        ok_dev = np.array_equal(dev_nc, want_nc) and np.array_equal(dev_ec, want_ec)
        if how == "host":
            ok_mirror = list(nc) == [-1] * 16 and list(ec) == [0] * 16
        else:
            ok_mirror = list(nc) == want_nc.tolist() and list(ec) == want_ec.tolist()
        print("client: batch %d (%s): mirror %s, device buffers %s" % (tag, how, "ok" if ok_mirror else "WRONG " + str(list(nc)), "ok" if ok_dev else "WRONG"), flush=True)
        bad += int(not ok_mirror) + int(not ok_dev)
        L.legion_ipc_client_post(c)
    L.legion_ipc_client_close(c)
    sys.exit(1 if bad else 0)
