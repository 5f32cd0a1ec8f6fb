"""Boundary B2 against the reference's REAL code: `make -C oracle ref` compiles /root/reference/src/helper_multiprocess.cpp -- the one plain
C++ file of the reference's path, the shm-slab helper its server and its trainer extension both call -- from where it lies into oracle/_ref/
(git-ignored; it travels to the GPU box as a built library, the reference tree does not).  tests/ref_shm_compat.py drives it against the
product's slab in both directions.  Skipped where the library was never built (no reference tree)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libhelper_multiprocess_ref.so")


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (make -C oracle ref needs /root/reference)")
def test_slab_is_the_references_own_struct_in_both_directions():
    ns = "refshm%d_" % os.getpid()
    env = {k: v for k, v in os.environ.items() if k not in ("LEGION_IPC_NAMESPACE", "LEGION_IPC_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_shm_compat.py"), ns], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "REF_SHM_COMPAT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert not [f for f in os.listdir("/dev/shm") if ns in f]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_LIB), reason="oracle/_ref not built (make -C oracle ref needs /root/reference)")
def test_reference_helper_reads_a_live_servers_handle_table():
    """The same check with a REAL device behind the slab (the CPU test above runs with $LEGION_IPC_NO_DEVICE=1, where every handle slot is zero):
    the product's server half registers its 7 hand-off buffers x 2 pipes of GPU 0 with hipMalloc + hipIpcGetMemHandle; the reference's own
    sharedMemoryCreate (what its trainer extension calls, ipc_cuda_kernel.cu:44-51 -> helper_multiprocess.cpp:5-47) then maps the object:
    still exactly 7180 bytes, the step counts where the reference reads them, 14 distinct non-zero 64-byte handles in the rows of GPU 0 and
    nothing in the rows of the GPUs this server does not drive (ipc_cuda_kernel.cu:63-69 indexes memHandle[device][pipe][buffer])."""
    import ctypes as C
    import numpy as np
    import legion1_amd.capi as K
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    L.SetGPUDevice(0)
    ns = "refshm_gpu%d_" % os.getpid()
    L.legion_ipc_set_namespace(ns.encode())
    ref = C.CDLL(REF_LIB)

    class ShmInfo(C.Structure):
        _fields_ = [("addr", C.c_void_p), ("size", C.c_size_t), ("fd", C.c_int)]
    create, close = ref._Z18sharedMemoryCreatePKcmP19sharedMemoryInfo_st, ref._Z17sharedMemoryCloseP19sharedMemoryInfo_st
    create.argtypes, create.restype = [C.c_char_p, C.c_size_t, C.POINTER(ShmInfo)], C.c_int
    close.argtypes = [C.POINTER(ShmInfo)]
    slab = 12 + 8 * 2 * 7 * 64
    path = "/dev/shm/" + ns + "simpleIPCshm"
    e = C.c_void_p(L.NewIPCEnv(2))
    try:
        info = K.LegionBuildInfo()
        nums = [np.array([4001, 4002], np.int32), np.array([600, 600], np.int32), np.array([100, 100], np.int32)]
        info.partition_count, info.epoch, info.raw_batch_size = 2, 3, 500
        info.training_set_num, info.validation_set_num, info.testing_set_num = [a.ctypes.data for a in nums]
        L.IPCEnv_Coordinate(e, C.byref(info))
        L.IPCEnv_InitializeSamplesBuffer(e, 500, 20000, 16, 0, 2)            # GPU 0 only: GPU 1's rows must stay empty
        L.IPCEnv_InitializeFeaturesBuffer(e, 0, 20000, 16, 0, 2)
        K.check()
        si = ShmInfo()
        assert create((ns + "simpleIPCshm").encode(), slab, C.byref(si)) == 0 and si.addr
        assert os.stat(path).st_size == slab
        assert list((C.c_int32 * 3).from_address(si.addr)) == [8, 2, 1]
        table = np.frombuffer((C.c_ubyte * (8 * 2 * 7 * 64)).from_address(si.addr + 12), np.uint8).reshape(8, 2, 7, 64)
        assert table[0].reshape(14, 64).any(axis=1).all(), "every buffer of GPU 0 carries a handle"
        assert len({bytes(h) for h in table[0].reshape(14, 64)}) == 14, "14 distinct allocations"
        assert not table[1:].any(), "GPUs without registered buffers: zero rows"
        close(C.byref(si))
    finally:
        L.IPCEnv_Finalize(e)
        L.legion_ipc_set_namespace(b"")
        K.check()
    assert not [f for f in os.listdir("/dev/shm") if ns in f]
