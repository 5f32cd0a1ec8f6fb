"""The hand-off protocol of csrc/ipc_env.cpp on the CPU: a fake producer (IPCEnv_*) and one fake consumer process per
logical GPU (legion_ipc_client_*) run the slab / semaphore / counter-mirror / poisoned-pipe protocol with $LEGION_IPC_NO_DEVICE=1 -- no GPU
buffer, zero handle slots.  tests/ipc_env_cpu.py is the workload (its own processes: the no-device switch is read once per process); the same
script runs under the host-sanitizer build in profiles/r05_robustness.sh."""
import os
import subprocess
import sys

from conftest import ROOT


def test_slab_semaphores_mirror_and_poisoned_pipe_without_a_gpu():
    ns = "cpuipc%d_" % os.getpid()
    env = {k: v for k, v in os.environ.items() if k not in ("LEGION_IPC_NAMESPACE", "LEGION_IPC_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ipc_env_cpu.py"), "producer", ns], env=env,
                       capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "PRODUCER_OK 40 batches x 2 consumers + poisoned pipe" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert not [f for f in os.listdir("/dev/shm") if ns in f]          # nothing left behind


def test_a_consumer_without_a_server_is_refused_or_blocks_nowhere():
    """A client that opens before any server created the slab: the slab is created empty (the reference does the same,
    ipc_cuda_kernel.cu:45), steps are 0, every handle slot is zero -- nothing is opened, nothing faults; closing removes nothing
    a server would need."""
    ns = "cpuipc_nosrv%d_" % os.getpid()
    code = ("import os, sys, ctypes as C; sys.path.insert(0, %r)\n"
            "os.environ['LEGION_IPC_NO_DEVICE'] = '1'; os.environ['LEGION_IPC_NAMESPACE'] = %r\n"
            "import legion1_amd.capi as K\n"
            "L = K.lib(); L.legion_set_error_mode(K.ERR_RETURN)\n"
            "c = C.c_void_p(L.legion_ipc_client_open(-1)); K.check(); assert c.value\n"
            "s = (C.c_int32 * 3)(); L.legion_ipc_client_steps(c, s); assert list(s) == [0, 0, 0]\n"
            "assert L.legion_ipc_client_hops(c) == 2 and L.legion_ipc_client_feature_rows(c) == 0\n"
            "assert all(not L.legion_ipc_client_buffer(c, w) for w in range(7))\n"
            "L.legion_ipc_client_close(c); K.check(); print('OK')\n") % (ROOT, ns)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr[-2000:]
    for f in os.listdir("/dev/shm"):                                    # the client's empty slab and semaphores: test litter, remove
        if ns in f:
            os.unlink(os.path.join("/dev/shm", f))


def test_a_stale_extension_object_is_ignored_and_a_killed_servers_leftovers_can_be_removed():
    """ADVICE r05 (low x 2).  (a) The extension object `<name>_ext` (hop count, counter mirror, row bound) is only unlinked by IPCEnv_Finalize:
    after a KILLED product server, a server without the extension (the reference's) re-creates the slab and leaves the stale object in place --
    a client then read stale hops and counters for good.  The extension now carries a copy of the slab's step counts and a checksum of the first
    handle per GPU; a client ignores one that does not match the slab it attached to and behaves as against a reference server (2 hops).
    (b) legion_ipc_unlink_namespace removes what the killed server left in /dev/shm: slab, extension object, 2 x depth semaphores per GPU."""
    ns = "cpuipc_stale%d_" % os.getpid()
    pre = ("import os, sys, ctypes as C; sys.path.insert(0, %r)\n"
           "os.environ['LEGION_IPC_NO_DEVICE'] = '1'; os.environ['LEGION_IPC_NAMESPACE'] = %r\n"
           "import legion1_amd.capi as K\n"
           "L = K.lib(); L.legion_set_error_mode(K.ERR_RETURN)\n") % (ROOT, ns)
    # a product server of 3 hops that is killed right after it became ready (no Finalize)
    server = pre + ("import numpy as np\n"
                    "e = L.NewIPCEnv(1)\n"
                    "info = K.LegionBuildInfo(); info.partition_count = 1; info.epoch = 1; info.raw_batch_size = 500\n"
                    "tr, va, te = (np.array([x], np.int32) for x in (3601, 700, 300))\n"
                    "info.training_set_num, info.validation_set_num, info.testing_set_num = tr.ctypes.data, va.ctypes.data, te.ctypes.data\n"
                    "L.IPCEnv_Coordinate(e, C.byref(info)); L.IPCEnv_InitializeSamplesBuffer(e, 500, 1000, 16, 0, 2); L.IPCEnv_SetHops(e, 3); K.check()\n"
                    "print('READY', flush=True); os._exit(0)\n")
    client = pre + ("c = C.c_void_p(L.legion_ipc_client_open(0)); K.check(); assert c.value\n"
                    "s = (C.c_int32 * 3)(); L.legion_ipc_client_steps(c, s)\n"
                    "print('HOPS', L.legion_ipc_client_hops(c), list(s), flush=True); os._exit(0)\n")     # no close: nothing posted matters here
    try:
        r = subprocess.run([sys.executable, "-c", server], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "READY" in r.stdout, r.stdout + r.stderr[-2000:]
        left = sorted(f for f in os.listdir("/dev/shm") if ns in f)
        assert ns + "simpleIPCshm" in left and ns + "simpleIPCshm_ext" in left and len(left) == 2 + 4, left      # slab, extension, sem_r/w x 2 pipes
        # against the live slab the extension is honoured ...
        r = subprocess.run([sys.executable, "-c", client], capture_output=True, text=True, timeout=60)
        assert "HOPS 3 [7, 2, 1]" in r.stdout, r.stdout + r.stderr[-2000:]
        # ... a reference-style server re-creates the slab (its own struct, other step counts) and knows nothing of the extension object
        with open("/dev/shm/" + ns + "simpleIPCshm", "r+b") as f:
            f.write((11).to_bytes(4, "little") + (3).to_bytes(4, "little") + (2).to_bytes(4, "little"))
        r = subprocess.run([sys.executable, "-c", client], capture_output=True, text=True, timeout=60)
        assert "HOPS 2 [11, 3, 2]" in r.stdout, r.stdout + r.stderr[-2000:]                                      # the stale extension is ignored
        # (b) what the killed server left behind
        r = subprocess.run([sys.executable, "-c", pre + "L.legion_ipc_unlink_namespace(%r.encode(), 8); print('UNLINKED')\n" % ns], capture_output=True, text=True, timeout=60)
        assert "UNLINKED" in r.stdout, r.stdout + r.stderr[-2000:]
        assert not [f for f in os.listdir("/dev/shm") if ns in f]
    finally:
        for f in os.listdir("/dev/shm"):
            if ns in f:
                os.unlink(os.path.join("/dev/shm", f))
