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
