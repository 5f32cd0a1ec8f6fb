"""The server <-> trainer hand-off protocol WITHOUT a GPU (tests/test_ipc_env_cpu.py; also the workload of the host-sanitizer run,
`make -C legion-1_amd/csrc asan-host` + profiles/r05_robustness.sh): the shm slab, the named semaphores, the host mirror of the
counters, the poisoned pipe and the teardown of csrc/ipc_env.cpp, driven through the C ABI by a fake producer (this process:
IPCEnv_*) and one fake consumer process per logical GPU (legion_ipc_client_*).  $LEGION_IPC_NO_DEVICE=1: no hand-off buffer is
allocated, the handle slots stay zero.

    python tests/ipc_env_cpu.py producer <namespace>        # spawns its consumers, prints PRODUCER_OK
    python tests/ipc_env_cpu.py consumer <namespace> <dev> <batches>
Reference code whose hazards are covered: src/CUDA_IPC_Service.cu:187-198 (sem_open without unlink: a crashed run poisons the
next start), :299-325 (Finalize), src/helper_multiprocess.cpp:5-82 (shm create / open)."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

N_DEV, BATCHES, HOPS = 2, 40, 3


def setup(ns):
    os.environ["LEGION_IPC_NO_DEVICE"] = "1"
    os.environ["LEGION_IPC_NAMESPACE"] = ns
    import legion1_amd.capi as K
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    return K, L


def consumer(ns, dev, batches):
    K, L = setup(ns)
    os.environ["LEGION_IPC_DEVICE"] = str(dev)
    c = C.c_void_p(L.legion_ipc_client_open(-1))
    K.check()
    assert c.value, "client refused"
    steps = (C.c_int32 * 3)()
    L.legion_ipc_client_steps(c, steps)
    assert list(steps) == [7, 2, 1], list(steps)              # (3601 - 1) // 500, (700 - 1) // 512 + 1, (300 - 1) // 512 + 1
    assert L.legion_ipc_client_hops(c) == HOPS
    for w in range(7):                                          # no device: every handle slot is zero, no buffer was opened
        assert not L.legion_ipc_client_buffer(c, w)
    nc, ec = (C.c_int32 * 16)(), (C.c_int32 * 16)()
    t_first = None
    for b in range(batches + 1):
        L.legion_ipc_client_wait(c)
        if t_first is None:
            t_first = time.time()
        L.legion_ipc_client_read_counters(c, nc, ec)
        if nc[0] == -1:                                         # the poisoned pipe: every node-counter word -1, edge counters zeroed
            assert b == batches and list(nc) == [-1] * 16 and list(ec) == [0] * 16, (b, list(nc), list(ec))
            L.legion_ipc_client_post(c)
            break
        want = 1000 * dev + b + 1
        assert list(nc) == [want] * 16 and list(ec) == [2 * want] * 16, (dev, b, list(nc), list(ec))
        L.legion_ipc_client_post(c)
    else:
        raise AssertionError("no poisoned pipe after %d batches" % batches)
    L.legion_ipc_client_close(c)
    K.check()
    print("CONSUMER_OK %d first_batch_at %.6f" % (dev, t_first), flush=True)


def producer(ns):
    K, L = setup(ns)
    sem_files = ["/dev/shm/sem.%ssem_%s_%d_%d" % (ns, rw, d, p) for rw in "rw" for d in range(N_DEV) for p in range(2)]
    shm_file = "/dev/shm/%ssimpleIPCshm" % ns
    # bad arguments come back as sticky errors, nothing is created
    assert not L.NewIPCEnv(0) and b"device_count must be 1..8" in L.legion_last_error()
    L.legion_clear_error()
    # ---- a crashed run: semaphores left behind with stale counts (the reference never unlinks before sem_open) ----
    dead = C.c_void_p(L.NewIPCEnv(N_DEV))
    for d in range(N_DEV):
        L.IPCEnv_InitializeSamplesBuffer(dead, 500, 1000, 16, d, 2)
        for p in range(2):
            for _ in range(3):
                L.IPCEnv_IPCPost(dead, d, p)                    # sem_w = 3: a new consumer would run ahead of its producer
    K.check()
    assert all(os.path.exists(f) for f in sem_files) and os.path.exists(shm_file)
    # (no Finalize: the process "died")
    # ---- the next start ----
    e = C.c_void_p(L.NewIPCEnv(N_DEV))
    K.check()
    L.IPCEnv_InitializeSamplesBuffer(e, 500, 1000, 16, 9, 2)    # device outside the env: refused, nothing touched
    assert b"InitializeSamplesBuffer: bad arguments" in L.legion_last_error()
    L.legion_clear_error()
    bad = K.LegionBuildInfo()                                   # the reference divides by raw_batch_size unchecked (CUDA_IPC_Service.cu:89)
    bad.partition_count, bad.epoch, bad.raw_batch_size = N_DEV, 2, 0
    L.IPCEnv_Coordinate(e, C.byref(bad))
    assert b"IPCEnv_Coordinate: partition_count must be" in L.legion_last_error()
    L.legion_clear_error()
    info = K.LegionBuildInfo()
    nums = [(C.c_int32 * N_DEV)(3601, 4000), (C.c_int32 * N_DEV)(700, 650), (C.c_int32 * N_DEV)(300, 10)]
    info.partition_count, info.epoch, info.raw_batch_size = N_DEV, 2, 500
    info.training_set_num, info.validation_set_num, info.testing_set_num = [C.cast(a, C.c_void_p) for a in nums]
    L.IPCEnv_Coordinate(e, C.byref(info))
    assert L.IPCEnv_GetTrainStep(e) == 7 and L.IPCEnv_GetMaxStep(e) == (7 + 2) * 2 + 1
    for d in range(N_DEV):
        L.IPCEnv_InitializeSamplesBuffer(e, 500, 1000, 16, d, 2)   # unlinks the stale semaphores first
        L.IPCEnv_InitializeFeaturesBuffer(e, 0, 1000, 16, d, 2)
    L.IPCEnv_SetHops(e, HOPS)
    K.check()
    assert not L.IPCEnv_GetIds(e, 0, 0) and not L.IPCEnv_GetFloatFeatures(e, 1, 1)      # no device: nothing was allocated
    for d in range(N_DEV):
        for p in range(2):
            assert L.IPCEnv_IPCTryWait(e, d, p, 0) == -1, "a fresh sem_r must be 0 (no client has freed the pipe yet)"
    t_spawn = time.time()
    cons = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "consumer", ns, str(d), str(BATCHES)], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True) for d in range(N_DEV)]
    time.sleep(1.0)                                             # the stale sem_w counts are gone: the consumers must be BLOCKED now
    assert all(c.poll() is None for c in cons)
    t_post = time.time()
    for b in range(BATCHES):
        for d in range(N_DEV):
            p = b % 2
            L.IPCEnv_IPCWait(e, d, p)                           # sem_r: the consumer opened (both pipes free) / handed the pipe back
            v = 1000 * d + b + 1
            L.IPCEnv_SetMirror(e, d, p, v, 2 * v)
            L.IPCEnv_IPCPost(e, d, p)
    for d in range(N_DEV):                                      # the failed batch: nc = -1 everywhere (runner.cpp post_poisoned)
        p = BATCHES % 2
        L.IPCEnv_IPCWait(e, d, p)
        L.IPCEnv_SetMirror(e, d, p, -1, 0)
        L.IPCEnv_IPCPost(e, d, p)
    for d, c in enumerate(cons):
        out, _ = c.communicate(timeout=60)
        assert c.returncode == 0 and ("CONSUMER_OK %d" % d) in out, out[-3000:]
        first = float(out.split("first_batch_at")[1].split()[0])
        assert first >= t_post - 0.05, "consumer %d ran on a stale semaphore: first batch %.3f s before the first post" % (d, t_post - first)
    assert time.time() - t_spawn < 60
    K.check()
    L.IPCEnv_Finalize(e)
    L.IPCEnv_Finalize(e)                                        # idempotent
    K.check()
    left = [f for f in sem_files + [shm_file] if os.path.exists(f)]
    assert not left, left
    print("PRODUCER_OK %d batches x %d consumers + poisoned pipe, stale semaphores replaced, slab and semaphores removed" % (BATCHES, N_DEV), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "consumer":
        consumer(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    else:
        producer(sys.argv[2])
