#!/usr/bin/env python3
"""Importer half of vmm_ipc_probe.cpp inside a PyTorch process (HIP runtime bundled with the torch wheel, ROCm 7.0):
receives the chunk fds over the unix socket, imports them, maps them into ONE virtual range and wraps it as a tensor.
usage: vmm_ipc_probe_torch.py <chunks> <chunk_bytes> <socket path>"""
import ctypes as C
import os
import socket
import sys
import threading
import time

chunks, chunk, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
import torch  # noqa: E402

torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
lib = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})[0]
hip = C.CDLL(lib)


def ck(rc, what):
    if rc != 0:
        print("%s failed: %d  [runtime %s]" % (what, rc, lib), flush=True)
        os._exit(2)


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class AccessDesc(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


threading.Thread(target=lambda: (time.sleep(40), print("importer (torch): watchdog after 40 s", flush=True), os._exit(3)), daemon=True).start()
total = chunks * chunk
va = C.c_void_p()
hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
print("reserve", flush=True)
ck(hip.hipMemAddressReserve(C.byref(va), total, 0, None, 0), "hipMemAddressReserve")
print("reserved", hex(va.value), flush=True)
s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
for _ in range(300):
    try:
        s.connect(path)
        break
    except OSError:
        time.sleep(0.1)
t0 = time.perf_counter()
hip.hipMemImportFromShareableHandle.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
for c in range(chunks):
    _, fds, _, _ = socket.recv_fds(s, 1, 1)
    h = C.c_void_p()
    print("import fd", fds[0], flush=True)
    # ROCm 7.0's runtime dereferences osHandle as an int* (7.2 takes the fd by value, like CUDA): BY_POINTER=1 selects the former
    if os.environ.get("BY_POINTER", "1") == "1":
        fdv = C.c_int(fds[0])
        ck(hip.hipMemImportFromShareableHandle(C.byref(h), C.cast(C.pointer(fdv), C.c_void_p), 1), "hipMemImportFromShareableHandle(&fd)")
    else:
        ck(hip.hipMemImportFromShareableHandle(C.byref(h), C.c_void_p(fds[0]), 1), "hipMemImportFromShareableHandle(fd)")   # 1 = PosixFileDescriptor
    ck(hip.hipMemMap(C.c_void_p(va.value + c * chunk), chunk, 0, h, 0), "hipMemMap")
    os.close(fds[0])
acc = AccessDesc(Loc(1, 0), 3)   # device 0, read + write
hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(AccessDesc), C.c_size_t]
ck(hip.hipMemSetAccess(va, total, C.byref(acc), 1), "hipMemSetAccess")
ms = (time.perf_counter() - t0) * 1e3
n = total // 8
idx = list(range(0, n, 1 << 20)) + [c * chunk // 8 + d for c in range(1, chunks) for d in (-1, 0)] + [n - 1]   # every 8 MiB + the chunk seams
# read the sampled words with hipMemcpy (no tensor view needed for the check)
bad = 0
for i in idx:
    w = C.c_ulonglong(0)
    ck(hip.hipMemcpy(C.byref(w), C.c_void_p(va.value + 8 * i), C.c_size_t(8), 2), "hipMemcpy")
    bad += int(w.value != ((i * 2654435761 + 11) & 0xFFFFFFFFFFFFFFFF))
print("importer (torch, %s): %d x %d bytes = %.2f GiB mapped contiguously in %.1f ms, %d of %d sampled words wrong" %
      (lib.split("/")[-3] + "/" + lib.split("/")[-1], chunks, chunk, total / 2 ** 30, ms, bad, len(idx)), flush=True)
s.send(b"d")
sys.exit(5 if bad else 0)
