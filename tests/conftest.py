import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Every GPU test runs under the logical-device audit (csrc/audit.h): the library of this process AND every child it starts (the
# `legion` server, bench.py, the trainer-side clients) tag their streams / events / allocations with the logical GPU they were
# created under and check every launch, copy, event and pointer table against it.  A violation is a sticky error (K.check() raises),
# fails the `legion` binary (exit code 4) and fails the test through the fixture below.  LEGION_DEVICE_AUDIT=0 switches it off.
os.environ.setdefault("LEGION_DEVICE_AUDIT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import json
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


BATCH_KEYS = ("nc", "ec", "ids", "labels", "src_off", "dst_off", "features")
KEYS_NO_FEATURES = BATCH_KEYS[:-1]      # a run that gathers nothing (pre-sampling batches, gather=False oracle runs): say so explicitly


def assert_batch_equal(ref, got, keys=BATCH_KEYS):
    """Every buffer named in `keys` must be present on BOTH sides and equal word for word.  A key missing from either side is a
    failure (round 4 skipped it silently: a renamed field of Engine.result() or of the oracle wrapper would have turned a parity
    test into a no-op for that buffer) -- a caller that means to leave a buffer out passes `keys=` without it."""
    for k in keys:
        assert k in ref, f"{k}: missing from the reference batch (has {sorted(ref)})"
        assert k in got, f"{k}: missing from the batch under test (has {sorted(got)})"
        a, b = np.asarray(ref[k]), np.asarray(got[k])
        assert a.shape == b.shape, f"{k}: shape {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            bad = np.nonzero(a.reshape(-1) != b.reshape(-1))[0]
            raise AssertionError(f"{k}: {len(bad)} mismatches, first at {bad[:5]}: {a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}")


AUDIT_TOTALS = {"tests": 0, "checks": 0, "unattributed": 0, "peer_launches": 0, "servers": []}      # evidence for the terminal summary


def note_server_audit(counts):
    """A `legion` server (child process) left this audit summary in its log (tests/test_gpu_ipc.py, tests/test_gpu_bench_legs.py)."""
    AUDIT_TOTALS["servers"].append(counts)


def pytest_terminal_summary(terminalreporter):
    t = AUDIT_TOTALS
    if t["tests"] == 0 and not t["servers"]:
        return
    terminalreporter.write_line("device audit (LEGION_DEVICE_AUDIT=1): %d GPU tests, %d checks in this process, 0 violations (a violation fails its test), "
                                "%d unattributed, %d launches with a peer's table" % (t["tests"], t["checks"], t["unattributed"], t["peer_launches"]))
    if t["servers"]:
        terminalreporter.write_line("device audit of %d `legion` server processes: %d checks, %d violations, %d unattributed, %d launches with a peer's table" % (
            len(t["servers"]), sum(s["checks"] for s in t["servers"]), sum(s["violations"] for s in t["servers"]),
            sum(s["unattributed"] for s in t["servers"]), sum(s["peer_launches"] for s in t["servers"])))


@pytest.fixture(autouse=True)
def _no_device_audit_violation(request):
    """After every test: the library of THIS process must not have recorded a wrong-device violation (tests that provoke one
    reset the audit themselves).  Before every test: the thread is back on logical GPU 0, wherever the previous test left it."""
    if request.node.get_closest_marker("gpu") is None:      # no device behind a CPU test
        yield
        return
    capi = sys.modules.get("legion1_amd.capi")
    if capi is not None and capi._lib is not None and capi._lib.legion_audit_enabled():
        capi._lib.SetGPUDevice(0)
        capi._lib.legion_audit_reset()          # per-test counts (summed below for the terminal summary)
    yield
    capi = sys.modules.get("legion1_amd.capi")
    if capi is None or capi._lib is None or not capi._lib.legion_audit_enabled():
        return
    import ctypes
    c = (ctypes.c_int64 * 4)()
    capi._lib.legion_audit_counts(c)
    AUDIT_TOTALS["tests"] += 1
    AUDIT_TOTALS["checks"] += c[0]
    AUDIT_TOTALS["unattributed"] += c[2]
    AUDIT_TOTALS["peer_launches"] += c[3]
    if c[1]:
        msgs = [capi._lib.legion_audit_message(i).decode() for i in range(min(5, capi._lib.legion_audit_message_count()))]
        capi._lib.legion_audit_reset()
        capi._lib.legion_clear_error()
        raise AssertionError("device audit: %d violation(s): %s" % (c[1], " | ".join(msgs)))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def synth():
    import legion1_amd.synth as S
    return S


@pytest.fixture(scope="session")
def small_ds(synth):
    spec = synth.spec_for("products", scale=0.01)
    return synth.generate(spec)
