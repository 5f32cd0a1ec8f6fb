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
