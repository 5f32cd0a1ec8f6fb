"""The RCCL branches of the multi-GPU plumbing on the one GPU of the test box (see tests/nccl_single_rank.py)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_rccl_code_paths_on_a_one_rank_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_single_rank.py"), str(port)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "nccl single rank ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
