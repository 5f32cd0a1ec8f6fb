"""Trainer-side process of the poisoned-pipe test: attaches through `ipc_service` and consumes batches.  The server fails
that batch and posts the pipe with every node-counter word at -1 (runner.cpp, post_poisoned): get_next must raise, not hand
out tensors with negative or stale sizes.  usage: ipc_client_poison.py <feature_dim> [expected error text]; exit 0 = raised as
specified (default text: "sampling server failed") after exactly one good batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "legion-1_amd", "ipc_service"))
import ipc_service  # noqa: E402

torch.cuda.set_device(0)
ipc_service.initialize()
print("ATTACHED", flush=True)
good = 0
try:
    for _ in range(3):       # the server hands over one good batch, then the poisoned pipe
        out = ipc_service.get_next(int(sys.argv[1]))
        assert out[0].numel() > 0 and out[1].shape[0] == out[0].shape[0]
        good += 1
        print("BATCH", [tuple(t.shape) for t in out], flush=True)
        ipc_service.synchronize()
except RuntimeError as e:
    print("RAISED after %d good batches:" % good, str(e).splitlines()[0], flush=True)
    ipc_service.finalize()
    want = sys.argv[2] if len(sys.argv) > 2 else "sampling server failed"
    sys.exit(0 if (want in str(e) and good == 1) else 5)
sys.exit(7)
