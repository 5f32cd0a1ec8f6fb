"""Trainer-side process of the IPC hand-off test: consumes every batch through the `ipc_service`
module exactly like legion_graphsage.py does (train_one_step, legion_graphsage.py:72-89) and writes
one digest record per batch.  usage: ipc_client.py <feature_dim> <epochs> <out.json>"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "legion-1_amd", "ipc_service"))
import ipc_service  # noqa: E402


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


def main():
    feat_dim, epochs, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.cuda.set_device(0)
    ipc_service.initialize()
    train_steps, valid_steps, test_steps = ipc_service.get_steps()
    hops = ipc_service.get_hops()
    total = (train_steps + valid_steps) * epochs + test_steps
    recs = []
    queued = os.environ.get("LEGION_CLIENT_QUEUED_WORK") == "1"
    busy = torch.randn(4096, 4096, device="cuda") * 0.01 if queued else None
    for b in range(total):
        if queued:
            # A reference-style trainer (legion_graphsage.py:93-116): it posts the pipe with its device work still QUEUED and never
            # synchronises on its own.  A long spin kernel sits in front of the copies that read the batch; the digests are taken from
            # those copies after the post.  If synchronize() handed the pipe back before the queued copies ran, the server would
            # overwrite it (depth 2: batch b + 2) and the digests would be another batch's.
            tensors = ipc_service.get_next(feat_dim)
            sizes = ipc_service.get_block_size()
            for _ in range(24):                                # tens of ms of queued device work (the "optimizer step")
                busy = torch.tanh(busy @ busy)
            # elementwise kernels reading the served buffers, queued behind that work and still unexecuted when we post
            # (+ 0 on the integer view keeps every bit of the float rows)
            snap = [(t.view(torch.int32) + 0).view(t.dtype) for t in tensors]
            ipc_service.synchronize()
            ids, feats, labels = snap[:3]
            blocks = snap[3:]
            recs.append(dict(b=b, n=int(ids.shape[0]), sizes=list(sizes), ids=sha(ids), features=sha(feats), labels=sha(labels),
                             edges=[int(blocks[2 * k].numel()) for k in range(hops)], src=sha(blocks[0]), dst=sha(blocks[1])))
            continue
        tensors = ipc_service.get_next(feat_dim)
        sizes = ipc_service.get_block_size()
        ids, feats, labels = tensors[:3]
        blocks = tensors[3:]
        assert len(blocks) == 2 * hops and feats.shape == (ids.shape[0], feat_dim)
        assert ids.device.type == "cuda" and ids.dtype == torch.int32 and feats.dtype == torch.float32
        # the DGL blocks of the trainers: block k is (src nodes, dst nodes, edges); b2_* alias b1_* prefixes
        for k in range(hops):
            src, dst = blocks[2 * k], blocks[2 * k + 1]
            assert src.data_ptr() == blocks[0].data_ptr() and dst.data_ptr() == blocks[1].data_ptr()
            if src.numel():
                assert int(src.max()) < sizes[2 * k] and int(dst.max()) < sizes[2 * k + 1]
        torch.cuda.synchronize()
        recs.append(dict(b=b, n=int(ids.shape[0]), sizes=list(sizes), ids=sha(ids), features=sha(feats), labels=sha(labels),
                         edges=[int(blocks[2 * k].numel()) for k in range(hops)],
                         src=sha(blocks[0]), dst=sha(blocks[1])))
        if os.environ.get("LEGION_CLIENT_DUMP_SEEDS"):   # the seed part of the batch: ids[0 : batch size]
            recs[-1]["seeds"] = ids[:labels.shape[0]].cpu().tolist()
        ipc_service.synchronize()
    ipc_service.finalize()
    with open(out, "w") as f:
        json.dump(dict(steps=[train_steps, valid_steps, test_steps], hops=hops, batches=recs), f)


if __name__ == "__main__":
    main()
