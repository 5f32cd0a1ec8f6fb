#!/usr/bin/env python3
"""GraphSAGE / GCN trainer on top of the sampling server, plain PyTorch (no DGL).

Consumes mini-batches from the `legion` server through `ipc_service` exactly like the reference trainer
(pytorch_extension/legion_graphsage.py:72-172): one process per GPU, `ipc_service.get_next(F)` ->
[ids, features, labels, (src, dst) x H], `get_block_size()` -> (num_src, num_dst) x H, `synchronize()` hands the
buffers back.  The reference builds DGL blocks + `SAGEConv(..., 'mean')`; dgl is not installable here, so the mean
aggregator is written with index_add_ -- the same arithmetic on the same COO blocks (duplicate edges, which sampling
with replacement produces, count twice as in DGL).  Works for any number of hops the server samples.

`--task lp` is the link-prediction variant (pytorch_extension/lp_sage.py:72-97): every training batch is laid out as
three equal thirds [src | pos | neg] (a `trainingset` file written by `legion1_amd.synth.lp_trainingset`), the loss is
-logsigmoid(<h_src, h_pos>) - logsigmoid(-<h_src, h_neg>); validation / test batches are only drained.

    LEGION_TABLES=auto legion-1_amd/csrc/legion 1 0 25,10 meta_config &          # or launch_server.py
    PYTHONPATH=legion-1_amd/ipc_service python examples/legion_sage_torch.py --features_num 100 --class_num 47
"""
import argparse
import os
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as Func
from torch.nn.parallel import DistributedDataParallel as DDP


class SageMean(nn.Module):
    """h_dst = W_self h_dst + W_neigh mean_{(s,d) in block} h_s + b.  The dst nodes of a block are the first
    num_dst src nodes (sampled_ids lists the seeds first, then the new nodes hop by hop)."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc_self = nn.Linear(in_feats, out_feats, bias=False)
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_feats))

    def forward(self, block, h):
        src, dst, num_src, num_dst = block
        assert h.shape[0] == num_src
        agg = torch.zeros(num_dst, h.shape[1], dtype=h.dtype, device=h.device).index_add_(0, dst, h.index_select(0, src))
        deg = torch.zeros(num_dst, dtype=h.dtype, device=h.device).index_add_(0, dst, torch.ones_like(dst, dtype=h.dtype))
        agg = agg / deg.clamp(min=1).unsqueeze(1)
        return self.fc_self(h[:num_dst]) + self.fc_neigh(agg) + self.bias


class GraphConvBoth(nn.Module):
    """DGL GraphConv(norm='both', allow_zero_in_degree=True) as legion_gcn.py:80-87 uses it:
    h_dst = sum_{(s,d)} h_s / sqrt(outdeg_s * indeg_d) W + b, degrees counted inside the block and clamped to >= 1."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc = nn.Linear(in_feats, out_feats, bias=True)

    def forward(self, block, h):
        src, dst, num_src, num_dst = block
        one = torch.ones_like(src, dtype=h.dtype)
        out_deg = torch.zeros(num_src, dtype=h.dtype, device=h.device).index_add_(0, src, one).clamp(min=1)
        in_deg = torch.zeros(num_dst, dtype=h.dtype, device=h.device).index_add_(0, dst, one).clamp(min=1)
        m = (h * out_deg.rsqrt().unsqueeze(1)).index_select(0, src)
        agg = torch.zeros(num_dst, h.shape[1], dtype=h.dtype, device=h.device).index_add_(0, dst, m)
        return self.fc(agg * in_deg.rsqrt().unsqueeze(1))


class SAGE(nn.Module):
    def __init__(self, in_feats, n_hidden, n_classes, n_layers, dropout, conv=SageMean):
        super().__init__()
        dims = [in_feats] + [n_hidden] * (n_layers - 1) + [n_classes]
        self.layers = nn.ModuleList(conv(dims[i], dims[i + 1]) for i in range(n_layers))
        self.dropout = nn.Dropout(dropout)

    def forward(self, blocks, x):
        h = x
        for l, (layer, block) in enumerate(zip(self.layers, blocks)):
            h = layer(block, h)
            if l != len(self.layers) - 1:
                h = self.dropout(Func.relu(h))
        return h


def next_batch(ipc_service, feat_len, hops):
    out = ipc_service.get_next(feat_len)            # zero-copy views of server-owned device memory
    sizes = ipc_service.get_block_size()
    features, labels = out[1], out[2]
    blocks = [(out[3 + 2 * k].long(), out[4 + 2 * k].long(), sizes[2 * k], sizes[2 * k + 1]) for k in range(hops)]
    return features, labels.long(), blocks


def worker(rank, world, args):
    import ipc_service
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "12355")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # RCCL on ROCm
    ipc_service.initialize()
    train_steps, valid_steps, test_steps = ipc_service.get_steps()
    hops = ipc_service.get_hops() if hasattr(ipc_service, "get_hops") else 2
    model = SAGE(args.features_num, args.hidden_dim, args.class_num, hops, args.drop_rate,
                 conv=GraphConvBoth if args.model == "gcn" else SageMean).to(device)
    if world > 1:
        model = DDP(model, device_ids=[rank])
    opt = torch.optim.Adam(model.parameters(), lr=args.learning_rate)
    loss_fn = nn.CrossEntropyLoss()

    def evaluate(steps):
        hit = tot = 0
        model.eval()
        with torch.no_grad():
            for _ in range(steps):
                x, y, blocks = next_batch(ipc_service, args.features_num, hops)
                ok = y >= 0                                  # -1 padded seeds of a short batch
                pred = model(blocks, x).argmax(1)
                hit += int((pred[ok] == y[ok]).sum())
                tot += int(ok.sum())
                torch.cuda.synchronize()
                ipc_service.synchronize()
        if world > 1:
            t = torch.tensor([hit, tot], device=device)
            dist.all_reduce(t)
            hit, tot = int(t[0]), int(t[1])
        return hit / max(tot, 1)

    def drain(steps):
        for _ in range(steps):
            ipc_service.get_next(args.features_num)
            ipc_service.synchronize()
        return float("nan")

    def lp_loss(h):                                          # lp_sage.py:87-90
        out, pos_out, neg_out = h.split(h.size(0) // 3, dim=0)
        return -Func.logsigmoid((out * pos_out).sum(-1)).mean() - Func.logsigmoid(-(out * neg_out).sum(-1)).mean()

    if args.task == "lp":
        evaluate = drain
    for epoch in range(args.epoch):
        model.train()
        t0, last = time.time(), float("nan")
        for _ in range(train_steps):
            x, y, blocks = next_batch(ipc_service, args.features_num, hops)
            loss = lp_loss(model(blocks, x)) if args.task == "lp" else loss_fn(model(blocks, x), y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            ipc_service.synchronize()                        # the server may refill this pipe now
            last = float(loss)
        cost = time.time() - t0
        acc = evaluate(valid_steps)
        if rank == 0:
            print("Epoch:{}, Cost:{:.3f} s, Train Loss:{:.4f}, Val Acc: {:.4f}".format(epoch, cost, last, acc), flush=True)
    acc = evaluate(test_steps)
    if rank == 0:
        print("Accuracy on test data: {:.4f}".format(acc), flush=True)
    ipc_service.finalize()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser("Train GraphSAGE on batches of the Legion server (plain PyTorch).")
    ap.add_argument("--model", default="sage", choices=["sage", "gcn"], help="legion_graphsage.py | legion_gcn.py layer")
    ap.add_argument("--task", default="nc", choices=["nc", "lp"], help="node classification | link prediction (lp_sage.py)")
    ap.add_argument("--class_num", type=int, default=47, help="nc: classes; lp: embedding width")
    ap.add_argument("--features_num", type=int, default=100)
    ap.add_argument("--hidden_dim", type=int, default=256)
    ap.add_argument("--drop_rate", type=float, default=0.5)
    ap.add_argument("--learning_rate", type=float, default=0.003)
    ap.add_argument("--epoch", type=int, default=100, help="must equal the epoch count in the server's meta_config")
    ap.add_argument("--gpu_num", type=int, default=1)
    ap.add_argument("--seed", type=int, default=None, help="torch.manual_seed (weights, dropout)")
    a = ap.parse_args()
    if a.gpu_num == 1:
        worker(0, 1, a)
    else:
        mp.spawn(worker, args=(a.gpu_num, a), nprocs=a.gpu_num, join=True)
