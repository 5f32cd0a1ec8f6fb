"""The DGL-free layers of examples/legion_sage_torch.py against dense restatements of DGL's definitions
(SAGEConv 'mean': fc_self(h_dst) + fc_neigh(mean of in-neighbours) + bias; GraphConv norm='both':
D_in^-1/2 A D_out^-1/2 H W + b with degrees clamped to 1) on a random COO block with duplicate edges, the shape the
server delivers (dst nodes = the first num_dst src nodes).  CPU only."""
import importlib.util
import os

import torch

from conftest import ROOT


def _load():
    spec = importlib.util.spec_from_file_location("legion_sage_torch", os.path.join(ROOT, "examples", "legion_sage_torch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _block(n_src=50, n_dst=20, n_edges=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n_src, (n_edges,), generator=g)
    dst = torch.randint(0, n_dst - 3, (n_edges,), generator=g)     # the last 3 dst nodes have no in-edge
    src[:10], dst[:10] = src[10:20], dst[10:20]                    # duplicate edges (sampling with replacement)
    A = torch.zeros(n_dst, n_src, dtype=torch.float64)
    A.index_put_((dst, src), torch.ones(n_edges, dtype=torch.float64), accumulate=True)
    h = torch.randn(n_src, 16, generator=g, dtype=torch.float64)
    return (src, dst, n_src, n_dst), A, h


def test_sage_mean_matches_dense_definition():
    m = _load()
    block, A, h = _block()
    layer = m.SageMean(16, 8).double()
    with torch.no_grad():
        layer.bias.normal_()
    deg = A.sum(1).clamp(min=1)
    want = layer.fc_self(h[:block[3]]) + layer.fc_neigh((A @ h) / deg[:, None]) + layer.bias
    assert torch.allclose(layer(block, h), want, atol=1e-12)


def test_graphconv_both_matches_dense_definition():
    m = _load()
    block, A, h = _block(seed=1)
    layer = m.GraphConvBoth(16, 8).double()
    d_in, d_out = A.sum(1).clamp(min=1), A.sum(0).clamp(min=1)
    want = layer.fc((A * d_in.rsqrt()[:, None] * d_out.rsqrt()[None, :]) @ h)
    assert torch.allclose(layer(block, h), want, atol=1e-12)


def test_model_runs_h_hop_blocks():
    m = _load()
    sizes = [(40, 25), (25, 12), (12, 5)]                          # 3 hops: each block's dst = the next block's src
    blocks = []
    for i, (ns, nd) in enumerate(sizes):
        b, _, _ = _block(ns, nd, 80, seed=2 + i)
        blocks.append(b)
    model = m.SAGE(16, 32, 7, 3, 0.0).double()
    out = model(blocks, torch.randn(40, 16, dtype=torch.float64))
    assert out.shape == (5, 7) and torch.isfinite(out).all()
