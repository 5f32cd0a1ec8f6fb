"""Size-independent properties every reference execution of a mini-batch satisfies, checked where the
oracle cannot run in seconds (BASELINE.json's full shapes).  Used by the -m gpu full-shape tests."""
import numpy as np


def check_batch(res, spec, synth, B, fan, indptr, indices, seeds, rs, distinct_seeds=True, n_edges_checked=300, n_rows_checked=2000):
    """res: Engine.result(); indptr / indices: torch tensors on the GPU (the whole CSR); seeds: the B seed ids of
    this batch (numpy).  With distinct_seeds=False the seed list may hold duplicates (link prediction): position_map
    keeps the LAST occurrence (Kernels.cu:92 in serial order), ids[:B] is still the list verbatim."""
    import torch
    import oracle as O
    dev = indptr.device
    H = len(fan)
    nc, ec, ids = res["nc"], res["ec"], res["ids"]
    levels = [int(nc[4 + 2 * l]) for l in range(H + 1)]
    assert nc[5 + 2 * H] == sum(levels) == len(ids) and levels[0] == B
    assert ids.min() >= 0 and ids.max() < spec.V
    assert np.array_equal(ids[:B], seeds)
    assert np.array_equal(res["labels"], synth.labels(spec, ids[:B]))
    new = ids[B:]
    assert len(np.unique(new)) == len(new) and not np.isin(new, ids[:B]).any()           # dedup
    if distinct_seeds:
        assert len(np.unique(ids)) == len(ids)
    # position of a seed id: its last occurrence in the seed list
    last_pos = {}
    if not distinct_seeds:
        for i, s in enumerate(ids[:B].tolist()):
            last_pos[s] = i
    src, dst = res["src_off"], res["dst_off"]
    cum_nodes, e0 = np.cumsum(levels), 0
    for h in range(1, H + 1):
        e1 = int(ec[2 + h])
        assert (dst[e0:e1] < cum_nodes[h - 1]).all() and (src[e0:e1] < cum_nodes[h]).all() and (src[e0:e1] >= 0).all()
        e0 = e1
    assert e0 == len(src) == len(dst)
    # first-seen order: a new node's position is increasing with the edge that discovered it
    first_edge = np.full(len(ids), -1, np.int64)
    order = np.arange(len(src) - 1, -1, -1)
    first_edge[src[order]] = order
    fe = first_edge[np.arange(B, len(ids))]
    assert (fe >= 0).all() and (np.diff(fe) > 0).all()
    # hop-1 edge multiset == closed form (the RNG stream depends only on the slot index)
    sid = torch.from_numpy(ids[:B].astype(np.int64)).to(dev)
    ip = indptr[sid].cpu().numpy()
    deg = indptr[sid + 1].cpu().numpy() - ip
    assert int(np.minimum(deg, fan[0]).sum()) == int(ec[3])
    checked = 0
    for i in list(range(0, B, 997)):
        base = int(np.minimum(deg[:i], fan[0]).sum())
        for j in range(min(int(deg[i]), fan[0])):
            k = O.sample_index(i * fan[0] + j, int(deg[i]))
            want = int(indices[int(ip[i]) + k].item())
            assert ids[src[base + j]] == want
            assert dst[base + j] == (i if distinct_seeds else last_pos[int(ids[i])])
            checked += 1
    assert checked > 0
    # every sampled edge is an edge of the graph (sample over all hops)
    for eidx in rs.choice(len(src), size=min(n_edges_checked, len(src)), replace=False):
        d_id, s_id = int(ids[dst[eidx]]), int(ids[src[eidx]])
        row = indices[int(indptr[d_id].item()):int(indptr[d_id + 1].item())].cpu().numpy()
        assert s_id in row
    # gathered rows are the table rows, byte for byte (generator closed form)
    if "features" in res:
        rows = rs.choice(len(ids), size=min(n_rows_checked, len(ids)), replace=False)
        assert np.array_equal(res["features"][rows], synth.features(spec, ids[rows]))
        assert np.array_equal(res["features"][:64], synth.features(spec, ids[:64]))
    return levels


def device_graph(K, synth, workload, skew=205):
    """(spec, indptr, indices, features, E) of a BASELINE shape, generated on cuda:0 by csrc/synth.hip."""
    import torch
    bench = __import__("bench")
    spec = synth.spec_for(workload)
    indptr, indices, feats, E = bench.build_graph_on_gpu(K, spec, torch.device("cuda", 0), skew)
    return spec, indptr, indices, feats, E


def device_seeds(K, spec, n_parts=1):
    """[(ids tensor, labels tensor)] per partition: the training ids with tid % n_parts == p, labels from the generator."""
    import torch
    L = K.lib()
    dev = torch.device("cuda", 0)
    tr = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, tr.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
    lab = torch.empty(spec.V, dtype=torch.int32, device=dev)
    L.legion_synth_labels(None, lab.data_ptr(), 0, spec.V, spec.classes)
    torch.cuda.synchronize()
    out = []
    for p in range(n_parts):
        ids = tr[(tr % n_parts) == p].contiguous()
        out.append((ids, lab[ids.long()].contiguous()))
    return out
