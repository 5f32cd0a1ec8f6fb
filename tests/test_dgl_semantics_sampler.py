"""The CPU baseline bench.py prints beside the GPU number when `dgl` is absent ("DGL-semantics CPU sampler (own)",
oracle/dgl_cpu_sampler.c) has to BE what its label says: dgl.sampling.sample_neighbors (uniform, without replacement,
take-all when deg <= fan-out) + dgl.to_block per layer (dst nodes first, new src nodes appended, local ids) +
index_select of the feature rows.  It is not a parity reference of the hot path (that is oracle/legion_oracle.c); these
tests pin the semantics the printed edges/s are quoted for.  DGL itself is not installed and is not needed here."""
import ctypes as C

import numpy as np
import pytest


def _graph(seed=3, V=400, F=6):
    """Rows of degree 0, 1, < f, == f, > f; neighbours DISTINCT inside a row, so 'without replacement' is visible in ids."""
    rs = np.random.RandomState(seed)
    deg = rs.choice([0, 1, 2, 3, 5, 8, 13, 40], size=V)
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = np.concatenate([rs.choice(V, size=d, replace=False) for d in deg]).astype(np.int32)
    feats = rs.rand(V, F).astype(np.float32)
    return V, F, indptr, indices, feats


def _run(oracle, g, seeds, fan, rng_seed=1, gather=True):
    V, F, indptr, indices, feats = g
    s = oracle.DglSemanticsSampler(indptr, indices, feats, V, F, len(seeds), fan)
    n, ne = s.run_batch(seeds, rng_seed=rng_seed, gather=gather)
    assert (s.local_map == -1).all()                        # scratch map restored for the next batch
    out = dict(n=n, ne=ne, nodes=s.nodes[:n].copy(), src=s.src[:ne].copy(), dst=s.dst[:ne].copy(), off=s.edge_off.copy())
    if gather:
        out["feat"] = s.feat[:n].copy()
    return out


@pytest.mark.parametrize("fan", [[5, 3], [8, 5, 2], [3]])
def test_sample_neighbors_and_to_block_semantics(oracle, fan):
    g = _graph()
    V, F, indptr, indices, feats = g
    rs = np.random.RandomState(1)
    seeds = rs.choice(V, size=37, replace=False).astype(np.int32)
    r = _run(oracle, g, seeds, fan)
    nodes, src, dst, off = r["nodes"], r["src"], r["dst"], r["off"]
    assert off[0] == 0 and off[len(fan)] == r["ne"] and (np.diff(off[:len(fan) + 1]) >= 0).all()
    # to_block: seeds keep their order as local ids 0..B-1; every node has ONE local id
    assert np.array_equal(nodes[:len(seeds)], seeds) and len(np.unique(nodes)) == r["n"]
    n_front = len(seeds)
    seen = len(seeds)
    for h, f in enumerate(fan):
        e0, e1 = int(off[h]), int(off[h + 1])
        s_h, d_h = src[e0:e1], dst[e0:e1]
        # the layer's destination (frontier) nodes are ALL nodes of the previous block, numbered first
        assert (d_h < n_front).all() and (np.diff(d_h) >= 0).all()
        for i in range(n_front):
            v = int(nodes[i])
            row = indices[indptr[v]:indptr[v + 1]]
            got = nodes[s_h[d_h == i]]
            assert len(got) == min(len(row), f)                     # min(deg, fan-out) edges per frontier node
            assert len(np.unique(got)) == len(got)                  # without replacement
            assert np.isin(got, row).all()                          # every edge is an edge of the graph
            if len(row) <= f:
                assert sorted(got.tolist()) == sorted(row.tolist())   # take-all
        # new source nodes get the next local ids in the order the layer's edges first mention them
        new_local = []
        for lu in s_h.tolist():
            if lu >= seen and lu not in new_local:
                new_local.append(lu)
        assert new_local == list(range(seen, seen + len(new_local)))
        seen += len(new_local)
        n_front = seen
    assert seen == r["n"]
    # index_select of the rows of the outermost source set
    assert np.array_equal(r["feat"], feats[nodes])


def test_duplicate_seeds_get_one_local_id(oracle):
    g = _graph()
    seeds = np.array([5, 9, 5, 11, 9], dtype=np.int32)
    r = _run(oracle, g, seeds, [3, 2], gather=False)
    assert r["nodes"][:3].tolist() == [5, 9, 11] and len(np.unique(r["nodes"])) == r["n"]


def test_deterministic_per_seed_and_uniform(oracle):
    g = _graph()
    V, F, indptr, indices, feats = g
    seeds = np.arange(0, V, 3, dtype=np.int32)
    a, b = _run(oracle, g, seeds, [5, 3], rng_seed=7), _run(oracle, g, seeds, [5, 3], rng_seed=7)
    for k in ("nodes", "src", "dst", "off"):
        assert np.array_equal(a[k], b[k])
    c = _run(oracle, g, seeds, [5, 3], rng_seed=8)
    assert not (np.array_equal(a["src"], c["src"]) and np.array_equal(a["nodes"], c["nodes"]))
    # uniform over the row: over many rng seeds every neighbour of a degree-40 row is picked about f/deg of the time
    v = int(np.flatnonzero(np.diff(indptr) == 40)[0])
    row = indices[indptr[v]:indptr[v + 1]]
    cnt = dict.fromkeys(row.tolist(), 0)
    trials, f = 600, 8
    for t in range(trials):
        r = _run(oracle, g, np.array([v], dtype=np.int32), [f], rng_seed=100 + t, gather=False)
        for u in r["nodes"][r["src"]].tolist():
            cnt[u] += 1
    freq = np.array(list(cnt.values())) / trials
    assert abs(freq.mean() - f / 40) < 1e-9 and freq.min() > 0.10 and freq.max() < 0.32     # p = 0.2, sigma = 0.016


def test_result_does_not_depend_on_the_thread_count(oracle):
    """The draws are a function of (rng_seed, layer, node) and the compaction is serial, so the edges/s printed for
    128 threads on the GPU box are for the same batches one thread produces."""
    gomp = C.CDLL("libgomp.so.1")
    g = _graph(seed=5, V=3000)
    seeds = np.arange(0, 3000, 7, dtype=np.int32)
    before = int(oracle.lib().dgl_threads())
    try:
        outs = []
        for nt in (1, 2, 5):
            gomp.omp_set_num_threads(nt)
            assert int(oracle.lib().dgl_threads()) == nt
            outs.append(_run(oracle, g, seeds, [8, 5, 2]))
    finally:
        gomp.omp_set_num_threads(before)
    for o in outs[1:]:
        for k in ("nodes", "src", "dst", "off", "feat"):
            assert np.array_equal(outs[0][k], o[k]), k
