"""The C oracle against the committed golden vectors, against the independent pure-Python
restatement (tests/pyref.py), and the invariants every reference execution satisfies."""
import numpy as np
import pytest

from conftest import assert_batch_equal, load_golden, sha
import pyref


def _toy():
    g = load_golden("toy_batches")
    V, F = g["V"], g["F"]
    return (g, V, F, np.array(g["indptr"], np.int64), np.array(g["indices"], np.int32),
            np.array(g["features"], np.float32).reshape(V, F), np.array(g["labels"], np.int32), np.array(g["seeds"], np.int32))


def test_toy_golden_and_pyref(oracle):
    g, V, F, indptr, indices, feats, labels, seeds = _toy()
    for case in g["cases"]:
        r = oracle.OracleRunner(indptr, indices, feats, V, F, case["batch"], case["fanout"])
        res = r.run_batch(seeds, labels[seeds], case["counter"])
        for k in ("nc", "ec", "ids", "labels", "src_off", "dst_off"):
            assert res[k].tolist() == case[k], (case["fanout"], k)
        assert sha(res["features"]) == case["features_sha256"]
        ref = pyref.run_batch(indptr, indices, feats, seeds, labels[seeds], case["batch"], case["counter"], case["fanout"])
        assert_batch_equal(ref, res)


def test_toy_hand_checked(oracle):
    """First slots by hand: seeds [0,4,9,13], fan-out 2: slot 0 -> node 0 (deg 3), k = int(48270/2147483646*3) = 0 -> 1."""
    g, V, F, indptr, indices, feats, labels, seeds = _toy()
    r = oracle.OracleRunner(indptr, indices, feats, V, F, 4, [2, 2])
    res = r.run_batch(seeds, labels[seeds], 0)
    assert res["ids"][:4].tolist() == [0, 4, 9, 13]
    assert res["ids"][4] == 1                       # first new node: neighbour 0 of node 0
    assert res["dst_off"][0] == 0 and res["src_off"][0] == 4
    assert res["nc"][5] == 4 and res["nc"][3] == 0 and res["nc"][4] == 4


def test_counter_layout_matches_literal_2hop_reference(oracle):
    """The H-hop counter layout reproduces the slots of the literal update_counter at H = 2."""
    g, V, F, indptr, indices, feats, labels, seeds = _toy()
    r = oracle.OracleRunner(indptr, indices, feats, V, F, 6, [3, 2])
    res = r.run_batch(seeds, labels[seeds], 0)
    nc, ec = [0] * 16, [0] * 16
    pyref.update_counter_reference_2hop(nc, ec, 0, 6)
    U1, E1 = int(res["nc"][6]), int(res["ec"][3])
    nc[1], ec[1] = U1, E1
    pyref.update_counter_reference_2hop(nc, ec, 2, 0)
    U2, E2 = int(res["nc"][8]), int(res["ec"][4]) - E1
    nc[1], ec[1] = U2, E2
    pyref.update_counter_reference_2hop(nc, ec, 4, 0)
    assert res["nc"].tolist() == nc and res["ec"].tolist() == ec


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_small_graphs_vs_pyref(oracle, seed):
    rng = np.random.RandomState(seed)
    V, F = 60, 3
    deg = rng.randint(0, 9, size=V)
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(-1, V, size=int(indptr[-1])).astype(np.int32)   # includes -1 entries
    feats = rng.rand(V, F).astype(np.float32)
    labels = rng.randint(0, 5, size=V).astype(np.int32)
    seeds = rng.permutation(V)[:23].astype(np.int32)
    for fan, B in (([3, 2], 8), ([2, 2, 2], 8), ([5], 23), ([4, 3], 23)):
        r = oracle.OracleRunner(indptr, indices, feats, V, F, B, fan)
        for counter in range(0, (len(seeds) + B - 1) // B):
            res = r.run_batch(seeds, labels[seeds], counter)
            ref = pyref.run_batch(indptr, indices, feats, seeds, labels[seeds], B, counter, fan)
            assert_batch_equal(ref, res)


def test_medium_digests(oracle, synth):
    g = load_golden("medium_digests")
    spec = synth.spec_for("products", scale=0.04)
    ds = synth.generate(spec)
    assert {"indptr": sha(ds.indptr), "indices": sha(ds.indices), "features": sha(ds.features), "labels": sha(ds.labels),
            "train": sha(ds.train)} == g["dataset_sha256"]
    lab = ds.labels[ds.train]
    runners = {}
    for case in g["cases"]:
        key = (case["batch"], tuple(case["fanout"]))
        if key not in runners:
            runners[key] = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, case["batch"], case["fanout"])
        res = runners[key].run_batch(ds.train, lab, case["counter"])
        assert res["nc"].tolist() == case["nc"] and res["ec"].tolist() == case["ec"]
        for k in ("ids", "labels", "src_off", "dst_off", "features"):
            assert sha(res[k]) == case[k + "_sha256"], (key, k)


def check_invariants(res, ds_indptr, ds_indices, feats, seeds_in_batch, fan):
    """Order-free properties any reference execution satisfies."""
    H = len(fan)
    nc, ec, ids = res["nc"], res["ec"], res["ids"]
    B = int(nc[4])
    levels = [int(nc[4 + 2 * l]) for l in range(H + 1)]
    assert nc[5 + 2 * H] == sum(levels) == len(ids)
    assert len(np.unique(ids)) == len(ids) and (ids >= 0).all()
    assert ids[:B].tolist() == list(seeds_in_batch)
    src, dst = res["src_off"], res["dst_off"]
    assert len(src) == ec[2 + H]
    cum_nodes = np.cumsum(levels)
    e0 = 0
    for h in range(1, H + 1):
        e1 = int(ec[2 + h])
        assert (dst[e0:e1] < cum_nodes[h - 1]).all() and (src[e0:e1] < cum_nodes[h]).all()
        e0 = e1
    # every edge is a real edge of the graph
    s_ids, d_ids = ids[src], ids[dst]
    for e in np.random.RandomState(0).choice(len(src), size=min(len(src), 2000), replace=False):
        row = ds_indices[ds_indptr[d_ids[e]]:ds_indptr[d_ids[e] + 1]]
        assert s_ids[e] in row
    # hop-1: per seed min(f, deg) edges (no negative neighbours in the synthetic graphs)
    deg = (ds_indptr[ids[:B] + 1] - ds_indptr[ids[:B]])
    assert int(np.minimum(deg, fan[0]).sum()) == int(ec[3])
    if "features" in res:
        assert np.array_equal(res["features"], feats[ids])


def test_invariants_on_synthetic(oracle, small_ds):
    ds = small_ds
    lab = ds.labels[ds.train]
    for fan, B in (([25, 10], 500), ([25, 10, 5], 300)):
        r = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, fan)
        res = r.run_batch(ds.train, lab, 1)
        check_invariants(res, ds.indptr, ds.indices, ds.features, ds.train[B:2 * B], fan)


def test_short_last_batch_quirk(oracle):
    """Kernels.cu:224-227 vs :81-85: the kernel's batch_size is the clamped size, so the read offset
    of a short last batch is size*counter (restated, not fixed)."""
    g, V, F, indptr, indices, feats, labels, seeds = _toy()
    r = oracle.OracleRunner(indptr, indices, feats, V, F, 4, [2])
    res = r.run_batch(seeds, labels[seeds], 1)       # 6 seeds, batch 4, counter 1 -> size 2, offset 2*1
    assert res["nc"][4] == 2
    assert res["ids"][:2].tolist() == seeds[2:4].tolist()


def test_link_prediction_seed_layout(oracle, synth, small_ds):
    """lp_sage.py:87-90 -- seed batches laid out as [src | pos | neg] thirds; duplicates inside a batch
    follow the reference's last-occurrence-wins position rule (position_map[src_id] = idx, Kernels.cu:92)."""
    ds = small_ds
    B = 96
    seeds = synth.lp_trainingset(ds, 200, B, seed=3)
    assert len(seeds) % B == 0
    k = B // 3
    first = seeds[:B]
    for i in range(k):
        row = ds.indices[ds.indptr[first[i]]:ds.indptr[first[i] + 1]]
        assert first[k + i] in row and 0 <= first[2 * k + i] < ds.spec.V
    # force duplicates inside a batch
    seeds[5] = seeds[40]
    seeds[B + 7] = seeds[B + 8]
    lab = ds.labels[seeds]
    r = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, B, [5, 3])
    for counter in (0, 1):
        res = r.run_batch(seeds, lab, counter)
        ref = pyref.run_batch(ds.indptr, ds.indices, ds.features, seeds, lab, B, counter, [5, 3])
        assert_batch_equal(ref, res)
        assert res["ids"][:B].tolist() == seeds[counter * B:(counter + 1) * B].tolist()   # thirds stay in place


def test_five_hops_vs_pyref(oracle):
    rng = np.random.RandomState(5)
    V, F = 80, 2
    deg = rng.randint(0, 6, size=V)
    indptr = np.zeros(V + 1, np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = rng.randint(0, V, size=int(indptr[-1])).astype(np.int32)
    feats = rng.rand(V, F).astype(np.float32)
    labels = np.zeros(V, np.int32)
    seeds = rng.permutation(V)[:10].astype(np.int32)
    fan = [2, 2, 2, 2, 2]
    res = oracle.OracleRunner(indptr, indices, feats, V, F, 10, fan).run_batch(seeds, labels[seeds], 0)
    assert_batch_equal(pyref.run_batch(indptr, indices, feats, seeds, labels[seeds], 10, 0, fan), res)
    assert res["nc"][15] == len(res["ids"]) and res["ec"][7] == len(res["src_off"])


def test_openmp_oracle_is_byte_identical_to_the_serial_one(oracle, synth):
    """bench.py's "reference-semantics CPU, OpenMP" row runs lo_run_batch_omp: parallel draws / COO
    offsets / row copies around the serial, order-defining compaction.  Same bytes as the canonical serial schedule."""
    spec = synth.spec_for("products", scale=0.01)
    ds = synth.generate(spec)
    for B, fan in ((500, [25, 10]), (300, [10, 5, 3]), (1966, [4, 3])):
        a = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
        b = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
        for it in (0, 2, 1):          # the last batch of B = 1966 is short (-1 padded): same clamp either way
            ref = a.run_batch(ds.train, ds.labels[ds.train], it)
            got = b.run_batch(ds.train, ds.labels[ds.train], it, omp=True)
            for k in ("nc", "ec", "ids", "labels", "src_off", "dst_off", "features"):
                assert np.array_equal(ref[k], got[k]), (B, fan, it, k)


def test_c_generator_is_the_numpy_generator(oracle, synth):
    """oracle/synth_gen.c (host, OpenMP: the full-shape fixtures) == legion-1_amd/synth.py (numpy) bit for bit; the HIP
    generator is compared with numpy in tests/test_gpu_parity.py."""
    for name, scale in (("products", 0.03), ("papers100M", 0.0005), ("uk-union", 0.0004)):
        spec = synth.spec_for(name, scale=scale)
        ds = synth.generate(spec, with_features=False)
        indptr, indices = oracle.synth_csr(spec)
        assert np.array_equal(indptr, ds.indptr) and np.array_equal(indices, ds.indices)
        assert np.array_equal(oracle.synth_seed_ids(spec, 0, spec.n_train), ds.train)
        assert np.array_equal(oracle.synth_seed_ids(spec, spec.n_train, spec.n_train + spec.n_valid), ds.valid)
        assert np.array_equal(oracle.synth_labels_of(spec, ds.train), ds.labels[ds.train])


def test_full_shape_digests_products(oracle, synth):
    """tests/golden/full_shape_digests.json (BASELINE shapes at full size, serial oracle): the two ogbn-products cases are
    regenerated here (the only shape whose CSR takes seconds on the CPU); the OpenMP oracle must give the same digests.
    The papers100M / uk-union cases are checked on the GPU box (tests/test_gpu_full_shape.py)."""
    from conftest import load_golden, sha
    gold = load_golden("full_shape_digests")
    assert set(gold["cases"]) == {"products-25,10", "products-25,10,5", "papers100M-25,10,5", "papers100M-lp", "uk-union-25,10"}
    spec = synth.spec_for("products")
    indptr, indices = oracle.synth_csr(spec)
    seeds = oracle.synth_seed_ids(spec, 0, spec.n_train)
    lab = oracle.synth_labels_of(spec, seeds)
    for name in ("products-25,10", "products-25,10,5"):
        case = gold["cases"][name]
        assert case["E"] == int(indptr[-1]) and sha(indptr) == case["indptr_sha256"] and sha(seeds) == case["seeds_sha256"]
        r = oracle.OracleRunner(indptr, indices, None, spec.V, spec.F, case["batch"], case["fanout"], with_features=False)
        assert [b["counter"] for b in case["batches"]] == [0, 12, 24] and case["batches"][-1]["size"] == 4615   # the short last batch
        for i, want in enumerate(case["batches"]):
            res = r.run_batch(seeds, lab, want["counter"], gather=False, omp=(i != 1))
            for f in gold["fields"]:
                assert sha(res[f]) == want[f + "_sha256"], (name, want["counter"], f)


def test_assert_batch_equal_is_strict_about_missing_buffers(oracle, small_ds):
    """a buffer missing from either side fails the comparison (it used to be skipped silently); leaving a buffer
    out is an explicit `keys=`."""
    import pytest
    from conftest import BATCH_KEYS, KEYS_NO_FEATURES
    ds = small_ds
    orc = oracle.OracleRunner(ds.indptr, ds.indices, ds.features, ds.spec.V, ds.spec.F, 64, [5, 3])
    ref = orc.run_batch(ds.train, ds.labels[ds.train], 0)
    assert set(BATCH_KEYS) <= set(ref)
    assert_batch_equal(ref, dict(ref))
    for k in BATCH_KEYS:
        dropped = {x: v for x, v in ref.items() if x != k}
        with pytest.raises(AssertionError, match=k + ": missing from the batch under test"):
            assert_batch_equal(ref, dropped)
        with pytest.raises(AssertionError, match=k + ": missing from the reference batch"):
            assert_batch_equal(dropped, ref)
    renamed = {("feats" if k == "features" else k): v for k, v in ref.items()}      # a renamed field is a missing field
    with pytest.raises(AssertionError, match="features: missing"):
        assert_batch_equal(ref, renamed)
    assert_batch_equal(ref, renamed, keys=KEYS_NO_FEATURES)                         # ... unless the caller says it does not compare it
    wrong = dict(ref, src_off=ref["src_off"].copy())
    wrong["src_off"][3] ^= 1
    with pytest.raises(AssertionError, match="src_off: 1 mismatches"):
        assert_batch_equal(ref, wrong)
