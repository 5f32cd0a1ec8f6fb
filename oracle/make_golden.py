#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.  TEST INFRASTRUCTURE.

The reference ships no tests, golden vectors or fixtures for this path, and
its CUDA sources cannot be built or run here, so the vectors are produced by

  * oracle/thrust_probe (the Thrust headers of this image, host side) for the RNG -- the one piece
    of third-party arithmetic on the path (rng_kat.json), and
  * the CPU oracle (oracle/legion_oracle.c) for everything else.  Those files pin the oracle
    against regressions and pin the HIP path at the listed sizes; they are not reference outputs.

Usage:  python oracle/make_golden.py        (from the repository root; the small fixtures, seconds)
        python oracle/make_golden.py full   (tests/golden/full_shape_digests.json: the five BASELINE shapes at full
                                             size, ~25 GB of host memory and a few minutes on 8 cores)
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import legion1_amd.synth as S  # noqa: E402
import oracle as O  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def rng_kat():
    subprocess.check_call(["make", "-C", HERE, "thrust_probe"], stdout=subprocess.DEVNULL)
    idxs = [0, 1, 2, 3, 24, 25, 26, 255, 256, 1023, 1024, 1025, 4095, 65535, 65536, 199999, 200000, 1048575,
            2147483, 2199999, 4999999, 9999999, 12199999, 12207999, 100000000, 2147483646]
    degs = [1, 2, 3, 5, 10, 24, 25, 26, 1000, 100000, 2147483647]
    pairs = [(i, d) for i in idxs for d in degs]
    inp = "".join(f"{i} {d}\n" for i, d in pairs)
    out = subprocess.run([os.path.join(HERE, "thrust_probe")], input=inp, capture_output=True, text=True, check=True).stdout
    rows = [[int(t) for t in line.split()] for line in out.strip().splitlines()]
    nth = int(subprocess.run([os.path.join(HERE, "thrust_probe"), "n", "10000"], capture_output=True, text=True,
                             check=True).stdout.strip())
    return {"source": "oracle/thrust_probe.cpp compiled against /opt/rocm/include/thrust (rocThrust, host side)",
            "columns": ["idx", "deg", "k", "x"], "rows": rows, "minstd_10000th": nth,
            "thrust_documented_10000th": 399268537}


def toy_graph():
    """V = 16, hand-checkable: includes a degree-0 node, degree < fan-out, a -1 neighbour entry and a hub."""
    adj = {
        0: [1, 2, 3], 1: [0, 4], 2: [5], 3: [], 4: [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1], 5: [2, -1, 9],
        6: [4], 7: [4, 8], 8: [7], 9: [5, 4, 10], 10: [9], 11: [4], 12: [13], 13: [12, 14], 14: [13, 15], 15: [14, 0],
    }
    V = 16
    indptr = np.zeros(V + 1, dtype=np.int64)
    for v in range(V):
        indptr[v + 1] = indptr[v] + len(adj[v])
    indices = np.array([d for v in range(V) for d in adj[v]], dtype=np.int32)
    F = 4
    feats = (np.arange(V * F, dtype=np.float32).reshape(V, F) * 0.5) - 3.0
    labels = (np.arange(V, dtype=np.int32) * 7) % 5
    return V, F, indptr, indices, feats, labels


def toy_cases():
    V, F, indptr, indices, feats, labels = toy_graph()
    seeds = np.array([0, 4, 9, 13, 3, 5], dtype=np.int32)
    out = {"V": V, "F": F, "indptr": indptr.tolist(), "indices": indices.tolist(), "features": feats.reshape(-1).tolist(),
           "labels": labels.tolist(), "seeds": seeds.tolist(), "cases": []}
    for fan, B, counter in (([2, 2], 4, 0), ([3, 2], 4, 1), ([2, 2, 2], 3, 0), ([25, 10], 6, 0), ([1], 6, 0)):
        r = O.OracleRunner(indptr, indices, feats, V, F, B, fan)
        res = r.run_batch(seeds, labels[seeds], counter)
        out["cases"].append({"fanout": fan, "batch": B, "counter": counter, "nc": res["nc"].tolist(), "ec": res["ec"].tolist(),
                             "ids": res["ids"].tolist(), "labels": res["labels"].tolist(), "src_off": res["src_off"].tolist(),
                             "dst_off": res["dst_off"].tolist(), "features_sha256": sha(res["features"])})
    return out


def medium_digests():
    spec = S.spec_for("products", scale=0.04)   # V ~ 98k, power-law degrees
    ds = S.generate(spec)
    lab = ds.labels[ds.train]
    out = {"spec": {"name": spec.name, "V": spec.V, "F": spec.F, "E": ds.E, "n_train": spec.n_train},
           "dataset_sha256": {"indptr": sha(ds.indptr), "indices": sha(ds.indices), "features": sha(ds.features),
                              "labels": sha(ds.labels), "train": sha(ds.train)}, "cases": []}
    for B in (1, 1000, 7000):
        for fan in ([25, 10], [25, 10, 5]):
            r = O.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
            for counter in (0, 1):
                res = r.run_batch(ds.train, lab, counter)
                out["cases"].append({"batch": B, "fanout": fan, "counter": counter, "nc": res["nc"].tolist(),
                                     "ec": res["ec"].tolist(), **{k + "_sha256": sha(res[k]) for k in
                                                                   ("ids", "labels", "src_off", "dst_off", "features")}})
    return out


def schedule_table():
    rows = []
    for name in ("products", "papers100M", "uk-union"):
        sh = S.SHAPES[name]
        for G in (1, 2, 4, 8):
            # tid % G split of a permutation-ordered list is near-even; use the exact counts of the generator
            spec = S.spec_for(name)
            n1, n2, n3 = spec.n_train, spec.n_valid, spec.n_test
            tr = np.bincount(S.seed_ids(spec, 0, n1) % G, minlength=G)
            va = np.bincount(S.seed_ids(spec, n1, n1 + n2) % G, minlength=G)
            te = np.bincount(S.seed_ids(spec, n1 + n2, n1 + n2 + n3) % G, minlength=G)
            steps, tb, vb, sb = O.coordinate(tr, va, te, 8000)
            epoch = 3
            probe = [0, 1, int(steps[0]) - 1, int(steps[0]), int(steps[0] + steps[1]) - 1, int(steps[0] + steps[1]),
                     int((steps[0] + steps[1]) * epoch) - 1, int((steps[0] + steps[1]) * epoch), O.max_step(steps, epoch) - 1]
            rows.append({"dataset": name, "G": G, "train_num": tr.tolist(), "valid_num": va.tolist(), "test_num": te.tolist(),
                         "steps": steps.tolist(), "valid_bs": vb.tolist(), "test_bs": sb.tolist(), "epoch": epoch,
                         "max_step": O.max_step(steps, epoch), "probe": probe,
                         "mode_local": [list(O.schedule(steps, epoch, b)) for b in probe]})
    return rows


def cache_fixture():
    """Hotness -> ranking -> cost model -> placement on a fixed small graph (Kg = 1, 2, 4)."""
    spec = S.spec_for("products", scale=0.002)
    ds = S.generate(spec)
    V = spec.V
    rng = np.random.RandomState(7)
    out = {"V": V, "cases": []}
    for Kg in (1, 2, 4):
        acc_n = [rng.zipf(1.6, V).astype(np.uint64) % 50 for _ in range(Kg)]
        acc_e = [rng.zipf(1.5, V).astype(np.uint64) % 40 for _ in range(Kg)]
        AF, QF = O.candidate_selection(acc_n, V)
        AT, QT = O.candidate_selection(acc_e, V)
        for budget in (200_000, 1_000_000, 3_000_000):
            cm = O.cost_model(AF, AT, QT, ds.indptr, V, spec.F, budget, Kg, [123456, 654321], [5000] * Kg, 24)
            out["cases"].append({"Kg": Kg, "budget": budget, "seed": 7, **cm, "QF_sha256": sha(QF), "QT_sha256": sha(QT),
                                 "AF_sha256": sha(AF), "AT_sha256": sha(AT), "QF_head": QF[:8].tolist(), "QT_head": QT[:8].tolist()})
    # placement rule (rank t, Kg) -> (owner, row, global slot), GPUCache.cu:88-108
    out["placement"] = [{"t": t, "Kg": Kg, "Ki": Ki, "capacity": cap, "owner": t % Kg + Ki * Kg, "row": t // Kg,
                         "slot": (t % Kg) * cap + t // Kg}
                        for Kg in (1, 2, 4, 8) for Ki in (0, 1) for cap in (5, 1000) for t in (0, 1, 2, 7, 8, 9, 39)]
    return out


PCM_FREE_CASE = dict(workload="products", batch=8000, fanout=[25, 10], presc_steps=8, budget_fracs=[0.10, 0.25, 0.60])


def cost_model_pcm_free(case=PCM_FREE_CASE):
    """The cache plan of a BASELINE shape WITHOUT the two Intel-PCM counters (counters == NULL: what the `legion` server of this repository
    always passes) -- the transaction estimate `sum over ranked rows of edge hotness x ceil((8 + 4 min(deg, 16)) / 64)` replaced round 3's
    "one transaction per sampled edge" and shifts alpha and both capacities for every such caller: this pins the numbers, so a
    later edit of the weight is visible.  Full products shape, 8 pre-sampling batches of {25,10}, B = 8000, G = Kg = 1."""
    spec = S.spec_for(case["workload"])
    indptr, indices = O.synth_csr(spec)
    B, fan, steps = case["batch"], case["fanout"], case["presc_steps"]
    orc = O.OracleRunner(indptr, indices, None, spec.V, spec.F, B, fan, with_features=False)
    train = S.seed_ids(spec, 0, spec.n_train)
    labels = S.labels(spec, train)
    max_ids = 0
    for it in range(steps):
        r = orc.run_batch(train, labels, it, is_presc=True, gather=False)
        max_ids = max(max_ids, int(r["nc"][5 + 2 * len(fan)]))
    AF, QF = O.candidate_selection([orc.node_access_time], spec.V)
    AT, QT = O.candidate_selection([orc.edge_access_time], spec.V)
    out = dict(case, V=spec.V, E=int(indptr[-1]), F=spec.F, max_ids=max_ids, node_hotness_sha256=sha(orc.node_access_time),
               edge_hotness_sha256=sha(orc.edge_access_time), QF_sha256=sha(QF), QT_sha256=sha(QT), plans=[])
    for frac in case["budget_fracs"]:
        budget = int(spec.V * spec.F * 4 * frac)
        cm = O.cost_model(AF, AT, QT, indptr, spec.V, spec.F, budget, 1, None, [max_ids], steps)
        out["plans"].append(dict(budget_frac=frac, cache_memory=budget, **cm))
    return out


def lp_seed_lists():
    """Link-prediction seed lists ([src | pos | neg] thirds per batch, lp_sage.py:87-90) of synth.lp_trainingset: the 1-GPU list
    and the two lists of a 2-GPU job (triples dealt by src % 2).  Pins the on-disk layout of `trainingset` / `trainingset_<G>_<g>`."""
    spec = S.spec_for("products", scale=0.002)
    ds = S.generate(spec, with_features=False)
    n, B = 50, 30
    out = {"workload": spec.name, "n_triples": n, "batch": B, "seed": 1, "lists": {}}
    out["lists"]["1of1"] = S.lp_trainingset(ds, n, B).tolist()
    for r in (0, 1):
        out["lists"]["%dof2" % r] = S.lp_trainingset(ds, n, B, rank=r, world=2).tolist()
    return out


# The five BASELINE.json configurations at their full shape (bench.py's workloads; B and fan-outs as benched).
FULL_SHAPES = [
    dict(name="products-25,10", workload="products", fanout=[25, 10], task="node", batch=8000),
    dict(name="products-25,10,5", workload="products", fanout=[25, 10, 5], task="node", batch=8000),
    dict(name="papers100M-25,10,5", workload="papers100M", fanout=[25, 10, 5], task="node", batch=8000),
    dict(name="papers100M-lp", workload="papers100M", fanout=[25, 10, 5], task="lp", batch=7998),
    dict(name="uk-union-25,10", workload="uk-union", fanout=[25, 10], task="node", batch=8000),
]


def full_shape_seeds(case, spec, indptr, indices):
    """(seed list, labels) of a FULL_SHAPES case: the training ids (1-GPU split) or the [src | pos | neg] list."""
    train = O.synth_seed_ids(spec, 0, spec.n_train)
    if case["task"] == "lp":
        ds = S.Dataset(spec, indptr, indices, None, None, train, None, None)
        train = S.lp_trainingset(ds, spec.n_train, case["batch"], seed=1)
    return train, O.synth_labels_of(spec, train)


def full_shape_counters(case, n_seeds):
    """Batch 0, a middle batch and the LAST batch generator call that still returns seeds: the short one of
    Kernels.cu:224 (size = total_cap - B * counter, read at offset size * counter) for node lists; LP lists are
    whole batches, so there it is the last full (padded) one."""
    B = case["batch"]
    last = (n_seeds - 1) // B
    return [0, last // 2, last]


def full_shape_digests(only=None):
    """SHA-256 of every output buffer of the canonical schedule (SERIAL oracle, lo_run_batch) at the BASELINE shapes.
    tests/test_gpu_full_shape.py compares the HIP path with these AND with the OpenMP oracle run on the GPU box."""
    out = {"generator": "oracle/make_golden.py full: oracle/synth_gen.c CSR (spec legion-1_amd/synth.py, skew 205) + lo_run_batch (serial)",
           "fields": ["nc", "ec", "ids", "labels", "src_off", "dst_off"], "cases": {}}
    graphs = {}
    for case in FULL_SHAPES:
        if only and case["name"] not in only:
            continue
        spec = S.spec_for(case["workload"])
        if case["workload"] not in graphs:
            graphs.clear()                      # one full-size CSR at a time (uk-union: 23 GB)
            graphs[case["workload"]] = O.synth_csr(spec)
        indptr, indices = graphs[case["workload"]]
        seeds, lab = full_shape_seeds(case, spec, indptr, indices)
        r = O.OracleRunner(indptr, indices, None, spec.V, spec.F, case["batch"], case["fanout"], with_features=False)
        entry = {"workload": case["workload"], "fanout": case["fanout"], "task": case["task"], "batch": case["batch"],
                 "V": spec.V, "E": int(indptr[-1]), "n_seeds": int(len(seeds)), "seeds_sha256": sha(seeds),
                 "indptr_sha256": sha(indptr), "indices_head_sha256": sha(indices[:1 << 24]), "batches": []}
        for counter in full_shape_counters(case, len(seeds)):
            res = r.run_batch(seeds, lab, counter, gather=False)
            H = len(case["fanout"])
            entry["batches"].append({"counter": counter, "size": int(res["nc"][4]), "n_nodes": int(res["nc"][5 + 2 * H]),
                                     "n_edges": int(res["ec"][2 + H]),
                                     **{k + "_sha256": sha(res[k]) for k in out["fields"]}})
            print(case["name"], entry["batches"][-1]["counter"], entry["batches"][-1]["size"], entry["batches"][-1]["n_nodes"],
                  entry["batches"][-1]["n_edges"], flush=True)
        out["cases"][case["name"]] = entry
        del r
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        path = os.path.join(GOLD, "full_shape_digests.json")
        data = full_shape_digests(only=sys.argv[2:] or None)
        if sys.argv[2:] and os.path.exists(path):      # regenerate selected cases only
            with open(path) as f:
                old = json.load(f)
            old["cases"].update(data["cases"])
            data = old
        with open(path, "w") as f:
            json.dump(data, f, indent=1)
        print("wrote full_shape_digests", os.path.getsize(path), "bytes")
        return
    for name, fn in (("rng_kat", rng_kat), ("toy_batches", toy_cases), ("medium_digests", medium_digests),
                     ("schedule_table", schedule_table), ("cache_fixture", cache_fixture), ("lp_seed_lists", lp_seed_lists),
                     ("cost_model_pcm_free", cost_model_pcm_free)):
        data = fn()
        with open(os.path.join(GOLD, name + ".json"), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(GOLD, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
