import sys, os, hashlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import legion1_amd.capi as K, legion1_amd.synth as S
import bench as B_
L = K.lib()
for workload, fan in (("papers100M", [25, 10, 5]), ("products", [25, 10, 5])):
    spec = S.spec_for(workload)
    dev = torch.device("cuda", 0)
    indptr, indices, feats, E = B_.build_graph_on_gpu(K, spec, dev)
    B = 8000
    tr = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
    L.legion_synth_seed_ids(None, tr.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
    lab = torch.zeros(spec.n_train, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    seeds = dict(train=[((tr.data_ptr(), spec.n_train), (lab.data_ptr(), spec.n_train))])
    eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), feats.data_ptr(), spec.V, spec.F, seeds, B, fan, E=E)
    ref = {}
    bad = 0
    N = 300
    for rep in range(N):
        c = rep % 3
        eng.run_batch(0, c, gather=False)
        r = eng.result(0, with_features=False)
        h = hashlib.sha256(r["ids"].tobytes() + r["src_off"].tobytes() + r["dst_off"].tobytes() + r["nc"].tobytes() + r["ec"].tobytes()).hexdigest()
        if c not in ref: ref[c] = h
        elif ref[c] != h: bad += 1
    print(workload, "repetitions", N, "mismatches", bad, "edges", int(r["ec"][2 + len(fan)]), flush=True)
    eng.close()
    del indptr, indices, feats
    torch.cuda.empty_cache()
