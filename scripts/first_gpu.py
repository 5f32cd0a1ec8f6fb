import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import legion1_amd.synth as S
import legion1_amd.capi as K
import oracle as O

def compare(a, b, tag):
    ok = True
    for k in ("nc", "ec", "ids", "labels", "src_off", "dst_off", "features"):
        if k not in a or k not in b: continue
        same = a[k].shape == b[k].shape and np.array_equal(a[k], b[k])
        if not same:
            ok = False
            print(tag, "MISMATCH", k, a[k].shape, b[k].shape)
            if a[k].shape == b[k].shape:
                bad = np.nonzero(a[k].reshape(-1) != b[k].reshape(-1))[0]
                print("  first bad", bad[:5], a[k].reshape(-1)[bad[:5]], b[k].reshape(-1)[bad[:5]])
            else:
                print(a[k][:16], b[k][:16])
    print(tag, "OK" if ok else "FAIL")
    return ok

L = K.lib()
print(L.legion_version())
L.legion_set_error_mode(K.ERR_RETURN)
# RNG probe
idx = np.array([0,1,2,24,25,1023,1024,199999,200000,2147483,4999999,12199999], dtype=np.int32)
deg = np.full_like(idx, 25)
di, dd, dk = K.DevBuf.from_numpy(idx), K.DevBuf.from_numpy(deg), K.DevBuf(idx.nbytes)
L.legion_rng_probe(None, di.ptr, dd.ptr, dk.ptr, len(idx)); L.d_stream_sync(None)
print("rng gpu", dk.to_numpy(np.int32, len(idx)), "oracle", [O.sample_index(int(i), 25) for i in idx])

allok = True
for name, scale, B, fan in (("products", 0.004, 64, [3, 2]), ("products", 0.01, 1000, [25, 10]), ("papers100M", 0.001, 8000, [25, 10, 5])):
    spec = S.spec_for(name, scale=scale)
    ds = S.generate(spec)
    lab = ds.labels[ds.train]
    orc = O.OracleRunner(ds.indptr, ds.indices, ds.features, spec.V, spec.F, B, fan)
    eng = K.Engine(ds.indptr, ds.indices, ds.features, spec.V, spec.F, dict(train=[(ds.train, lab)]), B, fan)
    eng.alloc_features()
    for it in range(min(3, max(1, len(ds.train) // B))):
        t0 = time.time(); ref = orc.run_batch(ds.train, lab, it); t1 = time.time()
        eng.run_batch(0, it); t2 = time.time()
        got = eng.result(0)
        allok &= compare(ref, got, f"{spec.name} B={B} fan={fan} it={it} oracle {t1-t0:.3f}s gpu {t2-t1:.3f}s nodes={ref['nc'][5+2*len(fan)]} edges={ref['ec'][2+len(fan)]}")
    eng.close()
print("ALL OK" if allok else "SOME FAILED")
sys.exit(0 if allok else 1)
