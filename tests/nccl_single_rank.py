"""Child of tests/test_gpu_nccl_single.py: the RCCL ("nccl") code paths of legion1_amd.dist and legion1_amd.exchange on a
1-rank process group (RCCL refuses two ranks on one GPU, and the test box has one).  What runs here is what bench.py runs
on N GPUs, with N = 1: group creation with device_id, barrier, float64 MAX / SUM all-reduces on device tensors, the int64
hotness all-reduce of CandidateSelection, all_gather_object, and the exchange gather's three device all-to-alls (int32
counts into a tensor view, int32 lists and f32 rows with split sizes -- all empty at N = 1: zero-byte collectives)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1", LEGION_DIST_FORCE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # bench.py worker(): same call
import legion1_amd.capi as K  # noqa: E402
import legion1_amd.dist as D  # noqa: E402
import legion1_amd.synth as S  # noqa: E402
import oracle as O  # noqa: E402
from conftest import assert_batch_equal  # noqa: E402

L = K.lib()
L.legion_set_error_mode(K.ERR_RETURN)
L.legion_set_device_map(0, 0)
L.SetGPUDevice(0)
D.barrier(1)
emax, sums = D.aggregate(1.25, [3, 4.5], 1, device=dev)
assert emax == 1.25 and sums == [3.0, 4.5]
assert D.aggregate_max_vec([0.5, 2.0], 1, device=dev) == [0.5, 2.0]
assert D.allgather_object({"a": 1}, 1) == [{"a": 1}]

ds = S.generate(S.spec_for("products", scale=0.01))
V, F = ds.spec.V, ds.spec.F
B, fan, cap = 250, [10, 5], 3000
steps = (len(ds.train) - 1) // B
eng = K.Engine(ds.indptr, ds.indices, ds.features, V, F, dict(train=[(ds.train, ds.labels[ds.train])]), B, fan, G=1, train_step=steps)
eng.alloc_features()
for it in range(steps):
    eng.run_batch(0, it, is_presc=True)
before = K.read_dev(L.GPUCache_GetNodeAccessedMap(eng.cache, 0), np.int64, V)
D.allreduce_device_u64(K, L.GPUCache_GetNodeAccessedMap(eng.cache, 0), V, 1, device=dev)      # int64 SUM over 1 rank: unchanged
assert np.array_equal(before, K.read_dev(L.GPUCache_GetNodeAccessedMap(eng.cache, 0), np.int64, V))
eng.build_cache(cache_agg_mode=0, node_capacity=cap, edge_capacity=0, train_step=steps)
orc = O.OracleRunner(ds.indptr, ds.indices, ds.features, V, F, B, fan)
from legion1_amd.exchange import ExchangeGather  # noqa: E402
xg = ExchangeGather(K, eng, 0, 1, F, dev, eng.num_ids)
assert xg.nccl
for it in range(2):
    ref = orc.run_batch(ds.train, ds.labels[ds.train], it)
    eng.run_batch(0, it, gather=False, plan=False)
    feat = eng.out[0][0]["feat"]
    L.d_memset_async(feat.ptr, 0xFF, feat.nbytes, None)
    L.d_stream_sync(None)
    info = xg.run(None, eng.pools[0])
    xg.wait()
    assert info == {"rows_requested": 0, "rows_served": 0, "per_owner": [0]}
    assert_batch_equal(ref, eng.result(0))
assert xg.host_syncs_per_batch == 1.0 and xg.staging_syncs == 0
xg.close()
eng.close()
dist.barrier()
dist.destroy_process_group()
print("nccl single rank ok")
