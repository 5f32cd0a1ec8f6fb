"""N > 1 path on CPU: two ranks (gloo) shard the seed list the way the reference splits it
(tid % G, GPUGraphStore.cu:332-346), run their batches independently (the oracle stands in for the
device here -- it is the checker, this test is about the sharding / aggregation plumbing that
bench.py uses) and aggregate with legion1_amd.dist: max-over-ranks time, summed units."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import legion1_amd.dist as D
    import legion1_amd.synth as S
    import oracle as O
    spec = S.spec_for("products", scale=0.004)
    ds = S.generate(spec)
    mine = D.shard_seeds(ds.train, rank, world)
    assert (mine % world == rank).all()
    B, fan = 100, [5, 3]
    steps = D.train_steps([len(D.shard_seeds(ds.train, r, world)) for r in range(world)], B)
    runner = O.OracleRunner(ds.indptr, ds.indices, None, spec.V, spec.F, B, fan, with_features=False)
    edges = nodes = 0
    for it in range(steps):
        res = runner.run_batch(mine, ds.labels[mine], it, gather=False)
        edges += int(res["ec"][2 + len(fan)])
        nodes += int(res["nc"][5 + 2 * len(fan)])
    D.barrier(world)
    elapsed = 1.0 + rank                     # synthetic clock: the MAX over ranks must win
    emax, (e_sum, n_sum) = D.aggregate(elapsed, [edges, nodes], world)
    # bench.py's timing: per window the MAX over ranks (every rank runs the same windows)
    wins = D.aggregate_max_vec([0.5 + rank, 2.0 - rank, 1.0], world)
    assert wins == [1.5, 2.0, 1.0]
    # link-prediction lists: triples dealt by src % world, thirds kept, pos/neg from the triple's GLOBAL stream position
    lp = S.lp_trainingset(ds, 600, 30, rank=rank, world=world)
    k = 10
    lp3 = lp.reshape(-1, 3, k)
    assert (lp3[:, 0, :] % world == rank).all()
    everyone = D.allgather_object(lp3.transpose(0, 2, 1).reshape(-1, 3).tolist(), world)
    # torch tensors shard the same way as numpy arrays
    t_mine = D.shard_seeds(torch.from_numpy(ds.train), rank, world)
    assert np.array_equal(t_mine.numpy(), mine)
    q.put((rank, steps, edges, nodes, emax, e_sum, n_sum, mine.tolist(), everyone))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, n0, emax0, es0, ns0, ids0, lp0), (r1, s1, e1, n1, emax1, es1, ns1, ids1, lp1) = out
    assert s0 == s1 > 0
    assert emax0 == emax1 == 2.0                      # max over ranks
    assert es0 == es1 == e0 + e1 and ns0 == ns1 == n0 + n1   # whole-job totals
    sys.path.insert(0, ROOT)
    import legion1_amd.synth as S
    ds = S.generate(S.spec_for("products", scale=0.004))
    assert sorted(ids0 + ids1) == sorted(ds.train.tolist()) and not set(ids0) & set(ids1)
    import legion1_amd.dist as D
    rate, ms = D.throughput_line(es0, emax0, s0)
    assert rate == es0 / 2.0 and ms == 2.0 / s0 * 1e3
    # the two per-GPU link-prediction lists together hold exactly the triples of the 1-GPU list (padding repeats aside)
    assert lp0 == lp1
    one = S.lp_trainingset(ds, 600, 30).reshape(-1, 3, 10).transpose(0, 2, 1).reshape(-1, 3)
    got = {tuple(t) for part in lp0 for t in part}
    assert {tuple(t) for t in one.tolist()} == got
