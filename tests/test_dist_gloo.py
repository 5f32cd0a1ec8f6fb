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


# ---------------------------------------------------------------------------------------------------------------------
# The owner-computes exchange gather at the clique size BASELINE.json states (Kg = 8): the orchestration of
# legion1_amd/exchange.py -- split sizes, buffer growth, order and arguments of the three all-to-alls -- run by 8 gloo ranks
# on CPU tensors.  The four device steps (plan / local / serve / scatter: HIP kernels, parity-tested on the GPU in
# tests/test_gpu_unified_ipc.py) are replaced by host stand-ins that follow the same contract; the GPU box allows at most 6
# processes on its card, so this is where all eight ranks of the protocol meet.
# ---------------------------------------------------------------------------------------------------------------------
class _HostOps:
    """Contract of exchange.HipOps on CPU tensors.  Shard row i of clique member j holds the F values j * 1e6 + i * 8 + c."""

    def __init__(self, me, world, F, cap, rs):
        self.me, self.world, self.F, self.cap, self.rs = me, world, F, cap, rs
        self.shard = (me * 1e6 + np.arange(cap)[:, None] * 8 + np.arange(F)[None, :]).astype(np.float32)
        self.feat = None

    def begin(self, sampler_stream):
        pass

    def plan(self, pool, req_row, req_dst, counts):
        n_rows = pool["rows"]
        owner = self.rs.randint(-1, self.world, size=n_rows)        # -1: backing table, me: own shard, else a peer
        row = self.rs.randint(0, self.cap, size=n_rows)
        self.owner, self.row = owner, row
        self.feat = np.full((n_rows, self.F), np.nan, np.float32)
        k = 0
        cnt = np.zeros(16, np.int32)
        for j in range(self.world):                                 # owner-major request lists
            if j == self.me:
                continue
            idx = np.flatnonzero(owner == j)
            req_row[k:k + len(idx)] = torch.from_numpy(row[idx].astype(np.int32))
            req_dst[k:k + len(idx)] = torch.from_numpy(idx.astype(np.int32))
            cnt[j] = len(idx)
            k += len(idx)
        counts.copy_(torch.from_numpy(cnt))
        local = (owner == self.me) | (owner < 0)                    # "legion_exchange_local"
        self.feat[local] = -1.0

    def counts_to_host(self, d_both, h_both):
        h_both.copy_(d_both)

    def serve(self, wanted, n, out_rows):
        out_rows[:n] = torch.from_numpy(self.shard[wanted[:n].numpy()])

    def scatter(self, pool, in_rows, req_dst, n):
        self.feat[req_dst[:n].numpy()] = in_rows[:n].numpy()

    def finish(self, sampler_stream):
        pass

    def wait(self):
        pass

    def close(self):
        pass


def _exchange_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from legion1_amd.exchange import ExchangeGather
        F, cap = 12, 5000
        ops = _HostOps(rank, world, F, cap, np.random.RandomState(100 + rank))
        xg = ExchangeGather(None, None, rank, world, F, "cpu", max_rows=40000, ops=ops)
        allocs = []
        for it, n_rows in enumerate((3000, 20000, 20000, 500, 0, 20000)):
            info = xg.run(None, {"rows": n_rows})
            allocs.append(xg.allocations)
            peer = (ops.owner >= 0) & (ops.owner != rank)
            assert info["rows_requested"] == int(peer.sum()) and info["per_owner"][rank] == 0
            assert info["per_owner"] == [int(((ops.owner == j) & peer).sum()) for j in range(world)]
            want = (ops.owner[:, None] * 1e6 + ops.row[:, None] * 8 + np.arange(F)[None, :]).astype(np.float32)
            assert np.array_equal(ops.feat[peer], want[peer])               # every requested row came from the right owner and row
            assert (ops.feat[~peer] == -1.0).all()                          # local rows were left to the local gather
        assert allocs[2] == allocs[1] and allocs[5] == allocs[1]            # no allocation once the buffers have grown
        assert xg.host_syncs_per_batch == 1.0 and xg.staging_syncs == 0
        served = dist.all_gather_object
        tot = [None] * world
        served(tot, (info["rows_requested"], info["rows_served"]))
        assert sum(a for a, _ in tot) == sum(b for _, b in tot)            # what is asked for is what is served, clique-wide
        xg.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))


def test_exchange_gather_protocol_with_eight_ranks():
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert [o[1] for o in out] == ["ok"] * world, [o for o in out if o[1] != "ok"]
