"""Owner-computes exchange variant of the unified-cache feature gather (SURVEY 5 option (b)), one process per GPU.

The reference reads the feature rows cached on the other GPUs of the clique in-kernel over NVLink
(zero_copy_with_aggregated_cache, Kernels.cu:662-702); ``get_feature_kernel`` does the same over xGMI.  This is the
collective formulation for the case that fine-grained peer loads fall short of the links: per batch

  1. plan     (HIP)   rows cached on another clique member are listed per owner; every other row gets its local
                      source address                                                 -> legion_exchange_plan
  1b. local   (HIP)   own-shard and backing-table rows are gathered on a side stream, concurrently with 2-5
                                                                                     -> legion_exchange_local
  2. counts   (RCCL)  all-to-all of the Kg request counts on the device, then ONE pinned asynchronous copy of
                      [sent | received] to the host -- the only host synchronisation of the batch: torch's
                      all_to_all_single needs its split sizes on the host
  3. requests (RCCL)  all-to-all of the request lists (4 bytes per row)
  4. serve    (HIP)   every owner gathers the requested rows from ITS shard in local HBM -> legion_exchange_serve
  5. rows     (RCCL)  all-to-all of the rows (4F bytes per row) -- the xGMI traffic, in bulk transfers
  6. scatter  (HIP)   rows to their place in the batch's feature buffer             -> legion_exchange_scatter

All buffers are allocated once and only ever grow (no allocation in the steady state); nothing waits at the end of a
batch: ``run`` records an event behind the scatter and makes the sampler stream wait for it, so the next batch's
launches queue up behind this one.  Results are bit-identical to the in-kernel variant (same rows, verbatim copies).
With gloo (CPU tests, the one-GPU rehearsal) the three collectives are staged through host memory; those extra
synchronisations are counted separately (``staging_syncs``) -- they do not exist under RCCL.

The device work sits behind a small ``ops`` object (HipOps: the C ABI calls + stream / event plumbing), so that the
orchestration -- split sizes, buffer growth, the order and arguments of the three all-to-alls -- is the same code in the
8-rank CPU test (tests/test_dist_gloo.py, gloo, host stand-ins for the four kernels) and on the GPUs.
"""
from __future__ import annotations


class HipOps:
    """The device side of the exchange: C ABI launches on torch's current stream (RCCL orders its work behind it), the
    local gather on a side stream, events instead of host waits."""

    def __init__(self, capi, eng, me: int, F: int):
        import torch
        self.K, self.L, self.eng, self.me, self.F, self.torch = capi, capi.lib(), eng, me, F, torch
        L = self.L
        self.ev_counts = torch.cuda.Event()
        self.ev_in, self.ev_plan, self.ev_local, self.done = (L.d_event_create() for _ in range(4))
        self.side = L.d_stream_create()
        self.sp = None

    def begin(self, sampler_stream):
        self.ts = self.torch.cuda.current_stream()
        self.sp = self.ts.cuda_stream
        self.L.d_event_record(self.ev_in, sampler_stream)
        self.L.d_stream_wait_event(self.sp, self.ev_in)

    def plan(self, pool, req_row, req_dst, counts):
        L = self.L
        if L.legion_exchange_plan(self.sp, self.eng.cache, self.eng.noder, pool, self.me, req_row.data_ptr(), req_dst.data_ptr(),
                                  counts.data_ptr()) != 0:
            self.K.check()
            raise RuntimeError("legion_exchange_plan failed")
        # own-shard and backing-table rows: on the side stream, while the counts travel and the peers' rows are exchanged
        L.d_event_record(self.ev_plan, self.sp)
        L.d_stream_wait_event(self.side, self.ev_plan)
        if L.legion_exchange_local(self.side, self.eng.cache, self.eng.noder, pool, self.me) != 0:
            self.K.check()
            raise RuntimeError("legion_exchange_local failed")
        L.d_event_record(self.ev_local, self.side)

    def counts_to_host(self, d_both, h_both):
        h_both.copy_(d_both, non_blocking=True)
        self.ev_counts.record(self.ts)
        self.ev_counts.synchronize()                 # THE host synchronisation of the batch

    def serve(self, wanted, n, out_rows):
        self.L.legion_exchange_serve(self.sp, self.eng.cache, self.me, wanted.data_ptr(), n, out_rows.data_ptr())

    def scatter(self, pool, in_rows, req_dst, n):
        self.L.legion_exchange_scatter(self.sp, pool, in_rows.data_ptr(), req_dst.data_ptr(), n, self.F)

    def finish(self, sampler_stream):
        L = self.L
        L.d_stream_wait_event(self.sp, self.ev_local)
        L.d_event_record(self.done, self.sp)
        L.d_stream_wait_event(sampler_stream, self.done)     # later launches on the sampler stream see the rows

    def wait(self):
        self.torch.cuda.current_stream().synchronize()
        self.L.d_stream_sync(self.side)
        self.K.check()

    def close(self):
        for ev in (self.ev_in, self.ev_plan, self.ev_local, self.done):
            self.L.d_event_destroy(ev)
        self.L.d_stream_destroy(self.side)


class ExchangeGather:
    def __init__(self, capi, eng, me: int, world: int, F: int, device, max_rows: int, init_rows: int = 0, ops=None):
        import torch
        import torch.distributed as dist
        self.me, self.world, self.F, self.dev = me, world, F, device
        self.torch, self.dist = torch, dist
        self.ops = ops if ops is not None else HipOps(capi, eng, me, F)
        self.nccl = dist.get_backend() == "nccl"
        self.on_gpu = torch.device(device).type == "cuda"
        self.max_rows = int(max_rows)
        self.req_row = torch.empty(self.max_rows, dtype=torch.int32, device=device)   # 4 B per row of the static bound
        self.req_dst = torch.empty(self.max_rows, dtype=torch.int32, device=device)
        self.counts = torch.zeros(16, dtype=torch.int32, device=device)               # int32[2 * LEGION_MAX_DEVICE]
        self.d_both = torch.zeros(2 * world, dtype=torch.int32, device=device)        # [sent | received]
        self.h_both = torch.zeros(2 * world, dtype=torch.int32)
        if self.on_gpu:
            self.h_both = self.h_both.pin_memory()
        # grow-only row buffers (served rows out, requested rows in) and the list of rows to serve
        self.cap_recv = self.cap_send = 0
        self.wanted = self.out_rows = self.in_rows = None
        self.batches = self.protocol_syncs = self.staging_syncs = self.allocations = 0
        self._reserve(int(init_rows), int(init_rows))
        self.last = {}

    # -- buffers --------------------------------------------------------------------------------------------------
    def _reserve(self, n_send: int, n_recv: int):
        torch = self.torch
        if n_recv > self.cap_recv or self.wanted is None:
            self.cap_recv = max(1024, min(self.max_rows * self.world, n_recv + n_recv // 4))
            self.wanted = torch.empty(self.cap_recv, dtype=torch.int32, device=self.dev)
            self.out_rows = torch.empty((self.cap_recv, self.F), dtype=torch.float32, device=self.dev)
            self.allocations += 1
        if n_send > self.cap_send or self.in_rows is None:
            self.cap_send = max(1024, min(self.max_rows, n_send + n_send // 4))
            self.in_rows = torch.empty((self.cap_send, self.F), dtype=torch.float32, device=self.dev)
            self.allocations += 1

    @property
    def host_syncs_per_batch(self):
        """Host synchronisations of the protocol itself (what an RCCL run pays), averaged over the batches so far."""
        return self.protocol_syncs / max(1, self.batches)

    # -- collectives ----------------------------------------------------------------------------------------------
    def _a2a(self, out, inp, out_splits, in_splits):
        torch, dist = self.torch, self.dist
        if self.nccl or not self.on_gpu:
            dist.all_to_all_single(out, inp, out_splits, in_splits)
            return
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), out_splits, in_splits)      # .cpu(): a staging synchronisation (gloo only)
        out.copy_(h_out)
        self.staging_syncs += 1

    def run(self, sampler_stream, pool):
        """Gather the features of the batch that was just sampled on `sampler_stream` into the current pipe's buffer.
        Collective: every rank of the clique calls it once per batch.  Returns as soon as everything is enqueued; work
        queued on `sampler_stream` afterwards runs behind the scatter (``wait()`` blocks the host until then)."""
        torch, ops, W = self.torch, self.ops, self.world
        ops.begin(sampler_stream)
        ops.plan(pool, self.req_row, self.req_dst, self.counts)
        # split sizes: the counts are exchanged on the device; one pinned copy brings [sent | received] to the host
        self.d_both[:W].copy_(self.counts[:W])
        if self.nccl:
            self.dist.all_to_all_single(self.d_both[W:], self.d_both[:W])
        ops.counts_to_host(self.d_both, self.h_both)
        self.protocol_syncs += 1
        if not self.nccl:                                      # gloo: the counts cross on the host
            recv = torch.empty(W, dtype=torch.int32)
            self.dist.all_to_all_single(recv, self.h_both[:W].clone())
            self.h_both[W:] = recv
        s_list, r_list = [int(x) for x in self.h_both[:W].tolist()], [int(x) for x in self.h_both[W:].tolist()]
        n_send, n_recv = sum(s_list), sum(r_list)              # rows I ask for / rows I serve
        self._reserve(n_send, n_recv)
        wanted, out_rows, in_rows = self.wanted[:n_recv], self.out_rows[:n_recv], self.in_rows[:n_send]
        self._a2a(wanted, self.req_row[:n_send], r_list, s_list)
        ops.serve(wanted, n_recv, out_rows)
        self._a2a(in_rows, out_rows, s_list, r_list)
        ops.scatter(pool, in_rows, self.req_dst, n_send)
        ops.finish(sampler_stream)
        self.batches += 1
        self.last = {"rows_requested": n_send, "rows_served": n_recv, "per_owner": s_list}
        return self.last

    def wait(self):
        """Block the host until the last batch's rows are in place (tests; the timed loop never calls it per batch)."""
        self.ops.wait()

    def close(self):
        self.wait()
        self.ops.close()
