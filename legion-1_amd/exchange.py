"""Owner-computes exchange variant of the unified-cache feature gather (SURVEY 5 option (b)), one process per GPU.

The reference reads the feature rows cached on the other GPUs of the clique in-kernel over NVLink
(zero_copy_with_aggregated_cache, Kernels.cu:662-702); ``get_feature_kernel`` does the same over xGMI.  This is the
collective formulation for the case that fine-grained peer loads fall short of the links: per batch

  1. plan     (HIP)   rows cached on another clique member are listed per owner; own-shard rows and backing-table
                      rows are gathered at once                                      -> legion_exchange_plan
  2. counts   (RCCL)  all-to-all of the Kg request counts (the only host round trip: the split sizes)
  3. requests (RCCL)  all-to-all of the request lists (4 bytes per row)
  4. serve    (HIP)   every owner gathers the requested rows from ITS shard in local HBM -> legion_exchange_serve
  5. rows     (RCCL)  all-to-all of the rows (4F bytes per row) -- the xGMI traffic, in bulk transfers
  6. scatter  (HIP)   rows to their place in the batch's feature buffer             -> legion_exchange_scatter

Results are bit-identical to the in-kernel variant (same rows, verbatim copies).  With gloo (CPU tests, the one-GPU
rehearsal) the three collectives are staged through host memory.
"""
from __future__ import annotations

import numpy as np


class ExchangeGather:
    def __init__(self, capi, eng, me: int, world: int, F: int, device, max_rows: int):
        import torch
        import torch.distributed as dist
        self.K, self.L, self.eng, self.me, self.world, self.F, self.dev = capi, capi.lib(), eng, me, world, F, device
        self.torch, self.dist = torch, dist
        self.nccl = dist.get_backend() == "nccl"
        self.req_row = torch.empty(max_rows, dtype=torch.int32, device=device)
        self.req_dst = torch.empty(max_rows, dtype=torch.int32, device=device)
        self.counts = torch.zeros(16, dtype=torch.int32, device=device)      # int32[2 * LEGION_MAX_DEVICE]
        self.ev = self.L.d_event_create()
        self.last = {}

    def _a2a(self, out, inp, out_splits, in_splits):
        torch, dist = self.torch, self.dist
        if self.nccl:
            dist.all_to_all_single(out, inp, out_splits, in_splits)
            return out
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), out_splits, in_splits)
        out.copy_(h_out)
        return out

    def run(self, sampler_stream, pool):
        """Gather the features of the batch that was just sampled on `sampler_stream` into the current pipe's buffer.
        Collective: every rank of the clique calls it once per batch.  Returns when the rows are in place."""
        torch, L, F, W = self.torch, self.L, self.F, self.world
        ts = torch.cuda.current_stream()                       # RCCL orders its work behind torch's current stream
        sp = ts.cuda_stream
        L.d_event_record(self.ev, sampler_stream)
        L.d_stream_wait_event(sp, self.ev)
        if L.legion_exchange_plan(sp, self.eng.cache, self.eng.noder, pool, self.me, self.req_row.data_ptr(), self.req_dst.data_ptr(),
                                  self.counts.data_ptr()) != 0:
            self.K.check()
            raise RuntimeError("legion_exchange_plan failed")
        send = self.counts[:W].cpu()                           # synchronises: the split sizes must be known on the host
        recv = torch.empty(W, dtype=torch.int32)
        if self.nccl:
            d_recv = torch.empty(W, dtype=torch.int32, device=self.dev)
            self.dist.all_to_all_single(d_recv, self.counts[:W].contiguous())
            recv = d_recv.cpu()
        else:
            self.dist.all_to_all_single(recv, send)
        s_list, r_list = [int(x) for x in send.tolist()], [int(x) for x in recv.tolist()]
        n_send, n_recv = sum(s_list), sum(r_list)              # rows I ask for / rows I serve
        wanted = torch.empty(n_recv, dtype=torch.int32, device=self.dev)
        self._a2a(wanted, self.req_row[:n_send], r_list, s_list)
        out_rows = torch.empty((n_recv, F), dtype=torch.float32, device=self.dev)
        L.legion_exchange_serve(sp, self.eng.cache, self.me, wanted.data_ptr(), n_recv, out_rows.data_ptr())
        in_rows = torch.empty((n_send, F), dtype=torch.float32, device=self.dev)
        self._a2a(in_rows, out_rows, s_list, r_list)
        L.legion_exchange_scatter(sp, pool, in_rows.data_ptr(), self.req_dst.data_ptr(), n_send, F)
        ts.synchronize()
        self.K.check()
        self.last = {"rows_requested": n_send, "rows_served": n_recv, "per_owner": s_list}
        return self.last

    def close(self):
        self.L.d_event_destroy(self.ev)
