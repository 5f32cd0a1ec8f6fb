#!/usr/bin/env python3
"""Streaming-copy sweep on one MI355X: which float4 copy shape reaches the HBM ceiling (bench.py prints the best one
as roofline.measured_copy_GBps beside the 8 TB/s vendor peak).  python3 profiles/copy_sweep.py > profiles/rXX_copy_sweep.md"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import legion1_amd.capi as K  # noqa: E402

L = K.lib()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print("| bytes | grid | chunks/lane | nt (1=st,2=ld) | block-strided | GB/s (read+write) |\n|---|---|---|---|---|---|")
best = {}
for nbytes in (1 << 30, 4 << 30):
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.fill_(1)
    e0, e1 = L.d_event_create(), L.d_event_create()
    for grid, unroll, nt, contig in itertools.product((2048, 8192, 32768, 131072, 0), (1, 2, 4, 8), (0, 1, 3), (0, 1)):
        L.legion_copy_f4_cfg(None, b.data_ptr(), a.data_ptr(), nbytes, grid, unroll, nt, contig)
        L.d_event_record(e0, None)
        for _ in range(5):
            L.legion_copy_f4_cfg(None, b.data_ptr(), a.data_ptr(), nbytes, grid, unroll, nt, contig)
        L.d_event_record(e1, None)
        L.d_stream_sync(None)
        gbps = 5 * 2 * nbytes / (L.d_event_elapsed_ms(e0, e1) * 1e-3) / 1e9
        best[(nbytes, grid, unroll, nt, contig)] = gbps
        print("| %d GiB | %s | %d | %d | %d | %.0f |" % (nbytes >> 30, grid or "1 iter/lane", unroll, nt, contig, gbps), flush=True)
    t = torch.empty_like(b)
    L.d_event_record(e0, None)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        t.copy_(a)
    torch.cuda.synchronize()
    print("| %d GiB | torch copy_ | | | | %.0f |" % (nbytes >> 30, 5 * 2 * nbytes / (time.perf_counter() - t0) / 1e9), flush=True)
    assert bool((b == 1).all())
    del a, b, t
top = sorted(best.items(), key=lambda kv: -kv[1])[:8]
print("\nbest:", ["%s: %.0f" % (k, v) for k, v in top])
