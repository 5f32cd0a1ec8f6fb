"""Random-access ceiling of the memory system: independent random 4-byte probes (torch's index kernel: 8-byte index read + 4-byte
result write per probe, both streamed) into an int32 table of the given size.  python3 profiles/probe_rate.py [MB ...]
Default sizes include the tables of the BASELINE shapes: 19.6 MB = position table / indptr at the products shape (V = 2 449 029 x 8 B),
495 MB = `indices` at the products shape, 444 MB / 889 MB = feat_map / position table at papers100M."""
import sys
import torch, time
dev = torch.device("cuda", 0)
n = 4_000_000
sizes = [float(x) for x in sys.argv[1:]] or [4, 9.8, 16, 19.6, 64, 128, 256, 444, 495, 889, 1024, 4096, 16384, 65536]
for mb in sizes:
    elems = int(mb * 1024 * 1024) // 4
    t = torch.zeros(elems, dtype=torch.int32, device=dev)
    idx = torch.randint(0, elems, (n,), device=dev, dtype=torch.int64)
    for _ in range(3): out = t[idx]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = t[idx]
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"table {mb:8.1f} MB: {n/1e6:.0f} M random 4-byte probes in {us:7.1f} us = {n/us/1e3:6.1f} G probes/s", flush=True)
    del t, idx, out
