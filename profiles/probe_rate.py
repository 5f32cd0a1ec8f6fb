import torch, time
dev = torch.device("cuda", 0)
n = 4_000_000
for mb in (16, 128, 444, 1024, 4096, 16384, 65536):
    elems = mb * 1024 * 1024 // 4
    t = torch.zeros(elems, dtype=torch.int32, device=dev)
    idx = torch.randint(0, elems, (n,), device=dev, dtype=torch.int64)
    for _ in range(3): out = t[idx]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = t[idx]
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"table {mb:6d} MB: {n/1e6:.0f} M random 4-byte probes in {us:7.1f} us = {n/us/1e3:6.1f} G probes/s", flush=True)
    del t, idx, out
