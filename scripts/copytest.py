import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, legion1_amd.capi as K
L = K.lib(); dev = torch.device("cuda", 0)
n = 4 << 30
a = torch.ones(n, dtype=torch.uint8, device=dev); b = torch.empty(n, dtype=torch.uint8, device=dev)
e0, e1 = L.d_event_create(), L.d_event_create()
L.legion_copy_f4(None, b.data_ptr(), a.data_ptr(), n)
L.d_event_record(e0, None)
for _ in range(10): L.legion_copy_f4(None, b.data_ptr(), a.data_ptr(), n)
L.d_event_record(e1, None)
print(os.environ.get("LEGION_COPY_GRID"), round(10*2*n/(L.d_event_elapsed_ms(e0,e1)*1e-3)/1e9,1), "GB/s")
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
b.copy_(a); t0.record()
for _ in range(10): b.copy_(a)
t1.record(); torch.cuda.synchronize()
print("torch copy_", round(10*2*n/(t0.elapsed_time(t1)*1e-3)/1e9,1), "GB/s")
