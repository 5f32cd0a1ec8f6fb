import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import legion1_amd.capi as K
L = K.lib()
L.legion_set_error_mode(K.ERR_RETURN)
calls = [
 ("batch_generator_kernel", lambda: L.batch_generator_kernel(None, None, None, None, 8, 0, 0, 0, 0)),
 ("GPU_Random_Sampling", lambda: L.GPU_Random_Sampling(None, None, None, None, 5, 2, 0)),
 ("get_feature_kernel", lambda: L.get_feature_kernel(None, None, None, None, 0, 1, 1)),
 ("get_feature_kernel_all", lambda: L.get_feature_kernel_all(None, None, None, None, 0, 1)),
 ("make_update_plan", lambda: L.make_update_plan(None, None, None, None, 0, 0)),
 ("update_cache", lambda: L.update_cache(None, None, None, None, 0, 0)),
 ("legion_exchange_plan", lambda: L.legion_exchange_plan(None, None, None, None, 0, None, None, None)),
 ("legion_exchange_local", lambda: L.legion_exchange_local(None, None, None, None, 0)),
 ("legion_exchange_serve", lambda: L.legion_exchange_serve(None, None, 0, None, 0, None)),
 ("legion_exchange_scatter", lambda: L.legion_exchange_scatter(None, None, None, None, 0, 4)),
 ("legion_peer_exchange_gather", lambda: L.legion_peer_exchange_gather(None, None, None, None, 0)),
 ("GPUGraphStorage_Build", lambda: L.GPUGraphStorage_Build(None, None)),
 ("GPUNodeStorage_Build", lambda: L.GPUNodeStorage_Build(None, None)),
 ("GPUCache_CandidateSelection", lambda: L.GPUCache_CandidateSelection(None, 0, None, None)),
 ("GPUCache_CostModel", lambda: L.GPUCache_CostModel(None, 0, None, None, None, 1)),
 ("GPUCache_FillUp", lambda: L.GPUCache_FillUp(None, 0, None, None)),
 ("GPUCache_HitSamplingDone", lambda: L.GPUCache_HitSamplingDone(None, 0, None)),
 ("NewIPCEnv(0)", lambda: L.NewIPCEnv(0)),
 ("NewIPCEnv(9)", lambda: L.NewIPCEnv(9)),
 ("Runner_Initialize", lambda: L.Runner_Initialize(None, None)),
 ("Server_SetFanout", lambda: L.Server_SetFanout(None, None, 0)),
 ("GPUMemoryPool_ReleasePeerExchange", lambda: L.GPUMemoryPool_ReleasePeerExchange(None)),
 ("Operator_run", lambda: L.Operator_run(None, None)),
]
for name, fn in calls:
    L.legion_clear_error()
    r = fn()
    print(name, "->", r, "|", (L.legion_last_error() or b"").decode()[:90], flush=True)
print("survived")
