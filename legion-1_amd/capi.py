"""ctypes binding of the C ABI (include/legion_amd.h -> csrc/liblegion_amd.so).

This is the host-side mirror used by tests and bench.py.  It contains no compute
and no fallback: if the HIP library is missing, importing :func:`lib` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB_PATH = os.path.join(_HERE, "csrc", "liblegion_amd.so")


def lib_path() -> str:
    """The library this process loads: csrc/liblegion_amd.so, or the ABSOLUTE path in $LEGION_LIB -- a variant build of an experiment
    (profiles/ab_kernels.sh) or the host-sanitizer build (make -C legion-1_amd/csrc asan-host).  Experiments point at their variant
    through this and never overwrite the shipped library (round 4 swapped it in place: a killed run left an invalid library behind)."""
    p = os.environ.get("LEGION_LIB")
    if not p:
        return DEFAULT_LIB_PATH
    if not os.path.isabs(p):
        raise RuntimeError(f"LEGION_LIB={p!r}: an absolute path is required (a relative one would depend on the working directory of every child process)")
    return p


LIB_PATH = DEFAULT_LIB_PATH      # kept for callers that print it; lib() resolves lib_path() when it loads
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "legion_amd.h")

TRAINMODE, VALIDMODE, TESTMODE = 0, 1, 2
LOC_HOST_PINNED, LOC_DEVICE, LOC_HOST_PAGEABLE = 0, 1, 2
ERR_EXIT, ERR_RETURN = 0, 1

_lib = None

vp, i32, i64, u32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_double


class LegionBuildInfo(C.Structure):
    _fields_ = [
        ("partition_count", i32),
        ("training_set_num", vp), ("training_set_ids", vp), ("training_labels", vp),
        ("validation_set_num", vp), ("validation_set_ids", vp), ("validation_labels", vp),
        ("testing_set_num", vp), ("testing_set_ids", vp), ("testing_labels", vp),
        ("total_num_nodes", i32), ("float_attr_len", i32),
        ("host_float_attrs", vp), ("features_location", i32),
        ("csr_node_index", vp), ("csr_dst_node_ids", vp), ("csr_location", i32),
        ("total_edge_num", i64), ("cache_edge_num", i64),
        ("epoch", i32), ("raw_batch_size", i32),
        ("float_attr_pitch", i32),
    ]


class LegionSynthSpec(C.Structure):
    _fields_ = [("V", i32), ("F", i32), ("classes", i32), ("n_train", i32), ("n_valid", i32), ("n_test", i32),
                ("M", u32), ("C", u32), ("M2", u32), ("C2", u32), ("ladder", i32 * 26), ("mean_degree", f64)]


class OpParams(C.Structure):
    _fields_ = [("device_id", C.c_int), ("stream", vp), ("event", vp), ("memorypool", vp), ("cache", vp),
                ("graph", vp), ("noder", vp), ("env", vp), ("neighbor_count", C.c_int), ("is_presc", C.c_int),
                ("in_memory", C.c_int)]


class RunnerParams(C.Structure):
    _fields_ = [("device_id", C.c_int), ("fanout", vp), ("hops", i32), ("cache", vp), ("graph", vp), ("noder", vp),
                ("env", vp), ("global_batch_id", C.c_int), ("in_memory", C.c_int)]


# (name, restype, argtypes) for everything that does not return void / take only ints+pointers
_SIGS = {
    "legion_version": (C.c_char_p, []),
    "legion_last_error": (C.c_char_p, []),
    "legion_set_error_mode": (None, [C.c_int]),
    "legion_set_device_map": (None, [i32, i32]),
    "legion_physical_device": (i32, [i32]),
    "legion_row_pitch": (i32, [i32]),
    "legion_shard_pitch": (i32, [i32, i64, i64]),
    "GPUCache_HitSamplingDone": (None, [vp, i32, vp]),
    "legion_set_remote_device": (None, [i32, C.c_int]),
    "legion_is_remote_device": (C.c_int, [i32]),
    "legion_audit_enabled": (C.c_int, []),
    "legion_audit_counts": (None, [vp]),
    "legion_audit_message_count": (i32, []),
    "legion_audit_message": (C.c_char_p, [i32]),
    "legion_audit_reset": (None, []),
    "legion_audit_report": (i64, []),
    "legion_audit_alias": (C.c_int, [i32, i32]),
    "d_alloc_space": (vp, [i64]),
    "d_free_space": (None, [vp]),
    "host_alloc_space64": (vp, [i64]),
    "host_free_space": (None, [vp]),
    "d_copy_h_2_d": (None, [vp, vp, i64]),
    "d_copy_d_2_h": (None, [vp, vp, i64]),
    "d_stream_sync": (None, [vp]),
    "d_stream_create": (vp, []),
    "d_stream_create_cu_mask": (vp, [vp, i32]),
    "d_stream_create_priority": (vp, [i32]),
    "d_stream_destroy": (None, [vp]),
    "d_copy_async": (None, [vp, vp, i64, vp]),
    "d_memset_async": (None, [vp, C.c_int, i64, vp]),
    "d_event_create": (vp, []),
    "d_event_destroy": (None, [vp]),
    "d_event_record": (None, [vp, vp]),
    "d_stream_wait_event": (None, [vp, vp]),
    "d_event_elapsed_ms": (C.c_float, [vp, vp]),
    "SetGPUDevice": (None, [i32]),
    "GetGPUDevice": (i32, []),
    "NewGPUMemoryGraphStorage": (vp, []),
    "GPUGraphStorage_Build": (None, [vp, vp]),
    "GPUGraphStorage_GraphCache": (None, [vp, vp, i32, i32, i32]),
    "GPUGraphStorage_ReplicateToDevices": (i64, [vp]),
    "GPUNodeStorage_ReplicateToDevices": (i64, [vp]),
    "GPUGraphStorage_Finalize": (None, [vp]),
    "GPUGraphStorage_Delete": (None, [vp]),
    "GPUGraphStorage_GetFragmentIndex": (vp, [vp, i32, i32]),
    "GPUGraphStorage_ExportFragment": (C.c_int, [vp, i32, vp, vp, vp]),
    "GPUGraphStorage_ImportFragment": (C.c_int, [vp, i32, i32, vp, vp, i32]),
    "GPUGraphStorage_GetFragmentMatrix": (vp, [vp, i32, i32]),
    "GPUMemoryPool_BeginBatchCapture": (C.c_int, [vp, vp]),
    "GPUMemoryPool_EndBatchCapture": (vp, [vp, vp]),
    "LegionBatchGraph_Launch": (C.c_int, [vp, vp, i32]),
    "LegionBatchGraph_Delete": (None, [vp]),
    "GPUGraphStorage_FragmentRows": (i32, [vp, i32]),
    "GPUGraphStorage_FragmentEdges": (C.c_int64, [vp, i32]),
    "GPUGraphStorage_FragmentChunkCount": (i32, [vp, i32, i32]),
    "GPUGraphStorage_FragmentChunkSpan": (C.c_int64, [vp, i32]),
    "GPUGraphStorage_GetFragmentChunk": (vp, [vp, i32, i32, i32]),
    "GPUGraphStorage_ExportFragmentChunk": (C.c_int, [vp, i32, i32, i32, vp]),
    "GPUGraphStorage_ImportFragmentChunk": (C.c_int, [vp, i32, i32, i32, i32, vp, i32, C.c_int64]),
    "NewGPUMemoryNodeStorage": (vp, []),
    "GPUNodeStorage_Build": (None, [vp, vp]),
    "GPUNodeStorage_Delete": (None, [vp]),
    "GPUNodeStorage_TrainingSetSize": (i32, [vp, i32]),
    "NewGPUMemoryPool": (vp, [i32]),
    "GPUMemoryPool_AllocateScratch": (None, [vp, i32, i32, vp, i32]),
    "GPUMemoryPool_NumIds": (i32, [vp]),
    "GPUMemoryPool_SetSampledIds": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetFloatFeatures": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetLabels": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetAggSrcOf": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetAggDstOf": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetNodeCounter": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetEdgeCounter": (None, [vp, vp, i32]),
    "GPUMemoryPool_SetFeatureRows": (None, [vp, i32]),
    "GPUMemoryPool_SetCurrentPipe": (None, [vp, i32]),
    "GPUMemoryPool_SetCurrentMode": (None, [vp, i32]),
    "GPUMemoryPool_SetIter": (None, [vp, i32]),
    "GPUMemoryPool_GetPositionMap": (vp, [vp]),
    "GPUMemoryPool_GetBatchSerial": (u32, [vp]),
    "GPUMemoryPool_SetBatchSerial": (None, [vp, u32]),
    "GPUMemoryPool_GetAggSrcId": (vp, [vp]),
    "GPUMemoryPool_GetCacheSearchBuffer": (vp, [vp]),
    "GPUMemoryPool_GetTmpPartIdx": (vp, [vp]),
    "GPUMemoryPool_GetTmpPartOff": (vp, [vp]),
    "GPUMemoryPool_Delete": (None, [vp]),
    "NewGPUCache": (vp, []),
    "GPUCache_Initialize": (None, [vp, i64, i32, i32, i32, i32]),
    "GPUCache_InitializeCacheController": (None, [vp, i32, i32]),
    "GPUCache_NodeCapacity": (i32, [vp, i32]),
    "GPUCache_EdgeCapacity": (i32, [vp, i32]),
    "GPUCache_FindFeat": (None, [vp, vp, vp, vp, i32, vp, i32]),
    "GPUCache_FindTopo": (None, [vp, vp, vp, vp, i32, i32, vp, i32]),
    "GPUCache_CandidateSelection": (None, [vp, C.c_int, vp, vp]),
    "GPUCache_CostModel": (None, [vp, C.c_int, vp, vp, vp, i32]),
    "GPUCache_SetCapacity": (None, [vp, i32, i32]),
    "GPUCache_SetPreSc": (None, [vp, C.c_int]),
    "GPUCache_FillUp": (None, [vp, C.c_int, vp, vp]),
    "GPUCache_MaxIdNum": (i32, [vp, i32]),
    "GPUCache_Float_Feature_Cache": (vp, [vp, i32]),
    "GPUCache_ExportFeatureShard": (C.c_int, [vp, i32, vp]),
    "GPUCache_ImportFeatureShard": (C.c_int, [vp, i32, vp]),
    "GPUCache_ShardChunkCount": (i32, [vp, i32]),
    "GPUCache_ShardChunkRows": (i32, [vp, i32]),
    "GPUCache_ShardPitch": (i32, [vp]),
    "GPUCache_GetShardChunk": (vp, [vp, i32, i32]),
    "GPUCache_ShardGeometry": (C.c_int, [vp, i32, vp]),
    "GPUCache_CheckShardGeometry": (C.c_int, [vp, i32, vp]),
    "GPUCache_ExportFeatureShardChunk": (C.c_int, [vp, i32, i32, vp]),
    "GPUCache_ImportFeatureShardChunk": (C.c_int, [vp, i32, i32, vp]),
    "GPUCache_HitSampling": (vp, [vp, i32, C.c_int, C.c_int]),
    "GPUCache_FeatureCacheHitRate": (f64, [vp, i32, vp, vp]),
    "GPUCache_GetNodeAccessedMap": (vp, [vp, i32]),
    "GPUCache_GetEdgeAccessedMap": (vp, [vp, i32]),
    "GPUCache_GetFeatureMap": (vp, [vp, i32]),
    "GPUCache_GetQF": (vp, [vp, i32]),
    "GPUCache_GetQT": (vp, [vp, i32]),
    "GPUCache_Kg": (i32, [vp]),
    "GPUCache_Kc": (i32, [vp]),
    "GPUCache_Alpha": (f64, [vp, i32]),
    "GPUCache_Delete": (None, [vp]),
    "batch_generator_kernel": (None, [vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    "GPU_Random_Sampling": (None, [vp, vp, vp, vp, i32, i32, C.c_int]),
    "get_feature_kernel": (None, [vp, vp, vp, vp, i32, i32, C.c_int]),
    "get_feature_kernel_all": (None, [vp, vp, vp, vp, i32, C.c_int]),
    "legion_exchange_plan": (C.c_int, [vp, vp, vp, vp, i32, vp, vp, vp]),
    "legion_exchange_local": (C.c_int, [vp, vp, vp, vp, i32]),
    "legion_exchange_serve": (None, [vp, vp, i32, vp, i32, vp]),
    "legion_exchange_scatter": (None, [vp, vp, vp, vp, i32, i32]),
    "legion_peer_exchange_gather": (C.c_int, [vp, vp, vp, vp, i32]),
    "legion_peer_exchange_stats": (None, [vp, vp]),
    "GPUMemoryPool_ReleasePeerExchange": (None, [vp]),
    "make_update_plan": (None, [vp, vp, vp, vp, i32, i32]),
    "update_cache": (None, [vp, vp, vp, vp, i32, i32]),
    "NewBatchGenerator": (vp, [C.c_int]), "NewRandomSampler": (vp, [C.c_int]), "NewFeatureExtractor": (vp, [C.c_int]),
    "NewCachePlanner": (vp, [C.c_int]), "NewCacheUpdater": (vp, [C.c_int]),
    "Operator_run": (None, [vp, vp]), "Operator_Delete": (None, [vp]),
    "NewIPCEnv": (vp, [i32]),
    "IPCEnv_MirrorCounters": (None, [vp, i32, i32, vp]),
    "IPCEnv_SetMirror": (None, [vp, i32, i32, i32, i32]),
    "IPCEnv_SlabPinned": (C.c_int, [vp]),
    "IPCEnv_Coordinate": (None, [vp, vp]),
    "IPCEnv_GetMaxStep": (i32, [vp]),
    "IPCEnv_InitializeSamplesBuffer": (None, [vp, i32, i32, i32, i32, i32]),
    "IPCEnv_InitializeFeaturesBuffer": (None, [vp, i32, i32, i32, i32, i32]),
    "IPCEnv_GetRawBatchsize": (i32, [vp]),
    "IPCEnv_GetLocalBatchId": (i32, [vp, i32]),
    "IPCEnv_GetCurrentBatchsize": (i32, [vp, i32, i32]),
    "IPCEnv_GetCurrentMode": (i32, [vp, i32]),
    "IPCEnv_GetIds": (vp, [vp, i32, i32]), "IPCEnv_GetFloatFeatures": (vp, [vp, i32, i32]),
    "IPCEnv_GetLabels": (vp, [vp, i32, i32]), "IPCEnv_GetAggSrc": (vp, [vp, i32, i32]),
    "IPCEnv_GetAggDst": (vp, [vp, i32, i32]), "IPCEnv_GetNodeCounter": (vp, [vp, i32, i32]),
    "IPCEnv_GetEdgeCounter": (vp, [vp, i32, i32]),
    "IPCEnv_IPCPost": (None, [vp, i32, i32]), "IPCEnv_IPCWait": (None, [vp, i32, i32]),
    "IPCEnv_IPCTryWait": (C.c_int, [vp, i32, i32, i32]),
    "IPCEnv_HandoffSpinUs": (C.c_int, []),
    "IPCEnv_Finalize": (None, [vp]), "IPCEnv_GetTrainStep": (i32, [vp]), "IPCEnv_SetHops": (None, [vp, i32]),
    "legion_ipc_set_namespace": (None, [C.c_char_p]),
    "legion_ipc_unlink_namespace": (None, [C.c_char_p, i32]),
    "IPCEnv_MirroredNodeCounter": (i32, [vp, i32, i32, i32]),
    "IPCEnv_SetFeatureRows": (None, [vp, i32, i32]),
    "legion_ipc_client_open": (vp, [i32]), "legion_ipc_client_wait": (None, [vp]),
    "legion_ipc_client_post": (None, [vp]), "legion_ipc_client_post_nosync": (None, [vp]), "legion_ipc_client_buffer": (vp, [vp, i32]),
    "legion_ipc_client_steps": (None, [vp, vp]), "legion_ipc_client_hops": (i32, [vp]), "legion_ipc_client_feature_rows": (i32, [vp]),
    "legion_ipc_client_read_counters": (None, [vp, vp, vp]), "legion_ipc_client_close": (None, [vp]),
    "legion_runner_gather_estimate": (C.c_int, [i32, f64, f64, vp, vp]),
    "NewGPURunner": (vp, []), "Runner_Initialize": (None, [vp, vp]),
    "Runner_InitializeFeaturesBuffer": (None, [vp, vp]), "Runner_RunPreSc": (None, [vp, vp]),
    "Runner_RunOnce": (None, [vp, vp]), "Runner_Finalize": (None, [vp, vp]), "Runner_GetMemoryPool": (vp, [vp]), "Runner_ShortBatches": (i64, [vp]),
    "Runner_Delete": (None, [vp]),
    "NewGPUServer": (vp, []), "Server_SetFanout": (None, [vp, vp, i32]),
    "Server_SetMetaConfigPath": (None, [vp, C.c_char_p]), "Server_Initialize": (None, [vp, C.c_int]),
    "Server_PreSc": (None, [vp, C.c_int]), "Server_Run": (None, [vp]), "Server_Finalize": (None, [vp]),
    "Server_Delete": (None, [vp]),
    "legion_synth_degrees": (None, [vp, vp, i32, i32, vp]),
    "legion_synth_neighbors": (None, [vp, vp, i64, i64, i32, u32, u32]),
    "legion_synth_neighbors_skew": (None, [vp, vp, i64, i64, i32, u32, u32, i32]),
    "legion_synth_lp_seeds": (None, [vp, vp, vp, vp, i64, i32, vp, vp, i32, u32]),
    "legion_synth_features": (None, [vp, vp, i64, i64, i32]),
    "legion_synth_features_pitched": (None, [vp, vp, i64, i64, i32, i32]),
    "legion_synth_labels": (None, [vp, vp, i32, i32, i32]),
    "legion_synth_seed_ids": (None, [vp, vp, i64, i64, i32, u32, u32, i32, i32]),
    "legion_synth_spec": (C.c_int, [C.c_char_p, f64, vp]),
    "legion_synth_label_host": (i32, [i32, i32]),
    "legion_synth_seed_id_host": (i32, [i64, i32, u32, u32]),
    "legion_copy_f4": (None, [vp, vp, vp, i64]),
    "legion_sum_words": (None, [vp, vp, i64, vp]),
    "GPUMemoryPool_GetCandidateBuffer": (vp, [vp]),
    "legion_copy_f4_cfg": (C.c_int, [vp, vp, vp, i64, i32, i32, i32, i32]),
    "legion_rng_probe": (None, [vp, vp, vp, vp, i32]),
}


def lib():
    """The loaded HIP library.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C legion-1_amd/csrc` "
                               "(or __graft_entry__.build()); there is no CPU fallback")
        _lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check():
    """Raise if the library recorded a (sticky) error."""
    msg = lib().legion_last_error()
    if msg:
        text = msg.decode()
        lib().legion_clear_error()
        raise RuntimeError(text)


# ------------------------------------------------------------------------------------------------------
# small device-memory helpers (raw hipMalloc through the C ABI; no torch needed)
# ------------------------------------------------------------------------------------------------------
class DevBuf:
    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = lib().d_alloc_space(max(self.nbytes, 16))
        check()

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        if a.nbytes:
            lib().d_copy_h_2_d(b.ptr, a.ctypes.data, a.nbytes)
        b.dtype, b.shape = a.dtype, a.shape
        return b

    def to_numpy(self, dtype, count: int, offset_bytes: int = 0) -> np.ndarray:
        out = np.empty(int(count), dtype=dtype)
        if out.nbytes:
            lib().d_copy_d_2_h(out.ctypes.data, self.ptr + offset_bytes, out.nbytes)
        return out

    def free(self):
        if self.ptr:
            lib().d_free_space(self.ptr)
            self.ptr = None


def read_dev(ptr: int, dtype, count: int) -> np.ndarray:
    out = np.empty(int(count), dtype=dtype)
    if out.nbytes:
        lib().d_copy_d_2_h(out.ctypes.data, ptr, out.nbytes)
    return out


def _ptr_array(ptrs):
    arr = (vp * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr


class Engine:
    """Host-side assembly of the reference objects for G logical GPUs (one process), the way
    GPUGraphStore::Initialze + GPURunner::Initialize do it (GPUGraphStore.cu:429-470,
    Server.cu:169-271), but driven from arrays instead of files.  Output buffers are raw device
    allocations (or externally supplied pointers) -- no IPC involved.

    ``indptr`` / ``indices`` / ``features`` may be numpy arrays (copied to the chosen location)
    or integer device pointers (``*_location`` = LOC_DEVICE).
    """

    def __init__(self, indptr, indices, features, V, F, seeds, batch_size, fanout, G=1,
                 csr_location=LOC_DEVICE, features_location=LOC_DEVICE, cache_memory=0, train_step=1, epoch=1,
                 pipeline_depth=1, E=None, local_devs=None, features_pitch=0):
        L = lib()
        # one process per GPU: the other members of the clique are remote (legion_set_remote_device)
        self.local_devs = list(range(int(G))) if local_devs is None else list(local_devs)
        for g in range(int(G)):
            L.legion_set_remote_device(g, 0 if g in self.local_devs else 1)
        self.L, self.G, self.V, self.F = L, int(G), int(V), int(F)
        L.SetGPUDevice(self.local_devs[0])      # the shared tables below live on the first local GPU
        self.fanout = np.asarray(fanout, dtype=np.int32)
        self.hops = len(self.fanout)
        self.batch_size = int(batch_size)
        self._keep = []
        self._owned = []

        def place(arr, location, dtype):
            if isinstance(arr, (int, np.integer)):
                return int(arr)
            arr = np.ascontiguousarray(arr, dtype=dtype)
            if location == LOC_DEVICE:
                b = DevBuf.from_numpy(arr)
                self._owned.append(b)
                return b.ptr
            p = L.host_alloc_space64(max(arr.nbytes, 16))
            C.memmove(p, arr.ctypes.data, arr.nbytes)
            self._owned.append(("host", p))
            return p

        self.E = int(E if E is not None else (indptr[-1] if not isinstance(indptr, (int, np.integer)) else 0))
        self.indptr_ptr = place(indptr, csr_location, np.int64)
        self.indices_ptr = place(indices, csr_location, np.int32)
        self.features_ptr = place(features, features_location, np.float32) if features is not None else None

        # seeds: dict(train=[(ids, labels)] * G, valid=..., test=...) with numpy arrays or (ptr, n) pairs
        info = LegionBuildInfo()
        info.partition_count = self.G
        for key, pre in (("train", "training"), ("valid", "validation"), ("test", "testing")):
            sets = seeds.get(key)
            if sets is None:
                sets = [(np.zeros(0, np.int32), np.zeros(0, np.int32))] * self.G
            nums = np.zeros(self.G, dtype=np.int32)
            idp, lbp = [], []
            for g, (ids, labs) in enumerate(sets):
                if isinstance(ids, tuple):          # (device_ptr, n)
                    nums[g] = ids[1]
                    idp.append(ids[0])
                    lbp.append(labs[0])
                else:
                    ids = np.ascontiguousarray(ids, dtype=np.int32)
                    labs = np.ascontiguousarray(labs, dtype=np.int32)
                    self._keep += [ids, labs]
                    nums[g] = len(ids)
                    idp.append(ids.ctypes.data)
                    lbp.append(labs.ctypes.data)
            ida, lba = _ptr_array(idp), _ptr_array(lbp)
            self._keep += [nums, ida, lba]
            setattr(info, pre + "_set_num", nums.ctypes.data)
            setattr(info, pre + "_set_ids", C.cast(ida, vp))
            setattr(info, pre + "_labels", C.cast(lba, vp))
            setattr(self, key + "_num", nums)
        info.total_num_nodes, info.float_attr_len = self.V, self.F
        info.host_float_attrs, info.features_location = self.features_ptr, features_location
        info.float_attr_pitch = int(features_pitch)        # 0 = dense rows; > F: the table was built with padded rows
        info.csr_node_index, info.csr_dst_node_ids, info.csr_location = self.indptr_ptr, self.indices_ptr, csr_location
        info.total_edge_num, info.cache_edge_num = self.E, 0
        info.epoch, info.raw_batch_size = epoch, self.batch_size
        self.info = info

        self.graph = L.NewGPUMemoryGraphStorage()
        L.GPUGraphStorage_Build(self.graph, C.byref(info))
        self.noder = L.NewGPUMemoryNodeStorage()
        L.GPUNodeStorage_Build(self.noder, C.byref(info))
        self.cache = L.NewGPUCache()
        L.GPUCache_Initialize(self.cache, int(cache_memory), 0, self.F, int(train_step), self.G)
        check()
        self.depth = int(pipeline_depth)
        self.pools, self.out = [None] * self.G, [None] * self.G
        for g in self.local_devs:
            L.SetGPUDevice(g)
            L.GPUCache_InitializeCacheController(self.cache, g, self.V)
            pool = L.NewGPUMemoryPool(self.depth)
            L.GPUMemoryPool_AllocateScratch(pool, self.V, self.batch_size, self.fanout.ctypes.data, self.hops)
            check()
            n = L.GPUMemoryPool_NumIds(pool)
            self.num_ids = n
            pipes = []
            for q in range(self.depth):
                o = dict(ids=DevBuf(n * 4), labels=DevBuf(self.batch_size * 4), src=DevBuf(n * 4), dst=DevBuf(n * 4),
                         nc=DevBuf(64), ec=DevBuf(64), feat=None)
                L.GPUMemoryPool_SetSampledIds(pool, o["ids"].ptr, q)
                L.GPUMemoryPool_SetLabels(pool, o["labels"].ptr, q)
                L.GPUMemoryPool_SetAggSrcOf(pool, o["src"].ptr, q)
                L.GPUMemoryPool_SetAggDstOf(pool, o["dst"].ptr, q)
                L.GPUMemoryPool_SetNodeCounter(pool, o["nc"].ptr, q)
                L.GPUMemoryPool_SetEdgeCounter(pool, o["ec"].ptr, q)
                pipes.append(o)
            self.pools[g] = pool
            self.out[g] = pipes
        self.streams = [None] * self.G
        self._graphs = []
        check()

    # ---- feature buffers ------------------------------------------------------------------------------
    def alloc_features(self, rows=None):
        rows = self.num_ids if rows is None else int(rows)
        for g in self.local_devs:
            self.L.SetGPUDevice(g)
            for q in range(self.depth):
                b = DevBuf(rows * self.F * 4)
                self.out[g][q]["feat"] = b
                self.L.GPUMemoryPool_SetFloatFeatures(self.pools[g], b.ptr, q)
            self.L.GPUMemoryPool_SetFeatureRows(self.pools[g], rows)
        self.feature_rows = rows
        check()

    # ---- one batch through the reference's launcher API -----------------------------------------------------
    def run_batch(self, dev=0, counter=0, mode=TRAINMODE, is_presc=False, gather=True, plan=True, pipe=0,
                  batch_size=None, per_level=True, stream=None, sync=True):
        L = self.L
        L.SetGPUDevice(dev)
        pool = self.pools[dev]
        L.GPUMemoryPool_SetCurrentPipe(pool, pipe)
        L.GPUMemoryPool_SetCurrentMode(pool, mode)
        L.GPUMemoryPool_SetIter(pool, counter)
        bs = self.batch_size if batch_size is None else batch_size
        L.batch_generator_kernel(stream, self.noder, self.cache, pool, bs, counter, dev, dev, mode)
        if gather and not is_presc and per_level:
            L.get_feature_kernel(stream, self.cache, self.noder, pool, dev, 1, 1)
        for h in range(self.hops):
            L.GPU_Random_Sampling(stream, self.graph, self.cache, pool, int(self.fanout[h]), 2 * h + 2, int(is_presc))
            if gather and not is_presc and per_level:
                L.get_feature_kernel(stream, self.cache, self.noder, pool, dev, 2 * h + 3, 1)
        if gather and not is_presc and not per_level:
            L.get_feature_kernel_all(stream, self.cache, self.noder, pool, dev, 1)
        if plan:
            L.make_update_plan(stream, self.graph, self.cache, pool, dev, mode)
            L.update_cache(stream, self.cache, self.noder, pool, dev, mode)
        if sync:
            L.d_stream_sync(stream)
            check()

    # ---- the same batch recorded once as a hipGraph (one launch per batch) ---------------------------------------
    def capture_batch(self, dev=0, mode=TRAINMODE, gather=True, plan=True, pipe=0, batch_size=None, per_level=True,
                      stream=None):
        """Record run_batch(dev, <any counter>, mode, ...) on `stream`; returns the graph handle for run_graph()."""
        L = self.L
        L.SetGPUDevice(dev)
        if stream is None:
            if self.streams[dev] is None:
                self.streams[dev] = L.d_stream_create()
            stream = self.streams[dev]
        if L.GPUMemoryPool_BeginBatchCapture(self.pools[dev], stream) != 0:
            check()
            raise RuntimeError("BeginBatchCapture failed")
        self.run_batch(dev, 0, mode=mode, gather=gather, plan=plan, pipe=pipe, batch_size=batch_size, per_level=per_level,
                       stream=stream, sync=False)
        g = L.GPUMemoryPool_EndBatchCapture(self.pools[dev], stream)
        check()
        if not g:
            raise RuntimeError("EndBatchCapture failed")
        self._graphs.append(g)
        return (g, stream, dev)

    def run_graph(self, handle, counter, sync=True):
        g, stream, dev = handle
        self.L.SetGPUDevice(dev)
        if self.L.LegionBatchGraph_Launch(g, stream, int(counter)) != 0:
            check()
            raise RuntimeError("LegionBatchGraph_Launch failed")
        if sync:
            self.L.d_stream_sync(stream)
            check()

    def result(self, dev=0, pipe=0, with_features=True):
        o = self.out[dev][pipe]
        self.L.SetGPUDevice(dev)
        nc = o["nc"].to_numpy(np.int32, 16)
        ec = o["ec"].to_numpy(np.int32, 16)
        H = self.hops
        n_nodes, n_edges = int(nc[5 + 2 * H]), int(ec[2 + H])
        res = dict(nc=nc, ec=ec, ids=o["ids"].to_numpy(np.int32, n_nodes), labels=o["labels"].to_numpy(np.int32, int(nc[4])),
                   src_off=o["src"].to_numpy(np.int32, n_edges), dst_off=o["dst"].to_numpy(np.int32, n_edges))
        if with_features and o["feat"] is not None:
            rows = min(n_nodes, getattr(self, "feature_rows", n_nodes))
            res["features"] = o["feat"].to_numpy(np.float32, rows * self.F).reshape(rows, self.F)
        return res

    # ---- cache pipeline (Server::PreSc, Server.cu:83-114) ----------------------------------------------------------
    def build_cache(self, cache_agg_mode=0, counters=None, node_capacity=None, edge_capacity=None, train_step=1):
        L = self.L
        L.GPUCache_CandidateSelection(self.cache, cache_agg_mode, self.noder, self.graph)
        if node_capacity is not None:
            L.GPUCache_SetCapacity(self.cache, int(node_capacity), int(edge_capacity))
        cp = None
        if counters is not None:
            counters = np.ascontiguousarray(counters, dtype=np.uint64)
            cp = counters.ctypes.data
        L.GPUCache_CostModel(self.cache, cache_agg_mode, self.noder, self.graph, cp, int(train_step))
        L.GPUCache_FillUp(self.cache, cache_agg_mode, self.noder, self.graph)
        check()

    # ---- one process per GPU: exchange the clique's cache shards / CSR fragments over HIP IPC -----------------
    def export_shards(self, dev):
        """(feature_chunk_handles, indptr_chunk_handles, indices_chunk_handles, (fragment_rows, fragment_edges), shard geometry) of a
        LOCAL clique member; every handle is a 64-byte HIP IPC handle of one chunk allocation; the geometry -- (pitch, rows per
        chunk, chunks, rows) the shard was built with -- travels with the handles and is checked by the importer."""
        L = self.L
        fh = []
        geom = None
        if L.GPUCache_Float_Feature_Cache(self.cache, dev):
            g4 = (i32 * 4)()
            if L.GPUCache_ShardGeometry(self.cache, dev, g4) == 0:
                geom = tuple(g4)
            for q in range(L.GPUCache_ShardChunkCount(self.cache, dev)):
                b = C.create_string_buffer(64)
                if L.GPUCache_ExportFeatureShardChunk(self.cache, dev, q, b) != 0:
                    break
                fh.append(b.raw)
        th = [[], []]
        rows, edges = L.GPUGraphStorage_FragmentRows(self.graph, dev), L.GPUGraphStorage_FragmentEdges(self.graph, dev)
        if rows > 0:
            for which in (0, 1):
                for q in range(L.GPUGraphStorage_FragmentChunkCount(self.graph, dev, which)):
                    b = C.create_string_buffer(64)
                    if L.GPUGraphStorage_ExportFragmentChunk(self.graph, dev, which, q, b) != 0:
                        break
                    th[which].append(b.raw)
        check()
        have_t = rows > 0 and th[0] and th[1]
        return (fh if fh else None, th[0] if have_t else None, th[1] if have_t else None, (rows, edges), geom)

    def import_shards(self, owner_dev, handles, viewer_devs=None):
        """Open a REMOTE member's shards so that the local members read them in-kernel (xGMI peer loads)."""
        L = self.L
        fh, ih, xh, (rows, edges) = handles[:4]
        geom = handles[4] if len(handles) > 4 else None
        if fh is not None:
            if geom is not None and L.GPUCache_CheckShardGeometry(self.cache, owner_dev, (i32 * 4)(*geom)) != 0:
                check()      # raises: the exporter's pitch / chunk geometry is not what this process derived
            for q, h in enumerate(fh):
                L.GPUCache_ImportFeatureShardChunk(self.cache, owner_dev, q, h)
        if ih is not None:
            for v in (self.local_devs if viewer_devs is None else viewer_devs):
                for which, hs in ((0, ih), (1, xh)):
                    for q, h in enumerate(hs):
                        L.GPUGraphStorage_ImportFragmentChunk(self.graph, owner_dev, v, which, q, h, rows, edges)
        check()

    def close(self):
        L = self.L
        for g in self._graphs:
            L.LegionBatchGraph_Delete(g)
        self._graphs = []
        for d, st in enumerate(self.streams):
            if st is not None:
                L.SetGPUDevice(d)
                L.d_stream_destroy(st)
        self.streams = [None] * self.G
        for g, pool in enumerate(self.pools):
            if pool is None:
                continue
            L.SetGPUDevice(g)
            L.GPUMemoryPool_Delete(pool)
            for pipes in self.out[g]:
                for b in pipes.values():
                    if b is not None:
                        b.free()
        L.GPUCache_Delete(self.cache)
        L.GPUGraphStorage_Delete(self.graph)
        L.GPUNodeStorage_Delete(self.noder)
        for b in self._owned:
            if isinstance(b, tuple):
                L.host_free_space(b[1])
            else:
                b.free()
        self.pools, self.out, self._owned = [], [], []
