"""ctypes front end of the CPU oracle (oracle/legion_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblegion_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, n) for n in ("legion_oracle.c", "dgl_cpu_sampler.c", "synth_gen.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liblegion_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.lo_minstd_value.restype = C.c_uint32
        _lib.lo_minstd_value.argtypes = [C.c_int32]
        _lib.lo_minstd_nth.restype = C.c_uint32
        _lib.lo_minstd_nth.argtypes = [C.c_uint64]
        _lib.lo_sample_index.restype = C.c_int32
        _lib.lo_sample_index.argtypes = [C.c_int32, C.c_int32]
        _lib.lo_get_max_step.restype = C.c_int32
        _lib.lo_get_current_mode.restype = C.c_int32
        _lib.lo_get_local_batch_id.restype = C.c_int32
    return _lib


def _p(a, ty=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def sample_index(idx: int, deg: int) -> int:
    return int(lib().lo_sample_index(int(idx), int(deg)))


def minstd_value(idx: int) -> int:
    return int(lib().lo_minstd_value(int(idx)))


def minstd_nth(n: int) -> int:
    return int(lib().lo_minstd_nth(int(n)))


class _BatchCtx(C.Structure):
    _fields_ = [
        ("V", C.c_int32), ("F", C.c_int32),
        ("indptr", C.c_void_p), ("indices", C.c_void_p), ("features", C.c_void_p),
        ("node_map", C.c_void_p), ("cache_ptrs", C.c_void_p), ("cache_capacity", C.c_int32),
        ("part_index_map", C.c_void_p), ("part_offset_map", C.c_void_p),
        ("frag_indptr", C.c_void_p), ("frag_indices", C.c_void_p), ("partition_count", C.c_int32),
        ("accessed_map", C.c_void_p), ("position_map", C.c_void_p),
        ("agg_src_ids", C.c_void_p), ("agg_dst_ids", C.c_void_p),
        ("tmp_part_ind", C.c_void_p), ("tmp_part_off", C.c_void_p), ("cache_index", C.c_void_p),
        ("sampled_ids", C.c_void_p), ("float_features", C.c_void_p), ("labels", C.c_void_p),
        ("agg_src_off", C.c_void_p), ("agg_dst_off", C.c_void_p), ("nc", C.c_void_p), ("ec", C.c_void_p),
        ("node_access_time", C.c_void_p), ("edge_access_time", C.c_void_p),
    ]


def max_ids(batch_size: int, fanout) -> int:
    """Server.cu:184-196: B * (1 + f0 + f0*f1 + ...)."""
    tot, cur = batch_size, batch_size
    for f in fanout:
        cur *= f
        tot += cur
    return tot


class OracleRunner:
    """Holds the scratch/output buffers of one GPU's runner and runs batches
    through the canonical schedule (GPURunner::RunOnce order, Server.cu:301-328)."""

    def __init__(self, indptr, indices, features, V, F, batch_size, fanout, partition_count=1,
                 with_features=True):
        self.V, self.F = int(V), int(F)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.features = None if features is None else np.ascontiguousarray(features, dtype=np.float32)
        self.fanout = np.asarray(fanout, dtype=np.int32)
        self.hops = len(fanout)
        self.batch_size = int(batch_size)
        self.P = int(partition_count)
        n = max_ids(batch_size, fanout)
        self.num_ids = n
        self.accessed_map = np.zeros(self.V // 32 + 1, dtype=np.uint32)
        self.position_map = np.zeros(self.V, dtype=np.int32)
        self.agg_src_ids = np.zeros(n, dtype=np.int32)
        self.agg_dst_ids = np.zeros(n, dtype=np.int32)
        self.tmp_part_ind = np.zeros(n, dtype=np.int8)
        self.tmp_part_off = np.zeros(n, dtype=np.int32)
        self.cache_index = np.zeros(n, dtype=np.int32)
        self.sampled_ids = np.zeros(n, dtype=np.int32)
        self.labels = np.zeros(batch_size, dtype=np.int32)
        self.agg_src_off = np.zeros(n, dtype=np.int32)
        self.agg_dst_off = np.zeros(n, dtype=np.int32)
        self.nc = np.zeros(16, dtype=np.int32)
        self.ec = np.zeros(16, dtype=np.int32)
        self.with_features = with_features and self.features is not None
        self.float_features = np.zeros((n if self.with_features else 1, self.F), dtype=np.float32)
        self.node_access_time = np.zeros(self.V, dtype=np.uint64)
        self.edge_access_time = np.zeros(self.V, dtype=np.uint64)
        # cache state (optional)
        self.node_map = None
        self.caches = None
        self.cache_capacity = 0
        self.part_index_map = None
        self.part_offset_map = None
        self.frag_indptr = None
        self.frag_indices = None
        self._keep = []

    # ---- cache construction (S8/S9) -------------------------------------
    def set_feature_cache(self, QF, capacity, Kg, caches=None):
        """Unified feature cache of one clique: node_map + per-GPU row caches."""
        L = lib()
        QF = np.ascontiguousarray(QF, dtype=np.int32)
        self.node_map = np.empty(self.V, dtype=np.int32)
        L.lo_build_feat_map(_p(self.node_map), C.c_int32(self.V), _p(QF), C.c_int32(capacity), C.c_int32(Kg))
        if caches is None:
            caches = []
            for j in range(Kg):
                buf = np.zeros((capacity, self.F), dtype=np.float32)
                L.lo_feat_fill_up(C.c_int32(capacity), C.c_int32(self.F), _p(buf), _p(self.features), _p(QF),
                                  C.c_int32(Kg), C.c_int32(j), C.c_int32(self.V))
                caches.append(buf)
        self.caches = caches
        self.cache_capacity = int(capacity)

    def set_topo_cache(self, QT, capacity, Kg, Ki=0):
        L = lib()
        QT = np.ascontiguousarray(QT, dtype=np.int32)
        self.part_index_map = np.empty(self.V, dtype=np.int8)
        self.part_offset_map = np.empty(self.V, dtype=np.int32)
        L.lo_build_topo_map(_p(self.part_index_map), _p(self.part_offset_map), C.c_int32(self.V), _p(QT),
                            C.c_int32(capacity), C.c_int32(Kg), C.c_int32(Ki))
        self.frag_indptr, self.frag_indices = [None] * self.P, [None] * self.P
        for j in range(Kg):
            fi = np.zeros(capacity + 1, dtype=np.int64)
            L.lo_graph_cache(_p(QT), C.c_int32(Kg), C.c_int32(j), C.c_int32(capacity), C.c_int32(self.V),
                             _p(self.indptr), _p(self.indices), _p(fi), None)
            fx = np.zeros(max(1, int(fi[-1])), dtype=np.int32)
            L.lo_graph_cache(_p(QT), C.c_int32(Kg), C.c_int32(j), C.c_int32(capacity), C.c_int32(self.V),
                             _p(self.indptr), _p(self.indices), _p(fi), _p(fx))
            self.frag_indptr[Ki * Kg + j] = fi
            self.frag_indices[Ki * Kg + j] = fx

    # ---- one batch ---------------------------------------------------------
    def _ptr_table(self, arrs):
        if arrs is None:
            return None
        t = (C.c_void_p * len(arrs))()
        for i, a in enumerate(arrs):
            t[i] = None if a is None else a.ctypes.data
        self._keep.append(t)
        return C.cast(t, C.c_void_p)

    def run_batch(self, all_ids, all_labels, counter, mode=0, is_presc=False, gather=True, batch_size=None, omp=False):
        """omp=True: the OpenMP run of the same canonical schedule (lo_run_batch_omp: draws, COO offsets and row copies in
        parallel, the order-defining compaction serial) -- resident form only; byte-identical output."""
        L = lib()
        if omp and (is_presc or mode != 0 or self.node_map is not None or self.part_index_map is not None):
            raise ValueError("the OpenMP oracle covers the resident train-mode batch only")
        if batch_size is not None and batch_size > self.batch_size:
            raise ValueError("batch larger than the runner was sized for (the reference sizes its buffers for raw_batch_size; "
                             "valid/test batches are up to 512, CUDA_IPC_Service.cu:101-117)")
        all_ids = np.ascontiguousarray(all_ids, dtype=np.int32)
        all_labels = np.ascontiguousarray(all_labels, dtype=np.int32)
        self._keep = []
        ctx = _BatchCtx()
        ctx.V, ctx.F = self.V, self.F
        ctx.indptr, ctx.indices = self.indptr.ctypes.data, self.indices.ctypes.data
        ctx.features = None if self.features is None else self.features.ctypes.data
        ctx.node_map = None if self.node_map is None else self.node_map.ctypes.data
        ctx.cache_ptrs = self._ptr_table(self.caches)
        ctx.cache_capacity = self.cache_capacity
        ctx.part_index_map = None if self.part_index_map is None else self.part_index_map.ctypes.data
        ctx.part_offset_map = None if self.part_offset_map is None else self.part_offset_map.ctypes.data
        ctx.frag_indptr = self._ptr_table(self.frag_indptr)
        ctx.frag_indices = self._ptr_table(self.frag_indices)
        ctx.partition_count = self.P
        for name in ("accessed_map", "position_map", "agg_src_ids", "agg_dst_ids", "tmp_part_ind", "tmp_part_off",
                     "cache_index", "sampled_ids", "float_features", "labels", "agg_src_off", "agg_dst_off", "nc",
                     "ec", "node_access_time", "edge_access_time"):
            setattr(ctx, name, getattr(self, name).ctypes.data)
        if omp:
            if getattr(self, "cand", None) is None:
                self.cand = np.zeros(self.num_ids, dtype=np.int32)
            L.lo_run_batch_omp(C.byref(ctx), _p(all_ids), _p(all_labels), C.c_int32(len(all_ids)),
                               C.c_int32(self.batch_size if batch_size is None else batch_size), C.c_int32(counter),
                               _p(self.fanout), C.c_int32(self.hops), C.c_int32(1 if (gather and self.with_features) else 0),
                               _p(self.cand))
            return self.result()
        L.lo_run_batch(C.byref(ctx), _p(all_ids), _p(all_labels), C.c_int32(len(all_ids)),
                       C.c_int32(self.batch_size if batch_size is None else batch_size), C.c_int32(counter),
                       _p(self.fanout), C.c_int32(self.hops), C.c_int32(mode), C.c_int32(1 if is_presc else 0),
                       C.c_int32(1 if (gather and self.with_features) else 0))
        return self.result()

    def result(self):
        H = self.hops
        n_nodes = int(self.nc[5 + 2 * H])
        n_edges = int(self.ec[2 + H])
        out = dict(nc=self.nc.copy(), ec=self.ec.copy(), ids=self.sampled_ids[:n_nodes].copy(),
                   labels=self.labels[:int(self.nc[4])].copy(), src_off=self.agg_src_off[:n_edges].copy(),
                   dst_off=self.agg_dst_off[:n_edges].copy())
        if self.with_features:
            out["features"] = self.float_features[:n_nodes].copy()
        return out


# ---- thin wrappers over the stand-alone pieces ---------------------------------
def candidate_selection(access_list, V):
    L = lib()
    arrs = [np.ascontiguousarray(a, dtype=np.uint64) for a in access_list]
    t = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    agg = np.empty(V, dtype=np.uint64)
    order = np.empty(V, dtype=np.int32)
    L.lo_candidate_selection(t, C.c_int32(len(arrs)), C.c_int32(V), _p(agg), _p(order))
    return agg, order


def cost_model(AF, AT, QT, indptr, V, F, cache_memory, Kg, counters, max_ids_list, train_step):
    L = lib()
    AF = np.ascontiguousarray(AF, dtype=np.uint64)
    AT = np.ascontiguousarray(AT, dtype=np.uint64)
    QT = np.ascontiguousarray(QT, dtype=np.int32)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    counters = None if counters is None else np.ascontiguousarray(counters, dtype=np.uint64)   # None: the PCM-free estimate
    mi = np.ascontiguousarray(max_ids_list, dtype=np.int32)
    nc_, ec_, a = C.c_int32(), C.c_int32(), C.c_int32()
    bt = C.c_float()
    L.lo_cost_model(_p(AF), _p(AT), _p(QT), _p(indptr), C.c_int32(V), C.c_int32(F), C.c_int64(cache_memory),
                    C.c_int32(Kg), _p(counters), _p(mi), C.c_int32(train_step), C.byref(nc_), C.byref(ec_),
                    C.byref(a), C.byref(bt))
    return dict(node_capacity=nc_.value, edge_capacity=ec_.value, alpha_idx=a.value, best_trans=bt.value)


def split_seeds(ids, G, partition_index=None, use_partition=0):
    L = lib()
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    n = len(ids)
    out = np.zeros((G, max(n, 1)), dtype=np.int32)
    num = np.zeros(G, dtype=np.int32)
    pi = None if partition_index is None else np.ascontiguousarray(partition_index, dtype=np.int32)
    L.lo_split_seeds(_p(ids), C.c_int32(n), C.c_int32(G), _p(pi), C.c_int32(use_partition), _p(out), _p(num))
    return [out[g, :num[g]].copy() for g in range(G)]


def coordinate(train_num, valid_num, test_num, raw_batch_size):
    L = lib()
    G = len(train_num)
    tn = np.asarray(train_num, dtype=np.int32)
    vn = np.asarray(valid_num, dtype=np.int32)
    sn = np.asarray(test_num, dtype=np.int32)
    steps = np.zeros(3, dtype=np.int32)
    tb, vb, sb = (np.zeros(G, dtype=np.int32) for _ in range(3))
    L.lo_coordinate(_p(tn), _p(vn), _p(sn), C.c_int32(G), C.c_int32(raw_batch_size), _p(steps), _p(tb), _p(vb),
                    _p(sb))
    return steps, tb, vb, sb


def schedule(steps, epoch, global_batch_id):
    L = lib()
    s = np.asarray(steps, dtype=np.int32)
    return (int(L.lo_get_current_mode(_p(s), C.c_int32(epoch), C.c_int32(global_batch_id))),
            int(L.lo_get_local_batch_id(_p(s), C.c_int32(epoch), C.c_int32(global_batch_id))))


def max_step(steps, epoch):
    s = np.asarray(steps, dtype=np.int32)
    return int(lib().lo_get_max_step(_p(s), C.c_int32(epoch)))


class DglSemanticsSampler:
    """DGL-semantics CPU baseline (oracle/dgl_cpu_sampler.c): uniform sampling without replacement,
    to_block compaction per layer, feature index_select; OpenMP.  Not a parity reference."""

    def __init__(self, indptr, indices, features, V, F, batch_size, fanout):
        L = lib()
        L.dgl_threads.restype = C.c_int32
        L.dgl_sample_batch.restype = C.c_int64
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.features = None if features is None else np.ascontiguousarray(features, dtype=np.float32)
        self.V, self.F = int(V), int(F)
        self.fanout = np.asarray(fanout, dtype=np.int32)
        nodes, scratch, edges = batch_size, 0, 0
        for f in fanout:
            scratch = max(scratch, nodes * f)
            edges += nodes * f
            nodes = nodes * (1 + f)
        self.local_map = np.full(self.V, -1, dtype=np.int32)
        self.nodes = np.empty(nodes, dtype=np.int32)
        self.src = np.empty(edges, dtype=np.int32)
        self.dst = np.empty(edges, dtype=np.int32)
        self.scratch = np.empty(scratch, dtype=np.int32)
        self.edge_off = np.zeros(len(fanout) + 1, dtype=np.int64)
        self.feat = None
        self.threads = int(L.dgl_threads())

    def run_batch(self, seeds, rng_seed=1, gather=True):
        L = lib()
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        if gather and self.features is not None and self.feat is None:
            self.feat = np.empty((len(self.nodes), self.F), dtype=np.float32)
        ne = C.c_int64(0)
        fo = self.feat if (gather and self.features is not None) else None
        n = L.dgl_sample_batch(_p(self.indptr), _p(self.indices), _p(self.features) if fo is not None else None,
                               C.c_int32(self.F), _p(seeds), C.c_int32(len(seeds)), _p(self.fanout),
                               C.c_int32(len(self.fanout)), C.c_uint64(rng_seed), _p(self.local_map), _p(self.nodes),
                               _p(self.src), _p(self.dst), _p(self.edge_off), _p(fo), _p(self.scratch), C.byref(ne))
        return int(n), int(ne.value)


# ---- synthetic datasets at full shape (oracle/synth_gen.c; spec: legion-1_amd/synth.py) --------------------------
def synth_csr(spec, skew=205):
    """(indptr int64[V+1], indices int32[E]) of a synthetic shape, generated on the host with OpenMP."""
    L = lib()
    L.sg_indptr.restype = C.c_int64
    indptr = np.empty(spec.V + 1, dtype=np.int64)
    ladder = np.ascontiguousarray(spec.ladder, dtype=np.int32)
    E = int(L.sg_indptr(_p(indptr), C.c_int32(spec.V), _p(ladder)))
    indices = np.empty(E, dtype=np.int32)
    L.sg_neighbors(_p(indices), C.c_int64(0), C.c_int64(E), C.c_uint32(spec.V), C.c_uint32(spec.M), C.c_uint32(spec.C),
                   C.c_uint32(skew))
    return indptr, indices


def synth_seed_ids(spec, i0, i1):
    out = np.empty(i1 - i0, dtype=np.int32)
    lib().sg_seed_ids(_p(out), C.c_int64(i0), C.c_int64(i1 - i0), C.c_uint32(spec.V), C.c_uint32(spec.M2), C.c_uint32(spec.C2))
    return out


def synth_labels_of(spec, ids):
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    out = np.empty(len(ids), dtype=np.int32)
    lib().sg_labels_of(_p(out), _p(ids), C.c_int64(len(ids)), C.c_int32(spec.classes))
    return out
