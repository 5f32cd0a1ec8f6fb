"""One-process-per-GPU plumbing shared by bench.py and the tests.

The path shards by independent units: rank r of N samples the seeds with ``tid % N == r``
(GPUGraphStore.cu:332-346, validation/test always use ``tid % N``) and there is no collective on
the data path.  ``torch.distributed`` (nccl == RCCL on ROCm, gloo on CPU) is only used for the
start/stop barriers and to add up the per-rank totals.
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_seeds(ids, rank: int, world: int, partition_index=None, use_partition: bool = False):
    """Seeds of ``rank``: order preserving filter.  Works on numpy arrays and torch tensors."""
    if use_partition and partition_index is not None:
        part = partition_index[ids.long()] if hasattr(ids, "long") else partition_index[ids]
    else:
        part = ids % world
    return ids[part == rank]


def train_steps(n_seeds_per_rank, batch_size: int) -> int:
    """train_step = (min_g n_train_g - 1) / B   (CUDA_IPC_Service.cu:81-89: floor, tail dropped)."""
    return (min(n_seeds_per_rank) - 1) // batch_size


def _active(world: int) -> bool:
    """Collectives run when there is more than one rank -- or when $LEGION_DIST_FORCE=1 asks for them on an initialised
    1-rank group (tests/test_gpu_nccl_single.py: the RCCL code paths of these helpers on a one-GPU box)."""
    return world > 1 or os.environ.get("LEGION_DIST_FORCE") == "1"


def barrier(world: int):
    if _active(world):
        import torch.distributed as dist
        dist.barrier()


def aggregate(elapsed_s: float, totals, world: int, device=None):
    """(max over ranks of elapsed, element-wise sum over ranks of totals)."""
    if not _active(world):
        return float(elapsed_s), [float(t) for t in totals]
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "nccl":
        device = None          # gloo: host tensors
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    s = torch.tensor([float(x) for x in totals], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in s.tolist()]


def aggregate_max_vec(values, world: int, device=None):
    """Element-wise MAX over ranks of a list of floats (per-window elapsed times: every rank runs the same windows)."""
    if not _active(world):
        return [float(v) for v in values]
    import torch
    import torch.distributed as dist
    if dist.get_backend() != "nccl":
        device = None
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.tolist()]


def throughput_line(job_units: float, elapsed_max_s: float, steps: int):
    """whole-job units/s and ms per step from the aggregated numbers."""
    return job_units / elapsed_max_s, elapsed_max_s / steps * 1e3


def allreduce_device_u64(capi, ptr: int, n: int, world: int, device=None):
    """In-place SUM over ranks of a u64[n] device array owned by the C library (the clique-wide hotness
    sum of CandidateSelection, GPUCache.cu:624-627, as a collective instead of peer reads).
    RCCL when the process group is nccl; with gloo (CPU tests) the array is staged through the host."""
    if not _active(world):
        return
    import numpy as np
    import torch
    import torch.distributed as dist
    L = capi.lib()
    if dist.get_backend() == "nccl":
        t = torch.empty(n, dtype=torch.int64, device=device)
        L.d_copy_async(t.data_ptr(), ptr, n * 8, None)
        L.d_stream_sync(None)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        L.d_copy_async(ptr, t.data_ptr(), n * 8, None)
        L.d_stream_sync(None)
    else:
        h = capi.read_dev(ptr, np.int64, n)
        t = torch.from_numpy(h)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        L.d_copy_h_2_d(ptr, h.ctypes.data, n * 8)


def allgather_object(obj, world: int):
    if not _active(world):
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out
