#!/usr/bin/env python3
"""would a per-workgroup LDS hash ("warp-cooperative hash dedup", BASELINE north_star) pay?

k_sample claims every drawn neighbour with one probe / atomicMin on the position table, except repeated draws of the SAME ROW, which an
8-lane __shfl_up window settles in-wave (kernels.hip k_sample).  A per-tile LDS hash could additionally settle every slot whose neighbour
was already drawn by an EARLIER slot of the same tile (any row): that slot would learn "lost to slot x" from LDS and skip its table probe
and its claim.  This script measures how many such slots there are: for each BASELINE shape, each hop, a few batches, it runs the real
sampler hop by hop on the GPU, reads the per-slot candidates (GPUMemoryPool_GetCandidateBuffer: cand[idx] = the neighbour slot idx drew,
-1 = no draw) and counts, per tile of T slots (T = 1024, the build's kTile, and 2048):

    drawn       slots with a neighbour
    window      ... whose neighbour was drawn by one of the <= 8 previous slots of the same row   (already free today)
    tile_dup    ... not in `window`, whose neighbour occurs in an earlier slot of the same tile   (what an LDS hash would save)

Reference hot spot: the per-element atomicOr on the bitmap + LDS atomicAdd compaction of kernel_random_sampler_2 (Kernels.cu:412-431).

    python3 profiles/dedup_census.py [--shapes products,papers100M,uk-union] [--batches 3] > gpurun_out/r06_dedup_census.md   (GPU box)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def census(cand, f, T):
    """(drawn, window, tile_dup) of one hop's candidate array (slot idx = row * f + j)."""
    n = len(cand)
    idx = np.arange(n, dtype=np.int64)
    drawn = cand >= 0
    j = idx % f
    win = np.zeros(n, bool)
    for d in range(1, min(f - 1, 8) + 1):
        prev = np.empty_like(cand)
        prev[:d] = -2
        prev[d:] = cand[:-d]
        win |= drawn & (j >= d) & (prev == cand)
    # first occurrence of (tile, neighbour) in slot order: everything else in the group is a duplicate inside the tile
    keep = np.flatnonzero(drawn)
    key = (keep // T) * (1 << 32) + cand[keep].astype(np.int64)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.ones(len(ks), bool)
    first[1:] = ks[1:] != ks[:-1]
    dup = np.zeros(n, bool)
    dup[keep[order[~first]]] = True
    return int(drawn.sum()), int((win & drawn).sum()), int((dup & ~win).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="products,papers100M,uk-union")
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    import torch
    import bench
    import legion1_amd.capi as K
    import legion1_amd.synth as S
    L = K.lib()
    L.legion_set_error_mode(K.ERR_RETURN)
    L.SetGPUDevice(0)
    dev = torch.device("cuda", 0)
    print("# r06 dedup census: slots a per-tile LDS hash could settle (profiles/dedup_census.py)\n")
    print("| shape | fan-out | hop | slots drawn per batch | same-row window (free today) | earlier slot of the same 1024-slot tile | ... of the same 2048-slot tile |")
    print("|---|---|---|---|---|---|---|")
    worst = 0.0
    for shape, fans in (("products", ([25, 10], [25, 10, 5])), ("papers100M", ([25, 10, 5],)), ("uk-union", ([25, 10],))):
        if shape not in a.shapes.split(","):
            continue
        spec = S.spec_for(shape, scale=a.scale)
        indptr, indices, feats, E = bench.build_graph_on_gpu(K, spec, dev, 205, spec.F)
        del feats
        torch.cuda.empty_cache()
        B = 8000
        ids = torch.empty(spec.n_train, dtype=torch.int32, device=dev)
        L.legion_synth_seed_ids(None, ids.data_ptr(), 0, spec.n_train, spec.V, spec.M2, spec.C2, 1, 0)
        labels = torch.zeros(spec.n_train, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for fan in fans:
            seeds = dict(train=[((ids.data_ptr(), spec.n_train), (labels.data_ptr(), spec.n_train))])
            eng = K.Engine(indptr.data_ptr(), indices.data_ptr(), None, spec.V, spec.F, seeds, B, fan, G=1, csr_location=K.LOC_DEVICE, E=E)
            L.GPUCache_SetPreSc(eng.cache, 0)
            pool = eng.pools[0]
            tot = np.zeros((len(fan), 4), np.int64)      # drawn, window, dup1024, dup2048
            for it in range(a.batches):
                L.GPUMemoryPool_SetCurrentPipe(pool, 0)
                L.GPUMemoryPool_SetCurrentMode(pool, K.TRAINMODE)
                L.GPUMemoryPool_SetIter(pool, it)
                L.batch_generator_kernel(None, eng.noder, eng.cache, pool, B, it, 0, 0, K.TRAINMODE)
                for h, f in enumerate(fan):
                    L.d_stream_sync(None)
                    n_in = int(eng.out[0][0]["nc"].to_numpy(np.int32, 16)[2])
                    L.GPU_Random_Sampling(None, eng.graph, eng.cache, pool, int(f), 2 * h + 2, 0)
                    L.d_stream_sync(None)
                    K.check()
                    cand = K.read_dev(L.GPUMemoryPool_GetCandidateBuffer(pool), np.int32, n_in * f)
                    d, w, t1 = census(cand, f, 1024)
                    _, _, t2 = census(cand, f, 2048)
                    tot[h] += (d, w, t1, t2)
            eng.close()
            for h, f in enumerate(fan):
                d, w, t1, t2 = (float(x) / a.batches for x in tot[h])
                worst = max(worst, t2 / max(d, 1.0))
                print("| %s | %s | %d | %.0f | %.0f (%.2f %%) | %.0f (**%.2f %%**) | %.0f (**%.2f %%**) |" % (
                    shape, fan, h + 1, d, w, 100 * w / max(d, 1), t1, 100 * t1 / max(d, 1), t2, 100 * t2 / max(d, 1)))
        del indptr, indices, ids, labels
        torch.cuda.empty_cache()
    print("\nLargest share of a hop's draws that a per-tile hash (2048 slots) could settle: **%.2f %%** (VERDICT's threshold for building it: 10 %%)." % (100 * worst))


if __name__ == "__main__":
    main()
