"""Second, independent restatement of the reference batch in plain Python (small cases only).

Written from the reference sources, not from oracle/legion_oracle.c, so that the two can be
checked against each other (the reference itself has no tests to pin either).
Follows Kernels.cu:68-96 (seeds), :112-150 (counters, the 2-hop state machine + the H-hop layout of
SURVEY 8a), :342-448 (sampler, canonical slot-ascending order), :450-463 (COO offsets),
:662-702 (feature rows).
"""
from __future__ import annotations

import numpy as np

M31 = 2147483647


def minstd_value(idx: int) -> int:
    """Value returned by dist(engine) after engine.discard(idx): 48271^(idx+1) mod (2^31-1)."""
    return pow(48271, idx + 1, M31)


def sample_index(idx: int, deg: int) -> int:
    x = minstd_value(idx)
    # IEEE double arithmetic exactly as thrust::uniform_real_distribution<double>
    r = np.float64(x - 1) / np.float64(2147483646.0)
    return int(np.float64(r * np.float64(deg)) + np.float64(0.0))


def update_counter_reference_2hop(nc, ec, op_id, size):
    """The reference's 2-hop counter state machine (update_counter, Kernels.cu:112-150) restated in Python."""
    if op_id == 0:
        nc[0] = size; nc[1] = 0; nc[2] = size; nc[3] = 0; nc[4] = size
        ec[0] = 0; ec[1] = 0; ec[2] = 0; ec[3] = 0
    elif op_id == 2:
        nc[0] += nc[1]; nc[5] = nc[3] + nc[4]; nc[6] = nc[1]; nc[1] = 0; nc[2] = ec[1]
        ec[3] += ec[1]; ec[2] = ec[0]; ec[0] += ec[1]; ec[1] = 0
    elif op_id == 4:
        nc[0] += nc[1]; nc[7] = nc[5] + nc[6]; nc[8] = nc[1]; nc[9] = nc[7] + nc[8]; nc[1] = 0; nc[2] = ec[1]
        ec[4] = ec[3] + ec[1]; ec[2] = ec[0]; ec[0] += ec[1]; ec[1] = 0


def run_batch(indptr, indices, feats, all_ids, all_labels, batch_size, counter, fanout):
    V = len(indptr) - 1
    H = len(fanout)
    total_cap = len(all_ids)
    nc = [0] * 16
    ec = [0] * 16
    seen = set()
    pos = {}
    ids, labels = [], []
    size = (total_cap - batch_size * counter) if batch_size * (counter + 1) >= total_cap else batch_size
    for i in range(size):
        g = size * counter + i                     # the kernel's batch_size parameter is `size`
        if g >= total_cap:
            ids.append(-1); labels.append(-1)
        else:
            s = int(all_ids[g % total_cap])
            ids.append(s); labels.append(int(all_labels[g % total_cap]))
            seen.add(s); pos[s] = i
    nc[0] = size; nc[2] = size; nc[3] = 0; nc[4] = size
    agg_src, agg_dst, src_off, dst_off = [], [], [], []
    for h in range(1, H + 1):
        f = fanout[h - 1]
        N = nc[2]
        inp = ids[:N] if h == 1 else agg_src[ec[2]:ec[2] + N]
        new_nodes, e_src, e_dst = [], [], []
        for idx in range(N * f):
            src = inp[idx // f]
            j = idx % f
            if src < 0:
                continue
            start = int(indptr[src]); deg = int(indptr[src + 1]) - start
            if j >= deg:
                continue
            dst = int(indices[start + sample_index(idx, deg)])
            if dst < 0:
                continue
            if dst not in seen:
                seen.add(dst)
                pos[dst] = nc[0] + len(new_nodes)
                new_nodes.append(dst)
            e_src.append(dst); e_dst.append(src)
        ids.extend(new_nodes)
        agg_src.extend(e_src); agg_dst.extend(e_dst)
        src_off.extend(pos[d] for d in e_src)
        dst_off.extend(pos[s] for s in e_dst)
        nc[1] = len(new_nodes); ec[1] = len(e_src)
        # update_counter, H-hop layout (bit compatible with the literal 2-hop code at H = 2)
        nc[0] += nc[1]
        nc[3 + 2 * h] = nc[1 + 2 * h] + nc[2 + 2 * h]
        nc[4 + 2 * h] = nc[1]
        if h == H:
            nc[5 + 2 * h] = nc[3 + 2 * h] + nc[4 + 2 * h]
        nc[1] = 0
        nc[2] = ec[1]
        ec[2 + h] = ec[0] + ec[1]
        ec[2] = ec[0]
        ec[0] += ec[1]
        ec[1] = 0
    n = nc[5 + 2 * H]
    F = feats.shape[1]
    out_feat = np.zeros((n, F), dtype=np.float32)
    for r in range(n):
        if ids[r] >= 0:
            out_feat[r] = feats[ids[r] % V]
    return dict(nc=np.array(nc, np.int32), ec=np.array(ec, np.int32), ids=np.array(ids[:n], np.int32),
                labels=np.array(labels, np.int32), src_off=np.array(src_off, np.int32),
                dst_off=np.array(dst_off, np.int32), features=out_feat)
