// peer_exchange.cpp -- the owner-computes gather inside the one-process `legion` server: the peers' rows of the unified
// cache arrive as BULK copies over xGMI (hipMemcpyPeerAsync) instead of 16-byte in-kernel peer loads.
//
// The reference's gather reads every cached row where it lives, in-kernel, over NVLink (zero_copy_with_aggregated_cache,
// Kernels.cu:662-702: cache_float_attrs[gidx / capacity] may be a peer's memory); k_gather does the same over xGMI.  This is
// the alternative BASELINE.json's north_star names ("xGMI hipMemcpyPeerAsync"), selected with $LEGION_PEER_GATHER=exchange
// (get_feature_kernel) or called directly (legion_peer_exchange_gather).  One process drives all GPUs, so the REQUESTER's
// thread does everything, nothing has to be agreed between the per-GPU runner threads:
//   1. plan      (requester GPU)  rows cached on another clique member are listed per owner (k_exch_count / k_exch_fill),
//                                 own-shard and backing-table rows are gathered locally at once (k_gather)
//   2. counts    one 64-byte pinned copy + ONE stream synchronisation: the copy sizes must be known on the host
//   3. per owner j with rows asked of it, on a stream of GPU j:
//        list  hipMemcpyPeerAsync   requester -> j   (4 bytes per row)
//        serve k_exch_rows          GPU j gathers the rows from ITS shard in its own HBM into a contiguous buffer
//        rows  hipMemcpyPeerAsync   j -> requester   (4F bytes per row: the xGMI traffic, one DMA per owner)
//   4. scatter   (requester GPU)  rows to their place in the feature buffer, behind the events of step 3
// Buffers are per (requester, owner), allocated on first use and only ever grown.  Bit-identical to the in-kernel gather.
#include "internal.h"

#include <algorithm>

#include "audit_hooks.h"

using namespace legion;

namespace {

struct OwnerLane {                 // staging on owner j for one requester
    hipStream_t stream = nullptr;  // on the owner's device
    hipEvent_t done = nullptr;
    int32_t* list = nullptr;       // rows of the owner's shard, device j
    float* rows = nullptr;         // served rows, device j
    int64_t cap = 0;
};

} // namespace

struct PeerExchange {
    int me = -1;                   // logical GPU of the requester
    int32_t F = 0;
    int32_t* req_row = nullptr;    // [num_ids] requester device
    int32_t* req_dst = nullptr;
    int32_t* counts = nullptr;     // device int32[2 * kMaxParts]
    int32_t* h_counts = nullptr;   // pinned
    float* in_rows = nullptr;      // requested rows as they arrive, owner-major (requester device)
    int64_t in_cap = 0;
    hipEvent_t planned = nullptr;
    OwnerLane lane[kMaxParts];
    int64_t batches = 0, rows_requested = 0, host_syncs = 0;
};

static void free_exchange(PeerExchange* x)
{
    if (!x) return;
    {
        DeviceGuard guard(x->me);
        (void)hipFree(x->req_row); (void)hipFree(x->req_dst); (void)hipFree(x->counts); (void)hipFree(x->in_rows);
        if (x->h_counts) (void)hipHostFree(x->h_counts);
        if (x->planned) (void)hipEventDestroy(x->planned);
    }
    for (int j = 0; j < kMaxParts; j++) {
        OwnerLane& l = x->lane[j];
        if (!l.stream) continue;
        DeviceGuard guard(j);
        (void)hipStreamSynchronize(l.stream);
        (void)hipFree(l.list); (void)hipFree(l.rows);
        (void)hipEventDestroy(l.done);
        (void)hipStreamDestroy(l.stream);
    }
    delete x;
}

extern "C" {

void GPUMemoryPool_ReleasePeerExchange(GPUMemoryPool* p)
{
    if (!p || !p->peer_exchange) return;
    free_exchange(p->peer_exchange);
    p->peer_exchange = nullptr;
}

// All rows [0, nc[0]) of the current pipe's batch.  Returns 0, or -1 with the sticky error set.
int legion_peer_exchange_gather(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* p, int32_t dev_id)
{
    if (!cache || !noder || !p || !p->owns_scratch) { LEGION_ARG_ERROR("legion_peer_exchange_gather: bad arguments"); return -1; }
    if (p->capturing) { LEGION_ARG_ERROR("legion_peer_exchange_gather: reads the request counts on the host, cannot be recorded into a batch graph"); return -1; }
    hipStream_t s = (hipStream_t)strm_hdl;
    const int Kg = cache->Kg, K0 = (dev_id / Kg) * Kg, me = dev_id % Kg;
    const int32_t F = noder->float_attr_len;
    for (int j = 0; j < Kg; j++)
        if (is_remote_device(K0 + j)) { LEGION_ARG_ERROR("legion_peer_exchange_gather: every clique member must be driven by this process (one process per GPU: legion_exchange_*)"); return -1; }
    PeerExchange* x = p->peer_exchange;
    if (!x) {
        x = new PeerExchange();
        x->me = dev_id; x->F = F;
        DeviceGuard guard(dev_id);
        HIP_CHECK(hipMalloc(&x->req_row, (size_t)p->num_ids * sizeof(int32_t)));
        HIP_CHECK(hipMalloc(&x->req_dst, (size_t)p->num_ids * sizeof(int32_t)));
        HIP_CHECK(hipMalloc(&x->counts, 2 * kMaxParts * sizeof(int32_t)));
        HIP_CHECK(hipHostMalloc((void**)&x->h_counts, kMaxParts * sizeof(int32_t), hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&x->planned, hipEventDisableTiming));
        p->peer_exchange = x;
    }
    if (legion_exchange_plan(strm_hdl, cache, noder, p, dev_id, x->req_row, x->req_dst, x->counts) != 0) return -1;
    HIP_CHECK(hipMemcpyAsync(x->h_counts, x->counts, kMaxParts * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipEventRecord(x->planned, s));
    if (legion_exchange_local(strm_hdl, cache, noder, p, dev_id) != 0) return -1;     // runs while the host waits for the counts
    HIP_CHECK(hipEventSynchronize(x->planned));                                       // the one host synchronisation
    x->host_syncs++;
    if (error_pending()) return -1;
    int64_t total = 0;
    for (int j = 0; j < Kg; j++) total += x->h_counts[j];
    if (total > p->num_ids || x->h_counts[me] != 0) { LEGION_ARG_ERROR("legion_peer_exchange_gather: inconsistent request counts"); return -1; }
    if (total > x->in_cap) {
        DeviceGuard guard(dev_id);
        HIP_CHECK(hipStreamSynchronize(s));
        (void)hipFree(x->in_rows);
        x->in_cap = std::min<int64_t>(p->num_ids, total + total / 4 + 1024);
        HIP_CHECK(hipMalloc(&x->in_rows, (size_t)x->in_cap * F * sizeof(float)));
    }
    const int phys_me = physical_device(dev_id);
    const int Ki = dev_id / Kg;
    int64_t off = 0;
    for (int j = 0; j < Kg; j++) {
        const int64_t n = x->h_counts[j];
        if (n == 0) continue;
        const int owner = K0 + j, phys_j = physical_device(owner);
        OwnerLane& l = x->lane[j];
        DeviceGuard guard(owner);
        if (!l.stream) {
            HIP_CHECK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
            HIP_CHECK(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
        }
        if (n > l.cap) {
            HIP_CHECK(hipStreamSynchronize(l.stream));
            (void)hipFree(l.list); (void)hipFree(l.rows);
            l.cap = std::min<int64_t>(p->num_ids, n + n / 4 + 1024);
            HIP_CHECK(hipMalloc(&l.list, (size_t)l.cap * sizeof(int32_t)));
            HIP_CHECK(hipMalloc(&l.rows, (size_t)l.cap * F * sizeof(float)));
        }
        HIP_CHECK(hipStreamWaitEvent(l.stream, x->planned, 0));
        HIP_CHECK(hipMemcpyPeerAsync(l.list, phys_j, x->req_row + off, phys_me, (size_t)n * sizeof(int32_t), l.stream));
        launch_exchange_rows(l.stream, false, cache->d_shard_tab[owner] + (size_t)j * cache->nchunks[Ki], cache->chunk_shift[Ki], l.list, (int32_t)n, F,
                             cache->shard_pitch, nullptr, l.rows, 0);
        HIP_CHECK(hipMemcpyPeerAsync(x->in_rows + off * F, phys_me, l.rows, phys_j, (size_t)n * F * sizeof(float), l.stream));
        HIP_CHECK(hipEventRecord(l.done, l.stream));
        off += n;
    }
    {
        DeviceGuard guard(dev_id);
        for (int j = 0; j < Kg; j++)
            if (x->h_counts[j] > 0) HIP_CHECK(hipStreamWaitEvent(s, x->lane[j].done, 0));
        if (total > 0) legion_exchange_scatter(strm_hdl, p, x->in_rows, x->req_dst, (int32_t)total, F);
    }
    x->batches++;
    x->rows_requested += total;
    return error_pending() ? -1 : 0;
}

// {batches, rows requested from peers, host synchronisations} since the pool was created
void legion_peer_exchange_stats(const GPUMemoryPool* p, int64_t out[3])
{
    out[0] = out[1] = out[2] = 0;
    if (!p || !p->peer_exchange) return;
    out[0] = p->peer_exchange->batches; out[1] = p->peer_exchange->rows_requested; out[2] = p->peer_exchange->host_syncs;
}

} // extern "C"
