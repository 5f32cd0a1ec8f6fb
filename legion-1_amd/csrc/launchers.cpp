// launchers.cpp -- the extern "C" operator launchers of the reference (src/Kernels.cuh:47-93,
// Kernels.cu:162-232,567-659,706-805) on top of kernels.hip, plus the Operator plugin classes
// (src/Operator.h:4-27, Operator.cu:10-124).
#include "internal.h"

#include <cstring>
#include <iostream>

#include "audit_hooks.h"

using namespace legion;

static inline int pool_dev(const GPUMemoryPool* p)
{
    if (p->device_id >= 0) return p->device_id;
    if (current_logical_device() >= 0) return current_logical_device();   // the logical GPU this thread selected (SetGPUDevice)
    int d = 0;
    (void)hipGetDevice(&d); // reference: dev_id = cudaGetDevice() (Kernels.cu:577-578)
    return d;
}

static bool pool_ready(const GPUMemoryPool* p, const char* who)
{
    if (!p || !p->owns_scratch) { LEGION_ARG_ERROR((std::string(who) + ": GPUMemoryPool_AllocateScratch was not called").c_str()); return false; }
    const int q = p->current_pipe;
    if (!p->sampled_ids[q] || !p->labels[q] || !p->agg_src_off[q] || !p->agg_dst_off[q] || !p->node_counter[q] || !p->edge_counter[q]) {
        LEGION_ARG_ERROR((std::string(who) + ": output buffers of the current pipe are not set").c_str());
        return false;
    }
    return true;
}

extern "C" {

// batch_generator_kernel, Kernels.cu:162-232
void batch_generator_kernel(void* strm_hdl, GPUNodeStorage* noder, GPUCache* cache, GPUMemoryPool* memorypool,
                            int32_t batch_size, int32_t counter, int32_t part_id, int32_t dev_id, int32_t mode)
{
    (void)cache; (void)part_id;
    if (!noder || !pool_ready(memorypool, "batch_generator_kernel")) return;
    hipStream_t s = (hipStream_t)strm_hdl;
    int32_t* all_ids = nullptr;
    int32_t* all_labels = nullptr;
    int32_t total_cap = 0;
    if (mode == LEGION_TRAINMODE) {
        all_ids = GPUNodeStorage_GetTrainingSetIds(noder, dev_id);
        all_labels = GPUNodeStorage_GetTrainingLabels(noder, dev_id);
        total_cap = GPUNodeStorage_TrainingSetSize(noder, dev_id);
    } else if (mode == LEGION_VALIDMODE) {
        all_ids = GPUNodeStorage_GetValidationSetIds(noder, dev_id);
        all_labels = GPUNodeStorage_GetValidationLabels(noder, dev_id);
        total_cap = GPUNodeStorage_ValidationSetSize(noder, dev_id);
    } else if (mode == LEGION_TESTMODE) {
        all_ids = GPUNodeStorage_GetTestingSetIds(noder, dev_id);
        all_labels = GPUNodeStorage_GetTestingLabels(noder, dev_id);
        total_cap = GPUNodeStorage_TestingSetSize(noder, dev_id);
    } else {
        log_out() << "invalid mode: " << mode << "\n";
    }
    if (all_ids == nullptr) { log_out() << "invalid src id ptr\n"; return; }
    if (all_labels == nullptr) { log_out() << "invalid label ptr\n"; return; }

    GPUMemoryPool* p = memorypool;
    const int q = p->current_pipe;
    if (p->device_id != dev_id && audit::on()) {   // the pool starts serving this GPU: its scratch and its output buffers must live there
        LEGION_AUDIT_OWNER(p->pos_map, dev_id, "batch_generator_kernel: scratch of the memory pool");
        LEGION_AUDIT_OWNER(p->cand, dev_id, "batch_generator_kernel: scratch of the memory pool");
        for (int i = 0; i < p->pipeline_depth; i++) {
            LEGION_AUDIT_OWNER(p->sampled_ids[i], dev_id, "batch_generator_kernel: output buffers of the memory pool");
            LEGION_AUDIT_OWNER(p->node_counter[i], dev_id, "batch_generator_kernel: output buffers of the memory pool");
        }
        LEGION_AUDIT_OWNER(all_ids, dev_id, "batch_generator_kernel: seed list");
    }
    p->device_id = dev_id;
    if (p->capturing) {
        // Recording a batch graph: cursor, epoch and the clamped size (Kernels.cu:224) are read / computed on the
        // device, `counter` is ignored, bounds are those of a full batch.
        if (batch_size > p->batch_size) { LEGION_ARG_ERROR("batch_generator_kernel: batch larger than the pool was sized for"); return; }
        launch_seed(s, p->sampled_ids[q], p->labels[q], batch_size, batch_size, 0, all_ids, all_labels, total_cap, p->pos_map, 0,
                    p->ctl, true, p->node_counter[q], p->edge_counter[q], p->aux2[1], p->fanout[0], p->max_slots);
        p->aux_ready_hop = 1; p->aux_ready_count = p->fanout[0];
        p->bound_n = batch_size > 0 ? batch_size : 0;
        p->bound_nodes = p->bound_n;
        return;
    }
    // A new batch = a new epoch of the position table: entries of older batches become stale without
    // touching them (replaces cudaMemsetAsync(accessed_map) + ClearPosMap, Kernels.cu:216,750-756).
    if (++p->batch_serial >= kSerialLimit) { // epoch space exhausted: wipe once and start over
        HIP_CHECK(hipMemsetAsync(p->pos_map, 0xFF, (size_t)p->V * sizeof(pos_t), s));
        p->batch_serial = 1;
    }
    const uint32_t epoch = kEpochTop - p->batch_serial;
    p->ctl_synced = false; // k_seed publishes (counter, epoch) of THIS batch: a batch graph must reset the cursor
    // Kernels.cu:224
    int32_t size = ((batch_size * (counter + 1)) >= total_cap) ? (total_cap - batch_size * counter) : batch_size;
    if (size > p->batch_size) { LEGION_ARG_ERROR("batch_generator_kernel: batch larger than the pool was sized for"); return; }
    launch_seed(s, p->sampled_ids[q], p->labels[q], batch_size, size, counter, all_ids, all_labels, total_cap, p->pos_map, epoch,
                p->ctl, false, p->node_counter[q], p->edge_counter[q], p->aux2[1], p->fanout[0], p->max_slots);
    p->aux_ready_hop = 1; p->aux_ready_count = p->fanout[0]; // k_seed prepared the slot states of hop 1
    p->bound_n = size > 0 ? size : 0;
    p->bound_nodes = p->bound_n;
}

// GPU_Random_Sampling, Kernels.cu:567-659
void GPU_Random_Sampling(void* strm_hdl, GPUGraphStorage* graph, GPUCache* cache, GPUMemoryPool* memorypool,
                         int32_t count, int32_t op_id, int is_presc)
{
    if (graph == nullptr) { log_out() << "invalid storage ptr\n"; return; }
    if (!pool_ready(memorypool, "GPU_Random_Sampling")) return;
    GPUMemoryPool* p = memorypool;
    const int hop = op_id / 2;
    if (op_id < 2 || (op_id & 1) || hop > p->hops) { LEGION_ARG_ERROR("GPU_Random_Sampling: op_id must be 2,4,..,2*hops"); return; }
    const int dev = pool_dev(p);
    const int P = graph->partition_count;
    if (dev < 0 || dev >= P) { LEGION_ARG_ERROR("GPU_Random_Sampling: device outside the partition table"); return; }
    const int64_t slots = (int64_t)p->bound_n * count;
    if (slots > p->max_slots) { LEGION_ARG_ERROR("GPU_Random_Sampling: fan-out exceeds what the pool was sized for"); return; }
    if (slots <= 0) return;

    CsrTables csr;
    csr.partition_count = P;
    // the whole CSR -- this GPU's HBM replica when there is one, else the (pinned host) table
    csr.indptr = graph->replica_indptr[dev] ? graph->replica_indptr[dev] : graph->csr_node_index_cpu;
    csr.indices = graph->replica_indices[dev] ? graph->replica_indices[dev] : graph->csr_dst_node_ids_cpu;
    csr.frag_indptr = nullptr; csr.frag_indices = nullptr;
    csr.ip_nch = csr.ix_nch = 1; csr.row_shift = graph->row_shift; csr.edge_shift = graph->edge_shift;
    csr.topo_owner = nullptr;
    csr.topo_row = nullptr;
    SamplerBuffers b;
    const int q = p->current_pipe;
    b.sampled_ids = p->sampled_ids[q]; b.agg_src_ids = p->agg_src_ids; b.agg_src_off = p->agg_src_off[q];
    b.agg_dst_off = p->agg_dst_off[q]; b.nc = p->node_counter[q]; b.ec = p->edge_counter[q];
    b.pos_map = p->pos_map; b.ctl = p->ctl; b.cand = p->cand; b.aux = p->aux2[hop & 1]; b.aux_next = p->aux2[(hop + 1) & 1]; b.tile_edge = p->tile_edge; b.tile_node = p->tile_node; b.tile_pre = p->tile_pre; b.chunk_tot = p->chunk_tot;
    b.hop_state = p->hop_state; b.edge_access_time = nullptr;
    if (is_presc) {
        // kernel_pre_sampler_optimized: host CSR only + topology hotness (Kernels.cu:636-649)
        if (!cache || dev >= cache->device_count || !cache->ctl[dev]->edge_access_time) { LEGION_ARG_ERROR("GPU_Random_Sampling: pre-sampling needs an initialised cache controller"); return; }
        b.edge_access_time = cache->ctl[dev]->edge_access_time;
    } else if (cache && dev < cache->device_count && cache->ctl[dev]->topo_owner && cache->ctl[dev]->edge_capacity > 0) {
        // every fragment the topology map of this GPU can name must be readable from here
        bool all = graph->d_frag_tab[dev] != nullptr;
        const int Kg = cache->Kg, K0 = (dev / Kg) * Kg;
        for (int g = K0; g < K0 + Kg && all; g++) all = graph->view[dev][g] && graph->frag[g].complete;
        if (all) {
            csr.frag_indptr = (const int64_t* const*)graph->d_frag_tab[dev];
            csr.frag_indices = (const int32_t* const*)(graph->d_frag_tab[dev] + (size_t)P * graph->ip_nch);
            csr.ip_nch = graph->ip_nch; csr.ix_nch = graph->ix_nch;
            csr.topo_owner = cache->ctl[dev]->topo_owner; csr.topo_row = cache->ctl[dev]->topo_row;
        }
    }
    b.aux_cap = p->max_slots;
    b.ids_cap = p->num_ids;
    b.V = p->V;
    b.aux_prepared = p->aux_ready_hop == hop && p->aux_ready_count == count;
    b.next_count = hop < p->hops ? p->fanout[hop] : 0;
    launch_sample_hop((hipStream_t)strm_hdl, csr, b, count, op_id, p->hops, (int32_t)slots, is_presc != 0);
    p->aux_ready_hop = hop + 1; p->aux_ready_count = b.next_count; // k_write prepared the next hop's slot states
    p->bound_n = (int32_t)slots;          // next hop expands every sampled edge endpoint
    p->bound_nodes += (int32_t)slots;
}

// Arguments of a gather over the rows (nc[off_idx], nc[size_idx]) of the current pipe; false (sticky error) if it cannot run.
static bool gather_args(GatherArgs& g, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* p, int32_t dev_id, int off_idx, int size_idx,
                        bool count_hits = true)
{
    const int32_t F = noder->float_attr_len;
    if (F < 0) log_out() << "error feature len\n"; // Kernels.cu:719-721
    const int q = p->current_pipe;
    if (!p->float_features[q]) { LEGION_ARG_ERROR("get_feature_kernel: feature buffer of the current pipe is not set"); return false; }
    const bool replica = dev_id >= 0 && dev_id < noder->partition_count && noder->replica_attrs[dev_id];
    g.table = replica ? noder->replica_attrs[dev_id] : noder->float_attrs;
    g.table_pitch = replica ? noder->replica_pitch : noder->float_attr_pitch;
    g.shard_pitch = cache ? cache->shard_pitch : 0;
    g.table_on_host = g.table == noder->float_attrs && noder->features_location != LEGION_LOC_DEVICE;
    g.shard_tab = nullptr; g.chunk_shift = 30; g.nchunks = 1;
    g.feat_map = nullptr;
    g.row_ptr = nullptr; g.row_ptr_ready = false;
    g.cache_capacity = 1;
    g.F = F;
    g.total_num_nodes = noder->total_num_nodes;
    g.sampled_ids = p->sampled_ids[q];
    g.nc = p->node_counter[q];
    g.dst = p->float_features[q];
    g.off_idx = off_idx;
    g.size_idx = size_idx;
    g.dst_rows = p->feature_rows;
    g.rows_seen = nullptr; g.rows_hint = 0;
    g.hit_stats = nullptr;
    if (p->rows_seen && p->rows_seen_dev) { // slot: level of a per-level gather, or the last one for "all rows of the batch"
        const int slot = off_idx < 0 ? LEGION_MAX_HOPS + 1 : (off_idx - 3) / 2;
        g.rows_hint = *(volatile int32_t*)(p->rows_seen + slot);
        g.rows_seen = p->rows_seen_dev + slot;
    }
    if (cache && dev_id >= 0 && dev_id < cache->device_count && cache->ctl[dev_id]->feat_map && cache->ctl[dev_id]->node_capacity > 0 &&
        cache->d_shard_tab[dev_id]) {
        const int Ki = dev_id / cache->Kg;
        g.feat_map = cache->ctl[dev_id]->feat_map;
        g.cache_capacity = cache->ctl[dev_id]->node_capacity;
        g.shard_tab = cache->d_shard_tab[dev_id]; // d_float_feature_cache_ptr_, GPUCache.cu:788-816
        g.chunk_shift = cache->chunk_shift[Ki];
        g.nchunks = cache->nchunks[Ki];
        g.row_ptr = p->row_ptr; // FindFeat + source selection as their own pass over the rows (k_row_ptrs)
        // the last gather of a batch: level `hops` of the per-level gathers, or the one gather over all rows
        // (a recorded batch graph would bake the decision in: graphs never sample)
        if (!p->capturing && count_hits) g.hit_stats = GPUCache_HitSampling(cache, dev_id, off_idx < 0 || off_idx == 3 + 2 * p->hops, off_idx < 0 || off_idx == 3);
    }
    if (!g.table && !g.feat_map) { LEGION_ARG_ERROR("get_feature_kernel: no feature table"); return false; }
    return true;
}

static void gather_common(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* p, int32_t dev_id,
                          int off_idx, int size_idx, int32_t rows_bound)
{
    GatherArgs g;
    if (!gather_args(g, cache, noder, p, dev_id, off_idx, size_idx)) return;
    launch_gather((hipStream_t)strm_hdl, g, rows_bound);
    // last counting launch of a sampled batch: the host may read the pinned {hits, rows} words once this event completes
    if (g.hit_stats && (off_idx < 0 || off_idx == 3 + 2 * p->hops)) GPUCache_HitSamplingDone(cache, dev_id, strm_hdl);
}

// $LEGION_PEER_GATHER=exchange and a filled clique cache (Kg > 1, every member in this process): the peers' rows travel as bulk
// copies (peer_exchange.cpp) instead of in-kernel xGMI loads
static bool use_peer_exchange(const GPUCache* cache, const GPUMemoryPool* p, int32_t dev_id)
{
    const char* e = getenv("LEGION_PEER_GATHER");
    if (!e || strcmp(e, "exchange") != 0 || !cache || p->capturing) return false;
    if (cache->Kg <= 1 || dev_id < 0 || dev_id >= cache->device_count || !cache->ctl[dev_id]->feat_map || cache->ctl[dev_id]->node_capacity <= 0) return false;
    const int K0 = (dev_id / cache->Kg) * cache->Kg;
    for (int j = 0; j < cache->Kg; j++) if (is_remote_device(K0 + j)) return false;
    return true;
}

// get_feature_kernel, Kernels.cu:706-748.  op_id 2l+1 gathers level l.
void get_feature_kernel(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool,
                        int32_t dev_id, int32_t op_id, int in_memory)
{
    if (!noder || !pool_ready(memorypool, "get_feature_kernel")) return;
    const int l = (op_id - 1) / 2;
    if (op_id < 1 || !(op_id & 1) || l > memorypool->hops) { LEGION_ARG_ERROR("get_feature_kernel: op_id must be 1,3,..,2*hops+1"); return; }
    if (!in_memory) return; // the reference only launches the in-memory path (Kernels.cu:737-746)
    if (use_peer_exchange(cache, memorypool, dev_id)) {   // one exchange per batch: the last level's op gathers every level
        if (l == memorypool->hops) legion_peer_exchange_gather(strm_hdl, cache, noder, memorypool, dev_id);
        return;
    }
    gather_common(strm_hdl, cache, noder, memorypool, dev_id, 3 + 2 * l, 4 + 2 * l, memorypool->level_bound[l]);
}

void get_feature_kernel_all(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool,
                            int32_t dev_id, int in_memory)
{
    if (!noder || !pool_ready(memorypool, "get_feature_kernel_all")) return;
    if (!in_memory) return;
    if (use_peer_exchange(cache, memorypool, dev_id)) { legion_peer_exchange_gather(strm_hdl, cache, noder, memorypool, dev_id); return; }
    gather_common(strm_hdl, cache, noder, memorypool, dev_id, -1, 0, memorypool->num_ids); // rows [0, nc[0])
}

// ---- owner-computes exchange variant of the gather (SURVEY 5 option b), one process per GPU ---------------------------------
// Step 1 on the requester.  Rows [0, nc[0]) of the batch: those cached on ANOTHER clique member are listed (req_row: row in the
// owner's shard, req_dst: row of the batch; contiguous per owner, owner-major; counts[j] rows for clique member j); every other row
// (own shard, backing table) gets its source address.  counts: device int32[2 * LEGION_MAX_DEVICE].  Nothing is gathered yet: the
// caller reads the counts back while legion_exchange_local (step 1b, any stream behind this one) gathers the local rows.
int legion_exchange_plan(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool, int32_t dev_id,
                         int32_t* req_row, int32_t* req_dst, int32_t* counts)
{
    if (!noder || !cache || !req_row || !req_dst || !counts || !pool_ready(memorypool, "legion_exchange_plan")) return -1;
    GPUMemoryPool* p = memorypool;
    GatherArgs g;
    if (!gather_args(g, cache, noder, p, dev_id, -1, 0, false)) return -1;      // the hit counter belongs to the lookup pass of k_row_ptrs
    if (!g.feat_map || !g.row_ptr || !p->cache_search_buffer) { LEGION_ARG_ERROR("legion_exchange_plan: needs a filled unified cache"); return -1; }
    launch_exchange_plan((hipStream_t)strm_hdl, g, dev_id % cache->Kg, cache->Kg, p->cache_search_buffer, counts, req_row, req_dst, p->num_ids);
    return error_pending() ? -1 : 0;
}
// Step 1b on the requester: gather the rows legion_exchange_plan resolved locally (own shard / backing table); the peers' rows
// have no source and are skipped -- legion_exchange_scatter fills them in.
int legion_exchange_local(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool, int32_t dev_id)
{
    if (!noder || !cache || !pool_ready(memorypool, "legion_exchange_local")) return -1;
    GPUMemoryPool* p = memorypool;
    GatherArgs g;
    if (!gather_args(g, cache, noder, p, dev_id, -1, 0, false)) return -1;
    if (!g.row_ptr) { LEGION_ARG_ERROR("legion_exchange_local: needs a filled unified cache"); return -1; }
    g.row_ptr_ready = true;   // k_exch_fill resolved the local rows (and left the peers' rows without a source)
    launch_gather((hipStream_t)strm_hdl, g, p->num_ids);
    return error_pending() ? -1 : 0;
}
// Step 2 on the owner: rows list[0..n) of THIS GPU's shard -> out[n x F] (a contiguous send buffer).
void legion_exchange_serve(void* strm_hdl, GPUCache* cache, int32_t dev_id, const int32_t* list, int32_t n, float* out)
{
    if (!cache || dev_id < 0 || dev_id >= cache->device_count || !cache->d_shard_tab[dev_id] || (n > 0 && (!list || !out))) { LEGION_ARG_ERROR("legion_exchange_serve: bad arguments"); return; }
    const int Ki = dev_id / cache->Kg, j = dev_id % cache->Kg;
    launch_exchange_rows((hipStream_t)strm_hdl, false, cache->d_shard_tab[dev_id] + (size_t)j * cache->nchunks[Ki], cache->chunk_shift[Ki], list, n,
                         cache->float_attr_len, cache->shard_pitch, nullptr, out, 0);
}
// Step 3 on the requester: rows[k] (as the owners returned them, in request order) -> feature row req_dst[k] of the current pipe.
void legion_exchange_scatter(void* strm_hdl, GPUMemoryPool* memorypool, const float* rows, const int32_t* req_dst, int32_t n, int32_t F)
{
    if (!pool_ready(memorypool, "legion_exchange_scatter")) return;
    float* dst = memorypool->float_features[memorypool->current_pipe];
    if (!dst || (n > 0 && (!rows || !req_dst))) { LEGION_ARG_ERROR("legion_exchange_scatter: bad arguments"); return; }
    launch_exchange_rows((hipStream_t)strm_hdl, true, nullptr, 0, req_dst, n, F, F, rows, dst, memorypool->feature_rows);
}

// make_update_plan, Kernels.cu:758-783
void make_update_plan(void* strm_hdl, GPUGraphStorage* graph, GPUCache* cache, GPUMemoryPool* memorypool,
                      int32_t dev_id, int32_t mode)
{
    (void)graph;
    if (!pool_ready(memorypool, "make_update_plan")) return;
    GPUMemoryPool* p = memorypool;
    const int q = p->current_pipe;
    hipStream_t s = (hipStream_t)strm_hdl;
    if (mode == LEGION_TRAINMODE && cache) // CacheProfiling: HotnessMeasure during pre-sampling
        GPUCache_CacheProfiling(cache, p->sampled_ids[q], p->node_counter[q], strm_hdl, dev_id);
    // ClearPosMap (Kernels.cu:780) needs no launch: the next batch starts a new table epoch.
    (void)s;
}

// update_cache, Kernels.cu:785-805: the reference body is commented out -- a no-op.
void update_cache(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool, int32_t dev_id,
                  int32_t mode)
{
    (void)strm_hdl; (void)cache; (void)noder; (void)memorypool; (void)dev_id; (void)mode;
}

} // extern "C"

// =================================== Operator plugin API ============================================
struct Operator {
    virtual ~Operator() = default;
    virtual void run(OpParams* params) = 0;
};

namespace {

inline void record(OpParams* p)
{
    if (p->event) HIP_CHECK(hipEventRecord((hipEvent_t)p->event, (hipStream_t)p->stream));
}

class Batch_Generator : public Operator { // Operator.cu:10-31
public:
    explicit Batch_Generator(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        GPUNodeStorage* noder = (GPUNodeStorage*)params->noder;
        GPUCache* cache = (GPUCache*)params->cache;
        GPUMemoryPool* memorypool = (GPUMemoryPool*)params->memorypool;
        int32_t mode = GPUMemoryPool_GetCurrentMode(memorypool);
        int32_t iter = GPUMemoryPool_GetIter(memorypool);
        IPCEnv* env = (IPCEnv*)params->env;
        int32_t device_id = params->device_id;
        int32_t batch_size = IPCEnv_GetCurrentBatchsize(env, device_id, mode);
        batch_generator_kernel(params->stream, noder, cache, memorypool, batch_size, iter, device_id, device_id, mode);
        record(params);
    }
private:
    [[maybe_unused]] int op_id_;
};

class Random_Sampler : public Operator { // Operator.cu:37-55
public:
    explicit Random_Sampler(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        GPU_Random_Sampling(params->stream, (GPUGraphStorage*)params->graph, (GPUCache*)params->cache,
                            (GPUMemoryPool*)params->memorypool, params->neighbor_count, op_id_, params->is_presc);
        record(params);
    }
private:
    [[maybe_unused]] int op_id_;
};

class Feature_Extractor : public Operator { // Operator.cu:61-76 (records no event)
public:
    explicit Feature_Extractor(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        get_feature_kernel(params->stream, (GPUCache*)params->cache, (GPUNodeStorage*)params->noder,
                           (GPUMemoryPool*)params->memorypool, params->device_id, op_id_, params->in_memory);
    }
private:
    [[maybe_unused]] int op_id_;
};

class Cache_Planner : public Operator { // Operator.cu:82-97
public:
    explicit Cache_Planner(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        GPUMemoryPool* memorypool = (GPUMemoryPool*)params->memorypool;
        int mode = GPUMemoryPool_GetCurrentMode(memorypool);
        make_update_plan(params->stream, (GPUGraphStorage*)params->graph, (GPUCache*)params->cache, memorypool,
                         params->device_id, mode);
        record(params);
    }
private:
    [[maybe_unused]] int op_id_;
};

class Cache_Updater : public Operator { // Operator.cu:103-119
public:
    explicit Cache_Updater(int op_id) : op_id_(op_id) {}
    void run(OpParams* params) override
    {
        GPUMemoryPool* memorypool = (GPUMemoryPool*)params->memorypool;
        int mode = GPUMemoryPool_GetCurrentMode(memorypool);
        update_cache(params->stream, (GPUCache*)params->cache, (GPUNodeStorage*)params->noder, memorypool,
                     params->device_id, mode);
        record(params);
    }
private:
    [[maybe_unused]] int op_id_;
};

} // namespace

extern "C" {
Operator* NewBatchGenerator(int op_id) { return new Batch_Generator(op_id); }
Operator* NewRandomSampler(int op_id) { return new Random_Sampler(op_id); }
Operator* NewFeatureExtractor(int op_id) { return new Feature_Extractor(op_id); }
Operator* NewCachePlanner(int op_id) { return new Cache_Planner(op_id); }
Operator* NewCacheUpdater(int op_id) { return new Cache_Updater(op_id); }
void Operator_run(Operator* op, OpParams* params) { if (op && params) op->run(params); }
void Operator_Delete(Operator* op) { delete op; }
}
