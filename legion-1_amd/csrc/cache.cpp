// cache.cpp -- host mirror of GPUCache / PreSCCacheController (GPUCache.cuh:9-161,
// GPUCache.cu:239-872): hotness profiling, candidate ranking, cost model, unified-cache fill-up
// and the id -> slot lookups.
//
// MI355X-first differences:
//  * the three BGHT cuckoo tables (GPUCache.cu:315-321) are replaced by direct-mapped tables over
//    [0, V) -- one 4-byte (1-byte) load per key instead of up to 3 x 128-byte bucket probes; the
//    key -> value / -1 function is identical.
//  * max_ids_ is tracked on the device (no blocking 64-byte D2H per batch, GPUCache.cu:291).
//  * ranks >= V are never dereferenced (the reference reads QF/QT out of bounds when
//    capacity*Kg > V).
//  * CostModel: when the budget covers all features and all adjacency the reference degenerates
//    (trans_* stay 0, GPUCache.cu:744-751); we then cache everything.  The Intel-PCM PCIe counter
//    input is optional: NULL selects an estimate from the edge hotness and the row degrees
//    (k_topo_transactions in kernels.hip has the per-row weight).
//  * feature shards use the line-aligned row pitch only when the padded shard fits the budget.
#include "internal.h"

#include <algorithm>
#include <cstring>
#include <iostream>

#include "audit_hooks.h"

using namespace legion;

#define MIN_INTERVAL 0.01 // GPUCache.cu:30
#define CLS 64            // GPUCache.cu:31

// first member of clique Ki that this process drives (-1: none)
static int clique_home(int Ki, int Kg)
{
    for (int j = 0; j < Kg; j++)
        if (!is_remote_device(Ki * Kg + j)) return Ki * Kg + j;
    return -1;
}

static int kg_of_mode(int cache_agg_mode)
{   // GPUCache.cu:593-607
    switch (cache_agg_mode) { case 0: return 1; case 1: return 2; case 2: return 4; case 3: return 8; default: return 0; }
}

extern "C" {

GPUCache* NewGPUCache(void) { return new GPUCache(); }

// GPUCache::Initialize, GPUCache.cu:508-535
void GPUCache_Initialize(GPUCache* c, int64_t cache_memory, int32_t int_attr_len, int32_t float_attr_len,
                         int32_t train_step, int32_t device_count)
{
    if (!c || device_count < 1 || device_count > kMaxParts) { LEGION_ARG_ERROR("GPUCache_Initialize: device_count must be 1..8"); return; }
    c->device_count = device_count;
    c->ctl.resize(device_count);
    for (int i = 0; i < device_count; i++) {
        c->ctl[i] = new CacheController();
        c->ctl[i]->train_step = train_step;
        c->ctl[i]->device_count = device_count;
    }
    c->float_feature_cache.assign(device_count, nullptr);
    c->cache_imported.assign(device_count, false);
    c->shard_chunks.assign(device_count, {});
    c->d_shard_tab.assign(device_count, nullptr);
    c->cache_memory = cache_memory;
    c->int_attr_len = int_attr_len;
    c->float_attr_len = float_attr_len;
    c->train_step = train_step;
    c->is_presc = true;
}

// PreSCCacheController::Initialize, GPUCache.cu:248-269
void GPUCache_InitializeCacheController(GPUCache* c, int32_t dev_id, int32_t total_num_nodes)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) { LEGION_ARG_ERROR("InitializeCacheController: bad dev_id"); return; }
    CacheController* k = c->ctl[dev_id];
    k->device_idx = dev_id;
    if (is_remote_device(dev_id)) return; // driven by another process: nothing lives here
    DeviceGuard guard(dev_id);
    k->total_num_nodes = total_num_nodes;
    HIP_CHECK(hipMalloc(&k->node_access_time, (size_t)total_num_nodes * sizeof(unsigned long long)));
    HIP_CHECK(hipMemset(k->node_access_time, 0, (size_t)total_num_nodes * sizeof(unsigned long long)));
    HIP_CHECK(hipMalloc(&k->edge_access_time, (size_t)total_num_nodes * sizeof(unsigned long long)));
    HIP_CHECK(hipMemset(k->edge_access_time, 0, (size_t)total_num_nodes * sizeof(unsigned long long)));
    HIP_CHECK(hipMalloc(&k->d_max_ids, sizeof(int32_t)));
    HIP_CHECK(hipMemset(k->d_max_ids, 0, sizeof(int32_t)));
    HIP_CHECK(hipMalloc(&k->d_global_count, sizeof(int32_t)));
    HIP_CHECK(hipHostMalloc((void**)&k->hit_stats, 4 * sizeof(int32_t), hipHostMallocMapped));
    if (k->hit_stats) {
        memset(k->hit_stats, 0, 4 * sizeof(int32_t));
        HIP_CHECK(hipHostGetDevicePointer((void**)&k->hit_stats_dev, k->hit_stats, 0));
    }
    for (int q = 0; q < 2; q++) { HIP_CHECK(hipEventCreateWithFlags(&k->hit_ev[q], hipEventDisableTiming)); k->hit_ev_armed[q] = false; }
    k->find_iter = 0; k->hit_samples = 0; k->last_hit_rate = -1.0;
    k->iter = 0;
    k->max_ids = 0;
    HIP_CHECK(hipDeviceSynchronize());
}

static void free_shard(GPUCache* c, int dev)
{
    for (float* p : c->shard_chunks[dev]) {
        if (!p) continue;
        if (c->cache_imported[dev]) (void)hipIpcCloseMemHandle(p);
        else (void)hipFree(p);
    }
    c->shard_chunks[dev].clear();
    c->float_feature_cache[dev] = nullptr;
    c->cache_imported[dev] = false;
}

// (re)write the device-side chunk-pointer tables of the local members of clique Ki
static void publish_shard_tables(GPUCache* c, int Ki)
{
    const int Kg = c->Kg, nch = c->nchunks[Ki];
    std::vector<float*> h((size_t)Kg * nch, nullptr);
    for (int j = 0; j < Kg; j++)
        for (size_t q = 0; q < c->shard_chunks[Ki * Kg + j].size() && (int)q < nch; q++) h[(size_t)j * nch + q] = c->shard_chunks[Ki * Kg + j][q];
    for (int j = 0; j < Kg; j++) {
        const int dev = Ki * Kg + j;
        if (is_remote_device(dev)) continue;
        DeviceGuard guard(dev);
        LEGION_AUDIT_TABLE(dev, h.data(), h.size(), "feature shard chunk table");
        if (!c->d_shard_tab[dev]) HIP_CHECK(hipMalloc(&c->d_shard_tab[dev], h.size() * sizeof(float*)));
        HIP_CHECK(hipMemcpy(c->d_shard_tab[dev], h.data(), h.size() * sizeof(float*), hipMemcpyHostToDevice));
    }
}

static void free_controller_maps(CacheController* k)
{
    if (k->feat_map) (void)hipFree(k->feat_map);
    if (k->topo_owner) (void)hipFree(k->topo_owner);
    if (k->topo_row) (void)hipFree(k->topo_row);
    k->feat_map = nullptr; k->topo_owner = nullptr; k->topo_row = nullptr;
}

void GPUCache_Finalize(GPUCache* c, int32_t dev_id)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) return;
    CacheController* k = c->ctl[dev_id];
    DeviceGuard guard(dev_id);
    free_controller_maps(k);
    if (k->node_access_time) (void)hipFree(k->node_access_time);
    if (k->edge_access_time) (void)hipFree(k->edge_access_time);
    if (k->d_max_ids) (void)hipFree(k->d_max_ids);
    if (k->d_global_count) (void)hipFree(k->d_global_count);
    if (k->hit_stats) { (void)hipHostFree(k->hit_stats); k->hit_stats = nullptr; k->hit_stats_dev = nullptr; }
    for (int q = 0; q < 2; q++) if (k->hit_ev[q]) { (void)hipEventDestroy(k->hit_ev[q]); k->hit_ev[q] = nullptr; k->hit_ev_armed[q] = false; }
    k->node_access_time = k->edge_access_time = nullptr;
    k->d_max_ids = k->d_global_count = nullptr;
    free_shard(c, dev_id);
    if (c->d_shard_tab[dev_id]) { (void)hipFree(c->d_shard_tab[dev_id]); c->d_shard_tab[dev_id] = nullptr; }
}

// "Feature Cache Hit" (GPUCache.cu:414-425): every $LEGION_CACHE_HIT_PERIOD-th batch (default 500, the reference's
// find_iter_ % 500) the lookup passes of the batch count their hits.  Called by the gather launchers with
// first / last = this is the first / last gather launch of the batch; returns the device pointer the lookup pass adds
// {hits, rows} to, or null when the batch is not sampled.  The ratio of sampling k - 1 is printed when sampling k starts
// (the reference prints after a blocking copy; here nothing waits for the GPU).
int32_t* GPUCache_HitSampling(GPUCache* c, int32_t dev_id, int last_launch_of_batch, int first_launch_of_batch)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) return nullptr;
    CacheController* k = c->ctl[dev_id];
    if (!k->hit_stats_dev) return nullptr;
    const char* e = getenv("LEGION_CACHE_HIT_PERIOD");
    const int period = e && atoi(e) > 0 ? atoi(e) : 500;
    const bool sampled = k->find_iter % period == 0;
    int32_t* out = nullptr;
    if (sampled) {
        const int slot = k->hit_samples & 1;
        // The host reads / clears the pinned words without synchronising, so a slot is only touched once the event recorded
        // behind the launches that counted into it has completed (GPUCache_HitSamplingDone): with pipelined batches and a
        // short period the previous sampling may still be in flight -- then its ratio is printed one sampling later, and
        // a slot that is still being written is not handed out again (this batch is simply not sampled).
        auto settled = [&](int q) { return !k->hit_ev_armed[q] || hipEventQuery(k->hit_ev[q]) == hipSuccess; };
        if (first_launch_of_batch) {
            int32_t* prev = k->hit_stats + 2 * (slot ^ 1);
            if (k->hit_samples > 0 && prev[1] > 0 && settled(slot ^ 1)) {
                k->last_hit_rate = (double)prev[0] / (double)prev[1];
                log_out() << dev_id << " Feature Cache Hit: " << k->last_hit_rate << std::endl;
                prev[1] = 0;   // printed once
            }
            if (settled(slot)) { k->hit_stats[2 * slot] = 0; k->hit_stats[2 * slot + 1] = 0; k->hit_ev_armed[slot] = false; }
        }
        (void)hipGetLastError();   // hipEventQuery's hipErrorNotReady is not an error
        if (!k->hit_ev_armed[slot]) out = k->hit_stats_dev + 2 * slot;
    }
    if (last_launch_of_batch) {
        if (sampled && out) k->hit_samples++;
        k->find_iter++;
    }
    return out;
}
// called by the launcher behind the last counting launch of a sampled batch (stream = the stream it ran on)
void GPUCache_HitSamplingDone(GPUCache* c, int32_t dev_id, void* stream)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) return;
    CacheController* k = c->ctl[dev_id];
    if (!k->hit_stats_dev || k->hit_samples == 0) return;
    const int slot = (k->hit_samples - 1) & 1;   // the sampling that just finished its launches
    if (!k->hit_ev[slot]) return;
    HIP_CHECK(hipEventRecord(k->hit_ev[slot], (hipStream_t)stream));
    k->hit_ev_armed[slot] = true;
}
// the newest completed sample (synchronises the device): hits / rows of the last sampled batch, or -1 if there is none
double GPUCache_FeatureCacheHitRate(GPUCache* c, int32_t dev_id, int32_t* hits_out, int32_t* rows_out)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count || !c->ctl[dev_id]->hit_stats || c->ctl[dev_id]->hit_samples == 0) return -1.0;
    CacheController* k = c->ctl[dev_id];
    { DeviceGuard guard(dev_id); HIP_CHECK(hipDeviceSynchronize()); }
    const int32_t* s = k->hit_stats + 2 * ((k->hit_samples - 1) & 1);
    if (hits_out) *hits_out = s[0];
    if (rows_out) *rows_out = s[1];
    return s[1] > 0 ? (double)s[0] / (double)s[1] : -1.0;
}

int32_t GPUCache_NodeCapacity(const GPUCache* c, int32_t dev_id)
{   // GPUCache.cu:549-551
    size_t ki = (size_t)(dev_id / (c->Kg > 0 ? c->Kg : 1));
    return ki < c->node_capacity.size() ? c->node_capacity[ki] : 0;
}
int32_t GPUCache_EdgeCapacity(const GPUCache* c, int32_t dev_id)
{
    size_t ki = (size_t)(dev_id / (c->Kg > 0 ? c->Kg : 1));
    return ki < c->edge_capacity.size() ? c->edge_capacity[ki] : 0;
}

// FindFeat, GPUCache.cu:387-432 (the 500-batch hit-rate print is metrics only and not restated)
void GPUCache_FindFeat(GPUCache* c, int32_t* sampled_ids, int32_t* cache_offset, int32_t* node_counter,
                       int32_t op_id, void* stream, int32_t dev_id)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) { LEGION_ARG_ERROR("FindFeat: bad dev_id"); return; }
    launch_find_feat((hipStream_t)stream, sampled_ids, cache_offset, node_counter, op_id, c->ctl[dev_id]->feat_map, 1 << 22);
}

// FindTopo, GPUCache.cu:434-443
void GPUCache_FindTopo(GPUCache* c, int32_t* input_ids, char* partition_index, int32_t* partition_offset,
                       int32_t batch_size, int32_t op_id, void* stream, int32_t dev_id)
{
    (void)op_id;
    if (!c || dev_id < 0 || dev_id >= c->device_count) { LEGION_ARG_ERROR("FindTopo: bad dev_id"); return; }
    CacheController* k = c->ctl[dev_id];
    launch_find_topo((hipStream_t)stream, input_ids, (int8_t*)partition_index, partition_offset, batch_size, k->topo_owner, k->topo_row);
}

// CacheProfiling, GPUCache.cu:275-303 (+ GPUCache::CacheProfiling :851-863)
void GPUCache_CacheProfiling(GPUCache* c, int32_t* sampled_ids, int32_t* node_counter, void* stream, int32_t dev_id)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) { LEGION_ARG_ERROR("CacheProfiling: bad dev_id"); return; }
    CacheController* k = c->ctl[dev_id];
    if (c->is_presc) {
        launch_hotness((hipStream_t)stream, sampled_ids, node_counter, 0, k->node_access_time, k->d_max_ids, 1 << 22);
        if (k->iter == (k->train_step - 1)) k->iter = 0;
    }
    k->iter++;
}

int32_t GPUCache_MaxIdNum(const GPUCache* c, int32_t dev_id)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count) return 0;
    CacheController* k = c->ctl[dev_id];
    if (k->d_max_ids) {
        DeviceGuard guard(dev_id);
        int32_t v = 0;
        HIP_CHECK(hipMemcpy(&v, k->d_max_ids, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (v > k->max_ids) k->max_ids = v;
    }
    return k->max_ids;
}

// CandidateSelection, GPUCache.cu:578-659
void GPUCache_CandidateSelection(GPUCache* c, int cache_agg_mode, GPUNodeStorage* noder, GPUGraphStorage* graph)
{
    (void)graph;
    if (!c || !noder) { LEGION_ARG_ERROR("CandidateSelection: null argument"); return; }
    const int Kg = kg_of_mode(cache_agg_mode);
    if (Kg == 0 || c->device_count % Kg != 0) { LEGION_ARG_ERROR("CandidateSelection: cache_agg_mode does not divide the device count"); return; }
    const int Kc = c->device_count / Kg;
    c->Kc = Kc;
    c->Kg = Kg;
    log_out() << "xGMI Clique: " << Kc << " GPU Per Clique: " << Kg << std::endl;
    const int32_t V = noder->total_num_nodes;
    for (auto p : c->QF) if (p) (void)hipFree(p);
    for (auto p : c->QT) if (p) (void)hipFree(p);
    for (auto p : c->AF) if (p) (void)hipFree(p);
    for (auto p : c->AT) if (p) (void)hipFree(p);
    c->QF.clear(); c->QT.clear(); c->AF.clear(); c->AT.clear();
    for (int i = 0; i < Kc; i++) {
        const int home = clique_home(i, Kg);
        if (home < 0) { // no member of this clique lives in this process
            c->QF.push_back(nullptr); c->AF.push_back(nullptr); c->QT.push_back(nullptr); c->AT.push_back(nullptr);
            continue;
        }
        DeviceGuard guard(home);
        for (int pass = 0; pass < 2; pass++) {
            int32_t* order = nullptr;
            unsigned long long* agg = nullptr;
            HIP_CHECK(hipMalloc(&order, (size_t)V * sizeof(int32_t)));
            HIP_CHECK(hipMalloc(&agg, (size_t)V * sizeof(unsigned long long)));
            HIP_CHECK(hipMemset(agg, 0, (size_t)V * sizeof(unsigned long long)));
            for (int j = 0; j < Kg; j++) { // peer reads of the clique members' hotness arrays (:624-627,644-647)
                CacheController* k = c->ctl[i * Kg + j];
                if (is_remote_device(i * Kg + j)) continue; // its hotness was summed in by the caller (RCCL all-reduce)
                launch_aggregate_access(nullptr, agg, pass == 0 ? k->node_access_time : k->edge_access_time, V);
            }
            launch_iota(nullptr, order, V);
            sort_by_hotness_desc(nullptr, agg, order, V);
            if (pass == 0) { c->QF.push_back(order); c->AF.push_back(agg); }
            else { c->QT.push_back(order); c->AT.push_back(agg); }
        }
        HIP_CHECK(hipDeviceSynchronize());
    }
    c->is_presc = false;
}

void GPUCache_SetPreSc(GPUCache* c, int is_presc) { if (c) c->is_presc = is_presc != 0; }

void GPUCache_SetCapacity(GPUCache* c, int32_t node_capacity, int32_t edge_capacity)
{
    c->capacity_forced = true;
    c->forced_node_capacity = node_capacity;
    c->forced_edge_capacity = edge_capacity;
}

// ---- CostModel, GPUCache.cu:661-767 --------------------------------------------------------------------------------
// What the reference computes, per clique: a budget of M = cache_memory * Kg bytes is split between adjacency rows
// (share alpha) and feature rows (share 1 - alpha) in 100 steps; for every split the PCIe transactions the cached rows
// would have saved during the pre-sampling epoch are estimated from the hotness curves, and the best split decides the
// two capacities.  The float arithmetic, its evaluation order and the two log lines are the observable contract (the
// oracle restates them from the reference); how the curves are held and walked is ours.
} // extern "C"
namespace {

// Inclusive prefix sums, in ranking order, of: node hotness, edge hotness, adjacency bytes of the ranked rows
struct HotnessCurves {
    std::vector<uint64_t> node, edge, edge_bytes;
    uint64_t node_total() const { return node.back(); }
    uint64_t edge_total() const { return edge.back(); }
    uint64_t adjacency_bytes() const { return edge_bytes.back(); }
};

// scan on the device (hipCUB), one staging buffer
HotnessCurves load_curves(const GPUCache* c, int clique, const GPUGraphStorage* graph, int32_t V, uint64_t* topo_trans_estimate)
{
    HotnessCurves h;
    h.node.resize(V); h.edge.resize(V); h.edge_bytes.resize(V);
    uint64_t *d_in = nullptr, *d_out = nullptr;
    HIP_CHECK(hipMalloc(&d_in, (size_t)V * sizeof(uint64_t)));
    HIP_CHECK(hipMalloc(&d_out, (size_t)V * sizeof(uint64_t)));
    auto fetch = [&](const uint64_t* src, std::vector<uint64_t>& dst) {
        inclusive_scan_u64(nullptr, src, d_out, V);
        HIP_CHECK(hipMemcpy(dst.data(), d_out, (size_t)V * sizeof(uint64_t), hipMemcpyDeviceToHost));
    };
    fetch((const uint64_t*)c->AF[clique], h.node);
    fetch((const uint64_t*)c->AT[clique], h.edge);
    launch_edge_mem(nullptr, c->QT[clique], d_in, V, graph->csr_node_index_cpu);   // GetEdgeMem, GPUCache.cu:35-41
    fetch(d_in, h.edge_bytes);
    if (topo_trans_estimate) {
        // PCM-free input: 64-byte transactions of the pre-sampling epoch's adjacency reads, from our own
        // hotness counters -- deterministic, no MSR access.  See topo_transactions_of() for the per-row weight.
        launch_topo_transactions(nullptr, c->QT[clique], (const uint64_t*)c->AT[clique], d_in, V, graph->csr_node_index_cpu);
        inclusive_scan_u64(nullptr, d_in, d_out, V);
        HIP_CHECK(hipMemcpy(topo_trans_estimate, d_out + (V - 1), sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    HIP_CHECK(hipFree(d_in));
    HIP_CHECK(hipFree(d_out));
    return h;
}

// transactions the `rows` hottest rows account for, out of `total` (GPUCache.cu:745,749: total * 1.0 / all * prefix, then float)
inline float saved_by(uint64_t total, const std::vector<uint64_t>& prefix, int32_t rows)
{
    const uint64_t covered = rows > 0 ? prefix[rows - 1] : 0;   // the reference reads prefix[-1] at rows == 0
    return (float)((double)total * 1.0 / (double)prefix.back() * (double)covered);
}

struct BudgetSplit { int64_t step; float transactions, feat_rows_per_gpu, topo_rows_per_gpu; };

// the alpha sweep of GPUCache.cu:721-761: split point s gives s budget steps to the adjacency and the rest to the features
BudgetSplit best_split(const HotnessCurves& h, int32_t V, int32_t F, int Kg, int64_t budget, uint64_t topo_trans, uint64_t feat_trans)
{
    int64_t step_bytes = (int64_t)((double)budget * MIN_INTERVAL);
    if (step_bytes < 1) step_bytes = 1;
    const int64_t n_steps = (budget - 1) / step_bytes + 1;
    const uint64_t table_bytes = (uint64_t)V * F * sizeof(float);
    const int64_t rows_per_step = step_bytes / (int64_t)(F * sizeof(float));
    // what s steps of budget buy on either side (rows per GPU kept as float: the reference stores them in a float vector)
    std::vector<float> topo_saved(n_steps + 1, 0.f), topo_rows(n_steps + 1, 0.f), feat_saved(n_steps + 1, 0.f), feat_rows(n_steps + 1, 0.f);
    for (int64_t s = 0; s < n_steps; s++) {
        const int64_t bytes = s * step_bytes;
        const int32_t nf = (uint64_t)bytes > table_bytes ? V : (int32_t)((s + 1) * rows_per_step);
        const int32_t nt = (uint64_t)bytes > h.adjacency_bytes()
                               ? V : (int32_t)(std::lower_bound(h.edge_bytes.begin(), h.edge_bytes.end(), (uint64_t)bytes) - h.edge_bytes.begin());
        if (nt < V) { topo_saved[s] = saved_by(topo_trans, h.edge, nt); topo_rows[s] = (float)(nt / Kg); }   // "everything fits" keeps 0: :744-751
        if (nf < V) { feat_saved[s] = saved_by(feat_trans, h.node, nf); feat_rows[s] = (float)(nf / Kg); }
    }
    // first maximum over s in [0, n_steps] with the end points pinned to 0 (std::max_element over trans_of_total, :757-760)
    BudgetSplit best{0, 0.f, 0.f, 0.f};
    for (int64_t s = 1; s < n_steps; s++) {
        const float t = topo_saved[s] + feat_saved[n_steps - 1 - s];
        if (t > best.transactions) { best.step = s; best.transactions = t; }
    }
    best.feat_rows_per_gpu = feat_rows[n_steps - 1 - best.step];
    best.topo_rows_per_gpu = topo_rows[best.step];
    return best;
}

} // namespace
extern "C" {

void GPUCache_CostModel(GPUCache* c, int cache_agg_mode, GPUNodeStorage* noder, GPUGraphStorage* graph,
                        const uint64_t* counters, int32_t train_step)
{
    (void)cache_agg_mode;
    if (!c || !noder || !graph) { LEGION_ARG_ERROR("CostModel: null argument"); return; }
    const int32_t V = noder->total_num_nodes;
    const int32_t F = noder->float_attr_len;
    const int Kg = c->Kg, Kc = c->Kc;
    c->node_capacity.clear(); c->edge_capacity.clear(); c->alpha.clear();
    log_out() << "Start solve cost model" << std::endl;
    for (int i = 0; i < Kc; i++) {
        const int home = clique_home(i, Kg);
        if (c->capacity_forced || home < 0) {
            c->node_capacity.push_back(c->capacity_forced ? c->forced_node_capacity : 0);
            c->edge_capacity.push_back(c->capacity_forced ? c->forced_edge_capacity : 0);
            c->alpha.push_back(-1.0);
            continue;
        }
        DeviceGuard guard(home);
        uint64_t topo_trans = 0;
        const HotnessCurves h = load_curves(c, i, graph, V, counters ? nullptr : &topo_trans);
        if (counters) topo_trans = counters[0] + counters[1];       // the two Intel-PCM PCIe read counters (Server.cu:100,108)
        const int64_t budget = c->cache_memory * Kg;
        // MI355X extension: the budget covers everything -> cache everything (see header comment)
        if ((uint64_t)budget >= (uint64_t)V * F * sizeof(float) + h.adjacency_bytes()) {
            c->node_capacity.push_back(V / Kg + 1);
            c->edge_capacity.push_back(V / Kg + 1);
            c->alpha.push_back((double)h.adjacency_bytes() / (double)budget);
            log_out() << "Budget covers all data: caching everything\n";
            continue;
        }
        // Feature transactions of the epoch.  The reference sums cache_controller_[j]->MaxIdNum() for j < Kg -- the members
        // of the FIRST clique -- for every clique i (GPUCache.cu:677-680: index j, not i * Kg + j).  Restated as is: the
        // capacities of cliques >= 1 depend on it whenever the GPUs saw different batch sizes.  (A member another process
        // drives: this clique's home GPU.)
        uint64_t feat_trans = 0;
        for (int j = 0; j < Kg; j++)
            feat_trans += (uint64_t)(((int64_t)GPUCache_MaxIdNum(c, is_remote_device(j) ? home : j) * train_step * F * (int64_t)sizeof(float)) / CLS);
        const BudgetSplit best = best_split(h, V, F, Kg, budget, topo_trans, feat_trans);
        log_out() << "Alpha: " << (best.step * MIN_INTERVAL) << " Transactions: " << best.transactions << std::endl;
        c->node_capacity.push_back((int32_t)(best.feat_rows_per_gpu + 1));
        c->edge_capacity.push_back((int32_t)(best.topo_rows_per_gpu + 1));
        c->alpha.push_back(best.step * MIN_INTERVAL);
        log_out() << "Feat capacity " << best.feat_rows_per_gpu << " topo capacity " << best.topo_rows_per_gpu << std::endl;
    }
}

// Row pitch of the feature shards.  The shards are the bytes `cache_memory` pays for (the reference's contract: dense rows,
// capacity = budget / (F * 4), GPUCache.cu:727), so the line-aligned pitch of legion_row_pitch (F = 100: 128 floats, +28 %)
// is only taken when the padded shard still fits the feature share of the budget; otherwise the rows stay dense.
// feat_budget_bytes <= 0: no budget known (capacity set by the caller, cache_memory == 0).
int32_t legion_shard_pitch(int32_t F, int64_t rows, int64_t feat_budget_bytes)
{
    const int32_t aligned = legion_row_pitch(F);
    if (aligned == F || feat_budget_bytes <= 0) return aligned;
    return rows * aligned * (int64_t)sizeof(float) <= feat_budget_bytes ? aligned : F;
}

// FillUp, GPUCache.cu:769-826 (+ InitializeMap :306-323, Insert :325-371, FeatFillUp :200-205)
void GPUCache_FillUp(GPUCache* c, int cache_agg_mode, GPUNodeStorage* noder, GPUGraphStorage* graph)
{
    (void)cache_agg_mode;
    if (!c || !noder || !graph) { LEGION_ARG_ERROR("FillUp: null argument"); return; }
    if ((int)c->node_capacity.size() != c->Kc) { LEGION_ARG_ERROR("FillUp: run CostModel / SetCapacity first"); return; }
    const int32_t V = noder->total_num_nodes;
    const int32_t F = noder->float_attr_len;
    const int Kg = c->Kg;
    // shard rows start on a 128-byte line (F = 100 -> 512-byte rows) when that fits the feature share of EVERY clique's budget
    int32_t pitch = legion_row_pitch(F);
    for (int i = 0; i < c->Kc && pitch != F; i++) {
        const double a = (i < (int)c->alpha.size() && c->alpha[i] >= 0.0) ? c->alpha[i] : 0.0;
        pitch = legion_shard_pitch(F, c->node_capacity[i], c->cache_memory > 0 ? (int64_t)((1.0 - a) * (double)c->cache_memory) : 0);
    }
    c->shard_pitch = pitch;
    c->chunk_shift.assign(c->Kc, 30);
    c->nchunks.assign(c->Kc, 1);
    // A shard is a list of chunk allocations (2^chunk_shift rows each, <= 1 GiB by default): a large single
    // allocation could not be imported over HIP IPC on the test pool (profiles/r01_unified_ipc_notes.md).
    const char* env_chunk = getenv("LEGION_SHARD_CHUNK_BYTES");
    const int64_t chunk_bytes = env_chunk ? atoll(env_chunk) : (1ll << 30);
    for (int i = 0; i < c->Kc; i++) {
        const int32_t ncap = c->node_capacity[i], ecap = c->edge_capacity[i];
        int shift = 0;
        while (shift < 30 && (2ll << shift) * pitch * (int64_t)sizeof(float) <= chunk_bytes) shift++;
        const int32_t rpc = 1 << shift;
        c->chunk_shift[i] = shift;
        c->nchunks[i] = ncap > 0 ? (ncap + rpc - 1) / rpc : 1;
        for (int j = 0; j < Kg; j++)
            if (c->d_shard_tab[i * Kg + j]) { (void)hipFree(c->d_shard_tab[i * Kg + j]); c->d_shard_tab[i * Kg + j] = nullptr; }
        for (int j = 0; j < Kg; j++) {
            const int dev = i * Kg + j;
            if (is_remote_device(dev)) continue; // its shard is imported (GPUCache_ImportFeatureShard)
            DeviceGuard guard(dev);
            CacheController* k = c->ctl[dev];
            free_controller_maps(k);
            k->node_capacity = ncap;
            k->edge_capacity = ecap;
            HIP_CHECK(hipMalloc(&k->feat_map, (size_t)V * sizeof(int32_t)));
            HIP_CHECK(hipMalloc(&k->topo_owner, (size_t)V));
            HIP_CHECK(hipMalloc(&k->topo_row, (size_t)V * sizeof(int32_t)));
            LEGION_AUDIT_OWNER(k->feat_map, dev, "FillUp: id -> slot map");
            launch_build_feat_map(nullptr, k->feat_map, c->QF[i], ncap, Kg, V);
            launch_build_topo_map(nullptr, k->topo_owner, k->topo_row, c->QT[i], ecap, Kg, i, V);
            free_shard(c, dev);
            if (F > 0 && ncap > 0) {
                for (int q = 0; q < c->nchunks[i]; q++) {
                    const int32_t row0 = q * rpc, rows = std::min(rpc, ncap - row0);
                    float* chunk = nullptr;
                    HIP_CHECK(hipMalloc(&chunk, (size_t)rows * pitch * sizeof(float)));
                    if (pitch != F) HIP_CHECK(hipMemsetAsync(chunk, 0, (size_t)rows * pitch * sizeof(float), nullptr));
                    launch_feat_fill_up(nullptr, row0, rows, F, pitch, noder->float_attr_pitch, chunk, noder->float_attrs, c->QF[i], Kg, j, V);
                    LEGION_AUDIT_OWNER(chunk, dev, "FillUp: feature shard chunk");
                    c->shard_chunks[dev].push_back(chunk);
                }
                c->float_feature_cache[dev] = c->shard_chunks[dev][0];
            }
            HIP_CHECK(hipDeviceSynchronize());
        }
        if (clique_home(i, Kg) >= 0) publish_shard_tables(c, i);
    }
    log_out() << "Finish load feature cache\n";
    for (int i = 0; i < c->Kc; i++)
        if (c->QT[i]) GPUGraphStorage_GraphCache(graph, c->QT[i], i, Kg, c->edge_capacity[i]);
    log_out() << "Finish load topology cache\n";
}

float* GPUCache_Float_Feature_Cache(const GPUCache* c, int32_t dev_id)
{
    return (dev_id >= 0 && dev_id < c->device_count) ? c->float_feature_cache[dev_id] : nullptr;
}
int32_t GPUCache_ShardChunkCount(const GPUCache* c, int32_t dev_id)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count || c->nchunks.empty()) return 0;
    return c->nchunks[dev_id / c->Kg];
}
int32_t GPUCache_ShardChunkRows(const GPUCache* c, int32_t dev_id)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count || c->chunk_shift.empty()) return 0;
    return 1 << c->chunk_shift[dev_id / c->Kg];
}
int32_t GPUCache_ShardPitch(const GPUCache* c) { return c ? (c->shard_pitch > 0 ? c->shard_pitch : c->float_attr_len) : 0; }
float* GPUCache_GetShardChunk(const GPUCache* c, int32_t dev_id, int32_t chunk)
{
    if (!c || dev_id < 0 || dev_id >= c->device_count || chunk < 0 || chunk >= (int)c->shard_chunks[dev_id].size()) return nullptr;
    return c->shard_chunks[dev_id][chunk];
}
int GPUCache_ExportFeatureShardChunk(GPUCache* c, int32_t dev_id, int32_t chunk, void* handle64)
{
    if (!c || !handle64 || dev_id < 0 || dev_id >= c->device_count || c->cache_imported[dev_id] || chunk < 0 ||
        chunk >= (int)c->shard_chunks[dev_id].size()) { LEGION_ARG_ERROR("ExportFeatureShardChunk: no such local chunk"); return -1; }
    DeviceGuard guard(dev_id);
    if (!ipc_export_ok(c->shard_chunks[dev_id][chunk], "ExportFeatureShardChunk")) return -1;
    HIP_CHECK(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, c->shard_chunks[dev_id][chunk]));
    return error_pending() ? -1 : 0;
}
int GPUCache_ImportFeatureShardChunk(GPUCache* c, int32_t dev_id, int32_t chunk, const void* handle64)
{
    if (!c || !handle64 || dev_id < 0 || dev_id >= c->device_count || !is_remote_device(dev_id) || c->nchunks.empty() || chunk < 0 ||
        chunk >= c->nchunks[dev_id / c->Kg]) { LEGION_ARG_ERROR("ImportFeatureShardChunk: dev_id must be a remote member, chunk in range"); return -1; }
    {   // what the exporter allocated for this chunk: 2^shift rows (the last chunk may be shorter)
        const int Ki = dev_id / c->Kg;
        const int64_t rows = std::min<int64_t>(1ll << c->chunk_shift[Ki], std::max<int64_t>(1, c->node_capacity[Ki]));
        if (c->shard_pitch <= 0) c->shard_pitch = legion_row_pitch(c->float_attr_len);
        if (!ipc_size_ok(rows * c->shard_pitch * (int64_t)sizeof(float), "ImportFeatureShardChunk")) return -1;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* p = nullptr;
    {
        const int home = clique_home(dev_id / c->Kg, c->Kg);
        DeviceGuard guard(home >= 0 ? home : dev_id);
        HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    }
    if (!p) return -1;
    auto& v = c->shard_chunks[dev_id];
    if ((int)v.size() <= chunk) v.resize(chunk + 1, nullptr);
    v[chunk] = (float*)p;
    c->cache_imported[dev_id] = true;
    if (chunk == 0) c->float_feature_cache[dev_id] = (float*)p;
    publish_shard_tables(c, dev_id / c->Kg);
    return 0;
}
// What an importer has to agree with before it opens a peer's shard: it addresses the peer's rows as
// chunk[r >> shift] + (r & mask) * pitch with ITS OWN pitch / chunk geometry (c->shard_pitch, chunk_shift, nchunks), each process having
// derived them from its own alpha and capacity.  They travel next to the exported handles; a disagreement would read the peer's rows at the
// wrong stride -- silently -- so it is refused.
int GPUCache_ShardGeometry(const GPUCache* c, int32_t dev_id, int32_t out4[4])
{
    if (!c || !out4 || dev_id < 0 || dev_id >= c->device_count || c->nchunks.empty()) { LEGION_ARG_ERROR("ShardGeometry: no shard geometry yet (run FillUp)"); return -1; }
    const int Ki = dev_id / c->Kg;
    out4[0] = GPUCache_ShardPitch(c);
    out4[1] = 1 << c->chunk_shift[Ki];
    out4[2] = c->nchunks[Ki];
    out4[3] = c->node_capacity[Ki];
    return 0;
}
int GPUCache_CheckShardGeometry(GPUCache* c, int32_t owner_dev, const int32_t geom4[4])
{
    int32_t mine[4];
    if (!geom4 || GPUCache_ShardGeometry(c, owner_dev, mine) != 0) { LEGION_ARG_ERROR("CheckShardGeometry: bad arguments"); return -1; }
    if (mine[0] == geom4[0] && mine[1] == geom4[1] && mine[2] == geom4[2] && mine[3] == geom4[3]) return 0;
    char msg[320];
    snprintf(msg, sizeof(msg), "CheckShardGeometry: the shard of clique member %d was built with (pitch %d floats, %d rows per chunk, %d chunks, %d rows), "
             "this process derived (%d, %d, %d, %d): refusing to import it (its rows would be read at the wrong stride)",
             owner_dev, geom4[0], geom4[1], geom4[2], geom4[3], mine[0], mine[1], mine[2], mine[3]);
    LEGION_ARG_ERROR(msg);
    return -1;
}
int GPUCache_ExportFeatureShard(GPUCache* c, int32_t dev_id, void* handle64)
{   // single-chunk shards only; chunked shards use the *Chunk calls
    if (GPUCache_ShardChunkCount(c, dev_id) != 1) { LEGION_ARG_ERROR("ExportFeatureShard: shard has several chunks, use ExportFeatureShardChunk"); return -1; }
    return GPUCache_ExportFeatureShardChunk(c, dev_id, 0, handle64);
}
int GPUCache_ImportFeatureShard(GPUCache* c, int32_t dev_id, const void* handle64)
{
    return GPUCache_ImportFeatureShardChunk(c, dev_id, 0, handle64);
}
uint64_t* GPUCache_GetNodeAccessedMap(const GPUCache* c, int32_t dev_id)
{
    return (dev_id >= 0 && dev_id < c->device_count) ? (uint64_t*)c->ctl[dev_id]->node_access_time : nullptr;
}
uint64_t* GPUCache_GetEdgeAccessedMap(const GPUCache* c, int32_t dev_id)
{
    return (dev_id >= 0 && dev_id < c->device_count) ? (uint64_t*)c->ctl[dev_id]->edge_access_time : nullptr;
}
int32_t* GPUCache_GetFeatureMap(const GPUCache* c, int32_t dev_id)
{
    return (dev_id >= 0 && dev_id < c->device_count) ? c->ctl[dev_id]->feat_map : nullptr;
}
int32_t* GPUCache_GetQF(const GPUCache* c, int32_t Ki) { return (Ki >= 0 && Ki < (int)c->QF.size()) ? c->QF[Ki] : nullptr; }
int32_t* GPUCache_GetQT(const GPUCache* c, int32_t Ki) { return (Ki >= 0 && Ki < (int)c->QT.size()) ? c->QT[Ki] : nullptr; }
int32_t GPUCache_Kg(const GPUCache* c) { return c->Kg; }
int32_t GPUCache_Kc(const GPUCache* c) { return c->Kc; }
double GPUCache_Alpha(const GPUCache* c, int32_t Ki) { return (Ki >= 0 && Ki < (int)c->alpha.size()) ? c->alpha[Ki] : -1.0; }

void GPUCache_Delete(GPUCache* c)
{
    if (!c) return;
    for (int i = 0; i < c->device_count; i++) {
        GPUCache_Finalize(c, i);
        delete c->ctl[i];
    }
    for (size_t i = 0; i < c->QF.size(); i++) {
        if (!c->QF[i]) continue;
        DeviceGuard guard(clique_home((int)i, c->Kg));
        if (c->QF[i]) (void)hipFree(c->QF[i]);
        if (c->QT[i]) (void)hipFree(c->QT[i]);
        if (c->AF[i]) (void)hipFree(c->AF[i]);
        if (c->AT[i]) (void)hipFree(c->AT[i]);
    }
    delete c;
}

} // extern "C"
