// internal.h -- shared declarations of liblegion_amd (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/legion_amd.h"
#include "audit.h"

namespace legion {

// ---- error handling (reference: cudaCheckError(), Kernels.cuh:14-22) -------------------------
void report_error(const char* file, int line, const char* msg, bool hip_failure);
bool error_pending();
bool error_is_fatal();                                     // LEGION_ERR_EXIT mode (the reference's behaviour)
int64_t ipc_max_bytes();                                   // $LEGION_IPC_MAX_BYTES, see runtime.cpp
bool ipc_size_ok(int64_t bytes, const char* who);          // sticky error + false above the limit
bool ipc_export_ok(const void* ptr, const char* who);      // same, for the allocation `ptr` belongs to
inline void check(hipError_t e, const char* file, int line)
{
    if (e != hipSuccess) report_error(file, line, hipGetErrorString(e), true);
}
#define HIP_CHECK(expr) ::legion::check((expr), __FILE__, __LINE__)
#define HIP_CHECK_LAST() ::legion::check(hipGetLastError(), __FILE__, __LINE__)
#define LEGION_ARG_ERROR(msg) ::legion::report_error(__FILE__, __LINE__, (msg), false)

// Where the library's progress prints go: stdout like the reference's (std::cout / printf all over Server.cu, GPUCache.cu,
// CUDA_IPC_Service.cu), or stderr under $LEGION_LOG=stderr (read once) -- for a host that owns stdout (bench.py's one JSON line).
std::ostream& log_out();
FILE* log_file();

int physical_device(int logical);
bool is_remote_device(int logical); // logical GPU driven by another process (one process per GPU)
// RAII: switch to a logical GPU (its physical device + the thread's logical current device), restore both on scope exit
struct DeviceGuard {
    int prev = -1, prev_logical = -1;
    explicit DeviceGuard(int logical);
    ~DeviceGuard();
};

// ---- constants --------------------------------------------------------------------------------
// Position-table entry: (epoch << kPosShift) | value.  epoch = kEpochTop - batch serial: entries of older batches compare GREATER than
// anything of the running batch (stale without being touched); value = kProvisional | slot idx while claimed in the running hop, else
// the node's final index in sampled_ids.  u64 entries, 32-bit epoch (never wraps in practice).  (A u32 entry -- 7-bit epoch, flag, 24-bit
// value, wiped every 126 batches -- was bit-identical and lost at two of three shapes: profiles/r05_sampler.md, r06_removed_experiments.patch.)
typedef unsigned long long pos_t;
constexpr int kPosShift = 32;
constexpr uint32_t kProvisional = 0x80000000u, kPosValueMask = 0x7FFFFFFFu;
constexpr uint32_t kEpochTop = 0xFFFFFFFFu, kSerialLimit = 0xFFFFFFF0u;
__host__ __device__ inline pos_t pos_entry(uint32_t epoch, uint32_t value) { return ((pos_t)epoch << kPosShift) | (pos_t)value; }
#ifndef LEGION_KTILE
#define LEGION_KTILE 1024
#endif
constexpr int kTile = LEGION_KTILE;            // sampler slots per workgroup tile
constexpr int kBlock = 256;                    // threads per workgroup
static_assert(kTile >= kBlock && kTile <= 2048 && (kTile & (kTile - 1)) == 0, "LEGION_KTILE: a power of two in [256, 2048] (k_sample stages 16 bytes of row descriptor per slot in static LDS)");
// k_mark runs one CONTIGUOUS chunk of tiles per workgroup and leaves the chunk totals in GPUMemoryPool::chunk_tot; k_write scans them in
// LDS, kMaxChunks / kBlock per thread.  One constant for the allocation (storage.cpp), the grid clamp (launch_sample_hop) and the LDS
// array (k_write): changing one of them alone would let k_mark write past the allocation.
constexpr int kMaxChunks = 2048;
static_assert(kMaxChunks % kBlock == 0 && kMaxChunks >= 256 * 8, "kMaxChunks: a multiple of the workgroup size, >= 256 CUs x 8 workgroups");
constexpr int ilog2_c(int v) { return v <= 1 ? 0 : 1 + ilog2_c(v >> 1); }
constexpr int kMaxParts = LEGION_MAX_DEVICE;

// minstd_rand arithmetic (thrust::minstd_rand: x <- 48271 x mod 2^31-1), shared by the sampler and the generators
constexpr uint32_t kP31 = 2147483647u; // minstd modulus 2^31 - 1
constexpr uint32_t kA = 48271u;        // minstd multiplier

__host__ __device__ inline uint32_t mulmod31(uint32_t a, uint32_t b)
{
    uint64_t p = (uint64_t)a * (uint64_t)b;
    uint32_t r = (uint32_t)(p & kP31) + (uint32_t)(p >> 31); // < 2^32
    r = (r & kP31) + (r >> 31);
    return r >= kP31 ? r - kP31 : r;
}

__host__ __device__ inline uint32_t powmod31(uint32_t base, uint64_t e)
{
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = mulmod31(r, base);
        base = mulmod31(base, base);
        e >>= 1;
    }
    return r;
}

// unsigned division by a runtime constant (host precomputed): q = (n * m) >> 32 >> s, n < 2^31
struct FastDiv {
    uint32_t d = 1, m = 0, s = 0;
    FastDiv() = default;
    explicit FastDiv(uint32_t div);
};

// ---- kernel launch API (kernels.hip) -------------------------------------------------------------
struct CsrTables {                 // GPU_Memory_Graph_Storage.cu:45-133: the whole CSR + the clique's fragments
    const int64_t* indptr;         // whole CSR (the reference's slot [P]): HBM replica or pinned host table
    const int32_t* indices;
    // fragments are lists of chunk allocations (see GPUGraphStorage): device-side pointer tables, part-major
    const int64_t* const* frag_indptr;  // [P * ip_nch]; chunk q of part p holds indptr entries [q<<row_shift, ((q+1)<<row_shift)]
    const int32_t* const* frag_indices; // [P * ix_nch]; chunk q holds the rows whose first edge lies in [q<<edge_shift, (q+1)<<edge_shift)
    int32_t ip_nch, ix_nch, row_shift, edge_shift;
    int32_t partition_count;
    const int8_t* topo_owner;      // int8[V]  owner logical GPU or -1 (edge_index_map), may be null
    const int32_t* topo_row;       // int32[V] row in the owner's fragment (edge_offset_map)
};

struct BatchCtl {                  // device-resident batch cursor (see k_seed)
    int32_t counter;               // batch index inside the seed list
    uint32_t epoch;                // position-table epoch of the running batch
};

struct HopState {                  // written by the scan kernel, read by the write/resolve kernels
    int32_t edge_base, node_base, n_edges, n_nodes, in_off, n_in, slots, pad;
};

struct SamplerBuffers {
    int32_t* sampled_ids;   // IPC buffer 0
    int32_t* agg_src_ids;   // per-edge neighbour id == next hop's input list
    int32_t* agg_src_off;   // IPC buffer 3
    int32_t* agg_dst_off;   // IPC buffer 4
    int32_t* nc;            // IPC buffer 5
    int32_t* ec;            // IPC buffer 6
    pos_t* pos_map;         // pos_t[V]: (epoch << kPosShift) | value
    const BatchCtl* ctl;    // ctl->epoch = 0xFFFFFFFF - batch serial: newer batches compare smaller
    int32_t* cand;          // i32[max slots of a hop]
    int32_t* aux;           // i32[max slots of a hop], slot state: -1 claim pending / won (k_mark: winner rank), >= 0 known position, <= -2 lost to slot -2-x
    int32_t* aux_next;      // the other buffer: k_write prepares it (-1) for the next hop
    int32_t next_count;     // fan-out of the next hop (0: none)
    int32_t aux_cap;        // elements per aux buffer
    int32_t ids_cap;        // elements of sampled_ids / agg_src_ids / agg_src_off / agg_dst_off (num_ids)
    int32_t V;              // entries of pos_map
    bool aux_prepared;      // aux already holds -1 for nc[2] * count slots
    int32_t* tile_edge;     // i32[max tiles]
    int32_t* tile_node;     // i32[max tiles]
    int2* tile_pre;         // int2[max tiles]: (edges, new nodes) in front of a tile inside its k_mark chunk
    int2* chunk_tot;        // int2[kMaxChunks]: totals of the k_mark chunks
    HopState* hop_state;
    unsigned long long* edge_access_time; // pre-sampling only (may be null)
};

void launch_seed(hipStream_t s, int32_t* batch_ids, int32_t* labels, int32_t batch_size, int32_t size, int32_t counter,
                 const int32_t* all_ids, const int32_t* all_labels, int32_t total_cap, pos_t* pos_map,
                 uint32_t epoch, BatchCtl* ctl, bool self_driven, int32_t* nc, int32_t* ec, int32_t* aux_next,
                 int32_t f_next, int32_t aux_cap);
void launch_set_cursor(hipStream_t s, BatchCtl* ctl, int32_t counter, uint32_t epoch);
void launch_advance(hipStream_t s, BatchCtl* ctl);
void warm_static_tables();   // per-device constant tables: must exist before a stream capture starts
void launch_sample_hop(hipStream_t s, const CsrTables& csr, const SamplerBuffers& b, int32_t count, int32_t op_id,
                       int32_t hops, int32_t slots_bound, bool is_presc);
void launch_find_feat(hipStream_t s, const int32_t* sampled_ids, int32_t* cache_offset, const int32_t* nc,
                      int32_t op_id, const int32_t* feat_map, int32_t bound);
void launch_find_topo(hipStream_t s, const int32_t* input_ids, int8_t* part_index, int32_t* part_offset,
                      int32_t batch_size, const int8_t* topo_owner, const int32_t* topo_row);
struct GatherArgs {
    const float* table;                       // V x F rows: the "cpu_float_attrs" of the reference
    // clique caches (Global_Float_Feature_Cache, the reference's float** cache_float_attrs): device table of
    // Kg x nchunks chunk pointers; shard row r lives in chunk r >> chunk_shift (shards are allocated in
    // chunks so that each piece can be exported over HIP IPC)
    const float* const* shard_tab;
    int32_t chunk_shift, nchunks;
    const int32_t* feat_map;                  // int32[V] global slot or -1; null = no cache
    const float** row_ptr;                    // scratch [rows]: address of each row's source (own shard / peer shard / backing
                                              // table / null), resolved by a lookup pass in front of the gather; null: resolve
                                              // inside the gather (no cache)
    int32_t cache_capacity;                   // rows per GPU
    int32_t F;
    int32_t table_pitch, shard_pitch;         // floats between two rows of the backing table / of a shard chunk (0: F, dense).
                                              // HBM copies we own are laid out with a 128-byte-aligned pitch when F * 4 is not
                                              // a multiple of 128 (F = 100: 512 bytes), so that every row read starts on a line
    int32_t total_num_nodes;
    const int32_t* sampled_ids;
    const int32_t* nc;
    float* dst;
    int32_t off_idx, size_idx;                // nc[] indices of (offset, size); off_idx < 0 => offset 0
    int32_t dst_rows;                         // capacity of dst in rows (<= 0: unbounded)
    int32_t* rows_seen;                       // host-mapped word: the launch leaves its actual row count here (may be null)
    int32_t* hit_stats;                       // host-mapped {hits, rows}: the lookup pass of a SAMPLED batch adds its counts (may be null)
    int32_t rows_hint;                        // row count of an earlier launch of this kind (0: unknown)
    bool table_on_host;                       // the backing table is pinned host memory (misses cross PCIe)
    bool row_ptr_ready;                       // row_ptr was filled by the caller (exchange plan): skip the lookup pass
};
void launch_gather(hipStream_t s, const GatherArgs& a, int32_t rows_bound);
// owner-computes exchange variant of the gather (kernels.hip "S5, owner-computes"): counts = int32[2 * kMaxParts] scratch
void launch_exchange_plan(hipStream_t s, const GatherArgs& g, int32_t me, int32_t Kg, int32_t* slot, int32_t* counts,
                          int32_t* req_row, int32_t* req_dst, int32_t rows_bound);
void launch_exchange_rows(hipStream_t s, bool scatter, const float* const* shard_chunks, int32_t chunk_shift, const int32_t* list,
                          int32_t n, int32_t F, int32_t shard_pitch, const float* in, float* out, int32_t out_rows);
void launch_hotness(hipStream_t s, const int32_t* ids, const int32_t* nc, int32_t hops, unsigned long long* access,
                    int32_t* max_ids, int32_t bound);
void launch_rng_probe(hipStream_t s, const int32_t* idx, const int32_t* deg, int32_t* k, int32_t n);
// cache construction helpers
void launch_aggregate_access(hipStream_t s, unsigned long long* agg, const unsigned long long* add, int32_t n);
void launch_iota(hipStream_t s, int32_t* out, int32_t n);
void launch_build_feat_map(hipStream_t s, int32_t* feat_map, const int32_t* QF, int32_t capacity, int32_t Kg, int32_t V);
void launch_build_topo_map(hipStream_t s, int8_t* owner, int32_t* row, const int32_t* QT, int32_t capacity, int32_t Kg,
                           int32_t Ki, int32_t V);
void launch_fill_i32(hipStream_t s, int32_t* p, int32_t v, int64_t n);
void launch_fill_i8(hipStream_t s, int8_t* p, int8_t v, int64_t n);
void launch_feat_fill_up(hipStream_t s, int32_t row0, int32_t rows, int32_t F, int32_t chunk_pitch, int32_t table_pitch, float* chunk,
                         const float* table, const int32_t* QF, int32_t Kg, int32_t Ki, int32_t V);
void launch_copy_rows_pitched(hipStream_t s, float* dst, int32_t dst_pitch, const float* src, int32_t src_pitch, int32_t F, int64_t rows);
void launch_neighbor_count(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                           const int64_t* indptr, int64_t* count_out);
void launch_topo_fill_up(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                         const int64_t* indptr, const int32_t* indices, const int64_t* frag_indptr,
                         int32_t* const* frag_chunks, int32_t edge_shift);
// ends[q] = end offset of the last row that starts before (q+1) << edge_shift   (q < nch - 1)
void launch_chunk_ends(hipStream_t s, const int64_t* frag_indptr, int32_t capacity, int32_t edge_shift, int32_t nch, int64_t* ends);
void launch_edge_mem(hipStream_t s, const int32_t* order, uint64_t* edge_mem, int32_t V, const int64_t* indptr);
void launch_topo_transactions(hipStream_t s, const int32_t* order, const uint64_t* hot, uint64_t* out, int32_t V, const int64_t* indptr);
void sort_by_hotness_desc(hipStream_t s, unsigned long long* keys, int32_t* ids, int32_t n);
void inclusive_scan_u64(hipStream_t s, const uint64_t* in, uint64_t* out, int32_t n);
void inclusive_scan_i64(hipStream_t s, const int64_t* in, int64_t* out, int32_t n);

} // namespace legion

// ---- the opaque handle types (host mirrors of the reference classes) ------------------------------
struct PeerExchange;               // peer_exchange.cpp: staging of the bulk-copy (hipMemcpyPeerAsync) gather
struct GPUMemoryPool {
    int32_t pipeline_depth = LEGION_PIPELINE_DEPTH;
    int32_t current_pipe = 0, iter = 0, mode = 0, op_id = 0;
    int32_t device_id = -1;           // logical GPU this pool serves (set by batch_generator_kernel)
    // scratch owned by the pool when AllocateScratch() was used
    bool owns_scratch = false;
    int32_t V = 0, batch_size = 0, hops = 0, num_ids = 0;
    int32_t fanout[LEGION_MAX_HOPS] = {0};
    int32_t max_slots = 0, max_tiles = 0;
    int32_t feature_rows = 0;         // capacity of the feature buffers in rows (0 = unbounded)
    legion::pos_t* pos_map = nullptr; // pos_t[V], see kernels.hip "position table"
    uint32_t batch_serial = 0;        // batches started on this pool; epoch = 0xFFFFFFFF - serial
    legion::BatchCtl* ctl = nullptr;  // device copy of (batch cursor, epoch): what the kernels read
    // Feedback for sizing the gather launches without a host round trip: pinned, device-mapped words the gather
    // kernels write their actual row count to ([l] = level-l gather, [LEGION_MAX_HOPS + 1] = all rows of the batch);
    // the host reads whatever an earlier batch left there.
    int32_t* rows_seen = nullptr;      // host view
    int32_t* rows_seen_dev = nullptr;  // device view of the same words
    bool capturing = false;           // between Begin/EndBatchCapture: launchers record a self-driven batch
    bool ctl_synced = false;          // ctl holds (ctl_counter, epoch of the NEXT batch): a batch graph can run as is
    int32_t ctl_counter = 0;
    int32_t* cand = nullptr;
    int32_t* aux2[2] = {nullptr, nullptr}; // slot states, one buffer per hop parity (hop h uses aux2[h & 1])
    int32_t aux_ready_hop = 0, aux_ready_count = 0; // the launch before prepared aux2[hop & 1] for this fan-out
    int32_t* tile_edge = nullptr;
    int32_t* tile_node = nullptr;
    int2* tile_pre = nullptr;
    int2* chunk_tot = nullptr;
    legion::HopState* hop_state = nullptr;
    int32_t* cache_search_buffer = nullptr;
    const float** row_ptr = nullptr;  // [num_ids] row source addresses of the running gather (cached configurations)
    PeerExchange* peer_exchange = nullptr; // created by the first legion_peer_exchange_gather of this pool
    int32_t* agg_src_ids = nullptr;
    int8_t* tmp_part_ind = nullptr;
    int32_t* tmp_part_off = nullptr;
    // per pipe (IPC buffers)
    std::vector<float*> float_features;
    std::vector<int32_t*> labels, node_counter, edge_counter, sampled_ids, agg_src_off, agg_dst_off;
    // host-side launch bounds (no device round trips)
    int32_t bound_n = 0;        // upper bound of the next hop's input count
    int32_t bound_nodes = 0;    // upper bound of nodes discovered so far
    int32_t level_bound[LEGION_MAX_HOPS + 1] = {0};
    explicit GPUMemoryPool(int32_t depth);
};

struct GPUGraphStorage {
    int32_t partition_count = 0;
    int32_t node_num = 0;
    int64_t edge_num = 0, cache_edge_num = 0;
    int64_t* csr_node_index_cpu = nullptr;   // device-visible pointer of the whole CSR (slot [P])
    int32_t* csr_dst_node_ids_cpu = nullptr;
    // HBM replicas of the whole CSR, one per logical GPU that has one (GPUGraphStorage_ReplicateToDevices)
    std::vector<int64_t*> replica_indptr;
    std::vector<int32_t*> replica_indices;
    int32_t csr_location = LEGION_LOC_HOST_PINNED;
    bool owns_csr = false;
    // CSR fragment of one logical GPU (device memory on that GPU's physical device).  Both arrays are lists of
    // chunk allocations (<= $LEGION_SHARD_CHUNK_BYTES, default 1 GiB, like the feature shards) so that every piece
    // can be opened over HIP IPC.  indptr chunk q: entries [q<<row_shift, min(rows, (q+1)<<row_shift)] (one entry of
    // overlap, so ip[r] and ip[r+1] come from the same chunk).  indices chunk q: every row whose first edge offset o
    // satisfies o >> edge_shift == q, whole, at element o & mask (the chunk is as long as its last row needs).
    struct Fragment {
        int32_t rows = 0;
        int64_t edges = 0;
        std::vector<int64_t*> ip;
        std::vector<int32_t*> ix;
        bool imported = false;               // opened from another process' IPC handles
        bool complete = false;               // every chunk present (built locally, or all chunks imported)
    };
    std::vector<Fragment> frag;
    int32_t row_shift = 27, edge_shift = 28;
    // which fragments logical GPU d may read (its clique): view[d][p]
    std::vector<std::vector<bool>> view;
    // device-side chunk-pointer tables per local viewer (P*ip_nch indptr pointers, then P*ix_nch indices pointers)
    std::vector<void**> d_frag_tab;
    int32_t ip_nch = 1, ix_nch = 1;
};

struct GPUNodeStorage {
    int32_t partition_count = 0, total_num_nodes = 0, float_attr_len = 0;
    float* float_attrs = nullptr;     // device-visible V x F table
    int32_t float_attr_pitch = 0;     // floats between two rows of float_attrs (>= float_attr_len)
    std::vector<float*> replica_attrs; // HBM replicas per logical GPU (GPUNodeStorage_ReplicateToDevices)
    int32_t replica_pitch = 0;        // ... of the replicas (legion_row_pitch)
    int32_t features_location = LEGION_LOC_HOST_PINNED;
    bool owns_features = false;
    std::vector<int32_t> training_set_num, validation_set_num, testing_set_num;
    std::vector<int32_t*> training_set_ids, validation_set_ids, testing_set_ids;
    std::vector<int32_t*> training_labels, validation_labels, testing_labels;
};

struct CacheController {              // PreSCCacheController, GPUCache.cu:239-500
    int32_t device_idx = 0, device_count = 1, total_num_nodes = 0, train_step = 0;
    unsigned long long* node_access_time = nullptr;
    unsigned long long* edge_access_time = nullptr;
    int32_t iter = 0, max_ids = 0;
    int32_t* d_max_ids = nullptr;     // device-side running max of nc[total]
    int32_t node_capacity = 0, edge_capacity = 0;
    // direct-mapped replacements of the three BGHT maps (GPUCache.cu:315-321)
    int32_t* feat_map = nullptr;      // node_map_:       id -> global cache slot | -1
    int8_t* topo_owner = nullptr;     // edge_index_map_: id -> owner logical GPU | -1
    int32_t* topo_row = nullptr;      // edge_offset_map_: id -> row in owner's fragment | -1
    int32_t* d_global_count = nullptr;
    // Feature-cache hit rate (GPUCache.cu:130-147,414-425: counted every 500th batch, printed at the last level).
    // Two pinned, device-mapped {hits, rows} slots: sampling k counts into slot k % 2 while the host prints what
    // sampling k - 1 left in the other one -- no device-to-host copy, no synchronisation.
    int32_t find_iter = 0;
    int32_t* hit_stats = nullptr;      // host view, 2 x {hits, rows}
    int32_t* hit_stats_dev = nullptr;  // device view
    hipEvent_t hit_ev[2] = {nullptr, nullptr}; // recorded behind the last counting launch of the sampling that used slot k
    bool hit_ev_armed[2] = {false, false};
    int32_t hit_samples = 0;           // samplings started
    double last_hit_rate = -1.0;       // what the last print showed
};

struct GPUCache {
    int32_t device_count = 0;
    std::vector<CacheController*> ctl;
    std::vector<int32_t*> QF, QT;                    // per clique, on the clique's first GPU
    std::vector<unsigned long long*> AF, AT;
    int Kc = 1, Kg = 1;
    std::vector<int32_t> node_capacity, edge_capacity;   // per clique
    std::vector<double> alpha;
    int64_t cache_memory = 0;
    int32_t int_attr_len = 0, float_attr_len = 0, train_step = 0;
    std::vector<float*> float_feature_cache;         // per logical GPU
    std::vector<bool> cache_imported;                // shard opened from another process' IPC handle
    std::vector<std::vector<float*>> shard_chunks;   // per logical GPU: the shard's chunk allocations
    std::vector<float**> d_shard_tab;                // per LOCAL logical GPU: device table [Kg x nchunks]
    std::vector<int32_t> chunk_shift, nchunks;       // per clique
    int32_t shard_pitch = 0;                         // floats between two rows of a shard chunk (legion_row_pitch(F))
    bool is_presc = true;
    bool capacity_forced = false;
    int32_t forced_node_capacity = 0, forced_edge_capacity = 0;
};
