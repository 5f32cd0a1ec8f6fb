/*
 * legion_amd.h -- C ABI of liblegion_amd.so, the MI355X-native (HIP, gfx950) implementation
 * of Legion's GPU-initiated mini-batch pipeline.
 *
 * The library is a drop-in for the reference's sampling-server hot path: every entry point
 * below replaces one interface of liayan/Legion-1 and cites it (paths relative to the
 * reference tree).  Where the reference already exports a C-linkage symbol
 * (src/Kernels.cuh:24-93, GPU_Graph_Storage.cuh:38-39, GPU_Node_Storage.cuh:60-61,
 * CUDA_IPC_Service.h:35) the SAME NAME and argument order are kept; its C++ class pointers
 * become opaque struct handles, and the C++ virtual methods a caller needs become
 * `Class_Method(handle, ...)` functions.  No torch / C++ types cross this boundary: plain
 * pointers, sizes and ints only.  Streams are `void*` (a hipStream_t).
 *
 * Error behaviour (reference: cudaCheckError() -> printf + exit(EXIT_FAILURE),
 * src/Kernels.cuh:14-22): by default a HIP failure prints "Hip failure <file>:<line>: '<msg>'"
 * and exits, exactly like the reference.  legion_set_error_mode(LEGION_ERR_RETURN) turns
 * that into a sticky per-thread error readable with legion_last_error() (for embedding /
 * tests).  Argument errors always take the sticky path.
 */
#ifndef LEGION_AMD_H
#define LEGION_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LEGION_MAX_DEVICE 8      /* CUDA_IPC_Service.cu:14 MAX_DEVICE */
#define LEGION_PIPELINE_DEPTH 2  /* Server.cu:15, CUDA_IPC_Service.cu:15 */
#define LEGION_MEMORY_USAGE 7    /* CUDA_IPC_Service.cu:16: buffers per (device, pipe) */
#define LEGION_MAX_HOPS 5        /* counter layout fits int32[16] up to 5 hops */

#define LEGION_TRAINMODE 0       /* Kernels.cu:10-12 */
#define LEGION_VALIDMODE 1
#define LEGION_TESTMODE 2

/* ---- opaque handles (reference C++ classes) -------------------------------------------- */
typedef struct GPUGraphStorage GPUGraphStorage; /* GPU_Graph_Storage.cuh:20-37 */
typedef struct GPUNodeStorage GPUNodeStorage;   /* GPU_Node_Storage.cuh:24-58 */
typedef struct GPUCache GPUCache;               /* GPUCache.cuh:64-161 */
typedef struct GPUMemoryPool GPUMemoryPool;     /* GPUMemoryPool.cuh:7-208 */
typedef struct IPCEnv IPCEnv;                   /* CUDA_IPC_Service.h:6-33 */
typedef struct Operator Operator;               /* Operator.h:18-21 */
typedef struct Runner Runner;                   /* Server.h:157-165 */
typedef struct Server Server;                   /* Server.h:148-155 */
typedef struct LegionBatchGraph LegionBatchGraph; /* one recorded mini-batch (hipGraph); no reference counterpart */

/* ---- library control --------------------------------------------------------------------- */
#define LEGION_ERR_EXIT 0   /* reference behaviour */
#define LEGION_ERR_RETURN 1
const char* legion_version(void);
void legion_set_error_mode(int mode);
const char* legion_last_error(void); /* "" when none; cleared by legion_clear_error */
void legion_clear_error(void);
/* Map a logical device id (the reference's dev_id / partition id) onto a physical HIP device.
 * Default: physical = logical % hipGetDeviceCount.  Lets a clique of Kg logical GPUs be
 * exercised on fewer physical devices. */
void legion_set_device_map(int32_t logical_dev, int32_t physical_dev);
int32_t legion_physical_device(int32_t logical_dev);
/* One process per GPU: mark the logical GPUs of a clique that another process drives.  Their
 * controllers hold nothing here; their cache shards / CSR fragments are imported over HIP IPC
 * (GPUCache_ImportFeatureShard, GPUGraphStorage_ImportFragment) and then read in-kernel over xGMI,
 * exactly like the reference's NVLink peer loads (Kernels.cu:395-409,698). */
void legion_set_remote_device(int32_t logical_dev, int is_remote);
int legion_is_remote_device(int32_t logical_dev);
/* Logical-device audit ($LEGION_DEVICE_AUDIT=1, read when the library loads; csrc/audit.h has the rules).  The reference runs one
 * thread per GPU under cudaSetDevice(own) and lets the clique's GPUs read each other's shards through peer access enabled all-pairs
 * (Server.cu:87-95,119-127, GPUGraphStore.cu:145-168, GPU_Memory_Graph_Storage.cu:98-133, GPUCache.cu:769-826).  With several
 * logical GPUs mapped onto ONE physical device (legion_set_device_map) a stream, event, allocation or launch made under the wrong
 * device still works; under the audit every such resource carries the logical GPU it was created under and every launch, copy,
 * event record, stream wait, graph launch and pointer table is checked against the logical GPU current on the calling thread
 * (SetGPUDevice; the library's own scopes).  A violation is a sticky error (legion_last_error) naming the call site.
 * counts = {checks, violations, resources / launches that could not be attributed (no logical GPU selected, foreign memory),
 * kernel launches with a table argument in a peer's memory}.  legion_audit_report prints one summary line (+ the first violations)
 * to the library's log and returns the number of violations. */
int legion_audit_enabled(void);
void legion_audit_counts(int64_t counts[4]);
int32_t legion_audit_message_count(void);
const char* legion_audit_message(int32_t i);
void legion_audit_reset(void);
int64_t legion_audit_report(void);
/* one process per GPU: the process's only device is logical GPU 0 of a replicated engine and logical GPU <rank> of a clique engine --
 * a and b (mapped to the same physical device) name the same device in THIS process; returns 0, -1 if they do not */
int legion_audit_alias(int32_t a, int32_t b);

/* ---- raw device helpers: src/Kernels.cuh:24-45 (same names) ------------------------------ */
void* d_alloc_space(int64_t num_bytes);
void* d_alloc_space_managed(unsigned int num_bytes);
void d_copy_2_h(void* h_ptr, void* d_ptr, unsigned int num_bytes);
void d_free_space(void* d_ptr);
void SetGPUDevice(int32_t shard_id);
int32_t GetGPUDevice(void);
void* host_alloc_space(unsigned int num_bytes);   /* pinned + mapped; returns the device alias */
void* host_alloc_space64(int64_t num_bytes);      /* same without the reference's 4 GiB limit */
void host_free_space(void* ptr);
void d_copy_h_2_d(void* d_ptr, const void* h_ptr, int64_t num_bytes);
void d_copy_d_2_h(void* h_ptr, const void* d_ptr, int64_t num_bytes);
void d_stream_sync(void* stream);
void* d_stream_create(void);
/* stream at the greatest (high != 0) or least priority of the device's range (hipStreamCreateWithPriority) */
void* d_stream_create_priority(int32_t high);
/* stream restricted to the compute units whose bit is set in cu_mask[words] (hipExtStreamCreateWithCUMask):
 * lets the two streams of the overlapped schedule own disjoint CU sets */
void* d_stream_create_cu_mask(const uint32_t* cu_mask, int32_t words);
void d_stream_destroy(void* stream);
void d_copy_async(void* dst, const void* src, int64_t num_bytes, void* stream); /* hipMemcpyDefault */
void d_memset_async(void* dst, int value, int64_t num_bytes, void* stream);
void* d_event_create(void);                       /* timing-enabled hipEvent_t */
void d_event_destroy(void* event);
void d_event_record(void* event, void* stream);
void d_stream_wait_event(void* stream, void* event); /* hipStreamWaitEvent */
float d_event_elapsed_ms(void* start, void* stop); /* synchronises on `stop` */

/* ---- BuildInfo: src/BuildInfo.h:5-70 (the fields the hot path uses; BaM/SSD fields dropped) */
typedef struct LegionBuildInfo {
    int32_t partition_count;                         /* == shard count == #logical GPUs */
    /* per-partition seed sets (host pointers, arrays of partition_count entries) */
    const int32_t* training_set_num;   const int32_t* const* training_set_ids;   const int32_t* const* training_labels;
    const int32_t* validation_set_num; const int32_t* const* validation_set_ids; const int32_t* const* validation_labels;
    const int32_t* testing_set_num;    const int32_t* const* testing_set_ids;    const int32_t* const* testing_labels;
    /* features */
    int32_t total_num_nodes;
    int32_t float_attr_len;
    float* host_float_attrs;      /* V*F floats.  features_location says where they live */
    int32_t features_location;    /* LEGION_LOC_* */
    /* CSR */
    int64_t* csr_node_index;      /* int64[V+1] */
    int32_t* csr_dst_node_ids;    /* int32[E]   */
    int32_t csr_location;         /* LEGION_LOC_* */
    int64_t total_edge_num;
    int64_t cache_edge_num;
    /* train */
    int32_t epoch;
    int32_t raw_batch_size;
    /* extension (after the reference's fields): floats between two rows of host_float_attrs; 0 = F (dense, the reference's
     * file layout).  A caller that builds the table itself may pad rows to legion_row_pitch(F).  Must be 0 or >= float_attr_len,
     * and a multiple of 4 floats whenever float_attr_len is (16-byte row alignment); Build() refuses anything else.  ZERO-INITIALISE
     * the struct: a caller compiled against the reference's shorter BuildInfo would pass an uninitialised word here. */
    int32_t float_attr_pitch;
} LegionBuildInfo;
/* Row pitch (in floats) the library gives the HBM copies it owns (table replicas, cache shards): F when a row is a whole
 * number of 128-byte lines, else F rounded up to 32 floats (F = 100 -> 128: every 400-byte row read then starts on a line).
 * The trainer-facing feature buffer stays dense [n, F]. */
int32_t legion_row_pitch(int32_t F);
/* Row pitch of the feature-cache shards: legion_row_pitch(F) when `rows` padded rows fit `feat_budget_bytes` (the feature
 * share of the reference's cache_memory contract, GPUCache.cu:674,727: capacity = budget / (F * 4)), else F (dense);
 * feat_budget_bytes <= 0: no budget known. */
int32_t legion_shard_pitch(int32_t F, int64_t rows, int64_t feat_budget_bytes);

/* Where a table handed to Build() lives.  The reference always uses pinned host memory read
 * by the GPUs through UVA (GPUGraphStore.cu:264-265,315).  On 288 GB parts the whole table
 * usually fits in HBM, so device-resident tables are first class here. */
#define LEGION_LOC_HOST_PINNED 0  /* pointer from host_alloc_space*: device-visible host memory */
#define LEGION_LOC_DEVICE 1       /* pointer is device memory on every logical GPU's physical device */
#define LEGION_LOC_HOST_PAGEABLE 2 /* plain host pointer: Build() copies it to a pinned mapping */

/* ---- graph storage: GPU_Graph_Storage.cuh:20-39, GPU_Memory_Graph_Storage.cu:45-133 ------- */
GPUGraphStorage* NewGPUMemoryGraphStorage(void);
void GPUGraphStorage_Build(GPUGraphStorage* g, const LegionBuildInfo* info);
void GPUGraphStorage_GraphCache(GPUGraphStorage* g, int32_t* QT, int32_t Ki, int32_t Kg, int32_t capacity);
void GPUGraphStorage_Finalize(GPUGraphStorage* g);
/* MI355X-first: copy the whole CSR into the HBM of every local GPU (one replica per physical device); the
 * sampler then reads its own replica instead of the pinned-host table.  Returns bytes per replica (0: nothing done). */
int64_t GPUGraphStorage_ReplicateToDevices(GPUGraphStorage* g);
int32_t GPUGraphStorage_GetPartitionCount(const GPUGraphStorage* g);
int64_t* GPUGraphStorage_GetCSRNodeIndexCPU(const GPUGraphStorage* g);
int32_t* GPUGraphStorage_GetCSRNodeMatrixCPU(const GPUGraphStorage* g);
/* fragment of logical GPU part_id as seen from dev_id (NULL when not cached): FIRST chunk of each array, i.e. the
 * whole array for fragments below the chunk size (GPU_Memory_Graph_Storage.cu:128-131 d_csr_node_index_/d_csr_dst_node_ids_) */
int64_t* GPUGraphStorage_GetFragmentIndex(const GPUGraphStorage* g, int32_t dev_id, int32_t part_id);
int32_t* GPUGraphStorage_GetFragmentMatrix(const GPUGraphStorage* g, int32_t dev_id, int32_t part_id);
/* A fragment is two lists of chunk allocations (<= $LEGION_SHARD_CHUNK_BYTES, default 1 GiB) so that each piece can be
 * opened over HIP IPC.  which = 0: indptr, chunk q = entries [q*span, min(rows, (q+1)*span)] (one entry of overlap);
 * which = 1: indices, chunk q = all rows whose first edge offset o has o / span == q, whole, at element o % span. */
int32_t GPUGraphStorage_FragmentRows(const GPUGraphStorage* g, int32_t dev_id);
int64_t GPUGraphStorage_FragmentEdges(const GPUGraphStorage* g, int32_t dev_id);
int32_t GPUGraphStorage_FragmentChunkCount(const GPUGraphStorage* g, int32_t dev_id, int32_t which);
int64_t GPUGraphStorage_FragmentChunkSpan(const GPUGraphStorage* g, int32_t which);
void* GPUGraphStorage_GetFragmentChunk(const GPUGraphStorage* g, int32_t dev_id, int32_t which, int32_t chunk);
int GPUGraphStorage_ExportFragmentChunk(GPUGraphStorage* g, int32_t dev_id, int32_t which, int32_t chunk, void* handle64);
int GPUGraphStorage_ImportFragmentChunk(GPUGraphStorage* g, int32_t owner_dev, int32_t viewer_dev, int32_t which, int32_t chunk,
                                        const void* handle64, int32_t rows, int64_t edges);
/* HIP-IPC exchange of a single-chunk fragment (64-byte handles); returns 0 on success */
int GPUGraphStorage_ExportFragment(GPUGraphStorage* g, int32_t dev_id, void* handle_indptr64, void* handle_indices64, int32_t* rows_out);
int GPUGraphStorage_ImportFragment(GPUGraphStorage* g, int32_t owner_dev, int32_t viewer_dev, const void* handle_indptr64,
                                   const void* handle_indices64, int32_t rows);
void GPUGraphStorage_Delete(GPUGraphStorage* g);

/* ---- node storage: GPU_Node_Storage.cuh:24-61, GPU_Memory_Node_Storage.cu:11-161 ---------- */
GPUNodeStorage* NewGPUMemoryNodeStorage(void);
void GPUNodeStorage_Build(GPUNodeStorage* n, const LegionBuildInfo* info);
void GPUNodeStorage_Finalize(GPUNodeStorage* n);
int32_t* GPUNodeStorage_GetTrainingSetIds(const GPUNodeStorage* n, int32_t part_id);
int32_t* GPUNodeStorage_GetValidationSetIds(const GPUNodeStorage* n, int32_t part_id);
int32_t* GPUNodeStorage_GetTestingSetIds(const GPUNodeStorage* n, int32_t part_id);
int32_t* GPUNodeStorage_GetTrainingLabels(const GPUNodeStorage* n, int32_t part_id);
int32_t* GPUNodeStorage_GetValidationLabels(const GPUNodeStorage* n, int32_t part_id);
int32_t* GPUNodeStorage_GetTestingLabels(const GPUNodeStorage* n, int32_t part_id);
int32_t GPUNodeStorage_TrainingSetSize(const GPUNodeStorage* n, int32_t part_id);
int32_t GPUNodeStorage_ValidationSetSize(const GPUNodeStorage* n, int32_t part_id);
int32_t GPUNodeStorage_TestingSetSize(const GPUNodeStorage* n, int32_t part_id);
/* same for the V x F feature table (the backing table of cache misses) */
int64_t GPUNodeStorage_ReplicateToDevices(GPUNodeStorage* n);
int32_t GPUNodeStorage_TotalNodeNum(const GPUNodeStorage* n);
float* GPUNodeStorage_GetAllFloatAttr(const GPUNodeStorage* n);
int32_t GPUNodeStorage_GetFloatAttrLen(const GPUNodeStorage* n);
void GPUNodeStorage_Delete(GPUNodeStorage* n);

/* ---- memory pool: GPUMemoryPool.cuh:7-208 ------------------------------------------------- */
/* The reference pool is a bag of raw pointers set by GPURunner::Initialize (Server.cu:216-247).
 * Here the pool can allocate its own scratch for a given (V, batch, fan-outs) and records the
 * static upper bounds that size the launches (no device->host round trips in the hot path). */
GPUMemoryPool* NewGPUMemoryPool(int32_t pipeline_depth);
/* Allocates scratch on the current device: dedup/position table u32[V], candidate buffer,
 * tile counters, cache_search_buffer, agg_src_ids; hops = #entries of fanout. */
void GPUMemoryPool_AllocateScratch(GPUMemoryPool* p, int32_t total_num_nodes, int32_t batch_size,
                                   const int32_t* fanout, int32_t hops);
int32_t GPUMemoryPool_NumIds(const GPUMemoryPool* p);   /* B*(1+f0+f0*f1+...), Server.cu:184-196 */
void GPUMemoryPool_SetSampledIds(GPUMemoryPool* p, int32_t* ptr, int32_t pipe);
void GPUMemoryPool_SetFloatFeatures(GPUMemoryPool* p, float* ptr, int32_t pipe);
void GPUMemoryPool_SetLabels(GPUMemoryPool* p, int32_t* ptr, int32_t pipe);
void GPUMemoryPool_SetAggSrcOf(GPUMemoryPool* p, int32_t* ptr, int32_t pipe);
void GPUMemoryPool_SetAggDstOf(GPUMemoryPool* p, int32_t* ptr, int32_t pipe);
void GPUMemoryPool_SetNodeCounter(GPUMemoryPool* p, int32_t* ptr, int32_t pipe);
void GPUMemoryPool_SetEdgeCounter(GPUMemoryPool* p, int32_t* ptr, int32_t pipe);
/* rows the feature buffers can hold (0 = unknown/unbounded); the gather never writes past it */
void GPUMemoryPool_SetFeatureRows(GPUMemoryPool* p, int32_t rows);
void GPUMemoryPool_SetCurrentPipe(GPUMemoryPool* p, int32_t pipe);
void GPUMemoryPool_SetCurrentMode(GPUMemoryPool* p, int32_t mode);
void GPUMemoryPool_SetIter(GPUMemoryPool* p, int32_t iter);
int32_t GPUMemoryPool_GetCurrentMode(const GPUMemoryPool* p);
int32_t GPUMemoryPool_GetIter(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetSampledIds(const GPUMemoryPool* p);
float* GPUMemoryPool_GetFloatFeatures(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetLabels(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetAggSrcOf(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetAggDstOf(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetNodeCounter(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetEdgeCounter(const GPUMemoryPool* p);
/* agg_src_ids (Server.cu:218-231 scratch): the neighbour id of every sampled edge = the next hop's input list.  Written for hops
 * 1 .. H-1 only: nothing reads the last hop's ids (the COO offsets and sampled_ids carry everything a consumer gets). */
int32_t* GPUMemoryPool_GetAggSrcId(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetCacheSearchBuffer(const GPUMemoryPool* p);
char* GPUMemoryPool_GetTmpPartIdx(const GPUMemoryPool* p);
int32_t* GPUMemoryPool_GetTmpPartOff(const GPUMemoryPool* p);
/* The dedup/position table (replaces accessed_map + position_map): u64[V],
 * entry = (epoch << 32) | value with epoch = 0xFFFFFFFF - batch serial; an entry whose epoch is not
 * the running batch's is "not in the batch"; value < 0x80000000 is the node's index in sampled_ids. */
uint64_t* GPUMemoryPool_GetPositionMap(const GPUMemoryPool* p);
/* cand[idx] of the hop that ran last: the neighbour sampler slot idx drew, -1 = no draw (degree < fan-out, padded source); what the
 * duplicate census reads (profiles/dedup_census.py) */
int32_t* GPUMemoryPool_GetCandidateBuffer(const GPUMemoryPool* p);
/* batches started on this pool (the table epoch is 0xFFFFFFFF - serial); settable to exercise the wrap-around */
uint32_t GPUMemoryPool_GetBatchSerial(const GPUMemoryPool* p);
void GPUMemoryPool_SetBatchSerial(GPUMemoryPool* p, uint32_t serial);
/* Batch graphs: record the launcher calls of ONE mini-batch (GPURunner::RunOnce's operator loop, Server.cu:309-323)
 * as a hipGraph and replay it with one launch per batch.  Between Begin and End call the usual launchers
 * (batch_generator_kernel .. get_feature_kernel .. make_update_plan) on `stream` (not the null stream; other
 * streams may join through events and must be joined back before End); batch_generator_kernel's `counter` is
 * ignored while recording -- the batch cursor and the table epoch are device-resident and stepped by the graph
 * itself.  The graph is bound to the pool, the pipe and the mode it was recorded with.  Launch(graph, stream,
 * counter) runs the batch that batch_generator_kernel(.., counter, ..) would; returns 0 on success. */
int GPUMemoryPool_BeginBatchCapture(GPUMemoryPool* p, void* stream);
LegionBatchGraph* GPUMemoryPool_EndBatchCapture(GPUMemoryPool* p, void* stream);
int LegionBatchGraph_Launch(LegionBatchGraph* g, void* stream, int32_t counter);
void LegionBatchGraph_Delete(LegionBatchGraph* g);
void GPUMemoryPool_Finalize(GPUMemoryPool* p);
void GPUMemoryPool_Delete(GPUMemoryPool* p);

/* ---- cache: GPUCache.cuh:64-161, GPUCache.cu:508-872 -------------------------------------- */
GPUCache* NewGPUCache(void);
void GPUCache_Initialize(GPUCache* c, int64_t cache_memory, int32_t int_attr_len, int32_t float_attr_len,
                         int32_t train_step, int32_t device_count);
void GPUCache_InitializeCacheController(GPUCache* c, int32_t dev_id, int32_t total_num_nodes);
void GPUCache_Finalize(GPUCache* c, int32_t dev_id);
int32_t GPUCache_NodeCapacity(const GPUCache* c, int32_t dev_id);
int32_t GPUCache_EdgeCapacity(const GPUCache* c, int32_t dev_id);
void GPUCache_FindFeat(GPUCache* c, int32_t* sampled_ids, int32_t* cache_offset, int32_t* node_counter,
                       int32_t op_id, void* stream, int32_t dev_id);
void GPUCache_FindTopo(GPUCache* c, int32_t* input_ids, char* partition_index, int32_t* partition_offset,
                       int32_t batch_size, int32_t op_id, void* stream, int32_t dev_id);
void GPUCache_CacheProfiling(GPUCache* c, int32_t* sampled_ids, int32_t* node_counter, void* stream, int32_t dev_id);
void GPUCache_CandidateSelection(GPUCache* c, int cache_agg_mode, GPUNodeStorage* noder, GPUGraphStorage* graph);
/* counters: the two PCIe read-transaction counts the reference takes from Intel PCM
 * (Server.cu:100, pcm-pcie.h:176-185).  Pass NULL to use the library's PCM-free estimate
 * derived from the edge hotness collected during pre-sampling. */
void GPUCache_CostModel(GPUCache* c, int cache_agg_mode, GPUNodeStorage* noder, GPUGraphStorage* graph,
                        const uint64_t* counters, int32_t train_step);
/* Skip the cost model and impose capacities (rows per GPU) for every clique. */
void GPUCache_SetCapacity(GPUCache* c, int32_t node_capacity, int32_t edge_capacity);
/* is_presc_ (GPUCache.cuh:160): true until CandidateSelection ran; a server that skips the
 * pre-sampling epoch (everything resident, no cache) clears it explicitly. */
void GPUCache_SetPreSc(GPUCache* c, int is_presc);
void GPUCache_FillUp(GPUCache* c, int cache_agg_mode, GPUNodeStorage* noder, GPUGraphStorage* graph);
int32_t GPUCache_MaxIdNum(const GPUCache* c, int32_t dev_id);
float* GPUCache_Float_Feature_Cache(const GPUCache* c, int32_t dev_id);
/* HIP-IPC size limit.  No single allocation of more than $LEGION_IPC_MAX_BYTES (default 2^31 - 2 MiB) is exported or
 * imported.  Cause (profiles/r02_ipc_limit.md): inside a PyTorch process the HIP runtime is the one bundled with the torch
 * wheel (ROCm 7.0.51831 for torch 2.10.0+rocm7.0 -- also for every library loaded later, liblegion_amd.so and the
 * ipc_service extension included), and its hipIpcOpenMemHandle never returns for an allocation of 2^31 bytes or more
 * (2^31 - 2 MiB opens in 0.2 ms, 2^31 hangs); the system runtime (ROCm 7.2, what the standalone `legion` binary links)
 * opens 8 GiB under 240 GiB of memory pressure without trouble.  Trainers are PyTorch processes, so the limit binds every
 * hand-off buffer.  The Export / Import calls and IPCEnv_InitializeSamplesBuffer refuse larger allocations with a sticky
 * error (LEGION_ERR_EXIT: the server exits non-zero instead of leaving a trainer stalled in ipc_service.initialize()).
 * Shards and fragments are lists of <= $LEGION_SHARD_CHUNK_BYTES (default 1 GiB) chunks and are not affected.
 * The FEATURE hand-off buffer (rows x F x 4 bytes per pipe -- the one that does outgrow the limit, 8.6 GB at the uk-union 3-hop
 * shape) is then built from <= $LEGION_SHARD_CHUNK_BYTES (default 1 GiB) physical chunks (HIP virtual memory management),
 * exported as POSIX file descriptors over an abstract unix socket and mapped back to back into one virtual range by
 * legion_ipc_client_open, so the trainer still sees one contiguous tensor; its 64-byte handle slot in the shm table carries
 * a descriptor ("LGNVMM01", total, chunk, count) instead of an IPC handle. */
#define LEGION_IPC_MAX_BYTES_DEFAULT 2145386496ll /* 2^31 - 2 MiB: the largest size verified with the torch-bundled runtime */
/* HIP-IPC exchange of a clique member's feature shard (64-byte handle); returns 0 on success */
int GPUCache_ExportFeatureShard(GPUCache* c, int32_t dev_id, void* handle64);
int GPUCache_ImportFeatureShard(GPUCache* c, int32_t dev_id, const void* handle64);
/* A shard is a list of chunk allocations (2^k rows each, <= $LEGION_SHARD_CHUNK_BYTES, default 1 GiB): shard row r is
 * row r % ChunkRows of chunk r / ChunkRows.  Each chunk is exported / imported on its own. */
int32_t GPUCache_ShardChunkCount(const GPUCache* c, int32_t dev_id);
int32_t GPUCache_ShardChunkRows(const GPUCache* c, int32_t dev_id);
int32_t GPUCache_ShardPitch(const GPUCache* c);   /* floats between two rows of a chunk: legion_shard_pitch(F, capacity, feature share of cache_memory) */
float* GPUCache_GetShardChunk(const GPUCache* c, int32_t dev_id, int32_t chunk);
int GPUCache_ExportFeatureShardChunk(GPUCache* c, int32_t dev_id, int32_t chunk, void* handle64);
int GPUCache_ImportFeatureShardChunk(GPUCache* c, int32_t dev_id, int32_t chunk, const void* handle64);
/* The geometry a shard was built with, to travel next to its exported handles: out4 = {pitch (floats), rows per chunk, chunks, rows}.
 * An importer addresses the peer's rows with ITS OWN derived geometry, so CheckShardGeometry(owner, geom of the exporter) must return 0
 * before the owner's chunks are imported; -1 (sticky error naming both) when they differ -- never a silent read at the wrong stride. */
int GPUCache_ShardGeometry(const GPUCache* c, int32_t dev_id, int32_t out4[4]);
int GPUCache_CheckShardGeometry(GPUCache* c, int32_t owner_dev, const int32_t geom4[4]);
/* "Feature Cache Hit" metric (GPUCache.cu:130-147,414-425: feature_cache_hit every 500th batch, printed at the last level).
 * Every $LEGION_CACHE_HIT_PERIOD-th batch (default 500) the FindFeat pass of the cached gathers counts its hits into pinned,
 * device-mapped words; "<dev> Feature Cache Hit: <ratio>" is printed when the next sampling starts (no blocking copy).
 * GPUCache_HitSampling is what the gather launchers call; GPUCache_FeatureCacheHitRate returns the newest completed
 * sample (hits / rows, -1 if none; synchronises the device).
 * The pinned words are read and cleared by the host without synchronising: GPUCache_HitSamplingDone records an event behind the
 * last counting launch of a sampled batch, and a slot is only printed / reused once its event has completed. */
int32_t* GPUCache_HitSampling(GPUCache* c, int32_t dev_id, int last_launch_of_batch, int first_launch_of_batch);
void GPUCache_HitSamplingDone(GPUCache* c, int32_t dev_id, void* stream);
double GPUCache_FeatureCacheHitRate(GPUCache* c, int32_t dev_id, int32_t* hits_out, int32_t* rows_out);
uint64_t* GPUCache_GetNodeAccessedMap(const GPUCache* c, int32_t dev_id);
uint64_t* GPUCache_GetEdgeAccessedMap(const GPUCache* c, int32_t dev_id);
/* ranked candidate lists of clique Ki (device pointers on the clique's first GPU): QF / QT */
/* direct-mapped id -> global cache slot table of dev_id (the reference's node_map_, GPUCache.cu:481): int32[V] */
int32_t* GPUCache_GetFeatureMap(const GPUCache* c, int32_t dev_id);
int32_t* GPUCache_GetQF(const GPUCache* c, int32_t Ki);
int32_t* GPUCache_GetQT(const GPUCache* c, int32_t Ki);
int32_t GPUCache_Kg(const GPUCache* c);
int32_t GPUCache_Kc(const GPUCache* c);
double GPUCache_Alpha(const GPUCache* c, int32_t Ki);
void GPUCache_Delete(GPUCache* c);

/* ---- mini-batch operators' launchers: src/Kernels.cuh:47-93 (same names, same order) ------ */
void batch_generator_kernel(void* strm_hdl, GPUNodeStorage* noder, GPUCache* cache, GPUMemoryPool* memorypool,
                            int32_t batch_size, int32_t counter, int32_t part_id, int32_t dev_id, int32_t mode);
void GPU_Random_Sampling(void* strm_hdl, GPUGraphStorage* graph, GPUCache* cache, GPUMemoryPool* memorypool,
                         int32_t count, int32_t op_id, int is_presc);
void get_feature_kernel(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool,
                        int32_t dev_id, int32_t op_id, int in_memory);
void make_update_plan(void* strm_hdl, GPUGraphStorage* graph, GPUCache* cache, GPUMemoryPool* memorypool,
                      int32_t dev_id, int32_t mode);
void update_cache(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool,
                  int32_t dev_id, int32_t mode);
/* Gather every level of the batch in one launch (rows [0, nc[5+2H]) ): same bytes as running
 * get_feature_kernel for op 1,3,..,2H+1; used when the per-level overlap is not wanted. */
void get_feature_kernel_all(void* strm_hdl, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool,
                            int32_t dev_id, int in_memory);
/* Owner-computes exchange variant of the feature gather (SURVEY 5 option b; the reference reads peer caches in-kernel over
 * NVLink, Kernels.cu:662-702 -- this is the collective formulation for one process per GPU, the all-to-all itself is RCCL /
 * hipMemcpyPeer in the caller: legion-1_amd/exchange.py).  plan (requester): rows of the batch cached on another clique member are
 * listed per owner -- req_row[k] = row in the owner's shard, req_dst[k] = row of the batch, contiguous per owner, owner-major,
 * counts[j] = rows asked of clique member j (device int32[2 * LEGION_MAX_DEVICE], the second half is scratch) -- and every other
 * row (own shard, backing table) gets its source address.  local (requester): gathers those rows; runs on any stream behind the
 * plan, e.g. while the counts travel to the host.  serve (owner): rows list[0..n) of this GPU's shard -> out[n x F].
 * scatter (requester): rows[k] -> feature row req_dst[k] of the current pipe.  All pointers are device pointers. */
int legion_exchange_plan(void* stream, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool, int32_t dev_id,
                         int32_t* req_row, int32_t* req_dst, int32_t* counts);
int legion_exchange_local(void* stream, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool, int32_t dev_id);
void legion_exchange_serve(void* stream, GPUCache* cache, int32_t dev_id, const int32_t* list, int32_t n, float* out);
void legion_exchange_scatter(void* stream, GPUMemoryPool* memorypool, const float* rows, const int32_t* req_dst, int32_t n, int32_t F);
/* The same owner-computes gather inside ONE process that drives every GPU of the clique (the `legion` server): the requester's
 * thread plans, reads the Kg request counts (one host synchronisation), and per owner copies the list over
 * (hipMemcpyPeerAsync), lets the owner's GPU gather from its own shard, copies the rows back (hipMemcpyPeerAsync: the xGMI
 * traffic as one DMA per owner) and scatters them -- all rows [0, nc[0]) of the current pipe's batch.  get_feature_kernel uses it
 * for cached clique configurations when $LEGION_PEER_GATHER=exchange (levels < H are then deferred to the last level's op);
 * the default is in-kernel peer loads, the reference's formulation.  stats: {batches, rows requested, host synchronisations}. */
int legion_peer_exchange_gather(void* stream, GPUCache* cache, GPUNodeStorage* noder, GPUMemoryPool* memorypool, int32_t dev_id);
void legion_peer_exchange_stats(const GPUMemoryPool* memorypool, int64_t out[3]);
void GPUMemoryPool_ReleasePeerExchange(GPUMemoryPool* memorypool);

/* ---- Operator plugin API: src/Operator.h:4-27 ------------------------------------------- */
typedef struct OpParams {
    int device_id;
    void* stream;       /* hipStream_t */
    void* event;        /* hipEvent_t  */
    void* memorypool;
    void* cache;
    void* graph;
    void* noder;
    void* env;
    int neighbor_count;
    int is_presc;       /* bool in the reference */
    int in_memory;      /* bool in the reference */
} OpParams;
Operator* NewBatchGenerator(int op_id);
Operator* NewRandomSampler(int op_id);
Operator* NewFeatureExtractor(int op_id);
Operator* NewCachePlanner(int op_id);
Operator* NewCacheUpdater(int op_id);
void Operator_run(Operator* op, OpParams* params);
void Operator_Delete(Operator* op);

/* ---- IPC service, server half: CUDA_IPC_Service.h:6-35, CUDA_IPC_Service.cu:34-360 -------- */
IPCEnv* NewIPCEnv(int32_t device_count);
void IPCEnv_Coordinate(IPCEnv* e, const LegionBuildInfo* info);
int32_t IPCEnv_GetMaxStep(IPCEnv* e);
void IPCEnv_InitializeSamplesBuffer(IPCEnv* e, int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                    int32_t device_id, int32_t pipeline_depth);
void IPCEnv_InitializeFeaturesBuffer(IPCEnv* e, int32_t batch_size, int32_t num_ids, int32_t feature_dim,
                                     int32_t device_id, int32_t pipeline_depth);
int32_t IPCEnv_GetRawBatchsize(IPCEnv* e);
int32_t IPCEnv_GetLocalBatchId(IPCEnv* e, int32_t global_batch_id);
int32_t IPCEnv_GetCurrentBatchsize(IPCEnv* e, int32_t dev_id, int32_t current_mode);
int32_t IPCEnv_GetCurrentMode(IPCEnv* e, int32_t global_batch_id);
int32_t* IPCEnv_GetIds(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
float* IPCEnv_GetFloatFeatures(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
int32_t* IPCEnv_GetLabels(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
int32_t* IPCEnv_GetAggSrc(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
int32_t* IPCEnv_GetAggDst(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
int32_t* IPCEnv_GetNodeCounter(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
int32_t* IPCEnv_GetEdgeCounter(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
/* extension: a second shm object ("<name>_ext"; the shared slab itself stays the reference's 7180-byte struct) carries a host mirror of both counter arrays of every (device, pipe), so that a
 * trainer reads them without a device copy (the reference's blocking cudaMemcpy, ipc_cuda_kernel.cu:195-196, synchronises the trainer
 * with its own queued GPU work once per batch).  IPCEnv_MirrorCounters queues the copy of the current batch's counters on `stream`
 * (behind the kernels that wrote them; wait for that stream before posting); IPCEnv_IPCPost copies synchronously when it was not called;
 * IPCEnv_SetMirror fills the mirror from the host (poisoned pipe). */
void IPCEnv_MirrorCounters(IPCEnv* e, int32_t dev_id, int32_t current_pipe, void* stream);
void IPCEnv_SetMirror(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t nc_fill, int32_t ec_fill);
/* nc[word] of the batch about to be posted, from the mirror IPCEnv_MirrorCounters queued (wait for that copy first); -1: not queued.
 * The runner compares nc[5 + 2H] with the rows of its feature buffer: a batch that reached more nodes had rows dropped by the bounded
 * gather (kernels.hip k_gather: "never write past the buffer") and its trainer will refuse it -- the server says so, once, and counts. */
int32_t IPCEnv_MirroredNodeCounter(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t word);
/* the row capacity of a device's feature buffers as published to its trainer (set by IPCEnv_InitializeFeaturesBuffer) */
void IPCEnv_SetFeatureRows(IPCEnv* e, int32_t device_id, int32_t rows);
int IPCEnv_SlabPinned(IPCEnv* e);   /* 1: the slab is page-locked (hipHostRegister), IPCEnv_MirrorCounters queues real asynchronous copies */
void IPCEnv_IPCPost(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
void IPCEnv_IPCWait(IPCEnv* e, int32_t dev_id, int32_t current_pipe);
int IPCEnv_IPCTryWait(IPCEnv* e, int32_t dev_id, int32_t current_pipe, int32_t timeout_ms); /* 0 = acquired */
/* Both sides poll the semaphore for this many microseconds (200) before they block on it: a futex
 * wake-up costs 10-60 us and sits on the depth-2 handshake of every batch (profiles/r05_handoff_spin.md). */
int IPCEnv_HandoffSpinUs(void);
void IPCEnv_Finalize(IPCEnv* e);
int32_t IPCEnv_GetTrainStep(IPCEnv* e);
/* extension: number of hops published to the trainer (stored after the reference's struct) */
void IPCEnv_SetHops(IPCEnv* e, int32_t hops);
/* Namespace for the POSIX shm / semaphore names ("" = the reference's literal names
 * "simpleIPCshm", "sem_r_D_P", "sem_w_D_P").  Also read from $LEGION_IPC_NAMESPACE. */
void legion_ipc_set_namespace(const char* ns);
/* Remove what a KILLED server of namespace `ns` left in /dev/shm -- the slab, its extension object, the 2 x depth named semaphores of
 * each of `devices` GPUs (the reference only unlinks in Finalize, CUDA_IPC_Service.cu:319-320).  Harmless when nothing is there. */
void legion_ipc_unlink_namespace(const char* ns, int32_t devices);

/* ---- IPC service, trainer half: pytorch_extension/ipc_service.h + ipc_cuda_kernel.cu ------ */
typedef struct LegionIPCClient LegionIPCClient;
LegionIPCClient* legion_ipc_client_open(int32_t device_id);  /* GPUIPCEnv::Initialize, ipc_cuda_kernel.cu:38-96 */
void legion_ipc_client_wait(LegionIPCClient* c);              /* Wait(), :98-101 */
/* Post(), :103-107 -- behind a hipDeviceSynchronize() of the calling process' current device: work the trainer has queued may still read
 * the pipe's buffers, and a posted pipe is overwritten by the server.  (The reference got the same bound from the blocking counter copy
 * of the next get_next; read_counters below reads a host mirror and synchronises nothing, so the wait sits here.) */
void legion_ipc_client_post(LegionIPCClient* c);
/* the bare sem_post: for a consumer that has already waited for its own device work (or queues none) */
void legion_ipc_client_post_nosync(LegionIPCClient* c);
/* buffer index: 0 ids 1 features 2 labels 3 agg_src 4 agg_dst 5 node_counter 6 edge_counter
 * (CUDA_IPC_Service.cu:169-175,209) of the current pipe */
void* legion_ipc_client_buffer(LegionIPCClient* c, int32_t which);
void legion_ipc_client_steps(LegionIPCClient* c, int32_t steps[3]);
int32_t legion_ipc_client_hops(LegionIPCClient* c);
/* rows the feature buffers of this client's device hold (0: the server did not say -- a reference server).  A batch whose node count
 * exceeds it must not be viewed as [n, F]: the reference sizes the buffer from the pre-sampling epoch's training batches (Server.cu:275)
 * and views it unchecked (ipc_cuda_kernel.cu:200). */
int32_t legion_ipc_client_feature_rows(LegionIPCClient* c);
/* both 16-int counters of the current pipe (ipc_cuda_kernel.cu:195-196): from the server's host mirror when it maintains one, else by
 * a blocking device copy like the reference */
void legion_ipc_client_read_counters(LegionIPCClient* c, int32_t h_node_counter[16], int32_t h_edge_counter[16]);
void legion_ipc_client_close(LegionIPCClient* c);             /* Finalize(), :141-156 */

/* ---- Runner / Server: Server.h:137-165, Server.cu:43-369 ---------------------------------- */
typedef struct RunnerParams {
    int device_id;
    const int32_t* fanout; int32_t hops;   /* std::vector<int> fanout in the reference */
    void* cache; void* graph; void* noder; void* env;
    int global_batch_id;
    int in_memory;
} RunnerParams;
/* The decision rule of $LEGION_RUNNER_GATHER=auto as a pure function (host only): 1 = ONE gather over all rows behind the last hop, 0 = one per level
 * (the reference's op list).  rows: unique nodes of a batch; slots: sum over the hops of (input nodes x fan-out).  profiles/r05_runner_gather.md */
int legion_runner_gather_estimate(int32_t F, double rows, double slots, double* gather_us, double* sampler_us);
Runner* NewGPURunner(void);
void Runner_Initialize(Runner* r, RunnerParams* params);
void Runner_InitializeFeaturesBuffer(Runner* r, RunnerParams* params);
void Runner_RunPreSc(Runner* r, RunnerParams* params);
/* RunOnce, Server.cu:301-328.  A batch an operator refused (sticky error): LEGION_ERR_EXIT exits like the reference;
 * LEGION_ERR_RETURN posts the pipe with every node-counter word = -1 (nc[0] == -1: "server failed", no valid batch has it)
 * so that no consumer blocks forever on sem_w, and returns with the error still set. */
void Runner_RunOnce(Runner* r, RunnerParams* params);
void Runner_Finalize(Runner* r, RunnerParams* params);
GPUMemoryPool* Runner_GetMemoryPool(Runner* r);
/* batches handed over with more nodes than the feature buffers hold rows (their last rows were not gathered; logged once) */
int64_t Runner_ShortBatches(const Runner* r);
void Runner_Delete(Runner* r);
/* Whole server driven by a meta_config file (legion_server.py:58-59, GPUGraphStore.cu:190-223).
 * fanout may be NULL (reference default {25,10}, Server.cu:68-69). */
Server* NewGPUServer(void);
void Server_SetFanout(Server* s, const int32_t* fanout, int32_t hops);
void Server_SetMetaConfigPath(Server* s, const char* path);
void Server_Initialize(Server* s, int global_shard_count);
void Server_PreSc(Server* s, int cache_agg_mode);
void Server_Run(Server* s);
void Server_Finalize(Server* s);
void Server_Delete(Server* s);

/* ---- synthetic datasets (generator spec: legion-1_amd/synth.py) ----------------------------- */
void legion_synth_degrees(void* stream, int64_t* deg_out, int32_t v0, int32_t n, const int32_t* ladder_host26);
void legion_synth_neighbors(void* stream, int32_t* indices_out, int64_t e0, int64_t n, int32_t V, uint32_t M, uint32_t C);
/* same with n/256 of the neighbours drawn from the Zipf-like skew (205 = the spec; 0 = uniform: the control run that
 * shows how much of the gather's rate is Infinity-Cache reuse of hot rows) */
void legion_synth_neighbors_skew(void* stream, int32_t* indices_out, int64_t e0, int64_t n, int32_t V, uint32_t M, uint32_t C,
                                 int32_t skew_of_256);
/* Link-prediction seed batches for lp_sage.py:87-90 ([src | pos | neg] thirds per batch of `batch_size`; the reference
 * server has no edge / negative sampler).  Triple j: src = srcs[j], pos / neg from the minstd stream at
 * 48271^(seed + 2*triple_no[j] + 1) -- triple_no = the triple's number in the global order, so the per-GPU lists of a
 * G-GPU job (triples dealt by src % G) hold exactly the triples of the 1-GPU list.  out: ceil(n/k)*batch_size ids,
 * k = batch_size/3; device pointers.  Same rule in numpy: legion-1_amd/synth.py lp_trainingset. */
void legion_synth_lp_seeds(void* stream, int32_t* out, const int32_t* srcs, const int64_t* triple_no, int64_t n_triples,
                           int32_t batch_size, const int64_t* indptr, const int32_t* indices, int32_t V, uint32_t seed);
void legion_synth_features(void* stream, float* out, int64_t v0, int64_t nrows, int32_t F);
/* the same values with `pitch` floats between two rows (LegionBuildInfo.float_attr_pitch); pad floats are not written */
void legion_synth_features_pitched(void* stream, float* out, int64_t v0, int64_t nrows, int32_t F, int32_t pitch);
void legion_synth_labels(void* stream, int32_t* out, int32_t v0, int32_t n, int32_t classes);
void legion_synth_seed_ids(void* stream, int32_t* out, int64_t i0, int64_t n, int32_t V, uint32_t M2, uint32_t C2, int32_t stride, int32_t phase);
/* The generator's parameters for a named shape ("products" | "papers100M" | "uk-union", legion_server.py:6-37), shrunk by
 * `scale` in (0, 1] (V and the seed sets; the mean degree stays): the C statement of synth.py spec_for(), field for field
 * (tests/test_host_logic.py).  Host only, no device call.  0, or -1 (sticky error) for an unknown name / scale. */
typedef struct LegionSynthSpec {
    int32_t V, F, classes;
    int32_t n_train, n_valid, n_test;   /* seed i of the permutation (i * M2 + C2) % V: train = [0, n_train), then valid, then test */
    uint32_t M, C, M2, C2;              /* neighbour-id scrambler and seed permutation (multipliers coprime with V) */
    int32_t ladder[26];                 /* degree ladder lo[b] of legion_synth_degrees */
    double mean_degree;
} LegionSynthSpec;
int legion_synth_spec(const char* workload, double scale, LegionSynthSpec* out);
int32_t legion_synth_label_host(int32_t v, int32_t classes);                           /* == legion_synth_labels, one id, on the host */
int32_t legion_synth_seed_id_host(int64_t i, int32_t V, uint32_t M2, uint32_t C2);     /* == legion_synth_seed_ids, one index */
/* streaming-copy kernel used by bench.py to report the measured HBM peak */
void legion_copy_f4(void* stream, void* dst, const void* src, int64_t bytes);
/* *acc += the sum of the 32-bit words of [src, src + bytes) (u64, wrap-around; src 16-byte aligned, bytes a multiple of 4): what a trainer's
 * read of a served batch costs without a model (legion_graphsage.py:72-89 reads every feature row and both COO arrays before it hands the
 * pipe back) -- bench.py's reading consumer; exact and order-independent, so it doubles as a checksum of what was read */
void legion_sum_words(void* stream, const void* src, int64_t bytes, uint64_t* acc);
/* the same copy with an explicit variant (profiles/copy_sweep.py): 256-thread workgroups, `unroll` 16-byte chunks in
 * flight per lane (1, 2, 4, 8), nt bit 0 = non-temporal stores, bit 1 = non-temporal loads, contig = block-strided;
 * grid <= 0: one iteration per lane.  Returns 0, or -1 for an unknown variant. */
int legion_copy_f4_cfg(void* stream, void* dst, const void* src, int64_t bytes, int32_t grid, int32_t unroll, int32_t nt, int32_t contig);

/* ---- self-description for tests ------------------------------------------------------------ */
/* RNG probe: k[i] = sample index for (idx[i], deg[i]) computed ON THE GPU with the kernel's
 * own arithmetic (Kernels.cu:402-405 semantics). */
void legion_rng_probe(void* stream, const int32_t* idx, const int32_t* deg, int32_t* k_out, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* LEGION_AMD_H */
