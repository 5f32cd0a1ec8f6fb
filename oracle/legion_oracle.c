/*
 * legion_oracle.c -- CPU restatement of Legion's GPU-initiated mini-batch hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (legion-1_amd/csrc)
 * never links or calls anything in this directory.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * upstream tree, liayan/Legion-1).  The reference's kernels race on LDS / global
 * atomics, so their output ORDER is schedule dependent; this oracle fixes the
 * canonical schedule = serial, idx-ascending execution of every grid-stride loop
 * (an admissible interleaving of the reference; SURVEY.md section 7, hard part 1).
 *
 * Parity pin: the reference has no tests / golden vectors for this path
 * (SURVEY.md section 4), and its CUDA sources cannot be compiled here.  The one
 * third-party piece of arithmetic on the path -- Thrust's minstd_rand +
 * uniform_int_distribution (Kernels.cu:402-405) -- is pinned by Thrust's own
 * documented known answer (10000th minstd_rand value = 399268537,
 * thrust/random/linear_congruential_engine.h) and by oracle/thrust_probe.cpp,
 * which compiles the Thrust headers shipped in this image (rocThrust, host side)
 * and was used to generate tests/golden/rng_kat.json.  Everything else is
 * integer indexing restated line by line; it is "pinned by restatement only".
 * PARITY vs. the RUNNING reference is therefore UNPINNED beyond the RNG: the reference cannot be built or run in
 * this image (CUDA-only) and holds no fixture of its own for this path.
 *
 * Plain C99, no dependencies.  Build: see oracle/Makefile.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define LO_TRAINMODE 0
#define LO_VALIDMODE 1
#define LO_TESTMODE 2

/* ------------------------------------------------------------------------- */
/* RNG: thrust::minstd_rand (a=48271, c=0, m=2^31-1), default seed 1          */
/* ------------------------------------------------------------------------- */

/* thrust/random/detail/linear_congruential_engine_discard.h:56-82 -- square
 * and multiply with 64-bit intermediate and a true `%`.  Returns the state
 * AFTER discard(z) starting from `state`. */
static uint32_t lo_minstd_discard(uint32_t state, unsigned long long z)
{
    const uint32_t modulus = 2147483647u;
    unsigned long long multiplier = 48271ull;
    unsigned long long multiplier_to_z = 1ull;
    while (z > 0) {
        if (z & 1ull) multiplier_to_z = (multiplier_to_z * multiplier) % modulus;
        z >>= 1;
        multiplier = (multiplier * multiplier) % modulus;
    }
    return (uint32_t)((multiplier_to_z * (unsigned long long)state) % modulus);
}

/* One engine step: linear_congruential_engine.h operator() -- x = a*x mod m. */
static uint32_t lo_minstd_next(uint32_t *state)
{
    *state = (uint32_t)((48271ull * (unsigned long long)(*state)) % 2147483647ull);
    return *state;
}

/* The value urng() returns inside dist(engine) after engine.discard(idx):
 * 48271^(idx+1) mod (2^31-1).  Kernels.cu:402-405. */
uint32_t lo_minstd_value(int32_t idx)
{
    uint32_t st = lo_minstd_discard(1u, (unsigned long long)idx);
    return lo_minstd_next(&st);
}

/* nth (1-based) output of a default-constructed minstd_rand; used for Thrust's
 * documented known answer (n = 10000 -> 399268537). */
uint32_t lo_minstd_nth(uint64_t n)
{
    uint32_t st = 1u;
    uint32_t v = 0;
    for (uint64_t i = 0; i < n; i++) v = lo_minstd_next(&st);
    return v;
}

/* thrust::uniform_int_distribution<int>(0, deg-1)(engine) after discard(idx):
 * uniform_int_distribution.inl:73-89 -> uniform_real_distribution<double>(0, deg):
 *   result = double(x - min) / (1.0 + double(max - min)) * (b - a) + a
 * with min = 1, max = 2147483646.  Kernels.cu:402-405. */
int32_t lo_sample_index(int32_t idx, int32_t deg)
{
    uint32_t x = lo_minstd_value(idx);
    double result = (double)(uint32_t)(x - 1u);
    result /= (1.0 + (double)(uint32_t)(2147483646u - 1u));
    double real_min = 0.0;
    double real_max = (double)(deg - 1) + 1.0;
    return (int32_t)((result * (real_max - real_min)) + real_min);
}

/* ------------------------------------------------------------------------- */
/* S2: update_counter (Kernels.cu:112-150), generalised to H hops             */
/* ------------------------------------------------------------------------- */
/* op_id 0 = after the batch generator; op_id 2h = after hop h (h = 1..H).
 * At H = 2 this writes exactly the slots the reference writes (nc[5..9],
 * ec[3], ec[4]); for general H the level-l gather pair lives at nc[3+2l],
 * nc[4+2l], the total at nc[5+2H], cumulative edges after hop h at ec[2+h]
 * (SURVEY.md section 8a, row S2). */
void lo_update_counter(int32_t *nc, int32_t *ec, int32_t op_id, int32_t size, int32_t hops)
{
    if (op_id == 0) {
        nc[0] = size; nc[1] = 0; nc[2] = size; nc[3] = 0; nc[4] = size;
        ec[0] = 0; ec[1] = 0; ec[2] = 0; ec[3] = 0;
    } else {
        int32_t h = op_id / 2;
        nc[0] += nc[1];
        nc[3 + 2 * h] = nc[1 + 2 * h] + nc[2 + 2 * h];
        nc[4 + 2 * h] = nc[1];
        if (h == hops) nc[5 + 2 * h] = nc[3 + 2 * h] + nc[4 + 2 * h];
        nc[1] = 0;
        nc[2] = ec[1];
        /* Kernels.cu:134 `ec[3] += ec[1]` (ec[3] is 0 there) and :145
         * `ec[4] = ec[3] + ec[1]` are both "cumulative edges through hop h". */
        ec[2 + h] = (h == 1 ? ec[3] : ec[1 + h]) + ec[1];
        ec[2] = ec[0];
        ec[0] += ec[1];
        ec[1] = 0;
    }
}

/* ------------------------------------------------------------------------- */
/* S1: batch_generator + batch_generator_kernel (Kernels.cu:68-96,163-232)    */
/* ------------------------------------------------------------------------- */
/* Restated literally, including the launcher's quirk that the kernel receives
 * `size` (the clamped batch size) as its `batch_size` parameter, so the read
 * offset of a short last batch is size*counter, not batch_size*counter
 * (Kernels.cu:224-227 vs :81-85).  Returns `size`. */
int32_t lo_batch_generator_kernel(
    int32_t *batch_ids, int32_t *labels, int32_t batch_size, int32_t counter,
    const int32_t *all_ids, const int32_t *all_labels, int32_t total_cap,
    int32_t total_node_num, int32_t *position_map, uint32_t *accessed_map,
    int32_t *nc, int32_t *ec, int32_t hops)
{
    /* Kernels.cu:216-221 */
    memset(accessed_map, 0, (size_t)((int64_t)(total_node_num / 32) + 1) * sizeof(uint32_t));
    memset(nc, 0, 16 * sizeof(int32_t));
    memset(ec, 0, 16 * sizeof(int32_t));
    /* Kernels.cu:224 */
    int32_t size = ((batch_size * (counter + 1)) >= total_cap) ? (total_cap - batch_size * counter) : batch_size;
    /* Kernels.cu:79-95, kernel's batch_size == size */
    for (int32_t idx = 0; idx < size; idx++) {
        if ((size * counter + idx) >= total_cap) {
            batch_ids[idx] = -1;
            labels[idx] = -1;
        } else {
            int32_t src_id = all_ids[(size * counter + idx) % total_cap];
            batch_ids[idx] = src_id;
            int32_t bitmap_idx = src_id / 32;
            int32_t bitmap_off = src_id % 32;
            accessed_map[bitmap_idx] |= (1u << bitmap_off);
            position_map[src_id] = idx;
            labels[idx] = all_labels[(size * counter + idx) % total_cap];
        }
    }
    lo_update_counter(nc, ec, 0, size, hops); /* Kernels.cu:230 */
    return size;
}

/* ------------------------------------------------------------------------- */
/* S3 / S3': kernel_random_sampler_2 (Kernels.cu:342-448) and                 */
/* kernel_pre_sampler_optimized (:468-564), canonical schedule                */
/* ------------------------------------------------------------------------- */
/* csr_node_index / csr_dst_node_ids: tables of partition_count+1 pointers;
 * slot [partition_count] is the whole (host) CSR, slot [g] GPU g's cached
 * fragment (GPU_Memory_Graph_Storage.cu:45-133).  partition_index may be NULL
 * (treated as all -1 = not cached).  edge_access_time != NULL selects the
 * pre-sampling variant (host CSR only, hotness bump, Kernels.cu:514-525). */
void lo_random_sampler(
    int32_t *sampled_ids, int32_t op_id,
    const int64_t *const *csr_node_index, const int32_t *const *csr_dst_node_ids,
    const signed char *partition_index, const int32_t *partition_offset,
    int32_t count, int32_t partition_count,
    int32_t *agg_src_ids, int32_t *agg_dst_ids,
    uint32_t *accessed_map, int32_t *position_map,
    int32_t *nc, int32_t *ec, uint64_t *edge_access_time)
{
    const int32_t *input_ids = NULL;
    int32_t batch_size = 0;
    if (op_id == 2) { input_ids = sampled_ids; batch_size = nc[2]; }
    else if (op_id > 2) { input_ids = agg_src_ids + ec[2]; batch_size = nc[2]; }
    /* int32 trip count exactly like the reference (Kernels.cu:375) */
    int32_t total = batch_size * count;
    for (int32_t idx = 0; idx < total; idx++) {
        int32_t sample_src_id = input_ids[idx / count];
        if (sample_src_id < 0) continue;
        int32_t neighbor_offset = idx % count;
        int32_t part_id = -1, part_offset = 0;
        if (!edge_access_time && partition_index) {
            part_id = partition_index[idx / count];   /* `char`, signed on x86 (Kernels.cu:387) */
            part_offset = partition_offset[idx / count];
        }
        int64_t start_index; int32_t col_size;
        const int32_t *adj;
        if (part_id < 0) {
            start_index = csr_node_index[partition_count][sample_src_id];
            col_size = (int32_t)(csr_node_index[partition_count][sample_src_id + 1] - start_index);
            adj = csr_dst_node_ids[partition_count];
        } else {
            start_index = csr_node_index[part_id][part_offset];
            col_size = (int32_t)(csr_node_index[part_id][part_offset + 1] - start_index);
            adj = csr_dst_node_ids[part_id];
        }
        if (neighbor_offset >= col_size) continue;            /* Kernels.cu:399-400 */
        int32_t dst_index = lo_sample_index(idx, col_size);    /* :402-405 */
        int32_t sample_dst_id = adj[start_index + (int64_t)dst_index];
        if (sample_dst_id < 0) continue;                       /* :411 */
        if (edge_access_time) edge_access_time[sample_src_id] += 1; /* :525 */
        int32_t bitmap_idx = sample_dst_id / 32;
        int32_t bitmap_off = sample_dst_id % 32;
        uint32_t old = accessed_map[bitmap_idx];
        accessed_map[bitmap_idx] = old | (1u << bitmap_off);
        if (((old >> bitmap_off) & 1u) == 0) {                 /* first time seen (:418-421,434-439) */
            int32_t p = nc[0] + nc[1]++;
            sampled_ids[p] = sample_dst_id;
            position_map[sample_dst_id] = p;
        }
        int32_t e = ec[0] + ec[1]++;                           /* :422-424,441-445 */
        agg_src_ids[e] = sample_dst_id;
        agg_dst_ids[e] = sample_src_id;
    }
}

/* S4: construct_graph (Kernels.cu:450-463) */
void lo_construct_graph(const int32_t *agg_src_ids, const int32_t *agg_dst_ids,
                        int32_t *agg_src_off, int32_t *agg_dst_off,
                        const int32_t *position_map, const int32_t *ec)
{
    int32_t edge_num = ec[1], edge_off = ec[0];
    for (int32_t idx = 0; idx < edge_num; idx++) {
        agg_src_off[edge_off + idx] = position_map[agg_src_ids[edge_off + idx]];
        agg_dst_off[edge_off + idx] = position_map[agg_dst_ids[edge_off + idx]];
    }
}

/* GPU_Random_Sampling (Kernels.cu:568-659): sampler -> construct_graph -> update_counter */
void lo_gpu_random_sampling(
    int32_t *sampled_ids, int32_t op_id,
    const int64_t *const *csr_node_index, const int32_t *const *csr_dst_node_ids,
    const signed char *partition_index, const int32_t *partition_offset,
    int32_t count, int32_t partition_count,
    int32_t *agg_src_ids, int32_t *agg_dst_ids, int32_t *agg_src_off, int32_t *agg_dst_off,
    uint32_t *accessed_map, int32_t *position_map,
    int32_t *nc, int32_t *ec, uint64_t *edge_access_time, int32_t hops)
{
    lo_random_sampler(sampled_ids, op_id, csr_node_index, csr_dst_node_ids, partition_index,
                      partition_offset, count, partition_count, agg_src_ids, agg_dst_ids,
                      accessed_map, position_map, nc, ec, edge_access_time);
    lo_construct_graph(agg_src_ids, agg_dst_ids, agg_src_off, agg_dst_off, position_map, ec);
    lo_update_counter(nc, ec, op_id, 0, hops);
}

/* ------------------------------------------------------------------------- */
/* S6: cache lookups.  BGHT bcht::find (include/detail/bcht_impl.cuh:249-273)  */
/* is observable only as key -> value-or-sentinel(-1); the table contents are */
/* the pairs built by InitPair / InitIndexPair / InitOffsetPair                */
/* (GPUCache.cu:88-108).  Restated as direct-mapped arrays over [0, V).       */
/* ------------------------------------------------------------------------- */
/* node_map[id] = (t % Kg) * capacity + t / Kg for t < capacity*Kg (t = rank of
 * id in QF), else -1.  GPUCache.cu:103-108.  Ranks t >= V are skipped (the
 * reference reads QF out of bounds there). */
void lo_build_feat_map(int32_t *node_map, int32_t V, const int32_t *QF, int32_t capacity, int32_t Kg)
{
    for (int32_t i = 0; i < V; i++) node_map[i] = -1;
    int64_t n = (int64_t)capacity * Kg;
    for (int64_t t = 0; t < n && t < V; t++)
        node_map[QF[t]] = (int32_t)((t % Kg) * capacity + t / Kg);
}

/* edge_index_map[id] = t % Kg + Ki*Kg ; edge_offset_map[id] = t / Kg. GPUCache.cu:88-100 */
void lo_build_topo_map(signed char *part_index, int32_t *part_offset, int32_t V, const int32_t *QT,
                       int32_t capacity, int32_t Kg, int32_t Ki)
{
    for (int32_t i = 0; i < V; i++) { part_index[i] = -1; part_offset[i] = -1; }
    int64_t n = (int64_t)capacity * Kg;
    for (int64_t t = 0; t < n && t < V; t++) {
        part_index[QT[t]] = (signed char)(t % Kg + Ki * Kg);
        part_offset[QT[t]] = (int32_t)(t / Kg);
    }
}

/* FindFeat (GPUCache.cu:387-432): cache_offset[r] = node_map.find(sampled_ids[off+r]).
 * Level offsets per SURVEY 8a S2: op_id = 2l+1 -> (nc[3+2l], nc[4+2l]).
 * A negative id is not a key of the table -> -1. */
void lo_find_feat(const int32_t *sampled_ids, int32_t *cache_offset, const int32_t *nc,
                  int32_t op_id, const int32_t *node_map)
{
    int32_t l = (op_id - 1) / 2;
    int32_t node_off = nc[3 + 2 * l], batch_size = nc[4 + 2 * l];
    for (int32_t r = 0; r < batch_size; r++) {
        int32_t id = sampled_ids[node_off + r];
        cache_offset[r] = (id < 0) ? -1 : node_map[id];
    }
}

/* FindTopo (GPUCache.cu:434-443) */
void lo_find_topo(const int32_t *input_ids, signed char *partition_index, int32_t *partition_offset,
                  int32_t batch_size, const signed char *part_index_map, const int32_t *part_offset_map)
{
    for (int32_t i = 0; i < batch_size; i++) {
        int32_t id = input_ids[i];
        partition_index[i] = (id < 0) ? -1 : part_index_map[id];
        partition_offset[i] = (id < 0) ? -1 : part_offset_map[id];
    }
}

/* ------------------------------------------------------------------------- */
/* S5: zero_copy_with_aggregated_cache (Kernels.cu:662-702)                   */
/* ------------------------------------------------------------------------- */
void lo_gather_features(
    const float *cpu_float_attrs, const float *const *cache_float_attrs, int32_t float_attr_len,
    const int32_t *sampled_ids, const int32_t *cache_index, int32_t cache_capacity,
    const int32_t *nc, float *dst_float_buffer, int32_t total_num_nodes, int32_t op_id)
{
    int32_t l = (op_id - 1) / 2;
    int32_t node_off = nc[3 + 2 * l], batch_size = nc[4 + 2 * l];
    if (float_attr_len <= 0) return;
    int64_t total = (int64_t)batch_size * float_attr_len;
    for (int64_t t = 0; t < total; t++) {
        int32_t gidx = cache_index[t / float_attr_len];
        int32_t foffset = (int32_t)(t % float_attr_len);
        int64_t out = (int64_t)node_off * float_attr_len + t;
        if (gidx < 0) {
            int32_t fidx = sampled_ids[node_off + (t / float_attr_len)];
            if (fidx >= 0)
                dst_float_buffer[out] = cpu_float_attrs[(int64_t)(fidx % total_num_nodes) * float_attr_len + foffset];
        } else {
            int32_t didx = gidx / cache_capacity;
            int32_t fidx = gidx % cache_capacity;
            dst_float_buffer[out] = cache_float_attrs[didx][(int64_t)fidx * float_attr_len + foffset];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* S7: ClearPosMap (Kernels.cu:750-756), HotnessMeasure (GPUCache.cu:227-235)  */
/* ------------------------------------------------------------------------- */
void lo_clear_pos_map(int32_t *position_map, const int32_t *sampled_ids, const int32_t *nc, int32_t hops)
{
    int32_t n = nc[5 + 2 * hops];
    for (int32_t i = 0; i < n; i++) position_map[sampled_ids[i]] = 0;
}

void lo_hotness_measure(const int32_t *new_batch_ids, const int32_t *nc, uint64_t *access_map, int32_t hops)
{
    int32_t n = nc[5 + 2 * hops];
    for (int32_t i = 0; i < n; i++) {
        int32_t cid = new_batch_ids[i];
        if (cid >= 0) access_map[cid] += 1;
    }
}

/* ------------------------------------------------------------------------- */
/* S8: CandidateSelection (GPUCache.cu:578-659)                                */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t key; int32_t id; } lo_rank_t;
static int lo_rank_cmp(const void *a, const void *b)
{
    const lo_rank_t *x = (const lo_rank_t *)a, *y = (const lo_rank_t *)b;
    if (x->key != y->key) return (x->key > y->key) ? -1 : 1;   /* thrust::greater */
    return (x->id < y->id) ? -1 : (x->id > y->id);              /* tie-break: ascending id (documented deviation:
                                                                 thrust::sort_by_key is not stable, ties unspecified) */
}

/* agg[v] = sum_j access[j][v] (aggregate_access, GPUCache.cu:44-48,624-627);
 * order = iota sorted by agg descending (:630-631).  On return agg is sorted too
 * (the reference sorts the keys in place and keeps them as AF_/AT_). */
void lo_candidate_selection(const uint64_t *const *access, int32_t Kg, int32_t V,
                            uint64_t *agg_sorted, int32_t *order)
{
    lo_rank_t *r = (lo_rank_t *)malloc((size_t)V * sizeof(lo_rank_t));
    for (int32_t v = 0; v < V; v++) {
        uint64_t s = 0;
        for (int32_t j = 0; j < Kg; j++) s += access[j][v];
        r[v].key = s; r[v].id = v;
    }
    qsort(r, (size_t)V, sizeof(lo_rank_t), lo_rank_cmp);
    for (int32_t v = 0; v < V; v++) { agg_sorted[v] = r[v].key; order[v] = r[v].id; }
    free(r);
}

/* ------------------------------------------------------------------------- */
/* S8: CostModel (GPUCache.cu:661-767), one clique                            */
/* ------------------------------------------------------------------------- */
/* Inputs: AF/AT = hotness sorted descending (from candidate selection), QT =
 * topology rank order, csr_index = host indptr, counters[2] = the two PCIe
 * read-transaction counters the reference takes from Intel PCM (Server.cu:100),
 * max_ids[j] = cache_controller_[j]->MaxIdNum(), j < Kg: the reference indexes
 * GPU j, i.e. the members of the FIRST clique, whichever clique is being solved
 * (GPUCache.cu:677-680) -- callers pass those.  Outputs node/edge capacity per GPU,
 * alpha step index and the winning transaction figure.
 * Restated literally: float accumulators, `steps` arithmetic, the `< V` guards
 * that leave trans_* at 0 once everything fits (GPUCache.cu:744-751), and
 * h_*_prefix[node_num - 1] (node_num == 0 is guarded here: the reference reads
 * index -1; we treat that prefix as 0). */
void lo_cost_model(
    const uint64_t *AF, const uint64_t *AT, const int32_t *QT, const int64_t *csr_index,
    int32_t V, int32_t float_attr_len, int64_t cache_memory, int32_t Kg,
    const uint64_t *counters, const int32_t *max_ids, int32_t train_step,
    int32_t *node_capacity, int32_t *edge_capacity, int32_t *alpha_idx, float *best_trans)
{
    const int max_payload_size = 64;                     /* CLS */
    int64_t memory_step = (int64_t)((double)(cache_memory * Kg) * 0.01); /* MIN_INTERVAL, :674 */
    uint64_t total_trans_of_topo = 0;
    if (counters) {
        total_trans_of_topo = counters[0] + counters[1];  /* the two Intel-PCM PCIe read counters, :675, Server.cu:100,108 */
    } else {
        /* No PCM here (Intel-only MSRs): the product estimates the epoch's adjacency transactions from its own hotness
         * counters (sum over rows of edge_access_time * (8 + 4 * min(deg, .)) / 64).  Restated
         * independently: a sampled edge reads the row's 8-byte offset and one neighbour id; rows whose offset + ids fit
         * one 64-byte line (deg <= 14) cost one transaction per sampled edge, longer rows two ("." = the 16 ids of a line).
         * Not reference arithmetic -- the reference always has the counters; parity of this branch is product-vs-oracle only. */
        for (int32_t i = 0; i < V; i++) {
            int64_t deg = csr_index[QT[i] + 1] - csr_index[QT[i]];
            int64_t ids = deg > 16 ? 16 : (deg < 0 ? 0 : deg);
            int64_t bytes = 8 + 4 * ids;
            total_trans_of_topo += AT[i] * (uint64_t)((bytes + 63) / 64);
        }
    }
    uint64_t total_trans_of_feat = 0;
    for (int j = 0; j < Kg; j++)
        total_trans_of_feat += (uint64_t)(((int64_t)max_ids[j] * train_step * float_attr_len * (int64_t)sizeof(float)) / max_payload_size);

    uint64_t *node_prefix = (uint64_t *)malloc((size_t)V * 8);
    uint64_t *edge_prefix = (uint64_t *)malloc((size_t)V * 8);
    uint64_t *edge_mem_prefix = (uint64_t *)malloc((size_t)V * 8);
    uint64_t a = 0, b = 0, c = 0;
    for (int32_t i = 0; i < V; i++) {
        a += AF[i]; node_prefix[i] = a;
        b += AT[i]; edge_prefix[i] = b;
        int32_t id = QT[i];
        int64_t nb = csr_index[id + 1] - csr_index[id];
        c += (uint64_t)(sizeof(int64_t) + sizeof(int32_t) * nb);   /* GetEdgeMem, :35-41 */
        edge_mem_prefix[i] = c;
    }

    int64_t current_mem = 0;
    int64_t total_mem = cache_memory * Kg;
    int64_t steps = (total_mem - 1) / memory_step + 1;
    int64_t current_steps = 0;
    int32_t node_num_topo = 0, node_num_feat = 0;
    float *trans_of_topo = (float *)calloc((size_t)steps + 1, sizeof(float));
    float *trans_of_feat = (float *)calloc((size_t)steps + 1, sizeof(float));
    float *cap_of_topo = (float *)calloc((size_t)steps + 1, sizeof(float));
    float *cap_of_feat = (float *)calloc((size_t)steps + 1, sizeof(float));
    float *trans_of_total = (float *)calloc((size_t)steps + 1, sizeof(float));

    for (; current_mem < total_mem; current_mem += memory_step) {
        if ((uint64_t)current_mem > (uint64_t)V * float_attr_len * sizeof(float)) {
            node_num_feat = V;
        } else {
            node_num_feat = (int32_t)((current_steps + 1) * (memory_step / (int64_t)(float_attr_len * sizeof(float))));
        }
        if ((uint64_t)current_mem > edge_mem_prefix[V - 1]) {
            node_num_topo = V;
        } else {
            /* std::lower_bound(prefix, prefix+V, current_mem) */
            int64_t lo = 0, hi = V;
            while (lo < hi) {
                int64_t mid = (lo + hi) / 2;
                if (edge_mem_prefix[mid] < (uint64_t)current_mem) lo = mid + 1; else hi = mid;
            }
            node_num_topo = (int32_t)lo;
        }
        if (node_num_topo < V) {
            uint64_t pref = node_num_topo > 0 ? edge_prefix[node_num_topo - 1] : 0;
            trans_of_topo[current_steps] = (float)(total_trans_of_topo * 1.0 / edge_prefix[V - 1] * pref);
            cap_of_topo[current_steps] = (float)(node_num_topo / Kg);
        }
        if (node_num_feat < V) {
            uint64_t pref = node_num_feat > 0 ? node_prefix[node_num_feat - 1] : 0;
            trans_of_feat[current_steps] = (float)(total_trans_of_feat * 1.0 / node_prefix[V - 1] * pref);
            cap_of_feat[current_steps] = (float)(node_num_feat / Kg);
        }
        current_steps++;
    }
    for (int64_t sidx = 1; sidx < steps; sidx++)
        trans_of_total[sidx] = trans_of_topo[sidx] + trans_of_feat[steps - 1 - sidx];
    int64_t max_sidx = 0;                                /* std::max_element: first maximum */
    for (int64_t s = 1; s <= steps; s++)
        if (trans_of_total[s] > trans_of_total[max_sidx]) max_sidx = s;
    *alpha_idx = (int32_t)max_sidx;
    *best_trans = trans_of_total[max_sidx];
    *node_capacity = (int32_t)(cap_of_feat[steps - 1 - max_sidx] + 1);   /* :763 */
    *edge_capacity = (int32_t)(cap_of_topo[max_sidx] + 1);               /* :764 */

    free(node_prefix); free(edge_prefix); free(edge_mem_prefix);
    free(trans_of_topo); free(trans_of_feat); free(cap_of_topo); free(cap_of_feat); free(trans_of_total);
}

/* ------------------------------------------------------------------------- */
/* S8/S9: FeatFillUp (GPUCache.cu:200-205), GraphCache                        */
/* (GPU_Memory_Graph_Storage.cu:14-34,98-133)                                  */
/* ------------------------------------------------------------------------- */
/* cache of clique GPU Ki: row r = features of QF[r*Kg + Ki].  Ranks >= V are
 * skipped (left untouched) -- the reference reads QF out of bounds there. */
void lo_feat_fill_up(int32_t capacity, int32_t float_attr_len, float *feature_cache,
                     const float *cpu_float_attrs, const int32_t *QF, int32_t Kg, int32_t Ki, int32_t V)
{
    for (int32_t r = 0; r < capacity; r++) {
        int64_t t = (int64_t)r * Kg + Ki;
        if (t >= V) continue;
        int32_t id = QF[t];
        memcpy(feature_cache + (int64_t)r * float_attr_len, cpu_float_attrs + (int64_t)id * float_attr_len,
               (size_t)float_attr_len * sizeof(float));
    }
}

/* Fragment of clique GPU Ki: frag_indptr[capacity+1] (exclusive scan of the
 * degrees of QT[r*Kg+Ki]) and frag_indices.  Call with frag_indices == NULL to
 * get only frag_indptr (and thereby the size to allocate). */
void lo_graph_cache(const int32_t *QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                    const int64_t *csr_node_index, const int32_t *csr_dst_node_ids,
                    int64_t *frag_indptr, int32_t *frag_indices)
{
    frag_indptr[0] = 0;
    for (int32_t r = 0; r < capacity; r++) {
        int64_t t = (int64_t)r * Kg + Ki;
        int64_t count = 0;
        if (t < V) { int32_t id = QT[t]; count = csr_node_index[id + 1] - csr_node_index[id]; }
        frag_indptr[r + 1] = frag_indptr[r] + count;
    }
    if (!frag_indices) return;
    for (int32_t r = 0; r < capacity; r++) {
        int64_t t = (int64_t)r * Kg + Ki;
        if (t >= V) continue;
        int32_t id = QT[t];
        int64_t count = csr_node_index[id + 1] - csr_node_index[id];
        for (int64_t i = 0; i < count; i++)
            frag_indices[frag_indptr[r] + i] = csr_dst_node_ids[csr_node_index[id] + i];
    }
}

/* ------------------------------------------------------------------------- */
/* S10: seed split (GPUGraphStore.cu:332-414)                                  */
/* ------------------------------------------------------------------------- */
/* part = partition_file[tid] if (have_partition && flag==1) else tid % G;
 * ids with part >= G are dropped.  out_ids must hold n entries per partition
 * (laid out [G][n]); out_num[G] receives the counts. */
void lo_split_seeds(const int32_t *ids, int32_t n, int32_t G, const int32_t *partition_index,
                    int32_t use_partition, int32_t *out_ids, int32_t *out_num)
{
    for (int32_t g = 0; g < G; g++) out_num[g] = 0;
    for (int32_t i = 0; i < n; i++) {
        int32_t tid = ids[i];
        int32_t part = (partition_index && use_partition == 1) ? partition_index[tid] : tid % G;
        if (part < G) out_ids[(int64_t)part * n + out_num[part]++] = tid;
    }
}

/* ------------------------------------------------------------------------- */
/* S11: step schedule (CUDA_IPC_Service.cu:66-134,136-138,219-259)             */
/* ------------------------------------------------------------------------- */
/* steps[0..2] = train/valid/test steps; *_bs[G] per-GPU batch sizes. */
void lo_coordinate(const int32_t *train_num, const int32_t *valid_num, const int32_t *test_num,
                   int32_t G, int32_t raw_batch_size, int32_t *steps,
                   int32_t *train_bs, int32_t *valid_bs, int32_t *test_bs)
{
    int32_t min_train_size = 1000000000;
    for (int32_t i = 0; i < G; i++) if (train_num[i] < min_train_size) min_train_size = train_num[i];
    int32_t train_step = (min_train_size - 1) / raw_batch_size;
    for (int32_t i = 0; i < G; i++) train_bs[i] = raw_batch_size;
    int32_t max_valid = 0, max_test = 0;
    for (int32_t i = 0; i < G; i++) if (valid_num[i] > max_valid) max_valid = valid_num[i];
    int32_t valid_step = (max_valid - 1) / 512 + 1;
    for (int32_t i = 0; i < G; i++) valid_bs[i] = (valid_num[i] - 1) / valid_step + 1;
    for (int32_t i = 0; i < G; i++) if (test_num[i] > max_test) max_test = test_num[i];
    int32_t test_step = (max_test - 1) / 512 + 1;
    for (int32_t i = 0; i < G; i++) test_bs[i] = (test_num[i] - 1) / test_step + 1;
    steps[0] = train_step; steps[1] = valid_step; steps[2] = test_step;
}

int32_t lo_get_max_step(const int32_t *steps, int32_t epoch)
{
    return ((steps[0] + steps[1]) * epoch) + steps[2];
}

int32_t lo_get_current_mode(const int32_t *steps, int32_t epoch, int32_t global_batch_id)
{
    if (global_batch_id < ((steps[0] + steps[1]) * epoch)) {
        int32_t e = global_batch_id % (steps[0] + steps[1]);
        return (e < steps[0]) ? LO_TRAINMODE : LO_VALIDMODE;
    }
    return LO_TESTMODE;
}

int32_t lo_get_local_batch_id(const int32_t *steps, int32_t epoch, int32_t global_batch_id)
{
    if (global_batch_id < ((steps[0] + steps[1]) * epoch)) {
        int32_t e = global_batch_id % (steps[0] + steps[1]);
        return (e < steps[0]) ? e : e - steps[0];
    }
    return (global_batch_id - ((steps[0] + steps[1]) * epoch)) % steps[2];
}

/* ------------------------------------------------------------------------- */
/* One whole batch in the canonical schedule (GPURunner::RunOnce order,       */
/* Server.cu:301-328; SURVEY.md section 8a "canonical-schedule restatement").  */
/* Resident single-partition form used by the parity tests and by bench.py's  */
/* cpu_baseline leg: CSR slot [P] only, optional feature map / caches.         */
/* ------------------------------------------------------------------------- */
typedef struct {
    /* graph */
    int32_t V; int32_t F;
    const int64_t *indptr; const int32_t *indices; const float *features;
    /* optional unified feature cache (may be NULL => all misses) */
    const int32_t *node_map; const float *const *cache_ptrs; int32_t cache_capacity;
    /* optional topology cache */
    const signed char *part_index_map; const int32_t *part_offset_map;
    const int64_t *const *frag_indptr; const int32_t *const *frag_indices; int32_t partition_count;
    /* scratch (caller allocated) */
    uint32_t *accessed_map; int32_t *position_map;
    int32_t *agg_src_ids; int32_t *agg_dst_ids;
    signed char *tmp_part_ind; int32_t *tmp_part_off; int32_t *cache_index;
    /* outputs (the 7 IPC buffers) */
    int32_t *sampled_ids; float *float_features; int32_t *labels;
    int32_t *agg_src_off; int32_t *agg_dst_off; int32_t *nc; int32_t *ec;
    /* hotness (pre-sampling) */
    uint64_t *node_access_time; uint64_t *edge_access_time;
} lo_batch_ctx;

/* mode: train/valid/test; is_presc: pre-sampling epoch (only even ops run,
 * Server.cu:284-299); gather!=0 runs the feature ops. */
void lo_run_batch(lo_batch_ctx *c, const int32_t *all_ids, const int32_t *all_labels, int32_t total_cap,
                  int32_t batch_size, int32_t counter, const int32_t *fanout, int32_t hops,
                  int32_t mode, int32_t is_presc, int32_t gather)
{
    int32_t P = c->partition_count;
    const int64_t **ip = (const int64_t **)calloc((size_t)P + 1, sizeof(void *));
    const int32_t **ix = (const int32_t **)calloc((size_t)P + 1, sizeof(void *));
    for (int32_t g = 0; g < P; g++) {
        ip[g] = c->frag_indptr ? c->frag_indptr[g] : NULL;
        ix[g] = c->frag_indices ? c->frag_indices[g] : NULL;
    }
    ip[P] = c->indptr; ix[P] = c->indices;

    lo_batch_generator_kernel(c->sampled_ids, c->labels, batch_size, counter, all_ids, all_labels, total_cap,
                              c->V, c->position_map, c->accessed_map, c->nc, c->ec, hops);
    for (int32_t op = 1; op <= 2 * hops + 1; op++) {
        if (op % 2 == 1) {
            if (is_presc || !gather) continue;
            if (c->node_map) lo_find_feat(c->sampled_ids, c->cache_index, c->nc, op, c->node_map);
            else {
                int32_t l = (op - 1) / 2;
                for (int32_t r = 0; r < c->nc[4 + 2 * l]; r++) c->cache_index[r] = -1;
            }
            lo_gather_features(c->features, c->cache_ptrs, c->F, c->sampled_ids, c->cache_index,
                               c->cache_capacity > 0 ? c->cache_capacity : 1, c->nc, c->float_features, c->V, op);
        } else {
            const signed char *pi = NULL; const int32_t *po = NULL;
            if (!is_presc && c->part_index_map) {
                const int32_t *input = (op == 2) ? c->sampled_ids : c->agg_src_ids + c->ec[2];
                lo_find_topo(input, c->tmp_part_ind, c->tmp_part_off, c->nc[2], c->part_index_map, c->part_offset_map);
                pi = c->tmp_part_ind; po = c->tmp_part_off;
            }
            lo_gpu_random_sampling(c->sampled_ids, op, ip, ix, pi, po, fanout[op / 2 - 1], P,
                                   c->agg_src_ids, c->agg_dst_ids, c->agg_src_off, c->agg_dst_off,
                                   c->accessed_map, c->position_map, c->nc, c->ec,
                                   is_presc ? c->edge_access_time : NULL, hops);
        }
    }
    /* Cache_Planner -> make_update_plan (Kernels.cu:759-783): train mode only */
    if (mode == LO_TRAINMODE) {
        lo_clear_pos_map(c->position_map, c->sampled_ids, c->nc, hops);
        if (is_presc && c->node_access_time) lo_hotness_measure(c->sampled_ids, c->nc, c->node_access_time, hops);
    }
    free(ip); free(ix);
}

/* ------------------------------------------------------------------------- */
/* The same batch with OpenMP ("reference-      */
/* semantics CPU oracle timed single-thread and OpenMP").  Resident form only  */
/* (whole CSR, no caches, train mode).  Per hop:                               */
/*   phase 1, parallel over slots: the draw of kernel_random_sampler_2         */
/*            (Kernels.cu:375-411) -- source row, degree, Thrust discard(idx), */
/*            neighbour -- parked in cand[idx] (-1: no edge);                  */
/*   phase 2, SERIAL in ascending slot order: bitmap test-and-set, node and    */
/*            edge append (:413-447) -- this is what fixes the canonical order; */
/*   phase 3, parallel over edges: construct_graph (:450-463);                 */
/* then the feature rows, parallel over rows (:662-702).  Output is byte for   */
/* byte what lo_run_batch produces (tests/test_oracle_batches.py).             */
/* cand: caller-allocated int32[max slots of a hop].                           */
/* ------------------------------------------------------------------------- */
void lo_run_batch_omp(lo_batch_ctx *c, const int32_t *all_ids, const int32_t *all_labels, int32_t total_cap,
                      int32_t batch_size, int32_t counter, const int32_t *fanout, int32_t hops,
                      int32_t gather, int32_t *cand)
{
    lo_batch_generator_kernel(c->sampled_ids, c->labels, batch_size, counter, all_ids, all_labels, total_cap,
                              c->V, c->position_map, c->accessed_map, c->nc, c->ec, hops);
    int32_t *nc = c->nc, *ec = c->ec;
    for (int32_t h = 1; h <= hops; h++) {
        const int32_t op_id = 2 * h, count = fanout[h - 1];
        const int32_t *input_ids = (op_id == 2) ? c->sampled_ids : c->agg_src_ids + ec[2];
        const int32_t total = nc[2] * count;
#pragma omp parallel for schedule(static, 4096)
        for (int32_t idx = 0; idx < total; idx++) {
            int32_t dst = -1;
            const int32_t src = input_ids[idx / count];
            if (src >= 0) {
                const int64_t start = c->indptr[src];
                const int32_t col = (int32_t)(c->indptr[src + 1] - start);
                if (idx % count < col) {
                    dst = c->indices[start + (int64_t)lo_sample_index(idx, col)];
                    if (dst < 0) dst = -1;
                }
            }
            cand[idx] = dst;
        }
        for (int32_t idx = 0; idx < total; idx++) {
            const int32_t dst = cand[idx];
            if (dst < 0) continue;
            const uint32_t bit = 1u << (dst % 32), old = c->accessed_map[dst / 32];
            c->accessed_map[dst / 32] = old | bit;
            if (!(old & bit)) {
                const int32_t p = nc[0] + nc[1]++;
                c->sampled_ids[p] = dst;
                c->position_map[dst] = p;
            }
            const int32_t e = ec[0] + ec[1]++;
            c->agg_src_ids[e] = dst;
            c->agg_dst_ids[e] = input_ids[idx / count];
        }
        {
            const int32_t e0 = ec[0], n = ec[1];
#pragma omp parallel for schedule(static, 4096)
            for (int32_t i = 0; i < n; i++) {
                c->agg_src_off[e0 + i] = c->position_map[c->agg_src_ids[e0 + i]];
                c->agg_dst_off[e0 + i] = c->position_map[c->agg_dst_ids[e0 + i]];
            }
        }
        lo_update_counter(nc, ec, op_id, 0, hops);
    }
    if (gather && c->features && c->F > 0) {
        const int32_t n = nc[5 + 2 * hops];
#pragma omp parallel for schedule(static, 256)
        for (int32_t r = 0; r < n; r++) {
            const int32_t id = c->sampled_ids[r];
            if (id >= 0) memcpy(c->float_features + (int64_t)r * c->F, c->features + (int64_t)(id % c->V) * c->F, (size_t)c->F * sizeof(float));
        }
    }
    lo_clear_pos_map(c->position_map, c->sampled_ids, nc, hops);   /* make_update_plan, train mode */
}
