// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the mini-batch hot path.
//
// Reference semantics restated (liayan/Legion-1, src/):
//   S1 batch_generator                Kernels.cu:68-96      -> k_seed
//   S2 update_counter                 Kernels.cu:112-150    -> folded into k_seed / k_write (last tile)
//   S3 kernel_random_sampler_2        Kernels.cu:342-448    -> k_sample + k_mark + k_write
//   S3' kernel_pre_sampler_optimized  Kernels.cu:468-564    -> k_sample<PRESC>
//   S4 construct_graph                Kernels.cu:450-463    -> k_write (both sides)
//   S5 zero_copy_with_aggregated_cache Kernels.cu:662-702   -> k_gather (+ k_row_ptrs in front of a cached gather)
//   S6 FindFeat/FindTopo (BGHT find)  GPUCache.cu:387-461   -> direct-mapped int32/int8[V] tables
//   S7 ClearPosMap / HotnessMeasure   Kernels.cu:750-756, GPUCache.cu:227-235
//
// Design (DESIGN.md has the long form):
//  * The reference's output ORDER depends on LDS/global atomicAdd races.  We produce the
//    canonical schedule (serial, slot-index ascending) deterministically: a hop is
//      k_sample : every slot draws its neighbour (same Thrust minstd arithmetic), parks it in
//                 cand[idx] and claims the node with atomicMin(pos[dst], PROVISIONAL|idx), so the
//                 LOWEST slot that touches a new node wins -- exactly the serial order.  Slot state
//                 aux[idx]: -1 claim pending / won, >= 0 the neighbour's known final position,
//                 <= -2 lost to slot -2-x.  Repeated draws of a row are settled in-wave; a claim that
//                 replaces a larger slot's claim writes that slot's state ("you lost to me");
//      k_mark   : streaming pass over the states: a slot still at -1 kept its claim = a new node; it gets its
//                 rank among the new nodes of its tile, the tile its count (no table probe); a workgroup runs a contiguous
//                 chunk of tiles and leaves each tile's (edges, new nodes) prefix inside the chunk + the chunk totals;
//      k_write  : every workgroup scans k_mark's <= 2048 chunk totals in LDS (no scan launch; + one 8-byte in-chunk prefix per tile),
//                 then ordered compaction (wave ballot + popcount prefix, one LDS exchange per tile)
//                 appends edges / new nodes at their canonical positions and both COO offsets -- an edge
//                 that lost its claim follows loser -> winner through the slot states and computes the
//                 winner's position from that tile's prefix + rank; it also sets the next hop's states
//                 to -1; the workgroup of the last tile applies update_counter (S2).
//    Three launches per hop (round 1: four), 11 per 3-hop batch with k_seed and the gather.
//  * One u64[V] "position table" replaces accessed_map (bitmap) + position_map.  Entry =
//    (epoch << 32) | value, epoch = 0xFFFFFFFF - batch serial, so entries of older batches compare
//    GREATER than anything of the running batch: they are stale without ever being cleared (no
//    V/8-byte memset, no ClearPosMap scatter).  value: 0x80000000|idx = claimed in the running hop,
//    else the final index in sampled_ids.
//  * Row descriptors (start, degree) of a tile are fetched once per source row and staged in
//    LDS -- the reference re-reads both int64 indptr words in each of the `count` lanes.
//  * RNG: x = 48271^(idx+1) mod (2^31-1).  Per thread: one table lookup and one Mersenne
//    mul-mod per tile instead of Thrust's discard() chain of 2*log2(idx) 64-bit `%`.
//    The final fp64 divide/multiply/truncate is kept verbatim -- it is what makes k bit exact.
//  * All loop bounds come from device counters; launches are sized by static upper bounds, so
//    there is not a single device->host copy in the batch (the reference does 7).
#include "internal.h"
#include <hipcub/hipcub.hpp>
#include <mutex>
#include <cstring>

#include "audit_hooks.h"

namespace legion {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// thrust::uniform_int_distribution<int>(0, deg-1) fed with x = minstd value (Kernels.cu:402-405;
// thrust/random/detail/uniform_int_distribution.inl:73-89, uniform_real_distribution.inl:71-79)
__device__ inline int32_t sample_index(uint32_t x, int32_t deg)
{
    double result = (double)(uint32_t)(x - 1u);
    result /= 2147483646.0;                  // 1.0 + double(max - min), max-min = 2147483645
    return (int32_t)(result * (double)deg + 0.0);
}

__device__ inline uint32_t fdiv(uint32_t n, const FastDiv& d)
{
    return d.d == 1 ? n : (uint32_t)(((uint64_t)n * d.m) >> d.s);
}

__device__ inline int lane_id() { return threadIdx.x & 63; }
__device__ inline int wave_id() { return threadIdx.x >> 6; }

// level (offset,size) slots of the generalised counter layout
__device__ inline int32_t total_nodes(const int32_t* nc, int32_t hops) { return nc[5 + 2 * hops]; }

// ------------------------------------------------------------------------------------------------
// S1 + S2(op 0): seed batch
// ------------------------------------------------------------------------------------------------
// Kernel's `batch_size` is the launcher's clamped `size` -- the reference passes `size`
// (Kernels.cu:227), so the read offset is size*counter (restated, not "fixed").
// SELF (captured batch graphs): the batch cursor and the table epoch live in device memory (BatchCtl),
// advanced by k_advance at the end of every graph launch, so that the captured launch has no per-batch
// arguments.  Host-driven launches pass both as arguments and publish them for the kernels that follow.
template <bool SELF>
__global__ __launch_bounds__(kBlock) void k_seed(int32_t* __restrict__ batch_ids, int32_t* __restrict__ labels,
                                                 int32_t batch_size, int32_t size, int32_t counter,
                                                 const int32_t* __restrict__ all_ids,
                                                 const int32_t* __restrict__ all_labels, int32_t total_cap,
                                                 pos_t* __restrict__ pos_map, uint32_t epoch,
                                                 BatchCtl* __restrict__ ctl, int32_t* __restrict__ nc,
                                                 int32_t* __restrict__ ec, int32_t* __restrict__ aux_next,
                                                 int32_t f_next, int32_t aux_cap)
{
    int32_t idx = threadIdx.x + blockDim.x * blockIdx.x;
    if (SELF) {
        counter = ctl->counter;
        epoch = ctl->epoch;
        // Kernels.cu:224 on the device (int64: counter is not bounded by the host here)
        const int64_t done = (int64_t)batch_size * counter;
        size = (done + batch_size >= total_cap) ? (int32_t)max((int64_t)0, min((int64_t)batch_size, (int64_t)total_cap - done)) : batch_size;
    } else if (idx == 0) {
        ctl->counter = counter;
        ctl->epoch = epoch;
    }
    if (idx < size) {
        int32_t g = size * counter + idx;
        if (g >= total_cap) {
            batch_ids[idx] = -1;
            labels[idx] = -1;
        } else {
            int32_t src_id = all_ids[g % total_cap];
            batch_ids[idx] = src_id;
            // position_map[src_id] = idx (Kernels.cu:92).  The reference assumes distinct seeds (:67); with
            // duplicates (link-prediction triples) its serial order lets the LAST occurrence win, so do the
            // same deterministically: the largest idx of the running epoch survives.
            const pos_t mine = pos_entry(epoch, (uint32_t)idx);
            pos_t cur = pos_map[src_id];
            while ((uint32_t)(cur >> kPosShift) != epoch || cur < mine) {
                const pos_t seen = atomicCAS(pos_map + src_id, cur, mine);
                if (seen == cur) break;
                cur = seen;
            }
            labels[idx] = all_labels[g % total_cap];
        }
    }
    if (idx < 16) { // cudaMemsetAsync(counters) + update_counter(op 0), Kernels.cu:220-221,118-127
        int32_t nv = 0;
        if (idx == 0 || idx == 2 || idx == 4) nv = size;
        nc[idx] = nv;
        ec[idx] = 0;
    }
    // slot states of hop 1 start as "claim pending" (-1), see k_sample
    const int64_t n_init = min((int64_t)max(size, 0) * f_next, (int64_t)aux_cap);
    for (int64_t i = idx; i < n_init; i += (int64_t)gridDim.x * blockDim.x) aux_next[i] = -1;
}
// fallback when a hop's fan-out differs from what the previous launch prepared the slot states for
__global__ void k_fill_aux(const int32_t* __restrict__ nc, int32_t count, int32_t* __restrict__ aux, int32_t aux_cap)
{
    const int64_t n = min((int64_t)nc[2] * count, (int64_t)aux_cap);
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) aux[i] = -1;
}
__global__ void k_set_cursor(BatchCtl* ctl, int32_t counter, uint32_t epoch) { ctl->counter = counter; ctl->epoch = epoch; }
// end of a captured batch: next batch, next (smaller) epoch
__global__ void k_advance(BatchCtl* ctl) { ctl->counter += 1; ctl->epoch -= 1; }

// S7: ClearPosMap (Kernels.cu:750-756) has no kernel here: position-table entries carry the batch epoch
// in their upper 32 bits, so entries of older batches are simply stale (see the table format below).

// ------------------------------------------------------------------------------------------------
// S3: sampler, pass 1 -- draw + claim
// ------------------------------------------------------------------------------------------------
struct SampleArgs {
    CsrTables csr;
    const int32_t* sampled_ids;
    const int32_t* agg_src_ids;
    const int32_t* nc;
    const int32_t* ec;
    pos_t* pos_map;
    int32_t* cand;
    int32_t* aux;
    int32_t* tile_edge;
    unsigned long long* edge_access_time;
    const BatchCtl* ctl;       // table epoch of the running batch
    const uint32_t* pow_tab;   // pow_tab[m] = 48271^(m+1), m < kTile
    uint32_t a_tile;           // 48271^kTile
    uint32_t a_step;           // 48271^(kTile * gridDim.x)
    FastDiv fdiv;              // / count
    int32_t count;
    int32_t op_id;
    int32_t window;            // lanes to look back for a repeated draw of the same row: min(count - 1, 8)
    int32_t prefilter_from_op; // first op_id whose claims are preceded by the pre-filter load (4: hop 2; hop 1 never)
};

// One slot's probe + claim on the position table.  Returns the slot's state: -1 = claim pending / won, >= 0 = the neighbour's known final
// position, <= -2 = lost to slot -2 - x.
__device__ inline int32_t claim_slot(const SampleArgs& a, uint32_t epoch, int32_t dst, int32_t idx)
{
    // claim: lowest idx wins.  Entries of older batches have a larger epoch field, i.e. compare
    // greater: unseen.  A stale (larger) pre-filter read only costs a redundant atomic.
    const pos_t prov0 = pos_entry(epoch, kProvisional);
    const pos_t mine = prov0 | (pos_t)(uint32_t)idx;
    // (plain loads: a non-temporal hint on this pre-filter load costs +11 % of k_sample, on the neighbour load
    // nothing, profiles/r02_sampler_experiments.md)
    // hop 1: nearly every neighbour is new, so the pre-filter load would only add a dependent round trip in front
    // of the claim -- go straight to the atomic (it returns the exact entry either way)
    pos_t cur = (a.op_id < a.prefilter_from_op) ? ~(pos_t)0 : a.pos_map[dst];
    if (cur > mine) {
        const pos_t old = atomicMin(a.pos_map + dst, mine);
        if (old > mine) {
            // the table holds this slot's claim now.  If it replaced a claim of this hop (a larger
            // slot that got there first), that slot has lost for good: tell it who beat it.  Its own
            // thread left aux at -1 (pending) and never writes it again, so this is the only store.
            if ((uint32_t)(old >> kPosShift) == epoch) a.aux[(uint32_t)old & kPosValueMask] = -2 - idx;
            cur = mine;
        } else {
            cur = old; // a smaller entry arrived between the load and the atomic: exact value
        }
    }
    // final positions are only written by earlier launches: if we see one it is exact
    if (cur < prov0) return (int32_t)((uint32_t)cur & kPosValueMask);
    // a smaller claim of this hop is in the table: this slot has lost for good (claims only
    // decrease).  Point at that slot; if it loses later too, its own aux points further, and
    // k_write follows the chain to the winner.
    if (cur < mine) return -2 - (int32_t)((uint32_t)cur & kPosValueMask);
    return -1;
}

template <bool PRESC, bool PARTITIONED>
__global__ __launch_bounds__(kBlock) void k_sample(SampleArgs a)
{
    __shared__ const int32_t* s_row[kTile + 2]; // pointer to the first neighbour of the staged row
    __shared__ int32_t s_deg[kTile + 2];
    __shared__ int32_t s_src[kTile + 2];
    __shared__ int32_t s_cnt[kBlock / 64];

    const int32_t N = a.nc[2];
    const int32_t f = a.count;
    const int32_t total = N * f; // int32 like the reference (Kernels.cu:375)
    const int32_t* __restrict__ input = (a.op_id == 2) ? a.sampled_ids : a.agg_src_ids + a.ec[2];
    const int32_t n_tiles = (total + kTile - 1) / kTile;
    const int tid = threadIdx.x;
    const uint32_t epoch = a.ctl->epoch;

    if ((int32_t)blockIdx.x >= n_tiles) return;

    // per-thread RNG state: x[s] = 48271^(tile*kTile + tid + 256*s + 1)
    uint32_t x[kTile / kBlock];
    {
        uint32_t base = powmod31(a.a_tile, (uint64_t)blockIdx.x); // uniform per workgroup
#pragma unroll
        for (int s = 0; s < kTile / kBlock; s++) x[s] = mulmod31(base, a.pow_tab[tid + kBlock * s]);
    }

    for (int32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int32_t tile_start = tile * kTile;
        const int32_t tile_end = min(tile_start + kTile, total);
        const int32_t i0 = (int32_t)fdiv((uint32_t)tile_start, a.fdiv);
        const int32_t i_last = (int32_t)fdiv((uint32_t)(tile_end - 1), a.fdiv);
        const int32_t nrows = i_last - i0 + 1;

        // stage the row descriptors of this tile in LDS (one global fetch per source row)
        for (int32_t r = tid; r < nrows; r += kBlock) {
            const int32_t src = input[i0 + r];
            const int32_t* rowp = nullptr;
            int32_t deg = -1;
            if (src >= 0) {
                const int64_t* ip = a.csr.indptr + src;
                const int32_t* ix = a.csr.indices;
                int8_t owner = -1;
                if (PARTITIONED && !PRESC) owner = a.csr.topo_owner[src]; // FindTopo fused (GPUCache.cu:434-443)
                if (owner >= 0) {   // cached row: chunk tables of the owner's fragment (local HBM or xGMI peer)
                    const int32_t row = a.csr.topo_row[src];
                    ip = a.csr.frag_indptr[owner * a.csr.ip_nch + (row >> a.csr.row_shift)] + (row & ((1 << a.csr.row_shift) - 1));
                    const int64_t start = ip[0];
                    ix = a.csr.frag_indices[owner * a.csr.ix_nch + (int32_t)(start >> a.csr.edge_shift)];
                    rowp = ix + (start & ((1ll << a.csr.edge_shift) - 1));
                    deg = (int32_t)(ip[1] - start); // int32 truncation as in Kernels.cu:393,396
                } else {
                    const int64_t start = ip[0];
                    rowp = ix + start;
                    deg = (int32_t)(ip[1] - start);
                }
            }
            s_row[r] = rowp;
            s_deg[r] = deg;
            s_src[r] = src;
        }
        __syncthreads();

        int32_t cnt = 0;
#pragma unroll
        for (int s = 0; s < kTile / kBlock; s++) {
            const int32_t idx = tile_start + tid + kBlock * s;
            int32_t dst = -1, known = -1, j = 0, r = 0;
            if (idx < tile_end) {
                const uint32_t i = fdiv((uint32_t)idx, a.fdiv);
                j = idx - (int32_t)i * f;
                r = (int32_t)i - i0;
                const int32_t deg = s_deg[r];
                if (j < deg) { // deg == -1 for padded (-1) sources; Kernels.cu:385,399
                    dst = s_row[r][sample_index(x[s], deg)];
                    if (dst < 0) dst = -1;
                }
            }
            // Draws are with replacement, so the f slots of a row repeat neighbours (f = 5 of ~14: every 7th
            // slot).  The slots of a row sit in adjacent lanes: a lane that finds its neighbour in an earlier
            // lane of the same row has lost to it for good -- no table probe, no claim, and k_mark skips it too.
            int32_t dup = 0;
            for (int d = 1; d <= a.window; d++) { // uniform trip count, executed by the whole wave
                const int32_t o = __shfl_up(dst, d);
                if (d <= j && d <= lane_id() && o == dst) dup = d; // keeps the earliest match: short chains
            }
            if (dst >= 0) {
                if (PRESC) atomicAdd(a.edge_access_time + s_src[r], 1ull); // Kernels.cu:525
                if (dup) known = -2 - (idx - dup);
                else known = claim_slot(a, epoch, dst, idx);
                cnt++;
            }
            if (idx < tile_end) {
                a.cand[idx] = dst;
                // aux[idx] was initialised to -1 ("claim pending") by the previous launch.  A pending slot must not
                // store here: the slot that replaces its claim writes aux[idx] from another XCD, and two L2s
                // holding different dirty bytes for one address would be written back in no defined order.
                if (dst < 0) a.aux[idx] = 0;            // no edge: never read as an edge, not counted as a winner
                else if (known != -1) a.aux[idx] = known;
            }
        }
        // tile edge count
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
        if (lane_id() == 0) s_cnt[wave_id()] = cnt;
        __syncthreads();
        if (tid == 0) {
            int32_t t = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) t += s_cnt[w];
            a.tile_edge[tile] = t;
        }
#pragma unroll
        for (int s = 0; s < kTile / kBlock; s++) x[s] = mulmod31(x[s], a.a_step);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// S3 pass 2 -- rank the winners inside their tile, count new nodes per tile
// ------------------------------------------------------------------------------------------------
// A slot whose state is still -1 after k_sample kept its claim: it discovered a new node.  Its state becomes
// "winner, r-th new node of this tile" (enc_win): with the per-tile counts that is the node's final position,
// computable by ANY workgroup of k_write -- which is what lets k_write resolve the edges that lost their claim
// itself (round 1 needed a fourth launch per hop, k_resolve, for that).
constexpr int32_t kWinBase = 0x40000000;   // loser states are -2 - slot with slot < 2^30; winner ranks sit below them

// (cached gather, same box, papers100M shape, 25 % of the rows cached: lookup pass k_row_ptrs + k_gather 416-423 us; k_gather_lookup<U> with U = 1: 455,
// U = 2: 412-414, U = 4: 418 (profiles/r04_cached_gather.md) -- the probes themselves (1.94 M random 4-byte reads of a 444 MB map) bound it: U = 2 is built)
__device__ inline int32_t enc_win(int32_t r) { return -2 - (kWinBase + r); }
__device__ inline bool is_win(int32_t v) { return v <= -2 - kWinBase; }
__device__ inline int32_t win_rank(int32_t v) { return -2 - v - kWinBase; }

// Tile prefixes without a scan launch and without every workgroup of k_write reading every tile count: k_mark gives each of its
// workgroups a CONTIGUOUS chunk of T = ceil(tiles / workgroups) tiles, so a workgroup knows the (edges, new nodes) counted in front of each of
// its tiles INSIDE its chunk (tile_pre) and the chunk's totals (chunk_tot); k_write scans the <= kMaxChunks chunk totals in LDS (8-16 KB of
// shared reads per workgroup instead of every tile count: 35-70 KB at 4-9 k tiles, re-read by all 1536 workgroups at once -- one such prefix
// build cost 6.8 / 16.3 us per launch at the papers100M / products hop 3, profiles/r04_sampler.md) and adds tile_pre[t] of any tile it needs.
// kMaxChunks (internal.h) >= the largest k_mark grid (256 CUs x 8 workgroups); GPUMemoryPool_AllocateScratch sizes chunk_tot with it.
__host__ __device__ inline FastDiv make_fastdiv(uint32_t div)
{
    FastDiv f;
    f.d = div ? div : 1;
    if (f.d == 1) { f.m = 0; f.s = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < f.d) l++;
    f.m = (uint32_t)(((1ull << (31 + l)) / f.d) + 1ull);
    f.s = 31 + l;
    return f;
}

__global__ __launch_bounds__(kBlock) void k_mark(const int32_t* __restrict__ nc, const int32_t* __restrict__ ec,
                                                 int32_t count, int32_t* __restrict__ aux, const int32_t* __restrict__ tile_edge,
                                                 int32_t* __restrict__ tile_node, int2* __restrict__ tile_pre,
                                                 int2* __restrict__ chunk_tot, HopState* __restrict__ hs)
{
    constexpr int S = kTile / kBlock, W = kBlock / 64;
    __shared__ int32_t s_c[S * W];
    const int32_t total = nc[2] * count;
    const int32_t n_tiles = (total + kTile - 1) / kTile;
    const int lane = lane_id(), wave = wave_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // hop-start snapshot of the counters: k_write's last tile applies update_counter in place, so
        // its other workgroups must not read the live nc/ec
        HopState h;
        h.edge_base = ec[0]; h.node_base = nc[0]; h.n_edges = 0; h.n_nodes = 0;
        h.in_off = ec[2]; h.n_in = nc[2]; h.slots = total; h.pad = 0;
        *hs = h;
    }
    const int32_t T = (n_tiles + (int32_t)gridDim.x - 1) / (int32_t)gridDim.x;     // tiles per chunk (k_write derives the same T)
    const int32_t t0 = (int32_t)blockIdx.x * T, t1 = min(t0 + T, n_tiles);
    if (t0 >= n_tiles) return;
    // the slot states of the NEXT tile of this workgroup are fetched before the current one is ranked: the loads overlap the
    // two barriers and the stores of the current tile (a workgroup runs 1-5 tiles; the pass is a chain of short latencies)
    int32_t nxt[S];
#pragma unroll
    for (int s = 0; s < S; s++) {
        const int64_t idx = (int64_t)t0 * kTile + threadIdx.x + kBlock * s;
        nxt[s] = idx < total ? aux[idx] : 0;
    }
    int32_t te_next = threadIdx.x == 0 ? tile_edge[t0] : 0, run_e = 0, run_n = 0;   // thread 0 keeps the chunk's running sums
    for (int32_t tile = t0; tile < t1; tile++) {
        bool win[S];
        int32_t rk[S];
        const int32_t te = te_next;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int32_t idx = tile * kTile + threadIdx.x + kBlock * s;
            win[s] = idx < total && nxt[s] == -1;
        }
        {
            const int64_t nt = (int64_t)tile + 1;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const int64_t idx = nt * kTile + threadIdx.x + kBlock * s;
                nxt[s] = (nt < t1 && idx < total) ? aux[idx] : 0;
            }
            if (threadIdx.x == 0 && nt < t1) te_next = tile_edge[nt];
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            const unsigned long long b = __ballot(win[s]);
            rk[s] = __popcll(b & lt);
            if (lane == 0) s_c[s * W + wave] = __popcll(b);
        }
        __syncthreads();
        int32_t run = 0, before[S];
#pragma unroll
        for (int q = 0; q < S * W; q++) { // slot order inside a tile: s-major, then wave, then lane
#pragma unroll
            for (int s = 0; s < S; s++)
                if (q == s * W + wave) before[s] = run;
            run += s_c[q];
        }
#pragma unroll
        for (int s = 0; s < S; s++)
            if (win[s]) aux[tile * kTile + threadIdx.x + kBlock * s] = enc_win(before[s] + rk[s]);
        if (threadIdx.x == 0) {
            tile_node[tile] = run;
            tile_pre[tile] = make_int2(run_e, run_n);      // in front of this tile inside its chunk
            run_e += te; run_n += run;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) chunk_tot[blockIdx.x] = make_int2(run_e, run_n);
}

// ------------------------------------------------------------------------------------------------
// S2: update_counter (Kernels.cu:128-149), H-hop layout: nc[1] == n_nodes, ec[1] == n_edges of this hop
// ------------------------------------------------------------------------------------------------
__device__ inline void apply_update_counter(int32_t* nc, int32_t* ec, int32_t op_id, int32_t hops, int32_t n_nodes,
                                            int32_t n_edges)
{
    const int32_t hh = op_id / 2;
    nc[0] += n_nodes;
    nc[3 + 2 * hh] = nc[1 + 2 * hh] + nc[2 + 2 * hh];
    nc[4 + 2 * hh] = n_nodes;
    if (hh == hops) nc[5 + 2 * hh] = nc[3 + 2 * hh] + nc[4 + 2 * hh];
    nc[1] = 0;
    nc[2] = n_edges;
    ec[2 + hh] = (hh == 1 ? ec[3] : ec[1 + hh]) + n_edges;
    ec[2] = ec[0];
    ec[0] += n_edges;
    ec[1] = 0;
}

// ------------------------------------------------------------------------------------------------
// S3 pass 3 + S4 -- ordered compaction: edges, new nodes, both COO offsets; next hop's slot states
// ------------------------------------------------------------------------------------------------
struct WriteArgs {
    HopState* hs;
    int32_t* nc;
    int32_t* ec;
    int32_t hops;
    const int32_t* cand;
    const int32_t* aux;     // slot states after k_mark: >= 0 known position, winner rank (enc_win), -2 - <slot it lost to>
    const int32_t* tile_edge;
    const int32_t* tile_node;
    int32_t* sampled_ids;
    int32_t* agg_src_ids;
    int32_t* agg_src_off;
    int32_t* agg_dst_off;
    pos_t* pos_map;
    FastDiv fdiv;
    int32_t op_id;
    const BatchCtl* ctl;
    int32_t last_hop;       // positions of the nodes found in the last hop are never looked up through the table
    const int2* tile_pre;   // k_mark: (edges, new nodes) in front of a tile inside its chunk
    const int2* chunk_tot;  // k_mark: totals of chunk c = tiles [c * T, (c + 1) * T)
    int32_t mark_grid;      // workgroups of k_mark: T = ceil(tiles / mark_grid)
    int32_t* aux_next;      // slot states of the next hop (the other buffer), set to "claim pending" here
    int32_t next_count;     // fan-out of the next hop (0: none)
    int32_t aux_cap;
    int32_t ids_cap;        // elements of sampled_ids / agg_src_ids / agg_*_off (GPUMemoryPool::num_ids): bound of every store below
    int32_t V;              // entries of pos_map
};

// Every store of k_write is addressed through the tile prefix (s_chunk[] + tile_pre[]) that k_mark left behind.  With a correct
// k_mark the offsets are below the buffers' capacity by construction (num_ids = the sum of the static per-hop bounds); an experiment
// that skips or breaks the prefix build writes through uninitialised offsets -- round 4's timing-only variant did, and hung its run
// (profiles/r04_sampler.md).  The bound check makes such a variant drop the store instead of running away; it is one unsigned compare
// per store in a kernel that waits for memory (+0.3-0.6 us per launch: profiles/r05_sampler.md, r05_ab_bounded_stores.log).
#define LEGION_STORE_OK(i, cap) ((uint32_t)(i) < (uint32_t)(cap))

__global__ __launch_bounds__(kBlock) void k_write(WriteArgs a)
{
    constexpr int S = kTile / kBlock, W = kBlock / 64;
    // (edges, new nodes) counted in front of a tile = exclusive prefix over the chunk totals of k_mark, built once per workgroup in
    // LDS (<= kMaxChunks entries: one coalesced round of loads), + tile_pre[t] (one 8-byte read, issued next to the other loads of
    // the tile or of the losing edge that needs it).  No scan launch, no inter-workgroup hand-off (device-scope fences cost an L2
    // write-back + invalidate per XCD: profiles/r01_gather_sweep.md).
    __shared__ int2 s_chunk[kMaxChunks];
    __shared__ int32_t s_e[S * W];
    __shared__ int2 s_scan[W];
    __shared__ uint32_t s_div_t[3];
    const HopState h = *a.hs;
    const uint32_t epoch = a.ctl->epoch;
    const int32_t total = h.slots;
    const int32_t n_tiles = (total + kTile - 1) / kTile;
    const int lane = lane_id(), wave = wave_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (n_tiles == 0) { // empty hop: only the counters move
        if (blockIdx.x == 0 && threadIdx.x == 0) apply_update_counter(a.nc, a.ec, a.op_id, a.hops, 0, 0);
        return;
    }
    if ((int32_t)blockIdx.x >= n_tiles) return;
    const int32_t T = (n_tiles + a.mark_grid - 1) / a.mark_grid;   // tiles per chunk, as k_mark derived it
    if (threadIdx.x == 0) {                                        // one 64-bit divide per workgroup (T is device-side: no host round trip)
        const FastDiv f = make_fastdiv((uint32_t)T);
        s_div_t[0] = f.d; s_div_t[1] = f.m; s_div_t[2] = f.s;
    }
    const int32_t n_chunks = (n_tiles + T - 1) / T;                // <= mark_grid <= kMaxChunks
    {
        constexpr int PER = kMaxChunks / kBlock;                    // consecutive chunks per thread
        const int32_t q0 = (int32_t)threadIdx.x * PER;
        int2 v[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) v[u] = a.chunk_tot[min(q0 + u, n_chunks - 1)];   // unconditional: all in flight together
        int32_t se = 0, sn = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (q0 + u >= n_chunks) v[u] = make_int2(0, 0);
            se += v[u].x; sn += v[u].y;
        }
        int32_t ie = se, in = sn;
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t ue = __shfl_up(ie, o), un = __shfl_up(in, o);
            if (lane >= o) { ie += ue; in += un; }
        }
        if (lane == 63) s_scan[wave] = make_int2(ie, in);
        __syncthreads();
        int32_t run_e = ie - se, run_n = in - sn;
#pragma unroll
        for (int w = 0; w < W; w++)
            if (w < wave) { run_e += s_scan[w].x; run_n += s_scan[w].y; }
#pragma unroll
        for (int u = 0; u < PER; u++) {
            if (q0 + u < n_chunks) s_chunk[q0 + u] = make_int2(run_e, run_n);
            run_e += v[u].x; run_n += v[u].y;
        }
        __syncthreads();
    }
    FastDiv div_t;
    div_t.d = s_div_t[0]; div_t.m = s_div_t[1]; div_t.s = s_div_t[2];
    // new nodes in front of tile t (what an edge that lost its claim needs of its winner's tile); `pre` = tile_pre[t]
    auto nodes_before = [&](int32_t t, int32_t pre_n_in_chunk) { return s_chunk[fdiv((uint32_t)t, div_t)].y + pre_n_in_chunk; };

    for (int32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int2 own = a.tile_pre[tile], own_chunk = s_chunk[fdiv((uint32_t)tile, div_t)];
        const int32_t pre_e = own_chunk.x + own.x, pre_n = own_chunk.y + own.y;
        const int32_t tile_e = a.tile_edge[tile];
        const int32_t ebase = h.edge_base + pre_e;
        const int32_t nbase = h.node_base + pre_n;
        // ---- loads first, all S slots of the thread in flight together (nothing below this block reads global memory) ----
        int32_t c[S], so[S], dpos[S], w[S], re[S], wpre[S];
        const int32_t* __restrict__ pre_n_of = reinterpret_cast<const int32_t*>(a.tile_pre) + 1;   // tile_pre[t].y at [2 * t]
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int32_t idx = tile * kTile + threadIdx.x + kBlock * s;
            c[s] = (idx < total) ? a.cand[idx] : -1;
            so[s] = (idx < total) ? a.aux[idx] : 0;
        }
#pragma unroll
        for (int s = 0; s < S; s++) {
            dpos[s] = 0; w[s] = -1;
            if (c[s] == -1) continue;
            const int32_t idx = tile * kTile + threadIdx.x + kBlock * s;
            // dst-side offset = position of the slot's source node; for hops > 1 the sources are the
            // previous hop's edge endpoints, whose positions are that hop's src-side offsets: same value
            // as position_map[src] (construct_graph, Kernels.cu:457-461) without the random read.
            const int32_t i = (int32_t)fdiv((uint32_t)idx, a.fdiv);
            // hop 1: the seed's position.  That is i unless the seed list holds duplicates (link-prediction
            // triples), where the reference's position_map keeps the last occurrence -- read it (<= B*f probes).
            dpos[s] = (a.op_id == 2) ? (int32_t)((uint32_t)a.pos_map[a.sampled_ids[i]] & kPosValueMask) : a.agg_src_off[h.in_off + i];
            // lost the claim: first link of loser -> (earlier loser ->)* winner or known node
            // (with the in-chunk node prefix of that slot's tile, should it turn out to be the winner: same round trip)
            wpre[s] = 0;
            if (so[s] < -1 && !is_win(so[s])) { w[s] = -2 - so[s]; so[s] = a.aux[w[s]]; wpre[s] = pre_n_of[2 * (w[s] / kTile)]; }
        }
#pragma unroll
        for (int s = 0; s < S; s++) // longer chains are rare: follow them one slot at a time
            while (w[s] >= 0 && so[s] < -1 && !is_win(so[s])) { w[s] = -2 - so[s]; so[s] = a.aux[w[s]]; wpre[s] = pre_n_of[2 * (w[s] / kTile)]; }
#pragma unroll
        for (int s = 0; s < S; s++) {
            const unsigned long long be = __ballot(c[s] != -1);
            re[s] = __popcll(be & lt);
            if (lane == 0) s_e[s * W + wave] = __popcll(be);
        }
        __syncthreads();
        if (tile == n_tiles - 1 && threadIdx.x == 0) { // hop totals: update_counter (S2)
            const int32_t n_edges = pre_e + tile_e, n_nodes = pre_n + a.tile_node[tile];
            a.hs->n_edges = n_edges;
            a.hs->n_nodes = n_nodes;
            apply_update_counter(a.nc, a.ec, a.op_id, a.hops, n_nodes, n_edges);
        }
        // ---- stores ----
#pragma unroll
        for (int s = 0; s < S; s++) {
            if (c[s] == -1) continue;
            int32_t pe = 0;
            for (int q = 0; q < s * W + wave; q++) pe += s_e[q];
            const int32_t dst = c[s];
            const int32_t e = ebase + pe + re[s];
            if (!LEGION_STORE_OK(e, a.ids_cap)) continue;
            if (!a.last_hop) a.agg_src_ids[e] = dst;   // the next hop's input list; nothing reads it after the last hop
            a.agg_dst_off[e] = dpos[s];
            // src-side offset (construct_graph, Kernels.cu:456-460) = position of the sampled neighbour
            int32_t p = so[s];
            if (w[s] >= 0) {         // an edge that lost its claim
                if (p < -1)         // ... to a new node: the winner's position from ITS tile's prefix and its rank
                    p = h.node_base + nodes_before(w[s] / kTile, wpre[s]) + win_rank(p);
            } else if (is_win(p)) {  // this slot discovered the node: k_mark ranked it inside the tile
                p = nbase + win_rank(p);
                if (LEGION_STORE_OK(p, a.ids_cap)) a.sampled_ids[p] = dst;
                // the scattered table store is only needed when a later hop may look the node up by id
                if (!a.last_hop && LEGION_STORE_OK(dst, a.V)) a.pos_map[dst] = pos_entry(epoch, (uint32_t)p);
            }
            a.agg_src_off[e] = p;
        }
        // The next hop expands this tile's edges into slots [e * f', (e + 1) * f'): their states (the other aux
        // buffer, last read two launches ago) start as "claim pending"
        if (a.next_count > 0) {
            const int64_t lo = min((int64_t)pre_e * a.next_count, (int64_t)a.aux_cap);
            const int64_t hi = min((int64_t)(pre_e + tile_e) * a.next_count, (int64_t)a.aux_cap);
            for (int64_t q = lo + threadIdx.x; q < hi; q += kBlock) a.aux_next[q] = -1;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// S6: stand-alone lookups (API parity with FindFeat / FindTopo; the hot path fuses them)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_find_feat(const int32_t* __restrict__ sampled_ids,
                                                      int32_t* __restrict__ cache_offset,
                                                      const int32_t* __restrict__ nc, int32_t off_idx,
                                                      int32_t size_idx, const int32_t* __restrict__ feat_map)
{
    const int32_t off = nc[off_idx], n = nc[size_idx];
    for (int32_t r = threadIdx.x + blockDim.x * blockIdx.x; r < n; r += gridDim.x * blockDim.x) {
        int32_t id = sampled_ids[off + r];
        cache_offset[r] = (id < 0 || !feat_map) ? -1 : feat_map[id];
    }
}

__global__ __launch_bounds__(kBlock) void k_find_topo(const int32_t* __restrict__ input_ids,
                                                      int8_t* __restrict__ part_index,
                                                      int32_t* __restrict__ part_offset, int32_t n,
                                                      const int8_t* __restrict__ owner,
                                                      const int32_t* __restrict__ row)
{
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int32_t id = input_ids[i];
        bool ok = id >= 0 && owner;
        part_index[i] = ok ? owner[id] : (int8_t)-1;
        part_offset[i] = ok ? row[id] : -1;
    }
}

// ------------------------------------------------------------------------------------------------
// S5: feature gather from the unified cache / the backing table
// ------------------------------------------------------------------------------------------------
struct GatherKArgs {
    GatherArgs g;
    FastDiv div_c;   // / chunks-per-row
    FastDiv div_cap; // / cache_capacity
};

typedef float v4f __attribute__((ext_vector_type(4)));

// FindFeat + source selection for the rows of one gather launch (GPUCache.cu:387-400, Kernels.cu:672-691): one thread
// per ROW resolves id -> cache slot -> (clique GPU, chunk, row) -> address, or the backing-table row on a miss.  The
// gather then starts every row with one coalesced 8-byte load; done inside the gather, the map probe and the
// chunk-table load sit on the critical path of every 16-byte chunk (profiles/r01_gather_sweep.md).
__global__ __launch_bounds__(kBlock) void k_row_ptrs(GatherKArgs a)
{
    constexpr int U = 4; // rows per thread and step: U independent map probes in flight
    const GatherArgs& g = a.g;
    const int32_t off = g.off_idx < 0 ? 0 : g.nc[g.off_idx];
    const int32_t rows = g.nc[g.size_idx];
    const int32_t stride = gridDim.x * blockDim.x;
    for (int32_t r0 = threadIdx.x + blockDim.x * blockIdx.x; r0 < rows; r0 += stride * U) {
        int32_t id[U], gidx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t r = r0 + u * stride;
            id[u] = r < rows ? g.sampled_ids[off + r] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; u++) gidx[u] = (id[u] >= 0 && g.feat_map) ? g.feat_map[id[u]] : -1;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int32_t r = r0 + u * stride;
            if (r >= rows) continue;
            const float* src = nullptr;
            if (gidx[u] >= 0) {
                const uint32_t didx = fdiv((uint32_t)gidx[u], a.div_cap);
                const uint32_t fidx = (uint32_t)gidx[u] - didx * (uint32_t)g.cache_capacity;
                const float* chunk = g.shard_tab[didx * (uint32_t)g.nchunks + (fidx >> g.chunk_shift)];
                src = chunk + (int64_t)(fidx & ((1u << g.chunk_shift) - 1u)) * (g.shard_pitch > 0 ? g.shard_pitch : g.F);
            } else if (id[u] >= 0 && g.table) {
                src = g.table + (int64_t)(id[u] % g.total_num_nodes) * (g.table_pitch > 0 ? g.table_pitch : g.F);
            }
            g.row_ptr[r] = src;
        }
        if (g.hit_stats) { // feature_cache_hit (GPUCache.cu:130-147): one atomic per wave and step
            int32_t h = 0;
#pragma unroll
            for (int u = 0; u < U; u++) h += (r0 + u * stride < rows && gidx[u] >= 0) ? 1 : 0;
            for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o);
            if (lane_id() == 0 && h) atomicAdd(g.hit_stats, h);
        }
    }
    if (g.hit_stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(g.hit_stats + 1, rows);
}

// Row layout of the flat work space: q -> (row, 16-byte chunk) = (q / C, q % C) -- a wave covers 64 consecutive chunks whatever
// the row length.  (A row-aligned layout -- 2^k lanes per row, the lanes >= C idle -- lost 5 % at F = 100:
// profiles/r03_other_shapes.md; kept as profiles/r06_removed_experiments.patch.)
template <typename VT, int UNROLL, int NT>
__global__ __launch_bounds__(kBlock) void k_gather(GatherKArgs a)
{
    constexpr int VEC = sizeof(VT) / 4;
    const GatherArgs& g = a.g;
    const int32_t off = g.off_idx < 0 ? 0 : g.nc[g.off_idx];
    const int32_t rows = g.nc[g.size_idx];
    const int32_t C = g.F / VEC;
    const int64_t total = (int64_t)rows * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t pitch = g.table_pitch > 0 ? g.table_pitch : g.F;   // floats between two rows of the backing table
    if (g.rows_seen && blockIdx.x == 0 && threadIdx.x == 0) *g.rows_seen = rows; // launch-size feedback for later batches
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 < total; q0 += stride * UNROLL) {
        const VT* src[UNROLL];
        VT val[UNROLL];
        int64_t dsti[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t q = q0 + u * stride;
            src[u] = nullptr;
            dsti[u] = 0;
            if (q < total) {
                const uint32_t r = fdiv((uint32_t)q, a.div_c); // rows*C < 2^31 is checked by the launcher
                const uint32_t ch = (uint32_t)q - r * (uint32_t)C;
                if (g.dst_rows > 0 && off + (int32_t)r >= g.dst_rows) continue; // never write past the buffer
                dsti[u] = ((int64_t)(off + (int32_t)r) * g.F) / VEC + ch;
                if (g.row_ptr) { // resolved by k_row_ptrs
                    const float* p = g.row_ptr[r];
                    if (p) src[u] = reinterpret_cast<const VT*>(p) + ch;
                    continue;
                }
                const int32_t id = g.sampled_ids[off + (int32_t)r]; // no cache: the backing table row (Kernels.cu:689)
                if (id >= 0) src[u] = reinterpret_cast<const VT*>(g.table + (int64_t)(id % g.total_num_nodes) * pitch) + ch;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            if (src[u]) val[u] = (NT >= 2) ? __builtin_nontemporal_load(src[u]) : *src[u];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            if (src[u]) {
                if (NT >= 1) __builtin_nontemporal_store(val[u], reinterpret_cast<VT*>(g.dst) + dsti[u]);
                else reinterpret_cast<VT*>(g.dst)[dsti[u]] = val[u];
            }
    }
}

// FindFeat inside the gather, ONE probe per row and wave: the first lane a wave gives to a row resolves
// id -> cache slot -> (clique GPU, chunk, row) -> source address (GPUCache.cu:387-400, Kernels.cu:672-691) and hands the address to
// the row's other lanes by __shfl -- F = 128: two probes per wave instead of 64 (round 1 probed in every chunk lane: 426 us), no
// row_ptr round trip through memory, no second launch.  The probe chain (id, map, chunk table) sits in front of the row loads of
// the wave, so U work items per lane are resolved together: U chains overlap instead of following each other.
template <typename VT, int U>
__global__ __launch_bounds__(kBlock) void k_gather_lookup(GatherKArgs a)
{
    constexpr int VEC = sizeof(VT) / 4;
    const GatherArgs& g = a.g;
    const int32_t off = g.off_idx < 0 ? 0 : g.nc[g.off_idx];
    const int32_t rows = g.nc[g.size_idx];
    const int32_t C = g.F / VEC;
    const int64_t total = (int64_t)rows * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tpitch = g.table_pitch > 0 ? g.table_pitch : g.F, spitch = g.shard_pitch > 0 ? g.shard_pitch : g.F;
    const int lane = lane_id();
    if (g.rows_seen && blockIdx.x == 0 && threadIdx.x == 0) *g.rows_seen = rows;
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q0 - lane < total; q0 += stride * U) {   // whole waves stay in the loop: shuffles below
        uint32_t r[U], ch[U];
        int32_t id[U], gidx[U];
        bool lead[U], live[U];
        int src_lane[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t q = q0 + u * stride;
            live[u] = q < total;
            r[u] = fdiv((uint32_t)(live[u] ? q : 0), a.div_c);           // rows * C < 2^31 is checked by the launcher
            ch[u] = (uint32_t)(live[u] ? q : 0) - r[u] * (uint32_t)C;
            // first lane of this wave that works on row r: the row's chunk 0 if it sits in this wave, else lane 0
            const int64_t first = (int64_t)r[u] * C - (q - lane);
            src_lane[u] = first > 0 ? (int)first : 0;
            lead[u] = live[u] && lane == src_lane[u];
            id[u] = lead[u] ? g.sampled_ids[off + (int32_t)r[u]] : -1;
        }
#pragma unroll
        for (int u = 0; u < U; u++) gidx[u] = (lead[u] && id[u] >= 0) ? g.feat_map[id[u]] : -1;
        const VT* src[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float* p = nullptr;
            if (lead[u]) {
                if (gidx[u] >= 0) {
                    const uint32_t didx = fdiv((uint32_t)gidx[u], a.div_cap);
                    const uint32_t fidx = (uint32_t)gidx[u] - didx * (uint32_t)g.cache_capacity;
                    const float* chunk = g.shard_tab[didx * (uint32_t)g.nchunks + (fidx >> g.chunk_shift)];
                    p = chunk + (int64_t)(fidx & ((1u << g.chunk_shift) - 1u)) * spitch;
                } else if (id[u] >= 0 && g.table) {
                    p = g.table + (int64_t)(id[u] % g.total_num_nodes) * tpitch;
                }
            }
            const unsigned long long bits = (unsigned long long)(uintptr_t)p;
            const uint32_t lo = __shfl((uint32_t)bits, src_lane[u]), hi = __shfl((uint32_t)(bits >> 32), src_lane[u]);
            const float* rp = (const float*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
            src[u] = (live[u] && rp && !(g.dst_rows > 0 && off + (int32_t)r[u] >= g.dst_rows)) ? reinterpret_cast<const VT*>(rp) + ch[u] : nullptr;
        }
        VT val[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (src[u]) val[u] = __builtin_nontemporal_load(src[u]);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (src[u]) __builtin_nontemporal_store(val[u], reinterpret_cast<VT*>(g.dst) + ((int64_t)(off + (int32_t)r[u]) * g.F) / VEC + ch[u]);
    }
    // (feature_cache_hit, GPUCache.cu:130-147, is counted by the lookup PASS: the launcher sends every batch whose hit rate is
    // sampled through k_row_ptrs, which sees each row exactly once -- here a row that straddles two waves is probed by both)
}

// ------------------------------------------------------------------------------------------------
// S5, owner-computes exchange variant (SURVEY 5 option b; one process per GPU): rows whose cache slot lives on another
// clique member are not read in-kernel over xGMI; the requester lists them per owner, the owners gather them from their
// own HBM and the rows come back in one all-to-all (RCCL / hipMemcpyPeer).  Everything else is gathered as usual.
// ------------------------------------------------------------------------------------------------
struct ExchArgs {
    GatherArgs g;
    FastDiv div_cap;
    int32_t me;               // this GPU's index inside its clique
    int32_t Kg;
    int32_t* slot;            // scratch [rows]: FindFeat result of the count pass
    int32_t* counts;          // [Kg]: rows requested from each clique member (me: 0)
    int32_t* cursor;          // [Kg]: fill cursors (zeroed by the count pass)
    int32_t* req_row;         // out, grouped by owner: row inside the owner's shard
    int32_t* req_dst;         // out, same order: destination row of the batch
};
// pass 1: FindFeat (GPUCache.cu:387-400) per row + rows per owner.  One LDS histogram per workgroup, Kg atomics per workgroup.
__global__ __launch_bounds__(kBlock) void k_exch_count(ExchArgs a)
{
    __shared__ int32_t s_cnt[kMaxParts];
    const GatherArgs& g = a.g;
    const int32_t rows = g.nc[g.size_idx];
    if (threadIdx.x < kMaxParts) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int32_t r = threadIdx.x + blockDim.x * blockIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        const int32_t id = g.sampled_ids[r];
        const int32_t gidx = (id >= 0 && g.feat_map) ? g.feat_map[id] : -1;
        a.slot[r] = gidx;
        if (gidx >= 0) {
            const int32_t owner = (int32_t)fdiv((uint32_t)gidx, a.div_cap);
            if (owner != a.me) atomicAdd(&s_cnt[owner], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < a.Kg && s_cnt[threadIdx.x]) atomicAdd(a.counts + threadIdx.x, s_cnt[threadIdx.x]);
}
// pass 2: the request lists (contiguous per owner, owner-major) and the source address of every row that is served
// locally (own shard / backing table); rows owned by a peer get no address: the local gather skips them.
__global__ __launch_bounds__(kBlock) void k_exch_fill(ExchArgs a)
{
    __shared__ int32_t s_cnt[kMaxParts], s_base[kMaxParts];
    const GatherArgs& g = a.g;
    const int32_t rows = g.nc[g.size_idx];
    const int32_t per_wg = (rows + gridDim.x - 1) / gridDim.x;
    const int32_t r0 = blockIdx.x * per_wg, r1 = min(r0 + per_wg, rows);
    if (threadIdx.x < kMaxParts) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int32_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
        const int32_t gidx = a.slot[r];
        if (gidx >= 0) {
            const int32_t owner = (int32_t)fdiv((uint32_t)gidx, a.div_cap);
            if (owner != a.me) atomicAdd(&s_cnt[owner], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < a.Kg) {
        int32_t off = 0;                                   // lists are owner-major: offset of owner j = sum of counts before it
        for (int j = 0; j < (int)threadIdx.x; j++) off += a.counts[j];
        s_base[threadIdx.x] = off + (s_cnt[threadIdx.x] ? atomicAdd(a.cursor + threadIdx.x, s_cnt[threadIdx.x]) : 0);
        s_cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int32_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
        const int32_t gidx = a.slot[r];
        const int32_t id = g.sampled_ids[r];
        const float* src = nullptr;
        if (gidx >= 0) {
            const uint32_t owner = fdiv((uint32_t)gidx, a.div_cap);
            const uint32_t fidx = (uint32_t)gidx - owner * (uint32_t)g.cache_capacity;
            if ((int32_t)owner == a.me) {
                const float* chunk = g.shard_tab[owner * (uint32_t)g.nchunks + (fidx >> g.chunk_shift)];
                src = chunk + (int64_t)(fidx & ((1u << g.chunk_shift) - 1u)) * (g.shard_pitch > 0 ? g.shard_pitch : g.F);
            } else {
                const int32_t k = s_base[owner] + atomicAdd(&s_cnt[owner], 1);
                a.req_row[k] = (int32_t)fidx;
                a.req_dst[k] = r;
            }
        } else if (id >= 0 && g.table) {
            src = g.table + (int64_t)(id % g.total_num_nodes) * (g.table_pitch > 0 ? g.table_pitch : g.F);
        }
        g.row_ptr[r] = src;
    }
}
// owner side: rows list[0..n) of this GPU's shard -> out[n x F]; requester side: rows[k] -> dst[req_dst[k]]
template <typename VT, bool SCATTER>
__global__ __launch_bounds__(kBlock) void k_exch_rows(const float* const* __restrict__ shard_chunks, int32_t chunk_shift,
                                                      const int32_t* __restrict__ list, int32_t n, int32_t F, int32_t shard_pitch,
                                                      const float* __restrict__ in, float* __restrict__ out, FastDiv div_c,
                                                      int32_t out_rows)
{
    constexpr int VEC = sizeof(VT) / 4;
    const int32_t C = F / VEC;
    const int64_t total = (int64_t)n * C, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const uint32_t k = fdiv((uint32_t)q, div_c), ch = (uint32_t)q - k * (uint32_t)C;
        const int32_t row = list[k];
        if (SCATTER) {
            if (out_rows > 0 && row >= out_rows) continue; // never write past the buffer
            const VT v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(in + (int64_t)k * F) + ch);
            __builtin_nontemporal_store(v, reinterpret_cast<VT*>(out + (int64_t)row * F) + ch);
        } else {
            const float* src = shard_chunks[row >> chunk_shift] + (int64_t)(row & ((1 << chunk_shift) - 1)) * shard_pitch;
            const VT v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(src) + ch);
            __builtin_nontemporal_store(v, reinterpret_cast<VT*>(out + (int64_t)k * F) + ch);
        }
    }
}

// S7: HotnessMeasure (GPUCache.cu:227-235)
__global__ __launch_bounds__(kBlock) void k_hotness(const int32_t* __restrict__ ids, const int32_t* __restrict__ nc,
                                                    int32_t hops, unsigned long long* __restrict__ access,
                                                    int32_t* __restrict__ max_ids)
{
    const int32_t n = hops > 0 ? total_nodes(nc, hops) : nc[0];
    // max_ids_ (GPUCache.cu:294-296) kept on the device: no blocking D2H per batch
    if (max_ids && blockIdx.x == 0 && threadIdx.x == 0) atomicMax(max_ids, n);
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int32_t cid = ids[i];
        if (cid >= 0) atomicAdd(access + cid, 1ull);
    }
}

__global__ void k_rng_probe(const int32_t* idx, const int32_t* deg, int32_t* k, int32_t n)
{
    int32_t i = threadIdx.x + blockDim.x * blockIdx.x;
    if (i < n) k[i] = sample_index(powmod31(kA, (uint64_t)idx[i] + 1ull), deg[i]);
}

// ------------------------------------------------------------------------------------------------
// cache construction kernels (one-off; S8 / S9)
// ------------------------------------------------------------------------------------------------
__global__ void k_aggregate_access(unsigned long long* agg, const unsigned long long* add, int32_t n)
{   // GPUCache.cu:44-48
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < n; i += gridDim.x * blockDim.x) agg[i] += add[i];
}
__global__ void k_iota(int32_t* out, int32_t n)
{   // init_cache_order, GPUCache.cu:50-54
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = i;
}
__global__ void k_fill_i32(int32_t* p, int32_t v, int64_t n)
{
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill_i8(int8_t* p, int8_t v, int64_t n)
{
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
// InitPair (GPUCache.cu:103-108) scattered into the direct-mapped table: rank t -> slot
__global__ void k_build_feat_map(int32_t* feat_map, const int32_t* QF, int32_t capacity, int32_t Kg, int32_t V)
{
    const int64_t n = min((int64_t)capacity * Kg, (int64_t)V);
    for (int64_t t = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        feat_map[QF[t]] = (int32_t)((t % Kg) * capacity + t / Kg);
}
// InitIndexPair / InitOffsetPair (GPUCache.cu:88-100)
__global__ void k_build_topo_map(int8_t* owner, int32_t* row, const int32_t* QT, int32_t capacity, int32_t Kg,
                                 int32_t Ki, int32_t V)
{
    const int64_t n = min((int64_t)capacity * Kg, (int64_t)V);
    for (int64_t t = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        owner[QT[t]] = (int8_t)(t % Kg + Ki * Kg);
        row[QT[t]] = (int32_t)(t / Kg);
    }
}
// FeatFillUp (GPUCache.cu:200-205): cache row r of clique GPU Ki = features of QF[r*Kg + Ki]
__global__ void k_feat_fill_up(int32_t row0, int32_t rows, int32_t F, int32_t chunk_pitch, int32_t table_pitch, float* cache,
                               const float* table, const int32_t* QF, int32_t Kg, int32_t Ki, int32_t V)
{
    const int64_t n = (int64_t)rows * F;
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lr = i / F, c = i % F, t = (row0 + lr) * Kg + Ki;
        if (t >= V) continue;
        cache[lr * chunk_pitch + c] = table[(int64_t)QF[t] * table_pitch + c];
    }
}
// dense / pitched row copy (HBM replica of a table with a line-aligned row pitch)
__global__ void k_copy_rows_pitched(float* dst, int32_t dst_pitch, const float* src, int32_t src_pitch, int32_t F, int64_t rows)
{
    const int64_t n = rows * F;
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / F, c = i % F;
        dst[r * dst_pitch + c] = src[r * src_pitch + c];
    }
}
// GetNeighborCount (GPU_Memory_Graph_Storage.cu:14-20)
__global__ void k_neighbor_count(const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                                 const int64_t* indptr, int64_t* count_out)
{
    for (int32_t r = threadIdx.x + blockDim.x * blockIdx.x; r < capacity; r += gridDim.x * blockDim.x) {
        const int64_t t = (int64_t)r * Kg + Ki;
        int64_t c = 0;
        if (t < V) { int32_t id = QT[t]; c = indptr[id + 1] - indptr[id]; }
        count_out[r] = c;
    }
}
// TopoFillUp (GPU_Memory_Graph_Storage.cu:22-34); one wave per row, lanes stride the neighbours
__global__ void k_topo_fill_up(const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                               const int64_t* indptr, const int32_t* indices, const int64_t* frag_indptr,
                               int32_t* const* frag_chunks, int32_t edge_shift)
{
    const int32_t wave = (threadIdx.x + blockDim.x * blockIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int32_t r = wave; r < capacity; r += nwaves) {
        const int64_t t = (int64_t)r * Kg + Ki;
        if (t >= V) continue;
        const int32_t id = QT[t];
        const int64_t s = indptr[id], c = indptr[id + 1] - s, o = frag_indptr[r];
        int32_t* __restrict__ out = frag_chunks[o >> edge_shift] + (o & ((1ll << edge_shift) - 1)); // a row never leaves its chunk
        for (int64_t i = lane_id(); i < c; i += 64) out[i] = indices[s + i];
    }
}
// end offset of the last row starting before each chunk boundary: lower bound of the boundary in frag_indptr
__global__ void k_chunk_ends(const int64_t* frag_indptr, int32_t capacity, int32_t edge_shift, int32_t nch, int64_t* ends)
{
    const int32_t q = threadIdx.x + blockDim.x * blockIdx.x;
    if (q >= nch - 1) return;
    const int64_t boundary = (int64_t)(q + 1) << edge_shift;
    int32_t lo = 0, hi = capacity; // frag_indptr[capacity] = total >= boundary
    while (lo < hi) {
        const int32_t mid = lo + (hi - lo) / 2;
        if (frag_indptr[mid] < boundary) lo = mid + 1; else hi = mid;
    }
    ends[q] = frag_indptr[lo];
}
// GetEdgeMem (GPUCache.cu:35-41)
__global__ void k_edge_mem(const int32_t* order, uint64_t* edge_mem, int32_t V, const int64_t* indptr)
{
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < V; i += gridDim.x * blockDim.x) {
        int32_t id = order[i];
        edge_mem[i] = (uint64_t)(sizeof(int64_t) + sizeof(int32_t) * (indptr[id + 1] - indptr[id]));
    }
}
// PCM-free input of the cost model: 64-byte read transactions of the pre-sampling epoch's adjacency
// accesses, estimated from the edge hotness.  AT[t] = sampled edges of the rank-t row QT[t] (Kernels.cu:525: +1 per sampled
// edge).  Every sampled edge reads the row's 8-byte offset and one neighbour id (Kernels.cu:392-409); in 64-byte units a
// row of deg <= 14 ids has both in one line, a longer row needs a second one: weight = ceil((8 + 4 * min(deg, 16)) / 64),
// the "min(deg, .)" of the survey's formula taken at the 16 ids one transaction holds.  Integer only, deterministic.
__host__ __device__ inline uint64_t topo_transactions_of(uint64_t sampled_edges, int64_t deg)
{
    const int64_t ids = deg < 16 ? (deg < 0 ? 0 : deg) : 16;
    return sampled_edges * (uint64_t)((8 + 4 * ids + 63) / 64);
}
__global__ void k_topo_transactions(const int32_t* order, const uint64_t* hot, uint64_t* out, int32_t V, const int64_t* indptr)
{
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < V; i += gridDim.x * blockDim.x) {
        const int32_t id = order[i];
        out[i] = topo_transactions_of(hot[i], indptr[id + 1] - indptr[id]);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: launch wrappers
// ------------------------------------------------------------------------------------------------
FastDiv::FastDiv(uint32_t div)
{
    d = div ? div : 1;
    if (d == 1) { m = 0; s = 0; return; }
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    m = (uint32_t)(((1ull << (31 + l)) / d) + 1ull);
    s = 31 + l;
}

static int g_cu_count = 0;
static int cu_count()
{
    if (!g_cu_count) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
            g_cu_count = p.multiProcessorCount;
        if (g_cu_count <= 0) g_cu_count = 256;
    }
    return g_cu_count;
}
static inline int grid_for(int64_t work_items, int per_block, int blocks_per_cu = 8)
{
    int64_t need = (work_items + per_block - 1) / per_block;
    int64_t cap = (int64_t)cu_count() * blocks_per_cu;
    if (need < 1) need = 1;
    return (int)(need < cap ? need : cap);
}

// 48271^(m+1) table for m < kTile, one copy per physical device
static uint32_t* pow_table()
{
    static uint32_t* tabs[64] = {nullptr};
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (!tabs[dev]) {
        std::vector<uint32_t> h(kTile);
        uint32_t x = 1;
        for (int m = 0; m < kTile; m++) { x = mulmod31(x, kA); h[m] = x; }
        HIP_CHECK(hipMalloc(&tabs[dev], kTile * sizeof(uint32_t)));
        HIP_CHECK(hipMemcpy(tabs[dev], h.data(), kTile * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    LEGION_AUDIT_SHARE(tabs[dev], current_logical_device());   // one table per PHYSICAL device, whichever logical GPUs map to it
    return tabs[dev];
}

void launch_seed(hipStream_t s, int32_t* batch_ids, int32_t* labels, int32_t batch_size, int32_t size, int32_t counter,
                 const int32_t* all_ids, const int32_t* all_labels, int32_t total_cap, pos_t* pos_map,
                 uint32_t epoch, BatchCtl* ctl, bool self_driven, int32_t* nc, int32_t* ec, int32_t* aux_next,
                 int32_t f_next, int32_t aux_cap)
{
    const int32_t bound = self_driven ? batch_size : size;
    int blocks = bound > 0 ? (bound - 1) / kBlock + 1 : 1;
    LEGION_AUDIT_LAUNCH(s, "k_seed", LEGION_AW(batch_ids), LEGION_AW(labels), LEGION_AW(pos_map), LEGION_AW(ctl), LEGION_AW(nc), LEGION_AW(ec), LEGION_AW(aux_next), LEGION_AL(all_ids), LEGION_AL(all_labels));
    if (self_driven) k_seed<true><<<blocks, kBlock, 0, s>>>(batch_ids, labels, batch_size, size, counter, all_ids, all_labels, total_cap, pos_map, epoch, ctl, nc, ec, aux_next, f_next, aux_cap);
    else k_seed<false><<<blocks, kBlock, 0, s>>>(batch_ids, labels, batch_size, size, counter, all_ids, all_labels, total_cap, pos_map, epoch, ctl, nc, ec, aux_next, f_next, aux_cap);
    HIP_CHECK_LAST();
}
void launch_set_cursor(hipStream_t s, BatchCtl* ctl, int32_t counter, uint32_t epoch)
{
    LEGION_AUDIT_LAUNCH(s, "k_set_cursor", LEGION_AW(ctl));
    k_set_cursor<<<1, 1, 0, s>>>(ctl, counter, epoch);
    HIP_CHECK_LAST();
}
void launch_advance(hipStream_t s, BatchCtl* ctl)
{
    LEGION_AUDIT_LAUNCH(s, "k_advance", LEGION_AW(ctl));
    k_advance<<<1, 1, 0, s>>>(ctl);
    HIP_CHECK_LAST();
}
void warm_static_tables() { (void)pow_table(); (void)cu_count(); }

void launch_sample_hop(hipStream_t s, const CsrTables& csr, const SamplerBuffers& b, int32_t count, int32_t op_id,
                       int32_t hops, int32_t slots_bound, bool is_presc)
{
    if (count <= 0 || slots_bound <= 0) { LEGION_ARG_ERROR("GPU_Random_Sampling: empty hop"); return; }
    const int max_tiles = (slots_bound + kTile - 1) / kTile;
    // Workgroups per CU of the persistent tile loops.  The memory system is saturated by the scattered probes long before the CUs
    // are full: 4 workgroups (16 waves) per CU beat 8 by 2-4 % on hop 3 at every shape and tie on the small hops; 3 lose on hop 2
    // (same-box sweep, profiles/r04_sampler.md).
    constexpr int wg_per_cu = 4;
    const int grid = std::min(grid_for(max_tiles, 1, wg_per_cu), kMaxChunks);   // one chunk of tiles per k_mark workgroup (see k_mark)
    if (!b.aux_prepared) { // the previous launch prepared the slot states for another fan-out (or there was none)
        LEGION_AUDIT_LAUNCH(s, "k_fill_aux", LEGION_AW(b.aux), LEGION_AL(b.nc));
        k_fill_aux<<<grid_for(slots_bound, kBlock * 4), kBlock, 0, s>>>(b.nc, count, b.aux, b.aux_cap);
        HIP_CHECK_LAST();
    }
    SampleArgs a;
    a.csr = csr;
    a.sampled_ids = b.sampled_ids; a.agg_src_ids = b.agg_src_ids; a.nc = b.nc; a.ec = b.ec;
    a.pos_map = b.pos_map; a.cand = b.cand; a.aux = b.aux; a.tile_edge = b.tile_edge;
    a.edge_access_time = b.edge_access_time;
    a.ctl = b.ctl;
    a.pow_tab = pow_table();
    a.a_tile = powmod31(kA, kTile);
    a.a_step = powmod31(kA, (uint64_t)kTile * (uint64_t)grid);
    a.fdiv = FastDiv((uint32_t)count);
    a.count = count; a.op_id = op_id;
    a.window = std::min(count - 1, 8);
    a.prefilter_from_op = 4;   // hop 1 goes straight to the atomic (see k_sample; moving the boundary lost: profiles/r04_sampler.md)
    const bool part = csr.topo_owner != nullptr;
    // the whole CSR may be a peer's / the host's table; the fragment chunk tables, the id -> (owner, row) maps and every buffer of the pool are this GPU's
    LEGION_AUDIT_LAUNCH(s, "k_sample", LEGION_AW(a.pos_map), LEGION_AW(a.cand), LEGION_AW(a.aux), LEGION_AW(a.tile_edge), LEGION_AW(a.edge_access_time), LEGION_AL(a.sampled_ids), LEGION_AL(a.agg_src_ids), LEGION_AL(a.nc), LEGION_AL(a.ec), LEGION_AL(a.ctl), LEGION_AL(a.pow_tab), LEGION_AL(csr.frag_indptr), LEGION_AL(csr.frag_indices), LEGION_AL(csr.topo_owner), LEGION_AL(csr.topo_row), LEGION_AR(csr.indptr), LEGION_AR(csr.indices));
    if (is_presc) k_sample<true, false><<<grid, kBlock, 0, s>>>(a);
    else if (part) k_sample<false, true><<<grid, kBlock, 0, s>>>(a);
    else k_sample<false, false><<<grid, kBlock, 0, s>>>(a);
    HIP_CHECK_LAST();
    LEGION_AUDIT_LAUNCH(s, "k_mark", LEGION_AW(b.aux), LEGION_AW(b.tile_node), LEGION_AW(b.tile_pre), LEGION_AW(b.chunk_tot), LEGION_AW(b.hop_state), LEGION_AL(b.nc), LEGION_AL(b.ec), LEGION_AL(b.tile_edge));
    k_mark<<<grid, kBlock, 0, s>>>(b.nc, b.ec, count, b.aux, b.tile_edge, b.tile_node, b.tile_pre, b.chunk_tot, b.hop_state);
    HIP_CHECK_LAST();
    WriteArgs w;
    w.hs = b.hop_state; w.nc = b.nc; w.ec = b.ec; w.hops = hops; w.cand = b.cand; w.aux = b.aux; w.ctl = b.ctl; w.tile_edge = b.tile_edge; w.tile_node = b.tile_node;
    w.sampled_ids = b.sampled_ids; w.agg_src_ids = b.agg_src_ids; w.agg_src_off = b.agg_src_off;
    w.agg_dst_off = b.agg_dst_off; w.pos_map = b.pos_map; w.fdiv = a.fdiv; w.op_id = op_id; w.last_hop = (op_id / 2 == hops) ? 1 : 0;
    w.aux_next = b.aux_next; w.next_count = b.next_count; w.aux_cap = b.aux_cap; w.ids_cap = b.ids_cap; w.V = b.V;
    w.tile_pre = b.tile_pre; w.chunk_tot = b.chunk_tot; w.mark_grid = grid;
    // 17 KB of static LDS (the chunk prefix): 8 workgroups per CU
    const int wgrid = grid_for(max_tiles, 1, 8);
    LEGION_AUDIT_LAUNCH(s, "k_write", LEGION_AW(w.hs), LEGION_AW(w.nc), LEGION_AW(w.ec), LEGION_AW(w.sampled_ids), LEGION_AW(w.agg_src_ids), LEGION_AW(w.agg_src_off), LEGION_AW(w.agg_dst_off), LEGION_AW(w.pos_map), LEGION_AW(w.aux_next), LEGION_AL(w.cand), LEGION_AL(w.aux), LEGION_AL(w.tile_edge), LEGION_AL(w.tile_node), LEGION_AL(w.tile_pre), LEGION_AL(w.chunk_tot), LEGION_AL(w.ctl));
    k_write<<<wgrid, kBlock, 0, s>>>(w);
    HIP_CHECK_LAST();
}

void launch_find_feat(hipStream_t s, const int32_t* sampled_ids, int32_t* cache_offset, const int32_t* nc,
                      int32_t op_id, const int32_t* feat_map, int32_t bound)
{
    const int l = (op_id - 1) / 2;
    LEGION_AUDIT_LAUNCH(s, "k_find_feat", LEGION_AW(cache_offset), LEGION_AL(sampled_ids), LEGION_AL(nc), LEGION_AL(feat_map));
    k_find_feat<<<grid_for(bound, kBlock), kBlock, 0, s>>>(sampled_ids, cache_offset, nc, 3 + 2 * l, 4 + 2 * l, feat_map);
    HIP_CHECK_LAST();
}

void launch_find_topo(hipStream_t s, const int32_t* input_ids, int8_t* part_index, int32_t* part_offset,
                      int32_t batch_size, const int8_t* topo_owner, const int32_t* topo_row)
{
    if (batch_size <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_find_topo", LEGION_AW(part_index), LEGION_AW(part_offset), LEGION_AL(input_ids), LEGION_AL(topo_owner), LEGION_AL(topo_row));
    k_find_topo<<<grid_for(batch_size, kBlock), kBlock, 0, s>>>(input_ids, part_index, part_offset, batch_size, topo_owner, topo_row);
    HIP_CHECK_LAST();
}

void launch_gather(hipStream_t s, const GatherArgs& g, int32_t rows_bound)
{
    if (g.F <= 0 || rows_bound <= 0) return;
    GatherKArgs a;
    a.g = g;
    a.div_cap = FastDiv((uint32_t)(g.cache_capacity > 0 ? g.cache_capacity : 1));
    // the backing table may be the host's or a peer's; the shard chunk TABLE and the id -> slot map are this GPU's (the chunks it names: audited where it is published)
    LEGION_AUDIT_LAUNCH(s, "k_gather", LEGION_AW(g.dst), LEGION_AW(g.row_ptr), LEGION_AW(g.rows_seen), LEGION_AW(g.hit_stats), LEGION_AL(g.sampled_ids), LEGION_AL(g.nc), LEGION_AL(g.feat_map), LEGION_AL(g.shard_tab), LEGION_AR(g.table));
    // cache chunks are hipMalloc'ed (256-byte aligned) and hold whole rows: F % 4 == 0 keeps rows 16-byte aligned
    bool vec4 = (g.F % 4 == 0) && (((uintptr_t)g.table | (uintptr_t)g.dst) % 16 == 0);
    const int C = vec4 ? g.F / 4 : g.F;
    const int lanes = C;
    if ((int64_t)rows_bound * lanes >= (1ll << 31)) { LEGION_ARG_ERROR("get_feature_kernel: rows*F exceeds 2^31 work items"); return; }
    a.div_c = FastDiv((uint32_t)C);
    // Tuned on MI355X (profiles/r01_gather_sweep.md): non-temporal loads and stores (rows are read once and
    // written once: keep them out of L2/MALL).
    // One 16-byte chunk per lane and iteration.  Best measured with 2-3 iterations per lane: fewer workgroups
    // run long serial loops, more leave most of the grid empty (profiles/r01_gather_sweep.md).  The static row
    // bound is typically filled 15-60 %, so the grid is sized from the row count an earlier launch of this kind
    // reported (rows_seen, no host round trip), + 25 %; without a report: the bound, at most 512 workgroups per CU.
    int grid;
    // cached configurations: FindFeat + source selection run inside the gather (k_gather_lookup: one probe per row and wave, two work
    // items per lane).  A batch whose hit rate is sampled (every 500th) and a backing table in host memory take the lookup pass
    // (k_row_ptrs in front of k_gather): it counts every row exactly once / keeps the PCIe rows at full width.
    bool fused = false;
    if (g.row_ptr && !g.row_ptr_ready) {
        fused = !g.hit_stats && !g.table_on_host;
        if (!fused) {
            const int64_t est_rows = g.rows_hint > 0 ? std::min<int64_t>(rows_bound, (int64_t)g.rows_hint + g.rows_hint / 4 + 1024) : rows_bound;
            k_row_ptrs<<<grid_for(est_rows, kBlock * 4, 64), kBlock, 0, s>>>(a);
            HIP_CHECK_LAST();
        }
    }
    if (g.rows_hint > 0) {
        // (re-swept in round 2, profiles/r02_gather_grid_sweep.md: 1-4 iterations per lane and a 3-25 % margin all land
        // within the run-to-run spread of 323-342 us at the papers100M shape)
        const int64_t est = std::min<int64_t>(rows_bound, (int64_t)g.rows_hint + g.rows_hint / 4 + 1024);
        grid = grid_for(est * lanes, kBlock * (g.table_on_host ? 1 : 3), 8192); // rows over PCIe: latency-bound, maximise lanes in flight
    } else {
        grid = grid_for((int64_t)rows_bound * lanes, kBlock, 512);
    }
    if (fused) { if (vec4) k_gather_lookup<v4f, 2><<<grid, kBlock, 0, s>>>(a); else k_gather_lookup<float, 2><<<grid, kBlock, 0, s>>>(a); }
    else if (vec4) k_gather<v4f, 1, 2><<<grid, kBlock, 0, s>>>(a);
    else k_gather<float, 1, 2><<<grid, kBlock, 0, s>>>(a);
    HIP_CHECK_LAST();
}

void launch_exchange_plan(hipStream_t s, const GatherArgs& g, int32_t me, int32_t Kg, int32_t* slot, int32_t* counts,
                          int32_t* req_row, int32_t* req_dst, int32_t rows_bound)
{
    if (rows_bound <= 0) return;
    ExchArgs a;
    a.g = g;
    a.div_cap = FastDiv((uint32_t)(g.cache_capacity > 0 ? g.cache_capacity : 1));
    a.me = me; a.Kg = Kg; a.slot = slot; a.counts = counts; a.cursor = counts + kMaxParts;
    a.req_row = req_row; a.req_dst = req_dst;
    LEGION_AUDIT_LAUNCH(s, "k_exch_count/k_exch_fill", LEGION_AW(slot), LEGION_AW(counts), LEGION_AW(req_row), LEGION_AW(req_dst), LEGION_AW(g.row_ptr), LEGION_AL(g.sampled_ids), LEGION_AL(g.nc), LEGION_AL(g.feat_map), LEGION_AL(g.shard_tab), LEGION_AR(g.table));
    HIP_CHECK(hipMemsetAsync(counts, 0, 2 * kMaxParts * sizeof(int32_t), s));
    const int64_t est = g.rows_hint > 0 ? std::min<int64_t>(rows_bound, (int64_t)g.rows_hint + g.rows_hint / 4 + 1024) : rows_bound;
    const int grid = grid_for(est, kBlock * 4, 64);
    k_exch_count<<<grid, kBlock, 0, s>>>(a);
    HIP_CHECK_LAST();
    k_exch_fill<<<grid, kBlock, 0, s>>>(a);
    HIP_CHECK_LAST();
}
void launch_exchange_rows(hipStream_t s, bool scatter, const float* const* shard_chunks, int32_t chunk_shift, const int32_t* list,
                          int32_t n, int32_t F, int32_t shard_pitch, const float* in, float* out, int32_t out_rows)
{
    if (n <= 0 || F <= 0) return;
    if (shard_pitch <= 0) shard_pitch = F;
    const bool vec4 = (F % 4 == 0) && (((uintptr_t)in | (uintptr_t)out) % 16 == 0);
    const int C = vec4 ? F / 4 : F;
    if ((int64_t)n * C >= (1ll << 31)) { LEGION_ARG_ERROR("legion_exchange: rows*F exceeds 2^31 work items"); return; }
    const FastDiv dc((uint32_t)C);
    LEGION_AUDIT_LAUNCH(s, "k_exch_rows", LEGION_AW(out), LEGION_AL(shard_chunks), LEGION_AL(list), LEGION_AL(in));
    const int grid = grid_for((int64_t)n * C, kBlock * 2, 8192);
    if (scatter) {
        if (vec4) k_exch_rows<v4f, true><<<grid, kBlock, 0, s>>>(shard_chunks, chunk_shift, list, n, F, shard_pitch, in, out, dc, out_rows);
        else k_exch_rows<float, true><<<grid, kBlock, 0, s>>>(shard_chunks, chunk_shift, list, n, F, shard_pitch, in, out, dc, out_rows);
    } else {
        if (vec4) k_exch_rows<v4f, false><<<grid, kBlock, 0, s>>>(shard_chunks, chunk_shift, list, n, F, shard_pitch, in, out, dc, out_rows);
        else k_exch_rows<float, false><<<grid, kBlock, 0, s>>>(shard_chunks, chunk_shift, list, n, F, shard_pitch, in, out, dc, out_rows);
    }
    HIP_CHECK_LAST();
}

void launch_hotness(hipStream_t s, const int32_t* ids, const int32_t* nc, int32_t hops, unsigned long long* access,
                    int32_t* max_ids, int32_t bound)
{
    LEGION_AUDIT_LAUNCH(s, "k_hotness", LEGION_AW(access), LEGION_AW(max_ids), LEGION_AL(ids), LEGION_AL(nc));
    k_hotness<<<grid_for(bound, kBlock), kBlock, 0, s>>>(ids, nc, hops, access, max_ids);
    HIP_CHECK_LAST();
}

void launch_rng_probe(hipStream_t s, const int32_t* idx, const int32_t* deg, int32_t* k, int32_t n)
{
    if (n <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_rng_probe", LEGION_AW(k), LEGION_AL(idx), LEGION_AL(deg));
    k_rng_probe<<<(n + 255) / 256, 256, 0, s>>>(idx, deg, k, n);
    HIP_CHECK_LAST();
}

void launch_aggregate_access(hipStream_t s, unsigned long long* agg, const unsigned long long* add, int32_t n)
{
    LEGION_AUDIT_LAUNCH(s, "k_aggregate_access", LEGION_AW(agg), LEGION_AR(add));
    k_aggregate_access<<<grid_for(n, 256), 256, 0, s>>>(agg, add, n);
    HIP_CHECK_LAST();
}
void launch_iota(hipStream_t s, int32_t* out, int32_t n)
{
    LEGION_AUDIT_LAUNCH(s, "k_iota", LEGION_AW(out));
    k_iota<<<grid_for(n, 256), 256, 0, s>>>(out, n);
    HIP_CHECK_LAST();
}
void launch_fill_i32(hipStream_t s, int32_t* p, int32_t v, int64_t n)
{
    if (n <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_fill_i32", LEGION_AW(p));
    k_fill_i32<<<grid_for(n, 256), 256, 0, s>>>(p, v, n);
    HIP_CHECK_LAST();
}
void launch_fill_i8(hipStream_t s, int8_t* p, int8_t v, int64_t n)
{
    if (n <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_fill_i8", LEGION_AW(p));
    k_fill_i8<<<grid_for(n, 256), 256, 0, s>>>(p, v, n);
    HIP_CHECK_LAST();
}
void launch_build_feat_map(hipStream_t s, int32_t* feat_map, const int32_t* QF, int32_t capacity, int32_t Kg, int32_t V)
{
    launch_fill_i32(s, feat_map, -1, V);
    if (capacity <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_build_feat_map", LEGION_AW(feat_map), LEGION_AR(QF));
    k_build_feat_map<<<grid_for((int64_t)capacity * Kg, 256), 256, 0, s>>>(feat_map, QF, capacity, Kg, V);
    HIP_CHECK_LAST();
}
void launch_build_topo_map(hipStream_t s, int8_t* owner, int32_t* row, const int32_t* QT, int32_t capacity, int32_t Kg,
                           int32_t Ki, int32_t V)
{
    launch_fill_i8(s, owner, (int8_t)-1, V);
    launch_fill_i32(s, row, -1, V);
    if (capacity <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_build_topo_map", LEGION_AW(owner), LEGION_AW(row), LEGION_AR(QT));
    k_build_topo_map<<<grid_for((int64_t)capacity * Kg, 256), 256, 0, s>>>(owner, row, QT, capacity, Kg, Ki, V);
    HIP_CHECK_LAST();
}
void launch_feat_fill_up(hipStream_t s, int32_t row0, int32_t rows, int32_t F, int32_t chunk_pitch, int32_t table_pitch, float* chunk,
                         const float* table, const int32_t* QF, int32_t Kg, int32_t Ki, int32_t V)
{
    if (rows <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_feat_fill_up", LEGION_AW(chunk), LEGION_AR(table), LEGION_AR(QF));
    k_feat_fill_up<<<grid_for((int64_t)rows * F, 256), 256, 0, s>>>(row0, rows, F, chunk_pitch > 0 ? chunk_pitch : F, table_pitch > 0 ? table_pitch : F,
                                                                  chunk, table, QF, Kg, Ki, V);
    HIP_CHECK_LAST();
}
void launch_copy_rows_pitched(hipStream_t s, float* dst, int32_t dst_pitch, const float* src, int32_t src_pitch, int32_t F, int64_t rows)
{
    if (rows <= 0 || F <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_copy_rows_pitched", LEGION_AW(dst), LEGION_AR(src));
    k_copy_rows_pitched<<<grid_for(rows * F, 256), 256, 0, s>>>(dst, dst_pitch, src, src_pitch, F, rows);
    HIP_CHECK_LAST();
}
void launch_neighbor_count(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                           const int64_t* indptr, int64_t* count_out)
{
    if (capacity <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_neighbor_count", LEGION_AW(count_out), LEGION_AR(QT), LEGION_AR(indptr));
    k_neighbor_count<<<grid_for(capacity, 256), 256, 0, s>>>(QT, Kg, Ki, capacity, V, indptr, count_out);
    HIP_CHECK_LAST();
}
void launch_topo_fill_up(hipStream_t s, const int32_t* QT, int32_t Kg, int32_t Ki, int32_t capacity, int32_t V,
                         const int64_t* indptr, const int32_t* indices, const int64_t* frag_indptr,
                         int32_t* const* frag_chunks, int32_t edge_shift)
{
    if (capacity <= 0) return;
    LEGION_AUDIT_LAUNCH(s, "k_topo_fill_up", LEGION_AL(frag_indptr), LEGION_AL(frag_chunks), LEGION_AR(QT), LEGION_AR(indptr), LEGION_AR(indices));
    k_topo_fill_up<<<grid_for((int64_t)capacity * 64, 256), 256, 0, s>>>(QT, Kg, Ki, capacity, V, indptr, indices, frag_indptr, frag_chunks, edge_shift);
    HIP_CHECK_LAST();
}
void launch_chunk_ends(hipStream_t s, const int64_t* frag_indptr, int32_t capacity, int32_t edge_shift, int32_t nch, int64_t* ends)
{
    if (nch <= 1) return;
    LEGION_AUDIT_LAUNCH(s, "k_chunk_ends", LEGION_AW(ends), LEGION_AL(frag_indptr));
    k_chunk_ends<<<(nch + 63) / 64, 64, 0, s>>>(frag_indptr, capacity, edge_shift, nch, ends);
    HIP_CHECK_LAST();
}
void launch_edge_mem(hipStream_t s, const int32_t* order, uint64_t* edge_mem, int32_t V, const int64_t* indptr)
{
    LEGION_AUDIT_LAUNCH(s, "k_edge_mem", LEGION_AW(edge_mem), LEGION_AR(order), LEGION_AR(indptr));
    k_edge_mem<<<grid_for(V, 256), 256, 0, s>>>(order, edge_mem, V, indptr);
    HIP_CHECK_LAST();
}
void launch_topo_transactions(hipStream_t s, const int32_t* order, const uint64_t* hot, uint64_t* out, int32_t V, const int64_t* indptr)
{
    LEGION_AUDIT_LAUNCH(s, "k_topo_transactions", LEGION_AW(out), LEGION_AR(order), LEGION_AR(hot), LEGION_AR(indptr));
    k_topo_transactions<<<grid_for(V, 256), 256, 0, s>>>(order, hot, out, V, indptr);
    HIP_CHECK_LAST();
}

// thrust::sort_by_key(keys, ids, greater) (GPUCache.cu:631,651).  The reference's sort is not
// stable, so the order among equal keys is unspecified there; we use a stable descending radix
// sort seeded with ascending ids => ties in ascending id order (the oracle's documented rule).
void sort_by_hotness_desc(hipStream_t s, unsigned long long* keys, int32_t* ids, int32_t n)
{
    if (n <= 0) return;
    unsigned long long* keys_out = nullptr;
    int32_t* ids_out = nullptr;
    HIP_CHECK(hipMalloc(&keys_out, (size_t)n * sizeof(unsigned long long)));
    HIP_CHECK(hipMalloc(&ids_out, (size_t)n * sizeof(int32_t)));
    size_t tmp_bytes = 0;
    LEGION_AUDIT_LAUNCH(s, "hipcub::DeviceRadixSort", LEGION_AW(keys), LEGION_AW(ids));
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, keys, keys_out, ids, ids_out, n, 0, 64, s));
    void* tmp = nullptr;
    HIP_CHECK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(tmp, tmp_bytes, keys, keys_out, ids, ids_out, n, 0, 64, s));
    HIP_CHECK(hipMemcpyAsync(keys, keys_out, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(ids, ids_out, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipFree(tmp));
    HIP_CHECK(hipFree(keys_out));
    HIP_CHECK(hipFree(ids_out));
}

template <typename T>
static void inclusive_scan_t(hipStream_t s, const T* in, T* out, int32_t n)
{
    if (n <= 0) return;
    size_t tmp_bytes = 0;
    LEGION_AUDIT_LAUNCH(s, "hipcub::DeviceScan", LEGION_AW(out), LEGION_AR(in));
    HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, in, out, n, s));
    void* tmp = nullptr;
    HIP_CHECK(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    HIP_CHECK(hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, in, out, n, s));
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipFree(tmp));
}
void inclusive_scan_u64(hipStream_t s, const uint64_t* in, uint64_t* out, int32_t n) { inclusive_scan_t(s, in, out, n); }
void inclusive_scan_i64(hipStream_t s, const int64_t* in, int64_t* out, int32_t n) { inclusive_scan_t(s, in, out, n); }

} // namespace legion
