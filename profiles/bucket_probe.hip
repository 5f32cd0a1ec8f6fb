// Microbenchmark for VERDICT r02 "next 5": would claiming inside XCD-L2-sized table windows make hop 3 of the sampler faster?
//
// Hop 3 at the papers100M shape: ~3.0 M (dst, slot) candidates, a u64[V = 111 M] position table (0.89 GB).  Today every
// candidate does one random table load (pre-filter) and ~1.4 M of them one memory-side atomicMin (k_sample: 110 us with the
// neighbour draws).  The bucketed formulation would
//   P1  histogram the candidates by window  w = dst >> 19  (512 K entries = 4 MiB of table per window, 212 windows)
//   P2  scatter the (dst, slot) pairs into window order
//   P3  let ONE workgroup per window resolve its pairs with the window's table lines L2-resident
//       (best case measured here: plain load + plain store per pair inside the window -- no atomics, no ordering logic;
//        and the realistic case: load + atomicMin per pair inside the window)
//   P4  write the 4-byte slot states back by slot index (scattered stores)
// against
//   B0  the table accesses of today's formulation: random load + atomicMin over the whole table, candidates in slot order.
// The neighbour draws (row descriptor + adjacency loads) are the same on both sides and left out.
//
//   hipcc -O3 --offload-arch=gfx950 profiles/bucket_probe.hip -o /tmp/bucket_probe && /tmp/bucket_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kWinShift = 19;            // 2^19 entries x 8 B = 4 MiB of table per window
constexpr int kBins = 256;               // >= number of windows (212 at V = 111 M)
constexpr int kChunk = 4096;             // candidates per workgroup in the partition passes

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; return x ^ (x >> 33); }

// candidates like the synthetic graph's neighbours: 80 % from the Zipf-like skew floor(V u^3), spread by an odd multiplier
__global__ void k_make(int32_t* dst, int32_t n, uint32_t V)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t h = mix(0x9E3779B97F4A7C15ull + i);
        const uint64_t a = h >> 32;
        uint64_t x = a;
        if ((h & 0xFF) < 205) x = (((a * a) >> 32) * a) >> 32;
        const uint64_t r = (x * (uint64_t)V) >> 32;
        dst[i] = (int32_t)((r * 2654435761ull + 12345ull) % V);
    }
}

// B0: today's table accesses in slot order
__global__ void k_claim_direct(const int32_t* __restrict__ dst, int32_t n, unsigned long long* table, int32_t* __restrict__ state, uint32_t epoch)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t d = dst[i];
        const unsigned long long mine = ((unsigned long long)epoch << 32) | 0x80000000u | (uint32_t)i;
        unsigned long long cur = table[d];
        if (cur > mine) { const unsigned long long old = atomicMin(table + d, mine); cur = old > mine ? mine : old; }
        state[i] = cur == mine ? -1 : (int32_t)(uint32_t)cur;
    }
}

// P1: per-chunk histogram of the window ids
__global__ __launch_bounds__(256) void k_hist(const int32_t* __restrict__ dst, int32_t n, int32_t* __restrict__ hist /* [bins][chunks] */, int32_t chunks)
{
    __shared__ int32_t h[kBins];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int32_t c = blockIdx.x, lo = c * kChunk, hi = min(lo + kChunk, n);
    for (int32_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[dst[i] >> kWinShift], 1);
    __syncthreads();
    hist[threadIdx.x * chunks + c] = h[threadIdx.x];
}
// exclusive scan of hist (bin-major), one workgroup; also window start offsets
__global__ __launch_bounds__(1024) void k_scan(int32_t* hist, int32_t total, int32_t* win_start, int32_t chunks)
{
    __shared__ int32_t part[1024];
    const int32_t per = (total + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, total);
    int32_t s = 0;
    for (int32_t i = lo; i < hi; i++) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int32_t run = 0; for (int i = 0; i < 1024; i++) { const int32_t v = part[i]; part[i] = run; run += v; } }
    __syncthreads();
    int32_t run = part[threadIdx.x];
    for (int32_t i = lo; i < hi; i++) { const int32_t v = hist[i]; hist[i] = run; if (i % chunks == 0) win_start[i / chunks] = run; run += v; }
    if (threadIdx.x == 1023) win_start[kBins] = run;
}
// P2: scatter (dst, slot) pairs into window order (order inside a window is not slot order: the pairs carry their slot)
__global__ __launch_bounds__(256) void k_scatter(const int32_t* __restrict__ dst, int32_t n, const int32_t* __restrict__ hist, int32_t chunks,
                                                 unsigned long long* __restrict__ pairs)
{
    __shared__ int32_t base[kBins];
    const int32_t c = blockIdx.x, lo = c * kChunk, hi = min(lo + kChunk, n);
    base[threadIdx.x] = hist[threadIdx.x * chunks + c];
    __syncthreads();
    for (int32_t i = lo + threadIdx.x; i < hi; i += 256) {
        const int32_t d = dst[i];
        const int32_t p = atomicAdd(&base[d >> kWinShift], 1);
        pairs[p] = ((unsigned long long)(uint32_t)d << 32) | (uint32_t)i;
    }
}
// P3: one workgroup per window walks its pairs; MODE 0: load + atomicMin (realistic), MODE 1: plain load + plain store (best case:
// the window belongs to this workgroup alone, so no atomic is needed -- ordering logic not included)
template <int MODE>
__global__ __launch_bounds__(1024) void k_claim_window(const unsigned long long* __restrict__ pairs, const int32_t* __restrict__ win_start,
                                                       unsigned long long* table, int32_t* __restrict__ out_state /* window order */, uint32_t epoch)
{
    const int32_t w = blockIdx.x, lo = win_start[w], hi = win_start[w + 1];
    for (int32_t k = lo + threadIdx.x; k < hi; k += 1024) {
        const unsigned long long pr = pairs[k];
        const int32_t d = (int32_t)(pr >> 32);
        const uint32_t slot = (uint32_t)pr;
        const unsigned long long mine = ((unsigned long long)epoch << 32) | 0x80000000u | slot;
        unsigned long long cur = table[d];
        if (MODE == 0) {
            if (cur > mine) { const unsigned long long old = atomicMin(table + d, mine); cur = old > mine ? mine : old; }
        } else {
            if (cur > mine) { table[d] = mine; cur = mine; }
        }
        out_state[k] = cur == mine ? -1 : (int32_t)(uint32_t)cur;
    }
}
// P4: slot states back to slot order
__global__ void k_writeback(const unsigned long long* __restrict__ pairs, const int32_t* __restrict__ st, int32_t n, int32_t* __restrict__ state)
{
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) state[(uint32_t)pairs[k]] = st[k];
}

int main(int argc, char** argv)
{
    const uint32_t V = 111059956u;
    const int32_t n = argc > 1 ? atoi(argv[1]) : 3000000;
    const int reps = 20;
    int32_t *dst, *state, *hist, *win_start, *st2;
    unsigned long long *table, *pairs;
    const int32_t chunks = (n + kChunk - 1) / kChunk;
    CK(hipMalloc(&dst, (size_t)n * 4)); CK(hipMalloc(&state, (size_t)n * 4)); CK(hipMalloc(&st2, (size_t)n * 4));
    CK(hipMalloc(&hist, (size_t)kBins * chunks * 4)); CK(hipMalloc(&win_start, (kBins + 1) * 4));
    CK(hipMalloc(&table, (size_t)V * 8)); CK(hipMalloc(&pairs, (size_t)n * 8));
    CK(hipMemset(table, 0xFF, (size_t)V * 8));
    k_make<<<2048, 256>>>(dst, n, V);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& fn) {
        std::vector<float> t;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0)); fn(0xFFFFFF00u - r); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%-62s median %7.1f us   min %7.1f us\n", name, t[reps / 2], t[0]);
        return t[reps / 2];
    };
    printf("n = %d candidates, table u64[%u] = %.2f GB, %d windows of 4 MiB, %d reps each (new epoch per rep: every rep claims afresh)\n",
           n, V, V * 8.0 / 1e9, (int)((V >> kWinShift) + 1), reps);
    const float b0 = timeit("B0  today: random load + atomicMin, slot order", [&](uint32_t ep) { k_claim_direct<<<2048, 256>>>(dst, n, table, state, ep); });
    const float p1 = timeit("P1  histogram by window (+ scan)", [&](uint32_t) { k_hist<<<chunks, 256>>>(dst, n, hist, chunks); k_scan<<<1, 1024>>>(hist, kBins * chunks, win_start, chunks); });
    const float p2 = timeit("P2  scatter (dst, slot) pairs into window order", [&](uint32_t) {
        k_hist<<<chunks, 256>>>(dst, n, hist, chunks); k_scan<<<1, 1024>>>(hist, kBins * chunks, win_start, chunks); k_scatter<<<chunks, 256>>>(dst, n, hist, chunks, pairs); }) - p1;
    const int wins = (int)((V >> kWinShift) + 1);
    const float p3a = timeit("P3a one workgroup per window: load + atomicMin in the window", [&](uint32_t ep) { k_claim_window<0><<<wins, 1024>>>(pairs, win_start, table, st2, ep); });
    const float p3b = timeit("P3b one workgroup per window: plain load + store (best case)", [&](uint32_t ep) { k_claim_window<1><<<wins, 1024>>>(pairs, win_start, table, st2, ep); });
    const float p4 = timeit("P4  slot states back to slot order (scattered 4-byte stores)", [&](uint32_t) { k_writeback<<<2048, 256>>>(pairs, st2, n, state); });
    printf("\nbucketed, realistic  P1+P2+P3a+P4 = %.1f us   vs   B0 = %.1f us   (%+.1f us)\n", p1 + p2 + p3a + p4, b0, p1 + p2 + p3a + p4 - b0);
    printf("bucketed, best case  P1+P2+P3b+P4 = %.1f us   vs   B0 = %.1f us   (%+.1f us)\n", p1 + p2 + p3b + p4, b0, p1 + p2 + p3b + p4 - b0);
    printf("partition + write-back alone (P1+P2+P4) = %.1f us; what the window saves on the probes: B0 - P3a = %.1f us, B0 - P3b = %.1f us\n",
           p1 + p2 + p4, b0 - p3a, b0 - p3b);
    return 0;
}
