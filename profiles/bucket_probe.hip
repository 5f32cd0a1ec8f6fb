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
#include <hipcub/hipcub.hpp>
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
// window start offsets from the scanned histogram (bin-major: entry [w][0] is the first pair of window w)
__global__ void k_win_start(const int32_t* __restrict__ hist, int32_t chunks, int32_t total, int32_t* __restrict__ win_start, int32_t n)
{
    const int32_t w = threadIdx.x;
    if (w < kBins) win_start[w] = hist[w * chunks];
    if (w == kBins) win_start[kBins] = n;
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
__global__ __launch_bounds__(256) void k_claim_window(const unsigned long long* __restrict__ pairs, const int32_t* __restrict__ win_start,
                                                      unsigned long long* table, int32_t* __restrict__ out_state /* window order */, uint32_t epoch, int32_t wg_per_win)
{
    const int32_t w = blockIdx.x / wg_per_win, part = blockIdx.x % wg_per_win, lo = win_start[w], hi = win_start[w + 1];
    for (int32_t k = lo + part * 256 + threadIdx.x; k < hi; k += 256 * wg_per_win) {
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
// the share of today's candidates the pre-filter settles without an atomic (nodes found in earlier hops, smaller claims):
// ~53 % at the papers100M shape (3.0 M slots, 1.4 M claims) -- pre-populate the table with final positions for them
__global__ void k_preseed(const int32_t* __restrict__ dst, int32_t n, unsigned long long* table, uint32_t epoch)
{
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (mix(i * 7919ull + 13) % 100 < 53) table[dst[i]] = ((unsigned long long)epoch << 32) | (uint32_t)(i & 0xFFFFF);
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
    void* scan_tmp = nullptr; size_t scan_bytes = 0;
    int32_t* hist_scanned;
    CK(hipMalloc(&hist_scanned, (size_t)kBins * chunks * 4));
    CK(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, hist, hist_scanned, kBins * chunks));
    CK(hipMalloc(&scan_tmp, scan_bytes));
    auto timeit = [&](const char* name, auto&& setup, auto&& fn) {
        std::vector<float> t;
        for (int r = 0; r < reps; r++) {
            const uint32_t ep = 0xFFFFFF00u - r;
            setup(ep);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); fn(ep); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%-78s median %7.1f us   min %7.1f us\n", name, t[reps / 2], t[0]);
        return t[reps / 2];
    };
    auto none = [&](uint32_t) {};
    auto preseed = [&](uint32_t ep) { k_preseed<<<2048, 256>>>(dst, n, table, ep); };
    auto partition_count = [&]() { k_hist<<<chunks, 256>>>(dst, n, hist, chunks); CK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, hist, hist_scanned, kBins * chunks));
                                   k_win_start<<<1, 512>>>(hist_scanned, chunks, kBins * chunks, win_start, n); };
    const int wins = (int)((V >> kWinShift) + 1);
    printf("n = %d candidates, table u64[%u] = %.2f GB, %d windows of 4 MiB, %d reps each (new table epoch per rep)\n", n, V, V * 8.0 / 1e9, wins, reps);
    printf("\n-- every candidate claims (upper bound of the atomics) --\n");
    const float b0a = timeit("B0  today: random load + atomicMin, slot order", none, [&](uint32_t ep) { k_claim_direct<<<2048, 256>>>(dst, n, table, state, ep); });
    printf("\n-- 53 %% of the candidates settled by the pre-filter load (today's mix at the papers100M shape: 3.0 M loads, ~1.4 M atomics) --\n");
    const float b0 = timeit("B0r today: random load + atomicMin where the pre-filter does not settle it", preseed, [&](uint32_t ep) { k_claim_direct<<<2048, 256>>>(dst, n, table, state, ep); });
    const float p1 = timeit("P1  histogram by window + device scan of the (window, chunk) counts", none, [&](uint32_t) { partition_count(); });
    const float p12 = timeit("P1+P2  ... + scatter of the (dst, slot) pairs into window order", none, [&](uint32_t) { partition_count(); k_scatter<<<chunks, 256>>>(dst, n, hist_scanned, chunks, pairs); });
    float p3a = 1e9f, p3b = 1e9f;
    for (int wpw : {1, 4, 16}) {
        char nm[128];
        snprintf(nm, sizeof(nm), "P3a %2d workgroup(s) per window: load + atomicMin inside the window", wpw);
        p3a = std::min(p3a, timeit(nm, preseed, [&](uint32_t ep) { k_claim_window<0><<<wins * wpw, 256>>>(pairs, win_start, table, st2, ep, wpw); }));
        snprintf(nm, sizeof(nm), "P3b %2d workgroup(s) per window: plain load + store (no atomics: best case)", wpw);
        p3b = std::min(p3b, timeit(nm, preseed, [&](uint32_t ep) { k_claim_window<1><<<wins * wpw, 256>>>(pairs, win_start, table, st2, ep, wpw); }));
    }
    const float p4 = timeit("P4  slot states back to slot order (scattered 4-byte stores)", none, [&](uint32_t) { k_writeback<<<2048, 256>>>(pairs, st2, n, state); });
    printf("\nbucketed, atomics kept     P1+P2+P3a+P4 = %6.1f us   vs   B0r = %6.1f us   (%+.1f us)\n", p12 + p3a + p4, b0, p12 + p3a + p4 - b0);
    printf("bucketed, no atomics (best) P1+P2+P3b+P4 = %6.1f us   vs   B0r = %6.1f us   (%+.1f us)\n", p12 + p3b + p4, b0, p12 + p3b + p4 - b0);
    printf("partition + write-back alone (P1+P2+P4) = %.1f us (P1 = %.1f); what the windows save on the probes: B0r - P3a = %.1f us, B0r - P3b = %.1f us; all-claim bound B0 = %.1f us\n",
           p12 + p4, p1, b0 - p3a, b0 - p3b, b0a);
    return 0;
}
