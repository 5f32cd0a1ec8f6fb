// synth.hip -- on-GPU generator of the synthetic datasets specified in legion-1_amd/synth.py
// (same closed forms, integer only, so host/numpy and device outputs are bit identical), and
// the streaming copy used by bench.py to report the measured HBM peak.
#include "internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>

#include "audit_hooks.h"

namespace legion {

__host__ __device__ inline uint64_t sm64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

constexpr uint64_t GEN_SEED = 0x1E610ull;
constexpr uint64_t S_DEG = (GEN_SEED << 32) ^ 0x0DE6ull;
constexpr uint64_t S_NBR = (GEN_SEED << 32) ^ 0x0EB2ull;
constexpr uint64_t S_FEAT = (GEN_SEED << 32) ^ 0xFEA7ull;
constexpr uint64_t S_LAB = (GEN_SEED << 32) ^ 0x1AB1ull;
constexpr int NBUCKET = 24;

struct Ladder { int32_t lo[NBUCKET + 2]; };

__global__ void k_synth_degrees(int64_t* out, int32_t v0, int32_t n, Ladder lad)
{
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t h = sm64(S_DEG + (uint64_t)(v0 + i));
        const uint32_t top = (uint32_t)(h >> 40);
        const int b = top ? (__clz(top) - 8) : NBUCKET;
        const int64_t lo = lad.lo[b], span = (int64_t)lad.lo[b + 1] - lo;
        out[i] = lo + (int64_t)((h & 0xFFFFFFFFull) % (uint64_t)span);
    }
}

__global__ void k_synth_neighbors(int32_t* out, int64_t e0, int64_t n, uint32_t V, uint32_t M, uint32_t C, uint32_t skew)
{
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = sm64(S_NBR + (uint64_t)(e0 + i));
        const uint64_t a = h >> 32;
        uint64_t x = a;
        if ((h & 0xFF) < skew) x = (((a * a) >> 32) * a) >> 32;
        const uint64_t r = (x * (uint64_t)V) >> 32;
        out[i] = (int32_t)((r * (uint64_t)M + (uint64_t)C) % (uint64_t)V);
    }
}

__global__ void k_synth_features(float* out, int64_t v0, int64_t nrows, int32_t F, int32_t pitch)
{
    const int64_t n = nrows * F;
    for (int64_t i = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = sm64(S_FEAT + (uint64_t)(v0 * F + i));
        // same values whatever the row pitch: element c of row r lands at r * pitch + c (pad floats are left as they are)
        out[pitch == F ? i : (i / F) * pitch + i % F] = (float)(uint32_t)(h >> 40) * 5.9604644775390625e-8f - 0.5f; // 2^-24
    }
}

__global__ void k_synth_labels(int32_t* out, int32_t v0, int32_t n, int32_t classes)
{
    for (int32_t i = threadIdx.x + blockDim.x * blockIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = (int32_t)(sm64(S_LAB + (uint64_t)(v0 + i)) % (uint64_t)classes);
}

// out[k] = (i*M2 + C2) % V for i = i0 + phase + k*stride  (stride/phase: the tid % G split of a
// permutation-ordered seed list cannot be written in closed form, so the split is done by the
// caller; stride = 1, phase = 0 yields the plain list)
__global__ void k_synth_seed_ids(int32_t* out, int64_t i0, int64_t n, uint32_t V, uint32_t M2, uint32_t C2,
                                 int32_t stride, int32_t phase)
{
    for (int64_t k = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t i = (uint64_t)(i0 + phase + k * stride);
        out[k] = (int32_t)((i * (uint64_t)M2 + (uint64_t)C2) % (uint64_t)V);
    }
}

// Link-prediction seed batches (lp_sage.py:87-90 splits a batch into [src | pos | neg] thirds; the reference server
// has no edge / negative sampler, so the seed list itself is laid out that way -- legion-1_amd/synth.py
// lp_trainingset is the numpy statement of the same rule).  Triple j of this list: src = srcs[j]; with
// t = triple_no[j] its number in the global (all ranks) triple order, pos = a neighbour of src drawn with minstd value
// 48271^(seed + 2t + 1), neg = a uniform node id from the next value.  k = batch / 3 triples per batch; the tail of the
// last batch repeats that batch's first triple.
__global__ void k_synth_lp_seeds(int32_t* out, const int32_t* srcs, const int64_t* triple_no, int64_t n, int32_t batch,
                                 const int64_t* indptr, const int32_t* indices, uint32_t V, uint32_t seed)
{
    const int64_t k = batch / 3, n_pad = (n + k - 1) / k * k;
    for (int64_t j = threadIdx.x + (int64_t)blockDim.x * blockIdx.x; j < n_pad; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = j / k, i = j - b * k;
        const int64_t jj = j < n ? j : b * k;       // padding: the first triple of the last batch
        const int32_t s = srcs[jj];
        const uint64_t t = (uint64_t)triple_no[jj];
        uint32_t x = powmod31(kA, (uint64_t)seed + 2ull * t + 1ull);
        const int64_t lo = indptr[s], hi = indptr[s + 1];
        int32_t pos = s;
        if (hi > lo) { pos = indices[lo + (int64_t)((x - 1u) % (uint64_t)(hi - lo))]; if (pos < 0) pos = s; }
        x = mulmod31(x, kA);
        const int32_t neg = (int32_t)((x - 1u) % V);
        out[b * batch + i] = s;
        out[b * batch + k + i] = pos;
        out[b * batch + 2 * k + i] = neg;
    }
}

typedef float copy_v4f __attribute__((ext_vector_type(4)));
// Streaming copy, the measured HBM ceiling bench.py prints beside the vendor peak.  U independent 16-byte loads in
// flight per lane; NT bit 0: non-temporal stores, bit 1: non-temporal loads.  CONTIG: a workgroup copies one
// contiguous block of U * blockDim chunks per step (block-strided) instead of striding every load by the whole grid.
template <int U, int NT, bool CONTIG>
__global__ __launch_bounds__(256) void k_copy_f4(copy_v4f* __restrict__ dst, const copy_v4f* __restrict__ src, int64_t n)
{
    const int64_t step = CONTIG ? (int64_t)blockDim.x : (int64_t)gridDim.x * blockDim.x;    // distance between a lane's U chunks
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;                              // distance between iterations
    int64_t i = CONTIG ? (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x : threadIdx.x + (int64_t)blockDim.x * blockIdx.x;
    for (; i < n; i += stride) {
        copy_v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i + u * step < n) v[u] = (NT & 2) ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (i + u * step < n) {
                if (NT & 1) __builtin_nontemporal_store(v[u], dst + i + u * step);
                else dst[i + u * step] = v[u];
            }
    }
}

// What a trainer's READ of a served batch costs, without a model (bench.py's reading consumer): legion_graphsage.py:72-89 reads every
// feature row and both COO arrays of the batch before it hands the pipe back.  Every 32-bit word of [src, src + 16 * n16) -- plus n_tail
// trailing words -- is added into *acc (u64, wrap-around): exact and order-independent, so the sum doubles as a checksum of what was read.
// Four non-temporal 16-byte loads in flight per lane, one atomic per workgroup.
typedef unsigned int sum_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_sum_words(const sum_v4u* __restrict__ src, int64_t n16, const uint32_t* __restrict__ tail, int32_t n_tail,
                                                   unsigned long long* __restrict__ acc)
{
    __shared__ unsigned long long s_part[4];
    unsigned long long sum = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        sum_v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; u++) sum += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n16; i += stride) {
        const sum_v4u v = __builtin_nontemporal_load(src + i);
        sum += (unsigned long long)v.x + v.y + v.z + v.w;
    }
    if (blockIdx.x == 0 && (int32_t)threadIdx.x < n_tail) sum += tail[threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

} // namespace legion

using namespace legion;

static inline int big_grid(int64_t n)
{
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (int)(b < 4096 ? b : 4096);
}

extern "C" {

void legion_synth_degrees(void* stream, int64_t* deg_out, int32_t v0, int32_t n, const int32_t* ladder_host26)
{
    if (n <= 0) return;
    Ladder lad;
    for (int i = 0; i < NBUCKET + 2; i++) lad.lo[i] = ladder_host26[i];
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_degrees", LEGION_AW(deg_out));
    k_synth_degrees<<<big_grid(n), 256, 0, (hipStream_t)stream>>>(deg_out, v0, n, lad);
    HIP_CHECK_LAST();
}
void legion_synth_neighbors_skew(void* stream, int32_t* indices_out, int64_t e0, int64_t n, int32_t V, uint32_t M, uint32_t C,
                                 int32_t skew_of_256)
{
    if (n <= 0) return;
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_neighbors", LEGION_AW(indices_out));
    k_synth_neighbors<<<big_grid(n), 256, 0, (hipStream_t)stream>>>(indices_out, e0, n, (uint32_t)V, M, C, (uint32_t)skew_of_256);
    HIP_CHECK_LAST();
}
void legion_synth_neighbors(void* stream, int32_t* indices_out, int64_t e0, int64_t n, int32_t V, uint32_t M, uint32_t C)
{
    legion_synth_neighbors_skew(stream, indices_out, e0, n, V, M, C, 205); // the spec'd 80 % Zipf-like neighbours
}
void legion_synth_lp_seeds(void* stream, int32_t* out, const int32_t* srcs, const int64_t* triple_no, int64_t n_triples,
                           int32_t batch_size, const int64_t* indptr, const int32_t* indices, int32_t V, uint32_t seed)
{
    if (n_triples <= 0) return;
    if (batch_size < 3 || batch_size % 3) { LEGION_ARG_ERROR("legion_synth_lp_seeds: batch size must be a multiple of 3"); return; }
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_lp_seeds", LEGION_AW(out), LEGION_AL(srcs), LEGION_AL(triple_no), LEGION_AR(indptr), LEGION_AR(indices));
    k_synth_lp_seeds<<<big_grid(n_triples), 256, 0, (hipStream_t)stream>>>(out, srcs, triple_no, n_triples, batch_size, indptr, indices,
                                                                           (uint32_t)V, seed);
    HIP_CHECK_LAST();
}
void legion_synth_features(void* stream, float* out, int64_t v0, int64_t nrows, int32_t F)
{
    if (nrows <= 0) return;
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_features", LEGION_AW(out));
    k_synth_features<<<big_grid(nrows * F), 256, 0, (hipStream_t)stream>>>(out, v0, nrows, F, F);
    HIP_CHECK_LAST();
}
void legion_synth_features_pitched(void* stream, float* out, int64_t v0, int64_t nrows, int32_t F, int32_t pitch)
{
    if (nrows <= 0) return;
    if (pitch < F) { LEGION_ARG_ERROR("legion_synth_features_pitched: pitch < F"); return; }
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_features", LEGION_AW(out));
    k_synth_features<<<big_grid(nrows * F), 256, 0, (hipStream_t)stream>>>(out, v0, nrows, F, pitch);
    HIP_CHECK_LAST();
}
void legion_synth_labels(void* stream, int32_t* out, int32_t v0, int32_t n, int32_t classes)
{
    if (n <= 0) return;
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_labels", LEGION_AW(out));
    k_synth_labels<<<big_grid(n), 256, 0, (hipStream_t)stream>>>(out, v0, n, classes);
    HIP_CHECK_LAST();
}
// ---- the generator's parameters (host; the C statement of legion-1_amd/synth.py spec_for / SynthSpec / degree_ladder) --------------
// Everything a `synth:` dataset source (runner.cpp, Server_Initialize) needs to regenerate, on the device, exactly the graph that
// synth.py / bench.py build from the same (workload, scale): tests/test_host_logic.py compares every field with the Python spec.
namespace {
struct Shape { const char* name; int64_t V, E; int32_t F, n_train, n_valid, n_test, classes; };
// legion_server.py:6-37 (V, E, F, seed-set sizes); classes as in synth.py SHAPES
const Shape kShapes[] = {
    {"products", 2449029, 123718280, 100, 196615, 39323, 2213091, 47},
    {"papers100M", 111059956, 1615685872, 128, 11105995, 100000, 100000, 172},
    {"uk-union", 133633040, 5507679822, 256, 13363304, 100000, 100000, 172},
};
uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; }
uint32_t coprime_multiplier(uint64_t V, uint64_t start)
{
    uint64_t m = start % V;
    if (m < 2) m = V > 2 ? 2 : 1;
    while (gcd_u64(m, V) != 1) m++;
    return (uint32_t)m;
}
// degree_ladder(): lo[b] = d0 * 1.6^b rounded half-to-even (Python's round), made strictly increasing; d0 by 80 bisection steps on
// the expected degree.  Plain IEEE double arithmetic in the order synth.py writes it; no fused multiply-add.
#pragma clang fp contract(off)
void ladder_of(double d0, int32_t lo[NBUCKET + 2])
{
    for (int b = 0; b < NBUCKET + 2; b++) {
        const double r = nearbyint(d0 * pow(1.6, (double)b));
        lo[b] = r < 1.0 ? 1 : (int32_t)r;
    }
    for (int b = 1; b < NBUCKET + 2; b++) if (lo[b] < lo[b - 1] + 1) lo[b] = lo[b - 1] + 1;
}
double ladder_mean(const int32_t lo[NBUCKET + 2])
{
    double m = 0.0;
    for (int b = 0; b < NBUCKET + 1; b++) {
        const double p = b < NBUCKET ? ldexp(1.0, -(b + 1)) : ldexp(1.0, -NBUCKET);
        m += p * ((double)lo[b] + (double)(lo[b + 1] - lo[b] - 1) / 2.0);
    }
    return m;
}
} // namespace

int legion_synth_spec(const char* workload, double scale, LegionSynthSpec* out)
{
    if (!workload || !out || !(scale > 0.0) || scale > 1.0) { LEGION_ARG_ERROR("legion_synth_spec: workload name and 0 < scale <= 1 expected"); return -1; }
    const Shape* sh = nullptr;
    for (const Shape& s : kShapes) if (strcmp(s.name, workload) == 0) sh = &s;
    if (!sh) { LEGION_ARG_ERROR("legion_synth_spec: unknown workload (products | papers100M | uk-union)"); return -1; }
    memset(out, 0, sizeof(*out));
    int64_t V = sh->V, n_train = sh->n_train, n_valid = sh->n_valid, n_test = sh->n_test;
    if (scale != 1.0) {                     // synth.py spec_for: V and the seed sets shrink, the mean degree stays
        V = std::max<int64_t>(64, (int64_t)((double)sh->V * scale));
        n_train = std::max<int64_t>(1, (int64_t)((double)sh->n_train * scale));
        n_valid = std::max<int64_t>(1, (int64_t)((double)sh->n_valid * scale));
        n_test = std::max<int64_t>(1, (int64_t)((double)sh->n_test * scale));
        if (n_train + n_valid + n_test > V) n_test = std::max<int64_t>(1, V - n_train - n_valid);
    }
    out->V = (int32_t)V; out->F = sh->F; out->classes = sh->classes;
    out->n_train = (int32_t)n_train; out->n_valid = (int32_t)n_valid; out->n_test = (int32_t)n_test;
    out->mean_degree = (double)sh->E / (double)sh->V;
    out->M = coprime_multiplier((uint64_t)V, 0x9E3779B1ull);  out->C = (uint32_t)(0x7F4A7C15ull % (uint64_t)V);
    out->M2 = coprime_multiplier((uint64_t)V, 0x85EBCA6Bull); out->C2 = (uint32_t)(0xC2B2AE35ull % (uint64_t)V);
    double a = 0.01, b = 1e4;
    int32_t lo[NBUCKET + 2];
    for (int it = 0; it < 80; it++) {
        const double mid = sqrt(a * b);
        ladder_of(mid, lo);
        if (ladder_mean(lo) < out->mean_degree) a = mid; else b = mid;
    }
    ladder_of(b, out->ladder);
    return 0;
}
// host statements of two of the closed forms (a `synth:` server builds its seed lists and their labels on the host)
int32_t legion_synth_label_host(int32_t v, int32_t classes) { return (int32_t)(sm64(S_LAB + (uint64_t)(uint32_t)v) % (uint64_t)classes); }
int32_t legion_synth_seed_id_host(int64_t i, int32_t V, uint32_t M2, uint32_t C2) { return (int32_t)(((uint64_t)i * (uint64_t)M2 + (uint64_t)C2) % (uint64_t)(uint32_t)V); }
void legion_synth_seed_ids(void* stream, int32_t* out, int64_t i0, int64_t n, int32_t V, uint32_t M2, uint32_t C2,
                           int32_t stride, int32_t phase)
{
    if (n <= 0) return;
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_synth_seed_ids", LEGION_AW(out));
    k_synth_seed_ids<<<big_grid(n), 256, 0, (hipStream_t)stream>>>(out, i0, n, (uint32_t)V, M2, C2, stride, phase);
    HIP_CHECK_LAST();
}
// variant: unroll in {1,2,4,8}, nt in 0..3, contig 0/1; grid <= 0: one iteration per lane
int legion_copy_f4_cfg(void* stream, void* dst, const void* src, int64_t bytes, int32_t grid, int32_t unroll, int32_t nt, int32_t contig)
{
    const int64_t n = bytes / 16;
    if (n <= 0) return 0;
    if (grid <= 0) grid = (int32_t)std::min<int64_t>((n + 256ll * unroll - 1) / (256ll * unroll), 0x7FFFFFFF);
    hipStream_t s = (hipStream_t)stream;
    copy_v4f* d = (copy_v4f*)dst;
    const copy_v4f* r = (const copy_v4f*)src;
    LEGION_AUDIT_LAUNCH(s, "k_copy_f4", LEGION_AW(dst), LEGION_AR(src));
#define LEGION_COPY_CASE(U, N, C) if (unroll == U && nt == N && contig == C) { k_copy_f4<U, N, (C != 0)><<<grid, 256, 0, s>>>(d, r, n); HIP_CHECK_LAST(); return 0; }
#define LEGION_COPY_NT(U, C) LEGION_COPY_CASE(U, 0, C) LEGION_COPY_CASE(U, 1, C) LEGION_COPY_CASE(U, 2, C) LEGION_COPY_CASE(U, 3, C)
    LEGION_COPY_NT(1, 0) LEGION_COPY_NT(2, 0) LEGION_COPY_NT(4, 0) LEGION_COPY_NT(8, 0)
    LEGION_COPY_NT(1, 1) LEGION_COPY_NT(2, 1) LEGION_COPY_NT(4, 1) LEGION_COPY_NT(8, 1)
#undef LEGION_COPY_NT
#undef LEGION_COPY_CASE
    LEGION_ARG_ERROR("legion_copy_f4_cfg: no such variant");
    return -1;
}
// *acc += the sum of the 32-bit words of [src, src + bytes); src 16-byte aligned, bytes a multiple of 4
void legion_sum_words(void* stream, const void* src, int64_t bytes, uint64_t* acc)
{
    if (bytes <= 0) return;
    if (!src || !acc || ((uintptr_t)src & 15) || (bytes & 3)) { LEGION_ARG_ERROR("legion_sum_words: src must be 16-byte aligned, bytes a multiple of 4"); return; }
    const int64_t n16 = bytes / 16;
    const int32_t n_tail = (int32_t)((bytes - n16 * 16) / 4);
    const int grid = (int)std::min<int64_t>(std::max<int64_t>((n16 + 1023) / 1024, 1), 2048);
    LEGION_AUDIT_LAUNCH((hipStream_t)stream, "k_sum_words", LEGION_AW(acc), LEGION_AR(src));
    k_sum_words<<<grid, 256, 0, (hipStream_t)stream>>>((const sum_v4u*)src, n16, (const uint32_t*)src + n16 * 4, n_tail, (unsigned long long*)acc);
    HIP_CHECK_LAST();
}
void legion_copy_f4(void* stream, void* dst, const void* src, int64_t bytes)
{
    // fastest variant of profiles/copy_sweep.py on MI355X (profiles/r02_copy_sweep.md): one 16-byte chunk per lane, no
    // loop (bytes / 4096 workgroups), non-temporal loads and stores: 6.52-6.56 TB/s; round 1's 32768 x 4 chunks: 5.3
    (void)legion_copy_f4_cfg(stream, dst, src, bytes, 0, 1, 3, 0);
}

} // extern "C"
