// Microbenchmark: random 64-bit atomicMin into a table, device scope (executes at the memory side on an 8-XCD part)
// vs workgroup scope (executes in the issuing XCD's L2), and the plain random load for reference.
//   hipcc -O3 --offload-arch=gfx950 profiles/atomic_scope_probe.hip -o /tmp/atomic_scope_probe && /tmp/atomic_scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; return x ^ (x >> 33); }

template <int MODE>
__global__ void probe(unsigned long long* table, uint64_t mask, uint64_t n, unsigned long long* sink)
{
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t slot = mix(i) & mask;
        const unsigned long long v = (i << 8) | 1;
        if (MODE == 0) acc += table[slot];
        else if (MODE == 1) atomicMin(table + slot, v);                                                              // device scope, no return
        else if (MODE == 2) acc += atomicMin(table + slot, v);                                                       // device scope, returning
        else if (MODE == 3) __hip_atomic_fetch_min(table + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // L2, no return
        else if (MODE == 4) acc += __hip_atomic_fetch_min(table + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // L2, returning
        else if (MODE == 5) atomicMin(reinterpret_cast<unsigned int*>(table) + slot, (unsigned int)v);               // 32-bit, device scope
        else if (MODE == 6) atomicOr(reinterpret_cast<unsigned int*>(table) + slot, (unsigned int)v);                // 32-bit or (the reference's bitmap op)
        else if (MODE == 7) atomicCAS(table + slot, ~0ull, v);                                                       // 64-bit CAS
        else reinterpret_cast<unsigned int*>(table)[slot] = (unsigned int)v;                                         // plain scattered store
    }
    if (acc == 0x1234567) *sink = acc;
}

template <int MODE>
float run(unsigned long long* t, uint64_t mask, uint64_t n, unsigned long long* sink)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<2048, 256>>>(t, mask, n, sink);
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) probe<MODE><<<2048, 256>>>(t, mask, n, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    const uint64_t n = 4u << 20;
    unsigned long long* sink;
    hipMalloc(&sink, 8);
    for (uint64_t mb : {16ull, 1024ull}) {
        const uint64_t elems = mb * 1024 * 1024 / 8;
        unsigned long long* t;
        hipMalloc(&t, elems * 8);
        hipMemset(t, 0xFF, elems * 8);
        const char* names[9] = {"plain load", "atomicMin device scope", "atomicMin device scope, returning", "atomicMin workgroup scope (L2)", "atomicMin workgroup scope (L2), returning",
                                "32-bit atomicMin device scope", "32-bit atomicOr device scope", "64-bit atomicCAS device scope", "plain 4-byte store"};
        float ms[9] = {run<0>(t, elems - 1, n, sink), run<1>(t, elems - 1, n, sink), run<2>(t, elems - 1, n, sink), run<3>(t, elems - 1, n, sink), run<4>(t, elems - 1, n, sink),
                       run<5>(t, elems - 1, n, sink), run<6>(t, elems - 1, n, sink), run<7>(t, elems - 1, n, sink), run<8>(t, elems - 1, n, sink)};
        for (int m = 0; m < 9; m++) printf("table %5llu MB  %-42s %8.1f us  %6.1f G ops/s\n", (unsigned long long)mb, names[m], ms[m] * 1e3, n / (ms[m] * 1e-3) / 1e9);
        hipFree(t);
    }
    return 0;
}
