// thrust_probe.cpp -- TEST INFRASTRUCTURE.  Host-side probe of the third-party RNG the
// reference's sampler calls (Kernels.cu:402-405: thrust::minstd_rand + discard(idx) +
// thrust::uniform_int_distribution<>(0, deg-1)).  Thrust is not vendored in the reference;
// this compiles the Thrust headers that ship with this image (rocThrust, /opt/rocm/include)
// for the host (CPP device system) and prints "idx deg k x" lines.  Its output is committed
// as tests/golden/rng_kat.json by oracle/make_golden.py; the oracle and the HIP kernel are
// both checked against it.
#include <thrust/random/linear_congruential_engine.h>
#include <thrust/random/uniform_int_distribution.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

int main(int argc, char** argv)
{
    if (argc >= 2 && argv[1][0] == 'n') {  // nth value of a default engine (Thrust's documented KAT)
        unsigned long long n = strtoull(argv[2], nullptr, 10);
        thrust::minstd_rand e;
        uint32_t v = 0;
        for (unsigned long long i = 0; i < n; i++) v = e();
        printf("%u\n", v);
        return 0;
    }
    // pairs on stdin: idx deg
    long long idx; int deg;
    while (scanf("%lld %d", &idx, &deg) == 2) {
        thrust::minstd_rand engine;
        engine.discard((int32_t)idx);
        thrust::minstd_rand e2 = engine;
        uint32_t x = e2();
        thrust::uniform_int_distribution<> dist(0, deg - 1);
        int32_t k = dist(engine);
        printf("%lld %d %d %u\n", idx, deg, k, x);
    }
    return 0;
}
