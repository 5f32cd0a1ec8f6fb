/* synth_gen.c -- TEST INFRASTRUCTURE.  Plain-C (OpenMP) generator of the synthetic datasets whose
 * specification is the docstring of legion-1_amd/synth.py (GEN_SEED 0x1E610, splitmix64 closed forms).
 *
 * Why it exists: the bit-exact full-shape parity fixtures (tests/golden/full_shape_digests.json) need the CSR of the
 * BASELINE shapes (up to E = 5.5e9) on a host without a GPU; the numpy form of the spec generates 2.6 M edges/s,
 * this one > 100 M/s.  It is a third, independent statement of the spec next to numpy (legion-1_amd/synth.py) and
 * HIP (csrc/synth.hip); tests/test_oracle_batches.py checks all pairs that can run on the CPU against each other.
 * Nothing of the reference is restated here (the reference ships no data). */
#include <stdint.h>
#include <stddef.h>

#define SG_SEED 0x1E610ull
#define SG_DEG ((SG_SEED << 32) ^ 0x0DE6ull)
#define SG_NBR ((SG_SEED << 32) ^ 0x0EB2ull)
#define SG_LAB ((SG_SEED << 32) ^ 0x1AB1ull)
#define SG_NBUCKET 24

static inline uint64_t sg_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* deg(v): bucket = leading zeros of a 24-bit uniform field (capped), degree uniform inside the bucket's ladder step.
 * Writes indptr[0..V] (exclusive prefix of the degrees); returns E. */
int64_t sg_indptr(int64_t *indptr, int32_t V, const int32_t *ladder)
{
    int64_t run = 0;
    indptr[0] = 0;
    for (int32_t v = 0; v < V; v++) { /* serial: one pass, ~0.3 s per 1e8 nodes */
        const uint64_t h = sg_mix(SG_DEG + (uint64_t)v);
        const uint32_t top = (uint32_t)(h >> 40);
        int b = SG_NBUCKET;
        if (top) { b = 0; while (!(top & (0x800000u >> b))) b++; }
        const int64_t lo = ladder[b], span = (int64_t)ladder[b + 1] - lo;
        run += lo + (int64_t)((h & 0xFFFFFFFFull) % (uint64_t)span);
        indptr[v + 1] = run;
    }
    return run;
}

/* nbr(e) for e in [e0, e0 + n): `skew` of 256 draws from the cubed (Zipf-like) variate, the rest uniform; scrambled by
 * the affine map (r * M + C) % V. */
void sg_neighbors(int32_t *out, int64_t e0, int64_t n, uint32_t V, uint32_t M, uint32_t C, uint32_t skew)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const uint64_t h = sg_mix(SG_NBR + (uint64_t)(e0 + i));
        const uint64_t a = h >> 32;
        uint64_t x = a;
        if ((h & 0xFF) < skew) x = (((a * a) >> 32) * a) >> 32;
        const uint64_t r = (x * (uint64_t)V) >> 32;
        out[i] = (int32_t)((r * (uint64_t)M + (uint64_t)C) % (uint64_t)V);
    }
}

/* seeds: id_i = (i * M2 + C2) % V */
void sg_seed_ids(int32_t *out, int64_t i0, int64_t n, uint32_t V, uint32_t M2, uint32_t C2)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = (int32_t)((((uint64_t)(i0 + i)) * M2 + C2) % V);
}

/* label(v) = sm64(S_LAB + v) % classes, for a list of ids */
void sg_labels_of(int32_t *out, const int32_t *ids, int64_t n, int32_t classes)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) out[i] = (int32_t)(sg_mix(SG_LAB + (uint64_t)(uint32_t)ids[i]) % (uint64_t)classes);
}
