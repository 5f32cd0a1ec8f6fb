/*
 * dgl_cpu_sampler.c -- "DGL-semantics CPU sampler (own implementation)".  TEST / BENCH INFRASTRUCTURE.
 *
 * BASELINE.json names DGL's CPU NeighborSampler as the reported CPU baseline of config 0; DGL is not
 * installed in this image and cannot be fetched, so bench.py falls back to this OpenMP implementation of
 * the same semantics and labels it as such:
 *   - per frontier node: uniform sampling WITHOUT replacement of min(deg, fanout) neighbours
 *     (all neighbours when deg <= fanout)           [dgl.sampling.sample_neighbors]
 *   - per layer: to_block compaction -- destination nodes first, then newly seen source nodes,
 *     relabelled to local ids                        [dgl.to_block]
 *   - feature rows of the outermost block's source nodes gathered into one dense matrix
 *     [index_select on the feature tensor]
 * It is NOT the reference's algorithm (that is oracle/legion_oracle.c) and is never checked for parity.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint64_t sm64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int32_t dgl_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* One mini-batch.  local_map: int32[V] filled with -1 by the caller (restored on return).
 * nodes_out: int32[>= B*(1+f0+f0*f1+...)] node ids of the final (outermost) source set, seeds first.
 * src_out/dst_out: local COO of all layers (concatenated); edge_off[hops+1] layer boundaries.
 * feat_out may be NULL.  Returns the number of nodes; *n_edges the total number of sampled edges. */
int64_t dgl_sample_batch(const int64_t *indptr, const int32_t *indices, const float *features, int32_t F,
                         const int32_t *seeds, int32_t n_seeds, const int32_t *fanout, int32_t hops,
                         uint64_t rng_seed, int32_t *local_map, int32_t *nodes_out, int32_t *src_out,
                         int32_t *dst_out, int64_t *edge_off, float *feat_out, int32_t *scratch, int64_t *n_edges)
{
    int64_t n_nodes = 0, e_total = 0;
    for (int32_t i = 0; i < n_seeds; i++) {
        if (local_map[seeds[i]] < 0) { local_map[seeds[i]] = (int32_t)n_nodes; nodes_out[n_nodes++] = seeds[i]; }
    }
    int64_t frontier_begin = 0, frontier_end = n_nodes;
    edge_off[0] = 0;
    for (int32_t h = 0; h < hops; h++) {
        const int32_t f = fanout[h];
        const int64_t nf = frontier_end; /* DGL blocks: every dst node of the previous block is a frontier node */
        (void)frontier_begin;
        /* phase 1 (parallel): draw into scratch[nf * f], -1 padded */
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t i = 0; i < nf; i++) {
            const int32_t v = nodes_out[i];
            const int64_t s = indptr[v];
            const int32_t deg = (int32_t)(indptr[v + 1] - s);
            int32_t *out = scratch + i * f;
            if (deg <= f) {
                for (int32_t j = 0; j < f; j++) out[j] = j < deg ? indices[s + j] : -1;
            } else { /* Floyd's algorithm: f distinct positions out of deg */
                int32_t pick[64];
                uint64_t st = sm64(rng_seed ^ ((uint64_t)h << 56) ^ (uint64_t)v);
                int32_t cnt = 0;
                for (int32_t j = deg - f; j < deg; j++) {
                    st = sm64(st);
                    int32_t t = (int32_t)(st % (uint64_t)(j + 1));
                    int dup = 0;
                    for (int32_t k = 0; k < cnt; k++) if (pick[k] == t) { dup = 1; break; }
                    pick[cnt++] = dup ? j : t;
                }
                for (int32_t j = 0; j < f; j++) out[j] = indices[s + pick[j]];
            }
        }
        /* phase 2 (serial, like dgl.to_block's hash-map pass): relabel + compact */
        for (int64_t i = 0; i < nf; i++) {
            const int32_t *in = scratch + i * f;
            for (int32_t j = 0; j < f; j++) {
                const int32_t u = in[j];
                if (u < 0) continue;
                int32_t lu = local_map[u];
                if (lu < 0) { lu = (int32_t)n_nodes; local_map[u] = lu; nodes_out[n_nodes++] = u; }
                src_out[e_total] = lu;
                dst_out[e_total] = (int32_t)i;
                e_total++;
            }
        }
        edge_off[h + 1] = e_total;
        frontier_begin = frontier_end;
        frontier_end = n_nodes;
    }
    if (feat_out && features) {
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n_nodes; r++)
            memcpy(feat_out + r * F, features + (int64_t)nodes_out[r] * F, (size_t)F * sizeof(float));
    }
    for (int64_t r = 0; r < n_nodes; r++) local_map[nodes_out[r]] = -1;
    *n_edges = e_total;
    return n_nodes;
}
