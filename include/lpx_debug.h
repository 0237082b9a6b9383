/*
 * lpx_debug.h -- stage-level test hooks of liblpx.so.  NOT part of the drop-in boundary.
 *
 * include/lpx.h is the surface that replaces the reference's calls; the functions here expose single
 * stages (sort, scan, kd-tree layout, neighbour lists, components, plane fit, frame counters) so that
 * tests/ can compare each of them with the oracle, and tools/ can read per-stage statistics.  A caller
 * of the hot path never needs them.  Host pointers, synchronous, same status codes as lpx.h.
 */
#ifndef LPX_DEBUG_H
#define LPX_DEBUG_H

#include "lpx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* stable LSD radix sort of (key, value) pairs on the device; bits = number of low key bits used */
int lpx_dbg_sort_pairs(lpx_ctx *ctx, uint32_t *keys, uint32_t *values, uint32_t n, uint32_t bits);
int lpx_dbg_sort_keys64(lpx_ctx *ctx, uint64_t *keys, uint32_t n, uint32_t bits);
/* exclusive scan of u32 -> u32, returns total in *total */
int lpx_dbg_scan(lpx_ctx *ctx, uint32_t *data, uint32_t n, uint64_t *total);
/* kd-tree array layout (reference KDTree::rebuild, src/kdtree.hpp:174-225): original index per node */
int lpx_dbg_kd_layout(lpx_ctx *ctx, const float *xyz, uint32_t m, uint32_t *layout_idx);
/* radius-neighbour lists of every point in kd-tree pre-order (src/kdtree.hpp:292-341) as CSR;
 * offsets[m+1]; idx/dist hold `capacity` entries.  Returns LPX_ERR_CAPACITY if too small
 * (offsets[m] still holds the required size).  The device keeps one word per neighbour (index | within-absorb-
 * radius bit); the distances handed back are recomputed on the host with the reference's expression. */
int lpx_dbg_neighbours(lpx_ctx *ctx, const float *xyz, uint32_t m, float r2, uint64_t *offsets, uint32_t *idx,
                       float *dist, uint64_t capacity);
/* connected-component root (smallest original index of the component) per point */
int lpx_dbg_components(lpx_ctx *ctx, const float *xyz, uint32_t m, float r2, uint32_t *root);
/* lpx_cluster calls of this context that found their clustering already enqueued by lpx_segment* (lpx_set_lookahead) */
uint64_t lpx_dbg_lookahead_hits(lpx_ctx *ctx);
/* statistics of the last frame processed by this context (synchronises): out[12] =
 * {n_ground, n_obstacle, n_clusters, status, neighbour entries lo/hi, components, expansions,
 *  entries read by the replay lo/hi, words of list storage handed out lo/hi} */
int lpx_dbg_frame_stats(lpx_ctx *ctx, uint32_t *out12);
/* the same for frame slot `slot` of a batch context */
int lpx_dbg_frame_stats_slot(lpx_ctx *ctx, uint32_t slot, uint32_t *out12);
/* plane from points through the device moment/Jacobi path */
int lpx_dbg_plane(lpx_ctx *ctx, const float *xyz, uint32_t n, float *plane);

/* expansion-driven search counters of frame slot `slot`: out[4] = {candidates distance-tested lo/hi, queue windows
 * with an expansion, searches redone by the sequencer (list larger than its LDS region)} */
int lpx_dbg_search_stats_slot(lpx_ctx *ctx, uint32_t slot, uint32_t *out4);

/* achievable HBM bandwidth of this device: GB/s (read + write) of a plain 16-byte-per-lane streaming copy of
 * `bytes` bytes, `reps` launches timed with HIP events */
int lpx_dbg_copy_bandwidth(lpx_ctx *ctx, size_t bytes, uint32_t reps, double *gb_per_s);

/* tools only: per-group statistics of the neighbour kernel, 8 words per kd group ({candidates, intervals,
 * queries, list words, cycles to allocation, cycles total, -, -}).  n_groups > 0 with out == NULL arms the
 * collection for the following calls, out != NULL copies what was collected, n_groups == 0 switches it off. */
int lpx_dbg_group_stats(lpx_ctx *ctx, uint32_t n_groups, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* LPX_DEBUG_H */
