/*
 * lidar_oracle.h -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call it.  The product library (liblpx.so) never links it.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference
 * repository YevgeniyEngineer/LiDAR-Processing).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - kd-tree order, radius search, FIFO clustering loop: PINNED against the reference's own
 *     src/kdtree.hpp + src/queue.hpp compiled unmodified (oracle/ref_driver.cpp ->
 *     oracle/_ref/libkdref.so), identical pre-order, neighbour lists and labels on real frames.
 *   - segmentation: PARITY UNPINNED at the Eigen boundary.  src/segmentation.cpp needs Eigen
 *     (absent, unpinned system package) and no reference test holds a segmentation result.
 *     The restatement fixes a canonical arithmetic (exact integer moments, float 3x3 Jacobi
 *     SVD following Eigen 3.4's JacobiSVD algorithm) and a canonical tie order (x, index).
 */
#ifndef LIDAR_ORACLE_H
#define LIDAR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors SegmentationConfiguration, src/segmentation.hpp:48-56 */
typedef struct
{
    float sensor_height_m;
    float orthogonal_distance_threshold;
    float initial_seed_threshold;
    uint32_t number_of_iterations;
    uint32_t number_of_planar_partitions;
    uint32_t number_of_lower_point_representatives;
} orc_seg_cfg;

/* mirrors ClusteringConfiguration, src/clustering.hpp:42-48 */
typedef struct
{
    float distance_squared;
    float cluster_quality;
    uint32_t min_cluster_size;
    uint32_t max_cluster_size;
} orc_clu_cfg;

#define ORC_LABEL_UNKNOWN 0u  /* src/segmentation.hpp:41-46 */
#define ORC_LABEL_GROUND 1u
#define ORC_LABEL_OBSTACLE 2u

#define ORC_CLUSTER_UNDEFINED INT32_MIN /* src/clustering.hpp:53 */
#define ORC_CLUSTER_INVALID (-1)        /* src/clustering.hpp:54 */

/* per-segment status written by orc_segment */
#define ORC_SEG_OK 0u
#define ORC_SEG_TOO_FEW_POINTS 1u /* <3 points: nothing labelled, src/segmentation.cpp:225-229 */
#define ORC_SEG_ALL_OBSTACLE 2u   /* <3 ground points at some iteration, :251-259 */

#define ORC_OK 0
#define ORC_ERR_RANGE (-2) /* a coordinate is NaN or infinite */
#define ORC_ERR_ARG (-1)

/*
 * Segmenter::segment, src/segmentation.cpp:311-345 (with :104-149, :151-217, :62-102, :219-309).
 * pts: AoS records, x,y,z float32 at byte offsets 0,4,8 of each stride_bytes record.
 * labels[n]           : written for every point (UNKNOWN for the N mod P tail, Q2/Q3).
 * ground_idx/obstacle_idx: original indices in output-cloud order (:331-343).
 * planes[P*4]         : a,b,c,d of the last plane fitted per segment (zeros if none).
 * seg_status[P]       : ORC_SEG_*.
 */
int orc_segment(const void *pts, size_t stride_bytes, uint32_t n, const orc_seg_cfg *cfg, uint32_t *labels,
                uint32_t *ground_idx, uint32_t *n_ground, uint32_t *obstacle_idx, uint32_t *n_obstacle,
                float *planes, uint32_t *seg_status);

/* threads for the two index sorts of orc_segment (the reference runs them with std::execution::par, its only
 * parallelism): 1 by default; results do not depend on it */
void orc_set_sort_threads(int n);

/* Clusterer::cluster, src/clustering.cpp:47-125 over src/kdtree.hpp:174-225,292-341. */
int orc_cluster(const void *pts, size_t stride_bytes, uint32_t m, const orc_clu_cfg *cfg, int32_t *labels,
                uint32_t *n_clusters);

/* Extra observables of the clustering loop for tests (expansions, neighbour visits). */
int orc_cluster_stats(const void *pts, size_t stride_bytes, uint32_t m, const orc_clu_cfg *cfg, int32_t *labels,
                      uint32_t *n_clusters, uint64_t *n_expansions, uint64_t *n_visits);

/* KDTree::rebuild (src/kdtree.hpp:174-225): original indices in array (= in-order) layout. */
int orc_kd_layout(const float *xyz, uint32_t m, uint32_t *layout_idx);
/* pre-order sequence of original indices (what an unbounded radius_search emits, :292-341) */
int orc_kd_preorder(const float *xyz, uint32_t m, uint32_t *preorder_idx);
/* KDTree::radius_search (:292-341), unsorted (sort_ == false, src/clustering.hpp:69) */
int orc_radius_search(const float *xyz, uint32_t m, const float *target, float r2, uint32_t *out_idx,
                      float *out_dist, uint32_t *count);

/* std::nth_element of libstdc++ 11 (bits/stl_algo.h:1964-1986) on (key,payload) pairs. */
void orc_nth_element_u32(float *keys, uint32_t *payload, uint32_t first, uint32_t nth, uint32_t last);

/* 3x3 plane solve used by the restatement: moments -> (a,b,c,d).  Exposed for KATs. */
int orc_plane_from_points(const float *xyz, uint32_t n, float *plane);

/* Eigen 3.4 JacobiSVD<Matrix3f> restated: V of a 3x3 float matrix (row-major in/out) */
void orc_jacobi_svd3(const float *a, float *v, float *sigma);
/* test instrumentation: record every plane fit (16 floats per fit: cov[9], plane[4], Jacobi sweeps, points, failed)
 * into buf until cap_records are full; NULL switches it off.  orc_trace_count(): fits recorded so far. */
void orc_trace_fits(float *buf, uint32_t cap_records);
uint32_t orc_trace_count(void);

/* N3: Andrew monotone chain, counter-clockwise, collinear points excluded (restated published algorithm; the
 * reference's own implementation lives in the un-vendored Convex-Hull submodule: PARITY UNPINNED).
 * xy: n (x, y) float pairs; out_idx: hull vertices as indices into xy. */
int orc_convex_hull(const float *xy, uint32_t n, uint32_t *out_idx, uint32_t *count);
/* hulls of the valid clusters with fewer than max_points points (src/polygon_simplification.cpp:96-115) */
int orc_cluster_hulls(const void *pts, size_t stride_bytes, uint32_t m, const int32_t *labels, uint32_t n_clusters,
                      uint32_t max_points, uint32_t *hull_offsets, uint32_t *hull_indices);

#ifdef __cplusplus
}
#endif
#endif
