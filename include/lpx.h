/*
 * lpx.h -- C-ABI of the MI355X-native ground-segmentation + obstacle-clustering hot path.
 *
 * Drop-in boundary for the reference's two calls in src/processor.cpp:
 *     segmenter_.segment(cloud_in_, segmentation_labels_, ground_points_, obstacle_points_);   (:150)
 *     clusterer_.cluster(*obstacle_cloud, cluster_labels);                                     (:178)
 * The header-only C++ classes in include/lidar_processing/{segmentation,clustering}.hpp keep the
 * reference's names and signatures (src/segmentation.hpp:58-70, src/clustering.hpp:50-75) and call
 * the functions below; INTEGRATION.md shows the binding.
 *
 * Conventions: plain pointers and sizes only, no C++ or torch types; every function returns
 * LPX_OK (0) or a negative LPX_ERR_* code and never throws; lpx_last_error() gives a message.
 * A context is bound to one HIP device and one stream; it is not thread-safe (one in-flight call
 * per context, like the reference objects).  "_device" entry points take device
 * pointers, enqueue on the context stream and do NOT synchronise; host entry points are
 * synchronous at return.
 */
#ifndef LPX_H
#define LPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lpx_ctx lpx_ctx;

/* mirrors lidar_processing::SegmentationConfiguration (reference src/segmentation.hpp:48-56) */
typedef struct
{
    float sensor_height_m;                          /* 1.73  */
    float orthogonal_distance_threshold;            /* 0.3   */
    float initial_seed_threshold;                   /* 0.6   */
    uint32_t number_of_iterations;                  /* 3     */
    uint32_t number_of_planar_partitions;           /* 2     */
    uint32_t number_of_lower_point_representatives; /* 5000  */
} lpx_seg_cfg;

/* mirrors lidar_processing::ClusteringConfiguration (reference src/clustering.hpp:42-48) */
typedef struct
{
    float distance_squared;    /* 0.18 */
    float cluster_quality;     /* 0.5  */
    uint32_t min_cluster_size; /* 4    */
    uint32_t max_cluster_size; /* UINT32_MAX */
} lpx_clu_cfg;

/* SegmentationLabel (reference src/segmentation.hpp:41-46) */
#define LPX_LABEL_UNKNOWN 0u
#define LPX_LABEL_GROUND 1u
#define LPX_LABEL_OBSTACLE 2u
/* Clusterer::UNDEFINED / INVALID (reference src/clustering.hpp:53-54) */
#define LPX_CLUSTER_UNDEFINED INT32_MIN
#define LPX_CLUSTER_INVALID (-1)

#define LPX_OK 0
#define LPX_ERR_ARG (-1)      /* bad argument */
#define LPX_ERR_RANGE (-2)    /* a coordinate is NaN or infinite.  Any finite cloud is processed, with two limits on
                               * what "like the reference" means far from the origin: (1) the plane-fit MOMENTS use a
                               * coordinate clamped to +-2^24 m (16.7e6 m; UTM / ECEF scale stays exact), the inlier
                               * test always the float itself -- beyond the clamp the planes are this library's, not
                               * the reference's float sums (the regression is locked by the repo's own oracle only);
                               * (2) the component grid saturates at +-2^20 cells (cell edge 0.57 d): farther points
                               * share a few edge cells, results stay exact but one giant point set serialises the
                               * replay, i.e. latency, not correctness, degrades for map-frame clouds with small d */
#define LPX_ERR_HIP (-3)      /* HIP runtime error, see lpx_last_error() */
#define LPX_ERR_CAPACITY (-4) /* workspace too small and could not be grown */
#define LPX_ERR_NO_DEVICE (-5)
#define LPX_ERR_INTERNAL (-6)

#ifndef LPX_MAX_BATCH
#define LPX_MAX_BATCH 64u
#endif
#define LPX_MAX_PARTITIONS 256u
#define LPX_MAX_ITERATIONS 64u

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Creates a context on HIP device `device` with its own stream.  Fails loudly (LPX_ERR_NO_DEVICE)
 * when no GPU is present: there is no CPU fallback. */
int lpx_create(int device, lpx_ctx **out);
/* A context with max_frames (1..LPX_MAX_BATCH) frame slots for lpx_segment_cluster_batch_device; every
 * other entry point uses slot 0.  Device scratch is max_frames times that of lpx_create. */
int lpx_create_batch(int device, uint32_t max_frames, lpx_ctx **out);
/* As lpx_create but enqueues on an existing hipStream_t (passed as void*; NULL = default stream). */
int lpx_create_on_stream(int device, void *hip_stream, lpx_ctx **out);
void lpx_destroy(lpx_ctx *ctx);
/* Segmenter::reserve_memory / Clusterer::reserve_memory (src/segmentation.cpp:44-60,
 * src/clustering.cpp:37-45): pre-size all device scratch for n points.  Scratch also grows on
 * demand.  neighbours_per_point sizes the exact-length region of the radius-neighbour lists of LPX_NEIGHBOURS_LISTS, one
 * 32-bit word per neighbour (0 = keep the default: 64, times ceil(distance_squared / 0.25) up to 4; a context of
 * lpx_create_batch with more than one frame slot starts at the 4 x, because nothing repeats its calls).  The list workspace
 * GROWS ON EVIDENCE: every list-mode clustering records what it asked for, and the next call on the context sizes both
 * regions to 1.25 x the largest demand seen (a 123k-point frame at d = 0.5 m: 126 MB).  A frame that outgrows the
 * workspace before that: the host entry points grow it and repeat the frame themselves, the device entry points report
 * LPX_ERR_CAPACITY in that frame's status word (the next call finds the workspace grown; reserve ahead for dense scenes). */
int lpx_reserve(lpx_ctx *ctx, uint32_t n_points, uint32_t neighbours_per_point);
/* Extra neighbour workspace (32-bit words per point, default 192, scaled and grown like the above) in which the
 * neighbour kernel may keep lists reserved by an upper bound of their length, which saves its counting pass.  It never
 * changes what
 * fits: a kd group that finds no room there counts first and uses the lpx_reserve workspace.  0 disables. */
int lpx_reserve_single_pass(lpx_ctx *ctx, uint32_t words_per_point);
/* Device memory the context holds right now, in bytes: bytes2[0] = the frame-slot arenas (~400 B per reserved point and
 * slot), bytes2[1] = the neighbour-list arena of LPX_NEIGHBOURS_LISTS (0 until a call has used that mode). */
int lpx_workspace_bytes(const lpx_ctx *ctx, uint64_t *bytes2);
/* How Clusterer::cluster finds neighbours.  Both modes give the reference's labels; they trade latency for work.
 *   LPX_NEIGHBOURS_LISTS : every radius list is materialised by the whole device at once, then the greedy loop
 *                          replays over them: shortest critical path for ONE frame alone on the device, but ~30x
 *                          the neighbour work and a large list workspace (lpx_reserve's neighbours_per_point).
 *   LPX_NEIGHBOURS_SEARCH: expansion-driven -- point sets from a uniform grid, and a radius search only when the
 *                          greedy loop expands a point; nothing is materialised, no list workspace.  Highest
 *                          throughput when many frames share the device.  Serves clouds of fewer than 2^30
 *                          obstacle points like the list path; very dense multi-million-point clouds are
 *                          faster with LISTS.
 *   LPX_NEIGHBOURS_AUTO  : (default) LISTS for a single-frame context (lpx_create), SEARCH for lpx_create_batch. */
#define LPX_NEIGHBOURS_AUTO 0
#define LPX_NEIGHBOURS_LISTS 1
#define LPX_NEIGHBOURS_SEARCH 2
int lpx_set_neighbour_mode(lpx_ctx *ctx, int mode);
const char *lpx_last_error(const lpx_ctx *ctx);
/* One line about the loaded library and the process it finds itself in: the build flavour -- the release library reads
 * NO environment variable that could change what it computes or how (the LPX_* development knobs exist only in
 * liblpx_dev.so, built with -DLPX_DEV_KNOBS) -- and GPU_MAX_HW_QUEUES as the process has it.  That HIP variable is the
 * host program's to set, before its first HIP call: HIP multiplexes its streams onto that many hardware queues
 * (default 4) and streams that share a queue serialise, so a process with more than four contexts, or with a feeder
 * beside its contexts, wants 16-32.  The library never modifies the environment. */
const char *lpx_build_info(void);
/* blocks until everything enqueued on the context stream has finished */
int lpx_synchronize(lpx_ctx *ctx);

/* Overlapped tail for batch contexts (lpx_create_batch): with `on`, consecutive lpx_segment_cluster_batch_device calls
 * alternate between two sets of frame slots (the workspace doubles), and the last part of a call -- the ordered
 * replay of Clusterer::cluster and the label kernels -- runs on a second stream while the context's stream already
 * takes the next call.  Results are the same; what changes is WHEN they are complete: the outputs of a call are
 * final after lpx_synchronize (which waits for both streams), not in the order of the context's stream -- do not use
 * it on a caller's stream whose later work reads the results without a host synchronisation.  Throughput only: a
 * single call is not faster. */
int lpx_set_overlap(lpx_ctx *ctx, int on);
/* Forked front end: with `on`, the component search of Clusterer::cluster (the clique-cell grid: eight short,
 * latency-bound launches) runs on a side stream of the device beside the kd-tree build and the chunk tables of the
 * same call -- both need only the obstacle cloud -- and the context's stream joins it before the ordered replay.
 * Results are the same; a call completes sooner (one frame in LPX_NEIGHBOURS_SEARCH mode, a chain of frames), at the
 * price of a second hardware queue while the fork is open.  Call it after lpx_set_overlap if both are wanted. */
int lpx_set_fork(lpx_ctx *ctx, int on);
/* Look-ahead of the two-call form.  The unchanged node calls Segmenter::segment and, a few lines later,
 * Clusterer::cluster on the obstacle cloud it got back (reference src/processor.cpp:150 and :178) -- two blocking calls
 * where the device could run one chain.  With the look-ahead on (the default), a context on which an lpx_cluster call
 * has just been served from the resident cloud of the lpx_segment* call before it remembers that call's configuration;
 * its next lpx_segment* call enqueues the clustering with that configuration right behind the segmentation and returns
 * as soon as the segmentation's own results are down (they are copied on a second stream).  An lpx_cluster call for
 * that cloud with that configuration then finds its chain running and only waits for it; any other call on the context
 * waits for the chain like for any earlier work, the guess is dropped, and nothing is guessed again until an
 * lpx_cluster call is served from a resident cloud once more.  Results never differ -- a mismatch takes the plain
 * path (upload and cluster) -- only the blocking time of the two calls does.  on = 0 turns it off. */
int lpx_set_lookahead(lpx_ctx *ctx, int on);
/* host wait until every batch call of an overlapped context but the LAST one is complete (a pipelined caller enqueues
 * call k, then collects call k - 1); without overlap the same as lpx_synchronize */
int lpx_wait_previous(lpx_ctx *ctx);

/* ---- host entry points (synchronous; what the C++ wrappers call) ---------------------------- */

/* Segmenter::segment (reference src/segmentation.cpp:311-345).
 * pts: n records of stride_bytes, float32 x,y,z at byte offsets 0,4,8 (pcl::PointXYZ = 16 B,
 * PointXYZI = 32 B).  labels[n]; ground_idx/obstacle_idx[n]: original indices in the order the
 * reference appends to ground_cloud / obstacle_cloud (:331-343).  planes may be NULL, else
 * [number_of_planar_partitions*4] (a,b,c,d of the last plane fitted per segment). */
int lpx_segment(lpx_ctx *ctx, const void *pts, size_t stride_bytes, uint32_t n, const lpx_seg_cfg *cfg,
                uint32_t *labels, uint32_t *ground_idx, uint32_t *n_ground, uint32_t *obstacle_idx,
                uint32_t *n_obstacle, float *planes);

/* Clusterer::cluster (reference src/clustering.cpp:47-125).  labels[m] (int32, dense 0..L-1 in
 * seed order, LPX_CLUSTER_INVALID for rejected groups); n_clusters may be NULL.
 * On the context whose last call was lpx_segment* (what two drop-in objects on one context do, reference
 * src/processor.cpp:150-178), a cloud of exactly that call's obstacle count whose position-bound checksum over all m
 * points (x, y, z words, computed here on the host) equals the one the device kept is clustered where it lies: no
 * upload.  Every other cloud -- another size, one changed coordinate, two swapped points, a second clustering of the
 * same cloud -- is uploaded.  "Equals" means: the size and TWO independent 64-bit position-bound wrapping sums over
 * every point's x, y, z words (lpx_internal.h: lpx_obstacle_mix, lpx_obstacle_mix2) agree -- a probabilistic witness
 * (two unrelated non-cryptographic mixes; ~2^-128 for clouds that are not built to collide), not a byte comparison:
 * a caller that must rule out even that gives its Clusterer a context of its own (the explicit constructor of
 * lidar_processing::Clusterer, or a second lpx_create), on which every cloud is uploaded.  Apart from that residual
 * the result never depends on which way a call went. */
int lpx_cluster(lpx_ctx *ctx, const void *pts, size_t stride_bytes, uint32_t m, const lpx_clu_cfg *cfg,
                int32_t *labels, uint32_t *n_clusters);

/* Counts the HOST clusterings (lpx_cluster, lpx_segment_cluster*) this context has completed; 0 while the labels of the
 * last one are NOT resident any more (any other call in between, a look-ahead clustering enqueued by lpx_segment
 * included).  lpx_cluster_groups / lpx_cluster_hulls serve exactly the clustering this number names and return
 * LPX_ERR_ARG when the labels are gone or m / n_clusters are not that call's. */
uint64_t lpx_cluster_epoch(const lpx_ctx *ctx);

/* Both calls back to back with the obstacle cloud kept on the device between them (what
 * Processor::process does at :150-178).  cluster_labels[i] belongs to obstacle_idx[i]. */
int lpx_segment_cluster(lpx_ctx *ctx, const void *pts, size_t stride_bytes, uint32_t n, const lpx_seg_cfg *seg_cfg,
                        const lpx_clu_cfg *clu_cfg, uint32_t *labels, uint32_t *ground_idx, uint32_t *n_ground,
                        uint32_t *obstacle_idx, uint32_t *n_obstacle, float *planes, int32_t *cluster_labels,
                        uint32_t *n_clusters);

/* ---- PointCloud2 wire format (N4) ------------------------------------------------------------ */

/* The reference's node decodes the incoming sensor_msgs/PointCloud2 on the host into a PointXYZI cloud with four
 * field iterators (src/conversions.cpp:62-85) before it calls segment().  These entry points take the message's
 * data[] buffer as it is: n = width * height records of point_step bytes, float32 x / y / z at the byte offsets
 * of the message's "x" / "y" / "z" PointField entries (little-endian, like the reference assumes; offsets and
 * point_step need not be multiples of 4).  Everything else is lpx_segment / lpx_segment_cluster. */
int lpx_segment_fields(lpx_ctx *ctx, const void *data, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                       uint32_t off_z, uint32_t n, const lpx_seg_cfg *cfg, uint32_t *labels, uint32_t *ground_idx,
                       uint32_t *n_ground, uint32_t *obstacle_idx, uint32_t *n_obstacle, float *planes);
int lpx_segment_cluster_fields(lpx_ctx *ctx, const void *data, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                               uint32_t off_z, uint32_t n, const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                               uint32_t *labels, uint32_t *ground_idx, uint32_t *n_ground, uint32_t *obstacle_idx,
                               uint32_t *n_obstacle, float *planes, int32_t *cluster_labels, uint32_t *n_clusters);

/* The two clouds the node publishes after segment(): ground points as pcl::PointXYZRGBL(x, y, z, 220, 220, 220,
 * label 0), obstacle points as (x, y, z, 0, 255, 0, label 1) (src/processor.cpp:152-163), whose 32-byte records
 * are memcpy'd into the data[] of the outgoing PointCloud2 messages (src/conversions.cpp:164-193).  Record layout
 * (PCL 1.12 point_types): float x, y, z, 1.0f | uint8 b, g, r, a = 255 | uint32 label | 8 zero bytes.  Records
 * of the LAST host segmentation call of this context (lpx_segment* / lpx_segment_cluster*), in output-cloud order;
 * each array needs room for 32 bytes per point of that cloud (n_ground / n_obstacle as returned by that call: the
 * library remembers those counts and never copies more).  Valid until the next call on the context other than
 * lpx_cluster_groups / lpx_cluster_hulls: a lpx_cluster of another cloud, a new segmentation or any device / batch
 * entry point re-initialises the frame state, and lpx_coloured_clouds then returns LPX_ERR_ARG. */
int lpx_coloured_clouds(lpx_ctx *ctx, void *ground_records, void *obstacle_records, uint32_t *n_ground,
                        uint32_t *n_obstacle);

/* Cluster regrouping done by the caller right after cluster() (reference src/processor.cpp:180-200):
 * the points of every valid cluster, clusters in label order, points in ascending index order,
 * INVALID dropped.  Works on the labels of the LAST lpx_cluster / lpx_segment_cluster call of this
 * context (still resident on the device): offsets[n_clusters + 1], indices[n_valid] (indices into the
 * clustered cloud), *n_valid = offsets[n_clusters].  m and n_clusters must be that call's, and no other call but
 * lpx_cluster_groups / lpx_cluster_hulls / lpx_coloured_clouds may have run on the context since (a look-ahead
 * lpx_segment overwrites the labels): otherwise LPX_ERR_ARG, never the groups of another cloud (lpx_cluster_epoch). */
int lpx_cluster_groups(lpx_ctx *ctx, uint32_t m, uint32_t n_clusters, uint32_t *offsets, uint32_t *indices,
                       uint32_t *n_valid);

/* Per-cluster 2-D convex hulls (N3): the counterpart of the convex branch of findOrderedConcaveOutlines, clusters
 * with fewer than 20 points (reference src/polygon_simplification.cpp:96-115; max_points = 20) -- Andrew's monotone
 * chain on (x, y), counter-clockwise from the lowest (x, y) point, collinear points and duplicates are not vertices.
 * RESTATED ALGORITHM, NOT VERIFIED AGAINST THE REFERENCE: the reference takes geom::constructConvexHull from its
 * Convex-Hull git submodule, which its checkout does not vendor (empty directory), so vertex order and the
 * collinear / duplicate conventions are this library's, checked against known answers and against
 * scipy's qhull vertex sets only.  Documented parity is limited to that < 20-point branch: findOrderedConvexOutlines
 * (:32-80) switches to Chan's algorithm above 1000 points in the same absent submodule -- max_points = UINT32_MAX
 * gives monotone-chain hulls of every cluster, with no claim about the reference's output there.  Clusters with
 * max_points or more points get an empty hull (the reference's concave branch is out of scope).  Works on the labels of the LAST clustering call of this context: hull_offsets[n_clusters + 1],
 * hull_indices (indices into the clustered cloud) and hull_xy (x, y pairs; may be NULL) hold up to m entries. */
int lpx_cluster_hulls(lpx_ctx *ctx, uint32_t m, uint32_t n_clusters, uint32_t max_points, uint32_t *hull_offsets,
                      uint32_t *hull_indices, float *hull_xy, uint32_t *n_hull_points);

/* ---- device-resident entry points (asynchronous on the context stream) ---------------------- */

/* Same as lpx_segment_cluster with every pointer a DEVICE pointer; counts[4] receives
 * {n_ground, n_obstacle, n_clusters, status(0 = ok, else -LPX_ERR_*)} on the device.
 * planes/cluster_labels may be NULL.  Nothing is copied to the host and nothing synchronises. */
int lpx_segment_cluster_device(lpx_ctx *ctx, const void *d_pts, size_t stride_bytes, uint32_t n,
                               const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg, uint32_t *d_labels,
                               uint32_t *d_ground_idx, uint32_t *d_obstacle_idx, float *d_planes,
                               int32_t *d_cluster_labels, uint32_t *d_counts);
/* n_frames clouds in ONE launch chain (a sensor rig's clouds of one tick, or a backlog of frames): every
 * kernel covers all frames, so the ~50 dependent launches of the chain are paid once per batch and each
 * launch has n_frames times the workgroups.  Per frame the results are identical to
 * lpx_segment_cluster_device.  Layout: frame b's records start at d_pts + b * frame_pitch * stride_bytes
 * and hold n_points[b] <= frame_pitch points (n_points is a HOST array); d_labels, d_ground_idx,
 * d_obstacle_idx and d_cluster_labels are pitched by frame_pitch elements, d_planes (may be NULL) by
 * 4 * number_of_planar_partitions floats, d_counts by 4 words.  Needs a context from lpx_create_batch with
 * max_frames >= n_frames. */
int lpx_segment_cluster_batch_device(lpx_ctx *ctx, uint32_t n_frames, const void *d_pts, size_t stride_bytes,
                                     uint32_t frame_pitch, const uint32_t *n_points, const lpx_seg_cfg *seg_cfg,
                                     const lpx_clu_cfg *clu_cfg, uint32_t *d_labels, uint32_t *d_ground_idx,
                                     uint32_t *d_obstacle_idx, float *d_planes, int32_t *d_cluster_labels,
                                     uint32_t *d_counts);
/* device forms of the PointCloud2 entry points above.  d_ground_idx / d_obstacle_idx are the index lists the
 * segmentation of the same frame(s) wrote; the batch form takes arrays pitched like the batch entry point's
 * (records pitched by frame_pitch * 32 bytes).  Counts stay on the device. */
int lpx_segment_cluster_fields_device(lpx_ctx *ctx, const void *d_data, uint32_t point_step, uint32_t off_x,
                                      uint32_t off_y, uint32_t off_z, uint32_t n, const lpx_seg_cfg *seg_cfg,
                                      const lpx_clu_cfg *clu_cfg, uint32_t *d_labels, uint32_t *d_ground_idx,
                                      uint32_t *d_obstacle_idx, float *d_planes, int32_t *d_cluster_labels,
                                      uint32_t *d_counts);
int lpx_segment_cluster_batch_fields_device(lpx_ctx *ctx, uint32_t n_frames, const void *d_data, uint32_t point_step,
                                            uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t frame_pitch,
                                            const uint32_t *n_points, const lpx_seg_cfg *seg_cfg,
                                            const lpx_clu_cfg *clu_cfg, uint32_t *d_labels, uint32_t *d_ground_idx,
                                            uint32_t *d_obstacle_idx, float *d_planes, int32_t *d_cluster_labels,
                                            uint32_t *d_counts);
/* LIFETIME OF THE INPUT.  The two coloured-cloud calls below read x, y, z of every point from the INPUT RECORDS of the
 * segmentation call they follow -- d_pts / d_data of that call, where the caller put them: the segmentation keeps no
 * copy of the cloud (16 bytes per point less written per frame).  That array must therefore stay allocated and
 * unmodified until the coloured-cloud call has been enqueued AND has run (lpx_synchronize, or stream order on a
 * caller-provided stream).  A caller that recycles its upload buffer earlier (a ring of staging buffers, a caching
 * allocator) calls lpx_set_record_copy(ctx, 1) once: the segmentation then keeps its own copy of the coordinates in the
 * context's arena and these calls read that instead.  The host form, lpx_coloured_clouds, is not affected (the host
 * calls own their staging buffer). */
int lpx_set_record_copy(lpx_ctx *ctx, int on);
int lpx_coloured_clouds_device(lpx_ctx *ctx, const uint32_t *d_ground_idx, const uint32_t *d_obstacle_idx,
                               void *d_ground_records, void *d_obstacle_records);
int lpx_coloured_clouds_batch_device(lpx_ctx *ctx, uint32_t n_frames, uint32_t frame_pitch,
                                     const uint32_t *d_ground_idx, const uint32_t *d_obstacle_idx,
                                     void *d_ground_records, void *d_obstacle_records);
int lpx_segment_device(lpx_ctx *ctx, const void *d_pts, size_t stride_bytes, uint32_t n, const lpx_seg_cfg *cfg,
                       uint32_t *d_labels, uint32_t *d_ground_idx, uint32_t *d_obstacle_idx, float *d_planes,
                       uint32_t *d_counts);
int lpx_cluster_device(lpx_ctx *ctx, const void *d_pts, size_t stride_bytes, uint32_t m, const lpx_clu_cfg *cfg,
                       int32_t *d_labels, uint32_t *d_counts);

/* device form of lpx_cluster_groups: d_offsets needs n_clusters + 1 (at most m + 1) entries, d_indices m */
int lpx_cluster_groups_device(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, uint32_t *d_offsets,
                              uint32_t *d_indices);

/* device form of lpx_cluster_hulls on the CSR lpx_cluster_groups_device wrote (same labels, same cloud) */
int lpx_cluster_hulls_device(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, const uint32_t *d_offsets,
                             const uint32_t *d_indices, uint32_t max_points, uint32_t *d_hull_offsets,
                             uint32_t *d_hull_indices, float *d_hull_xy);

/* ---- N1: frames from binary PCD files, pinned memory, double-buffered stream ------------------ */

/* Counterpart of the reference's input harness (Dataloader::preload_point_clouds, src/dataloader.cpp:128-153, which
 * loads every .pcd file of data/ with pcl::io::loadPCDFile): binary PCD v0.7, header as in data/0000000000.pcd:1-11.  The
 * payload is used AS STORED (records of point_step bytes, float32 x / y / z at off_x/y/z) through the *_fields
 * entry points, so a file goes disk -> pinned host memory -> device without a decode or a copy in between. */
typedef struct
{
    uint32_t n_points;   /* POINTS (or WIDTH * HEIGHT) */
    uint32_t point_step; /* bytes per record = sum of SIZE * COUNT */
    uint32_t off_x, off_y, off_z;
    uint32_t n_fields;
} lpx_pcd_info;

/* page-locked host memory (hipHostMalloc) for lpx_pcd_load destinations and result arrays */
int lpx_host_alloc(void **out, size_t bytes);
void lpx_host_free(void *p);
/* header only / header + exactly POINTS records into dst (trailing bytes of the file are ignored, as PCL does).
 * LPX_ERR_ARG: unreadable, not `DATA binary`, or x / y / z are not float32 scalars; LPX_ERR_CAPACITY: dst too small */
int lpx_pcd_info_read(const char *path, lpx_pcd_info *info);
int lpx_pcd_load(const char *path, void *dst, size_t dst_capacity_bytes, lpx_pcd_info *info);

/* A feeder preloads a list of files into ONE pinned arena (like the reference preloads all clouds) ... */
typedef struct lpx_feeder lpx_feeder;
int lpx_feeder_create(int device, const char *const *paths, uint32_t n_files, lpx_feeder **out);
void lpx_feeder_destroy(lpx_feeder *f);
uint32_t lpx_feeder_frames(const lpx_feeder *f);
const void *lpx_feeder_frame(const lpx_feeder *f, uint32_t i, lpx_pcd_info *info); /* pinned records of frame i */
const char *lpx_feeder_last_error(const lpx_feeder *f);

/* ... and runs frames through a batch context with two device buffer sets: while chain k computes, the records
 * of chain k + 1 travel H2D on a copy stream and the results of chain k - 1 travel D2H (exact sizes) on another,
 * so PCIe in both directions overlaps the kernels.  Host result arrays (ordinary or pinned memory) are pitched by
 * frame_pitch elements per frame; planes (may be NULL) by 4 * number_of_planar_partitions floats, counts by 4
 * words {n_ground, n_obstacle, n_clusters, status}.  Per frame the results equal lpx_segment_cluster's.  A frame
 * the device flags (status != 0: non-finite coordinates, LPX_ERR_RANGE; neighbour lists that do not fit a context
 * in LPX_NEIGHBOURS_LISTS mode, LPX_ERR_CAPACITY -- batch contexts default to LPX_NEIGHBOURS_SEARCH, which has no
 * list workspace) makes the run return that code for the first such frame; counts[] holds every frame's status. */
typedef struct
{
    uint32_t *labels, *ground_idx, *obstacle_idx;
    int32_t *cluster_labels;
    float *planes;
    uint32_t *counts;
    uint32_t frame_pitch;
} lpx_stream_out;
int lpx_feeder_run(lpx_feeder *f, lpx_ctx *batch_ctx, const uint32_t *frame_ids, uint32_t n_frames,
                   const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg, const lpx_stream_out *out);
/* The same over n_ctx batch contexts of equal slot count B: chain k (frames [k B, (k+1) B) of frame_ids) runs on
 * context k % n_ctx, every context with its own buffer sets and host thread, so several chains compute at once while
 * PCIe moves the others' data; the copy streams are shared by the pipelines (two per direction; nothing ever waits
 * on them).  Same results, same output layout.  Keep GPU_MAX_HW_QUEUES above the number of contexts (the library
 * sets 32 when it is loaded and the variable is unset): streams that share a hardware queue run one after the other. */
int lpx_feeder_run_multi(lpx_feeder *f, lpx_ctx *const *batch_ctxs, uint32_t n_ctx, const uint32_t *frame_ids,
                         uint32_t n_frames, const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                         const lpx_stream_out *out);

/* ---- measurement ---------------------------------------------------------------------------- */

/* Stage timing with HIP events on the context stream.  enable=1 records an event pair around
 * every stage of subsequent calls (adds a few microseconds per stage; leave off when timing
 * whole-frame throughput). */
int lpx_profile_enable(lpx_ctx *ctx, int enable);
/* number of stages; names are static strings */
int lpx_profile_stage_count(void);
const char *lpx_profile_stage_name(int stage);
/* Synchronises, then returns for each stage the accumulated milliseconds and number of launches
 * since the last reset; arrays of lpx_profile_stage_count() entries. */
int lpx_profile_read(lpx_ctx *ctx, float *ms, uint32_t *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* LPX_H */
