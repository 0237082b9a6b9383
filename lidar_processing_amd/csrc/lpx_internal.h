// lpx_internal.h -- shared declarations of the MI355X (gfx950) hot-path library.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "lpx.h"

// ------------------------------------------------------------------------------------------------
// stages (for lpx_profile_*)
// ------------------------------------------------------------------------------------------------
enum lpx_stage
{
    ST_INGEST = 0,   // AoS -> SoA + x keys
    ST_XSORT,        // radix sort by (x, index)
    ST_GATHER,       // x-sorted SoA + composite (segment, z) keys
    ST_ZSORT,        // radix sort of (segment, z) keys
    ST_SEEDS,        // per-segment seed thresholds
    ST_PLANE,        // fused inlier test + moment accumulation + 3x3 solve (I launches + seed pass)
    ST_COMPACT,      // final flags -> labels, ground/obstacle lists, obstacle SoA
    ST_KD_BUILD,     // kd-tree layout (libstdc++ introselect order) + pre-order rank layout
    ST_NB_SCAN,      // union-find hooking over the neighbour lists
    ST_NB_FILL,      // radius neighbour lists (count + allocate + fill, one launch)
    ST_CC,           // flatten roots, sort members by (root, index), component ranges
    ST_REPLAY,       // ordered FEC replay, one wavefront per component
    ST_LABELS,       // dense relabel
    ST_GROUPS,       // cluster regrouping (CSR by label), reference src/processor.cpp:180-200
    ST_COUNT
};

// ------------------------------------------------------------------------------------------------
// device-side frame state (one per context, lives in HBM)
// ------------------------------------------------------------------------------------------------
#define LPX_RS_STRIPES 16u
// Every word that kernels bump atomically sits on a 128-byte line of its own: atomics serialise per line at L2, and
// the sizes in the first line are read by every workgroup of every kernel -- they must not queue behind the bumps of
// a counter that happens to share their line.
struct alignas(128) FrameStripe
{
    uint64_t v;
};
struct alignas(128) FrameState
{
    // line 0: sizes and status, written a handful of times per frame
    uint32_t n_ground;
    uint32_t n_obstacle;  // M: number of points handed to clustering
    uint32_t n_clusters;
    uint32_t n_in;        // points of the input cloud of this frame slot
    uint32_t has_far;     // some coordinate has |v| >= 2048 m: the plane kernels take the wide-moment path for it
    uint32_t status;      // 0 ok, else -LPX_ERR_*
    alignas(128) uint64_t nb_total;     // words of exact-length list storage required (workspace region [0, cap_nb))
    alignas(128) uint32_t n_roots;      // connected components of the d-graph
    alignas(128) uint32_t root_cursor;  // work queue head of the replay
    alignas(128) uint32_t n_cells;      // occupied cells of the component grid
    alignas(128) uint32_t cell_cursor;  // points handed out to the cells' contiguous runs
    // statistics (a few bumps per wavefront at the end of a kernel)
    alignas(128) uint64_t replay_entries;  // neighbour entries read by the replay (lists of expanded points)
    uint64_t nb_entries;                   // (unstriped remainder of the entries written, see ent_stripe)
    uint64_t cand_total;      // candidates distance-tested by the replay's searches (expansion-driven path)
    uint32_t n_expansions;    // radius_search calls the reference would have made
    uint32_t n_windows;       // queue windows with at least one expansion (expansion-driven path)
    uint32_t n_overflow;      // searches redone by the sequencer because the list did not fit its LDS region
    uint32_t n_single;        // single-point sets, settled by cc_ranges_kernel (not in the work list of n_roots)
    uint32_t max_obstacle;    // slot 0 only: largest n_obstacle of the frames of the call (sizes the next call's LDS bitmaps)
    uint32_t pad_stats;
    // position-bound checksum of the obstacle cloud the compaction wrote (single-frame calls): what lets lpx_cluster
    // recognise the cloud lpx_segment left on the device and skip the upload (lpx_obstacle_mix)
    alignas(128) uint64_t obs_hash;
    uint64_t obs_hash2;       // second, independent witness (lpx_obstacle_mix2): same line, same atomics
    // The single-pass region of the list workspace is handed out from LPX_RS_STRIPES sub-regions with a cursor each
    // (group g bumps cursor g % LPX_RS_STRIPES): thousands of bumps of ONE word per frame serialise at L2.
    FrameStripe rs_stripe[LPX_RS_STRIPES];
    FrameStripe ent_stripe[LPX_RS_STRIPES];  // neighbour entries written, likewise striped
};

// One point's share of the obstacle-cloud checksum: its three coordinate words bound to its position in the cloud.
// The checksum is the wrapping sum of the shares (order of accumulation irrelevant: the device adds per wavefront).
__host__ __device__ static inline uint64_t lpx_obstacle_mix(uint32_t i, uint32_t xb, uint32_t yb, uint32_t zb)
{
    uint64_t h = (((uint64_t)xb << 32) | yb) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h += ((uint64_t)zb << 32) | i;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

// The second witness of the same cloud: other constants, another structure (the position is bound by a multiplication
// of its own, z enters first, x and y are rotated apart), so that a crafted or accidental collision has to hit two
// unrelated 64-bit wrapping sums at once.  lpx_cluster takes the resident path only when BOTH sums (and the size) agree.
__host__ __device__ static inline uint64_t lpx_obstacle_mix2(uint32_t i, uint32_t xb, uint32_t yb, uint32_t zb)
{
    uint64_t h = ((uint64_t)(i + 1u) * 0xD6E8FEB86659FD93ull) ^ (((uint64_t)zb << 32) | xb);
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
    h += ((uint64_t)yb << 17) | ((uint64_t)yb >> 15);
    h ^= (uint64_t)xb << 41;
    h *= 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 31;
    return h;
}

static inline uint64_t lpx_entries_written(const FrameState &f)
{
    uint64_t s = f.nb_entries;
    for (int i = 0; i < 16; ++i)
        s += f.ent_stripe[i].v;
    return s;
}

// Occupancy bitmap of the component grid: 2^16 bits (8 KiB: small enough to sit in the LDS of every linking workgroup)
// addressed by the low bits of a cell's position -- 64 x 64 x 16 positions (x, y, z) -- with the bit inside the word
// permuted by a hash of the 64 x 64 tile the cell lies in.  A set bit says "some cell maps here" (a false positive costs
// one probe of the table, nothing else); a clear bit says the partner cell does not exist, without touching the table.
// (Rounds 4-5 used 64 x 32 x 32 with bit = z & 31: the obstacle cells of a street scene span ~10 levels of z, so two
// thirds of every word stayed empty and aliases from different tiles met in the same few bits -- 55 % of the far pass's
// partner tests passed where 20 % of the partners exist.)
#define LPX_CELL_BITS_WORDS 2048u
#define LPX_CELL_BITS_BYTES (4u * LPX_CELL_BITS_WORDS)
__host__ __device__ static inline uint32_t lpx_cell_bit_index(uint32_t cx, uint32_t cy, uint32_t cz)
{
    const uint32_t tile = (cx >> 6) * 0x9E3779B1u + (cy >> 6) * 0x85EBCA77u;
    const uint32_t word = ((cx & 63u) << 5) | ((cy >> 1) & 31u);
    const uint32_t bit = ((((cy & 1u) << 4) | (cz & 15u)) ^ (tile >> 27)) & 31u;
    return (word << 5) | bit;  // word index in the high bits, bit inside the word in the low five
}

#define LPX_ACC_WORDS 16  // n, sx, sy, sz, 6 x (hi, lo)
// blocks of a plane pass (and rows of seg_part / blk_counts) a cloud of n points can need at most: a segment's blocks
// start on a multiple of four points (16-byte loads), which can add one block per segment
#define LPX_SEG_MAX_BLOCKS(n) ((size_t)(n) / 2048 + 2 * LPX_MAX_PARTITIONS + 2)
#define LPX_FAR_WORDS 24  // moments of the points beyond +-2048 m: n, sx, sy, sz, 6 x (hh, hl, ll) limbs, 2 spare

struct SegState  // per segment
{
    float lo_excl;   // seeds: z > lo_excl
    float hi_incl;   //        z <= hi_incl
    uint32_t has_seeds;
    uint32_t failed;  // sticky: treat everything as obstacle (reference src/segmentation.cpp:251-259)
    float plane[4];
    uint32_t fitted;  // at least one plane was fitted
    float thr;        // orthogonal_distance_threshold * |normal| of `plane`
    uint32_t pad[2];
};

// one candidate chunk of a kd group: `count` consecutive pre-order ranks from `rank`, and their bounding box
struct ChunkRec
{
    uint32_t rank, count;
    float lo[3], hi[3];
};

// what a list-mode clustering asked of the list workspace (one record per frame slot, in pinned host memory, written by
// thread 0 of flatten_kernel once the neighbour kernel of the call is done; seq last)
struct LpxListStat
{
    uint64_t nb_total;       // words the exact-length region was asked for (groups that found no single-pass room)
    uint64_t stripe_max;     // largest demand on one of the LPX_RS_STRIPES sub-regions of the single-pass region
    uint32_t status, n_obstacle;
    uint32_t seq;            // lpx_ctx::list_seq of the call that wrote the record
    uint32_t pad;
    uint64_t entries;        // list entries written (what decides the form of the edge check: lpx_lists.hip)
    uint64_t pad2;
};

struct Buf
{
    void *p = nullptr;
    size_t bytes = 0;
};

// ------------------------------------------------------------------------------------------------
// Frame batching.  A context owns `batch` identical frame slots.  Every internal buffer is a
// sub-range of ONE arena per slot, so slot b of any buffer is at (char *)buf.p + b * fstride (the
// neighbour lists live in a second arena with stride nb_fstride so they can grow on their own).
// A launch covers the slots of a call with gridDim.z; caller-provided arrays are pitched by
// `upitch` elements per frame.  With one frame everything is offset 0.
// ------------------------------------------------------------------------------------------------
#define LPX_NB_BUCKET_DEFAULT 64u
#define LPX_SORT_TILE 2048u  // keys per radix-sort block (lpx_primitives.hip)
#define LPX_GROUP_CHUNKS 64u // candidate chunks kept per kd group: one per lane of the searching wavefront

struct FV
{
    size_t fs;        // bytes between the slots of the frame arena
    size_t fs_nb;     // bytes between the slots of the neighbour arena
    uint32_t upitch;  // elements between the frames of caller arrays
    uint32_t pad;
};

struct NArr
{
    uint32_t v[LPX_MAX_BATCH];
};

struct lpx_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // ---- overlapped tail (lpx_set_overlap, batch contexts): the replay and the label kernels of a chain run on a
    // second stream while the context's stream already takes the front end of the next chain, which works on a SECOND
    // set of frame slots (`twin`, a context of its own on the same stream).  See lpx_api.hip: overlap_*.
    bool overlap = false;          // primary: calls alternate between this context and its twin
    lpx_ctx *twin = nullptr;       // primary: the second slot set
    lpx_ctx *last = nullptr;       // primary: the slot set of the last batch call (statistics, coloured clouds)
    uint32_t flip = 0;
    hipStream_t tail_stream = nullptr;  // where this slot set's tail runs (from the device's pool)
    hipEvent_t ev_front = nullptr, ev_tail = nullptr;
    // ---- forked front end (lpx_set_fork): the component grid of a chain runs on a side stream beside the kd build and
    // the chunk tables of the same chain -- both need nothing but the obstacle cloud, and both are chains of
    // latency-bound launches
    bool fork = false;
    hipStream_t fork_stream = nullptr;  // from the device's pool of side streams
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool split_tail = false;       // the call being enqueued hands its tail to tail_stream
    bool enqueued = false;         // the batch call in progress got past its checks and uses this slot set
    bool tail_pending = false;     // a tail was enqueued since the stream last waited for ev_tail
    char err[512] = {0};

    uint32_t cap_n = 0;        // points per frame slot
    uint64_t cap_nb = 0;       // neighbour entries per frame slot
    // List workspace per point of the slot: 64 words of exact-length lists + 192 words of single-pass lists (126 MB for a
    // 123k-point frame, 5.1 GB for a 5M-point one; rounds 1-5: 256 + 512, 379 MB / 15.4 GB) -- enough for the reference's
    // frames at d = 0.5 m but for the densest and for BASELINE's synthetic clouds.  It GROWS on evidence: every list-mode
    // clustering leaves what it asked for in pinned memory (LpxListStat, written by flatten_kernel), the next call on the
    // context sizes the regions to 1.25 x the largest demand seen (lists_grow_on_evidence, lpx_api.hip).  A frame that
    // outgrows the workspace before that still reports LPX_ERR_CAPACITY: the host calls repeat it (cluster_resident),
    // a device call's caller sees the status of that ONE frame and may lpx_reserve() ahead for dense scenes.
    uint32_t nb_per_point = 64;
    uint64_t cap_rs = 0;       // words of the single-pass list region per frame slot (behind the cap_nb words)
    uint32_t rs_per_point = 192;
    struct LpxListStat *h_liststat = nullptr;  // pinned, one record per frame slot
    uint32_t list_seq = 0;         // list-mode clusterings enqueued on this context
    uint32_t list_seq_seen = 0;    // ... whose demand has been looked at
    int list_short = -1;           // the lists of the last frame looked at were short (1) / long (0); -1: not known yet
    uint32_t batch = 1;        // frame slots
    uint32_t cur_b = 1;        // frames of the call being enqueued (gridDim.z)
    uint32_t in_off[3] = {0, 4, 8};  // byte offsets of x, y, z inside a record of the call being enqueued
    uint32_t upitch = 0;       // pitch of the caller arrays of that call
    void *arena = nullptr, *nb_arena = nullptr;
    size_t fstride = 0, nb_fstride = 0;
    size_t fs_tag = 0;         // fstride | families whose launches re-read their workgroup number (what kernels get)

    // ---- segmentation buffers (cap_n) ----
    Buf in_aos;                // staging for host input
    Buf rec_out;               // staging for the coloured-cloud records handed to the host
    uint32_t last_n = 0;       // points of the last single-frame segmentation (bounds n_ground + n_obstacle)
    // host-side memory of the last HOST segmentation call: what lpx_coloured_clouds may copy out.  Cleared by every
    // other call that re-initialises the frame state or may move the arena (cluster, device and batch entry points).
    bool seg_valid = false;
    uint32_t seg_ground = 0, seg_obstacle = 0;
    // the obstacle cloud of that call is still resident exactly as the compaction left it (SoA + kd input): lpx_cluster
    // of a cloud with this count and checksum runs on it without an upload; cleared once a clustering has consumed it
    bool seg_fresh = false;
    uint64_t seg_hash = 0, seg_hash2 = 0;
    // host-side memory of the last HOST clustering (lpx_cluster, lpx_segment_cluster*): what d_clabels holds, i.e. what
    // lpx_cluster_groups / lpx_cluster_hulls may regroup.  Cleared by begin_call -- every other call, a look-ahead
    // clustering enqueued by lpx_segment included, may overwrite the labels -- and checked against the caller's m /
    // n_clusters, so that groups or hulls of ANOTHER cloud are an LPX_ERR_ARG, never a silent answer.
    bool clu_valid = false;
    uint32_t clu_m = 0, clu_clusters = 0;
    uint64_t clu_epoch = 0;        // counts the host clusterings of this context (lpx_cluster_epoch)
    // Look-ahead of the two-call form (lpx_set_lookahead): once an lpx_cluster call has found the obstacle cloud of the
    // lpx_segment call before it resident, the next lpx_segment enqueues the clustering with that call's configuration
    // right behind its own kernels and returns as soon as ITS results are down; a matching lpx_cluster then only waits
    // for the chain that is already running.
    int lookahead = 1;             // 1: learn and look ahead, 0: never
    bool la_armed = false;         // the last lpx_cluster call was served from the resident cloud: la_cfg is what it asked for
    bool la_pending = false;       // a clustering enqueued by lpx_segment is (or was) running on the resident cloud
    bool la_failed = false;        // ... could not be enqueued (the segmentation call still succeeds)
    lpx_clu_cfg la_cfg = {};
    uint64_t la_hits = 0;          // lpx_cluster calls that found their clustering already enqueued (tests)
    hipStream_t copy_stream = nullptr;  // downloads of the segmentation beside the look-ahead clustering
    hipEvent_t ev_seg = nullptr;        // the segmentation's kernels (and the snapshot of its frame state) are done
    void *h_frame = nullptr;            // pinned: that snapshot
    Buf pts4;                  // the cloud in original order, float4 {x, y, z, 0} per point (only when rec_direct is off)
    // where the records of the last segmentation call lie (device memory: the caller's array or the staging buffer of a
    // host call) and how (stride, field offsets, records between the frames of a batch): with rec_direct the last pass
    // of the x sort and lpx_coloured_clouds* read the coordinates from there and no copy is made
    const void *rec_ptr = nullptr;
    size_t rec_stride = 0;
    uint32_t rec_off[3] = {0, 4, 8};
    uint32_t rec_pitch = 0;
    bool rec_direct = false;
    bool keep_copy = false;    // lpx_set_record_copy: the segmentation keeps its own copy of the coordinates (pts4), so that
                               // lpx_coloured_clouds*_device never read the caller's input buffer after the call returned
    Buf key_a, key_b;          // u32 keys ping-pong
    Buf val_a, val_b;          // u32 values ping-pong
    Buf key64_a, key64_b;      // u64 keys ping-pong
    Buf XS, YS, ZS;            // x-sorted SoA
    Buf flags;                 // u8 per sorted position
    Buf hist;                  // radix histograms / scan scratch
    Buf seg_state;             // SegState[2][LPX_MAX_PARTITIONS]: the state pass t works with lives in set t & 1
    Buf seg_part;              // int64 [2][blocks of a pass][LPX_ACC_WORDS]: the moment partials pass t leaves for pass t + 1
    Buf seg_far;               // int64 [3][LPX_MAX_PARTITIONS][LPX_FAR_WORDS]: far-point moments of pass t in set t % 3
    Buf blk_counts;            // per block ground / obstacle counts
    Buf d_labels, d_gidx, d_oidx, d_planes, d_counts;  // outputs for host API
    // ---- clustering buffers ----
    Buf OX, OY, OZ;            // obstacle SoA (cap_n)
    Buf nodes;                 // float4 kd nodes, array (in-order) layout
    Buf nodes_pre;             // the same nodes in pre-order rank layout
    Buf lpos, rpos;            // partition scratch
    uint32_t ix_bucket = LPX_NB_BUCKET_DEFAULT;  // most nodes of a kd group of the search tables: 64, or 32 on scenes whose
                                                 // searches test many candidates per hit (adapted from h_search)
    uint64_t *h_search = nullptr;  // pinned: {hits, -, candidates, expansions | windows, overflows | single sets, largest
                                   // obstacle count} of the previous search-mode call (slot 0's statistics)
    int reg_index = -1;        // entry of this context in the registry behind lpx_active_frame_slots
    Buf kd_state;              // introselect state of the ranges of a top kd level (multi-workgroup rounds)
    Buf nb_len, nb_off;        // u32 len, u32 off (cap_n + 1)
    Buf nb_idx;                // cap_nb words: neighbour index | (within the absorb radius) << 31
    Buf parent;                // union-find
    Buf cc_lo, cc_hi;          // member range per root
    Buf state;                 // u8 replay state
    Buf seed_of;               // i32
    Buf queue;                 // u32
    Buf valid;                 // u32 per seed
    Buf d_clabels;
    // ---- expansion-driven search (default path): no neighbour lists at all ----
    Buf grp_of;                // float4 per point {x, y, z, kd group (bucket or upper node) the point is a query of}: what a
                               // queue window of the replay gathers per point, in ONE 16-byte record
    Buf chunks;                // ChunkRec [groups][LPX_GROUP_CHUNKS]: candidate chunks (pre-order rank, count, box) of a group
    Buf cell_key;              // u64 [cell_cap]: occupied cells of the component grid (open addressing), then the
                               // occupancy bitmap of the cells (LPX_CELL_BITS_BYTES, lpx_cell_bit)
    Buf cell_rep, cell_parent; // u32 [cell_cap]: points of the cell / union-find over cells
    Buf cell_start;            // u32 [cell_cap]: where the cell's points begin in the cell-ordered copy
    Buf cell_of;               // u32 per point: its cell slot
    Buf cell_xyz;              // float4 [3][cell_cap]: the point that claimed the cell; low / high corner of the box of its points
    Buf cell_pts;              // float4 per point: the cell-ordered copy of the cloud ({x, y, z, index} runs per cell)
    Buf cell_list;             // u32 per point: the occupied cells' table slots
    uint32_t cell_cap = 0;     // slots per frame slot (power of two >= 2 * cap_n)
    bool arena_has_search = false;  // the arena holds the search tables above (contexts that have been in search mode)
    bool use_lists = false;    // lpx_dbg_use_lists: materialise every radius list (the round-1 path, kept for tests)
    Buf frame;                 // FrameState
    // ---- pinned host staging ----
    void *h_pinned = nullptr;
    size_t h_pinned_bytes = 0;

    void *dbg_buf = nullptr;   // optional per-group statistics of the neighbour kernel (tools only)
    Buf dbg_store;
    bool attr_kd = false, attr_replay = false, attr_search = false;  // hipFuncSetAttribute done for this context's device
    bool exact_lists_only = false;  // capacity retry: count every list, so nb_total is the exact requirement

    // profiling
    bool profiling = false;
    hipEvent_t ev_a[ST_COUNT], ev_b[ST_COUNT];
    bool ev_ready = false;
    float st_ms[ST_COUNT] = {0};
    uint32_t st_launches[ST_COUNT] = {0};
    // deferred event pairs recorded in the current call
    struct Pending { int stage; hipEvent_t a, b; };
    Pending *pending = nullptr;
    int n_pending = 0, cap_pending = 0;
};

static inline FV lpx_fv(const lpx_ctx *ctx)
{
    FV fv;
    fv.fs = ctx->fs_tag;
    fv.fs_nb = ctx->nb_fstride;
    fv.upitch = ctx->upitch;
    fv.pad = 0;
    return fv;
}

int lpx_fail(lpx_ctx *ctx, int code, const char *fmt, ...);
int lpx_ensure(lpx_ctx *ctx, Buf &b, size_t bytes);
int lpx_ensure_capacity(lpx_ctx *ctx, uint32_t n, uint64_t nb);

// Environment knobs (LPX_SKIP, LPX_POISON, LPX_REMAP, LPX_RS_*, LPX_KD_*, ...) exist only in the DEVELOPMENT build
// (make dev -> liblpx_dev.so, -DLPX_DEV_KNOBS: what tools/ and a few tests select with LPX_LIB).  The release library
// reads no LPX_* variable at all: nothing in a host's environment can change what it computes or how.
#ifdef LPX_DEV_KNOBS
#define LPX_KNOB(name) getenv(name)
#else
#define LPX_KNOB(name) ((const char *)nullptr)
#endif

#define LPX_HIP(ctx, call)                                                                                          \
    do                                                                                                              \
    {                                                                                                               \
        hipError_t e_ = (call);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            return lpx_fail((ctx), LPX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                            __LINE__);                                                                              \
    } while (0)

struct StageTimer
{
    lpx_ctx *ctx;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(lpx_ctx *c, int s);
    ~StageTimer();
};

// ------------------------------------------------------------------------------------------------
// primitives (lpx_primitives.hip)
// ------------------------------------------------------------------------------------------------
// stable LSD radix sort, 8 bits per pass over key bits [0, bits); result ends in the *_a buffers
// (function copies if the pass count is odd).  n is a host upper bound; d_n (optional) the device count.
// gather (optional): the sorted values index a table of 16-byte records {x, y, z, .}; the LAST pass writes x / y / z of
// the records in sorted order itself (and leaves the sorted keys unwritten: *keys_out is then not to be read)
// records: either the arena's table of 16-byte records {x, y, z, .} in input order (stride 0: one table per frame slot) or
// the CALLER's records themselves (stride > 0: float32 x / y / z at byte offsets off[] of every stride-byte record, frame
// b of the call `pitch` records behind frame 0) -- then nobody has to write a copy of the cloud first
struct LpxSortGather
{
    const void *records;
    float *x, *y, *z;
    size_t stride = 0;
    uint32_t off[3] = {0, 4, 8};
    uint32_t pitch = 0;
};
// how a kernel reads x, y, z of record i (LpxSortGather / lpx_ctx::rec_*): mode 0 = 16-byte table entries, 1 = records
// whose x, y, z are the first three floats and which are 16-byte aligned (every PCL point type: ONE 16-byte load),
// 2 = 4-byte aligned fields, 3 = bytes (a PointCloud2 buffer with odd offsets)
struct LpxRecLayout
{
    size_t stride = 0;
    uint32_t ox = 0, oy = 4, oz = 8, pitch = 0, mode = 0;
};
static inline LpxRecLayout lpx_rec_layout(const void *records, size_t stride, const uint32_t *off, uint32_t pitch)
{
    LpxRecLayout l;
    l.stride = stride;
    l.ox = off[0], l.oy = off[1], l.oz = off[2];
    l.pitch = pitch;
    if (stride == 0)
        l.mode = 0;
    else if (off[0] == 0 && off[1] == 4 && off[2] == 8 && (((uintptr_t)records | stride) & 15u) == 0)
        l.mode = 1;
    else if ((((uintptr_t)records | stride | off[0] | off[1] | off[2]) & 3u) == 0)
        l.mode = 2;
    else
        l.mode = 3;
    return l;
}
#ifdef __HIPCC__
__device__ __forceinline__ float lpx_ld_bytes_f32(const char *p)
{
    const unsigned char *b = (const unsigned char *)p;
    return __uint_as_float((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
}
__device__ __forceinline__ float4 lpx_rec_xyz(const float4 *rec, uint32_t i, const LpxRecLayout &lay)
{
    if (lay.mode == 0)
        return rec[i];
    const char *p = (const char *)rec + (size_t)i * lay.stride;
    if (lay.mode == 1)
        return *(const float4 *)p;
    if (lay.mode == 2)
        return make_float4(*(const float *)(p + lay.ox), *(const float *)(p + lay.oy), *(const float *)(p + lay.oz), 0.0f);
    return make_float4(lpx_ld_bytes_f32(p + lay.ox), lpx_ld_bytes_f32(p + lay.oy), lpx_ld_bytes_f32(p + lay.oz), 0.0f);
}
#endif

// first_hist_ready: the producer of keys_a has left the tile histograms of the lowest key byte in lpx_sort_first_hist()
// iota_vals: vals_a[i] == i is MEANT, the array is never read (nobody has to write it)
// keys_below_n: every key of a frame is below that frame's element count *d_n (passes above log2 of it only copy);
// ignored together with `gather` (the gathering last pass has no copy-only form)
int lpx_sort_pairs(lpx_ctx *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n,
                   const uint32_t *d_n, uint32_t bits, uint32_t **keys_out, uint32_t **vals_out,
                   bool first_hist_ready = false, const LpxSortGather *gather = nullptr, bool iota_vals = false,
                   bool keys_below_n = false);
// where a kernel that produces the keys of an n-element sort may leave the first pass's tile histograms (block-major,
// 256 words per LPX_SORT_TILE keys), or null when the sort would not use them (tables beyond the fused-scan limit)
uint32_t *lpx_sort_first_hist(lpx_ctx *ctx, uint32_t n);
int lpx_sort_keys64(lpx_ctx *ctx, uint64_t *keys_a, uint64_t *keys_b, uint32_t n, const uint32_t *d_n, uint32_t bits,
                    uint64_t **keys_out);
// exclusive scan (u32 in, u32 out, in place allowed); total (u64) written to *d_total if not null.
int lpx_exclusive_scan(lpx_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *d_n,
                       uint64_t *d_total);

// ------------------------------------------------------------------------------------------------
// pipeline stages
// ------------------------------------------------------------------------------------------------
int lpx_run_segment(lpx_ctx *ctx, const void *d_pts, size_t stride, const uint32_t *n_points, const lpx_seg_cfg *cfg,
                    uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes);
// clustering of the obstacle SoA already in ctx->OX/OY/OZ, count in frame->n_obstacle (bound m_max);
// d_counts (optional) receives {n_ground, n_obstacle, n_clusters, status} from the last kernel;
// kd_ready: the tree of a previous attempt on the same cloud is kept
int lpx_run_cluster(lpx_ctx *ctx, uint32_t m_max, const lpx_clu_cfg *cfg, int32_t *d_labels, uint32_t *d_counts,
                    bool kd_ready);
// PointXYZRGBL records of the ground / obstacle clouds of the frames last segmented (X/Y/Z still resident)
int lpx_run_colour(lpx_ctx *ctx, uint32_t n_max, const uint32_t *d_gidx, const uint32_t *d_oidx, void *d_grec,
                   void *d_orec);
// AoS (device) -> ctx->OX/OY/OZ, sets frame->n_obstacle = m
int lpx_ingest_obstacles(lpx_ctx *ctx, const void *d_pts, size_t stride, uint32_t m);
// resets the FrameState of every slot of the call and stores the per-frame input sizes
int lpx_frame_init(lpx_ctx *ctx, const uint32_t *n_points, bool as_obstacles);
// {n_ground, n_obstacle, n_clusters, status} of every slot -> caller array (4 words per frame)
int lpx_write_counts(lpx_ctx *ctx, uint32_t *d_counts);

// CSR of the valid clusters from d_labels (m entries): d_offsets[n_clusters + 1], d_indices[n_valid]
int lpx_run_groups(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, uint32_t *d_offsets, uint32_t *d_indices);

// batch launch chain; offs = byte offsets of x, y, z in a record, or null for PCL records
int lpx_batch_impl(lpx_ctx *ctx, uint32_t n_frames, const void *d_pts, size_t stride, const uint32_t *offs,
                   uint32_t frame_pitch, const uint32_t *n_points, const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                   uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes, int32_t *d_clabels,
                   uint32_t *d_counts);
// N3: convex hulls of the small clusters from that CSR (results: hull offsets / point indices / xy)
int lpx_run_hulls(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, const uint32_t *d_offsets,
                  const uint32_t *d_indices, uint32_t max_points, uint32_t *d_hull_off, uint32_t *d_hull_idx,
                  float *d_hull_xy);

int lpx_kd_build(lpx_ctx *ctx, uint32_t m_max);
// thr_f: absorb threshold of the clustering (largest float <= (1-q)^2 d^2), stored as bit 31 of every list
// word; hook: also build the connected components
int lpx_neighbours(lpx_ctx *ctx, uint32_t m_max, float r2, float thr_f, bool hook);
// expansion-driven path: candidate chunks per kd group + the point -> group map; components from a uniform grid
uint32_t lpx_active_frame_slots(int device);  // frame slots of the contexts of the device that enqueued work lately
void lpx_note_enqueue(lpx_ctx *ctx);
// clear_grid: also empty the cell table of the component grid (lpx_grid_components(..., cleared = true) follows)
int lpx_group_index(lpx_ctx *ctx, uint32_t m_max, float r2, bool clear_grid);
bool lpx_cc_from_chunks(uint32_t m_max);  // components of the search path: chunk tables (large frames) or clique-cell grid
#ifdef LPX_DEV_KNOBS
bool lpx_cc_from_sweep(uint32_t m_max);   // ... or (development build, LPX_CC=sweep) a sweep over y-sorted x slabs
int lpx_sweep_components(lpx_ctx *ctx, uint32_t m_max, float r2);  // leaves the forest in ctx->parent
#endif
int lpx_grid_components(lpx_ctx *ctx, uint32_t m_max, float r2, uint32_t *d_root, uint32_t *d_iota, bool cleared);
// the last kernel of the grid path, on ctx->stream: roots per point (+ the first histogram of the sort that follows)
int lpx_grid_flatten(lpx_ctx *ctx, uint32_t m_max, uint32_t *d_root, uint32_t *d_iota, uint32_t *first_hist);
// the neighbour-list workspace is only allocated for the list path
int lpx_ensure_lists(lpx_ctx *ctx);

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__

#define WAVE 64

// XCD-affine launch geometry.  A launch covers the frames of a call with gridDim.z, and the hardware deals the
// workgroups of a launch round-robin over the 8 XCDs in linear order (x fastest, z slowest): with the plain
// (blockIdx.x, blockIdx.z) = (tile, frame) reading, every XCD works on every frame, so each of the eight private,
// mutually non-coherent 4 MiB L2s fetches its own copy of every table a frame's workgroups share (hash cells, kd
// nodes, chunk tables, label lines that sixteen scattered stores fill) and writes back its own partial lines.
// lpx_block() re-reads the linear workgroup number so that the workgroups an XCD receives belong to the frames
// z = c, c + 8, c + 16, ... of ITS residue c: a frame then lives in one L2.  A bijection on the grid for the first
// gridDim.z & ~7 frames, the identity for the rest (and for single-frame launches); placement is a matter of
// speed only, nothing depends on which XCD a workgroup really gets.
// Kernel families (template argument of lpx_block): 0 streaming kernels of the segmentation, 1 sorts and scans, 2 seeds
// and plane passes, 3 kd build, 4 neighbour tables, 5 cell linking of the component grid, 6 the grid's per-point /
// per-cell kernels, component ranges / labels / groups, 7 replay.
// Which families re-read their workgroup number is a property of the LAUNCH (every workgroup of a launch must agree):
// the host passes it in the low byte of the frame-arena stride that every kernel receives (the stride is a multiple of
// 256): bit g = family g.  The host's choice: lpx_remap_mask() in lpx_api.hip.
#define LPX_FS_TAG_MASK ((size_t)255)
struct LpxBlock
{
    uint32_t x, y, z;
};
template <int FAMILY>
__device__ __forceinline__ LpxBlock lpx_block(size_t tagged_stride)
{
    LpxBlock b = {blockIdx.x, blockIdx.y, blockIdx.z};
    if ((((uint32_t)tagged_stride >> FAMILY) & 1u) && blockIdx.z < (gridDim.z & ~7u))
    {
        // the 8 frames z = 8 g + s (s = 0..7) of a group occupy 8 nxy consecutive linear workgroup numbers that start at
        // a multiple of 8: number l inside the group lands on XCD l & 7 and becomes tile l >> 3 of frame 8 g + (l & 7)
        const uint32_t s = blockIdx.z & 7u;
        if (gridDim.y == 1u)
        {
            const uint32_t l = blockIdx.x + gridDim.x * s;
            b.x = l >> 3;
            b.z = (blockIdx.z & ~7u) + (l & 7u);
        }
        else
        {
            const uint32_t l = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * s), r = l >> 3;
            b.y = r / gridDim.x;
            b.x = r - b.y * gridDim.x;
            b.z = (blockIdx.z & ~7u) + (l & 7u);
        }
    }
    return b;
}

// Every kernel starts with `const LpxBlock lpx_blk = lpx_block();` (one evaluation per workgroup); lpx_slot /
// lpx_user and the tile indices read that.
// slot `lpx_blk.z` of an arena buffer (null stays null)
template <class T>
__device__ __forceinline__ T *lpx_slot_z(T *p, size_t stride_bytes, uint32_t z)
{
    return p ? (T *)((char *)p + (size_t)z * (stride_bytes & ~LPX_FS_TAG_MASK)) : p;
}
// frame `lpx_blk.z` of a caller array pitched by `pitch` elements
template <class T>
__device__ __forceinline__ T *lpx_user_z(T *p, uint32_t pitch, uint32_t z)
{
    return p ? p + (size_t)z * pitch : p;
}
#define lpx_slot(p, stride_bytes) lpx_slot_z((p), (stride_bytes), lpx_blk.z)
#define lpx_user(p, pitch) lpx_user_z((p), (pitch), lpx_blk.z)

// order-preserving key of a float under operator< with -0 == +0 (ties are broken by index later)
__device__ __forceinline__ uint32_t lpx_float_key(float f)
{
    f = f + 0.0f;  // -0 -> +0
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float lpx_key_float(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

__device__ __forceinline__ unsigned long long lpx_lanemask_lt()
{
    const unsigned lane = __lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

__device__ __forceinline__ long long lpx_wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

__device__ __forceinline__ uint32_t lpx_wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o, 64);
    return v;
}

// inclusive scan inside a wave with DPP row shifts / broadcasts (VALU only, no LDS crossbar): the
// wave64 sequence of GFX9 -- row_shr 1,2,4,8 inside each row of 16, then row_bcast:15 into rows 1 and 3,
// then row_bcast:31 into rows 2 and 3.
__device__ __forceinline__ uint32_t lpx_wave_incl_scan_u32(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return (uint32_t)x;
}

// wave-wide min / max with the same DPP sequence: the result is valid in lane 63.  Lanes without a DPP
// source keep their own value (old == src), which is neutral for min and max.
#define LPX_DPP_REDUCE(T, NAME, OP, TOI, FROMI)                                                        \
    __device__ __forceinline__ T NAME(T v)                                                             \
    {                                                                                                  \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x111, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x112, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x114, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x118, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x142, 0xa, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x143, 0xc, 0xf, false)));         \
        return v;                                                                                      \
    }
__device__ __forceinline__ uint32_t lpx_umin(uint32_t a, uint32_t b)
{
    return a < b ? a : b;
}
// the same inside every row of 16 lanes: the result of row r is valid in lane 16 * r + 15
#define LPX_DPP_ROW_REDUCE(T, NAME, OP, TOI, FROMI)                                                    \
    __device__ __forceinline__ T NAME(T v)                                                             \
    {                                                                                                  \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x111, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x112, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x114, 0xf, 0xf, false)));         \
        v = OP(v, FROMI(__builtin_amdgcn_update_dpp(TOI(v), TOI(v), 0x118, 0xf, 0xf, false)));         \
        return v;                                                                                      \
    }
LPX_DPP_ROW_REDUCE(float, lpx_row_min15_f32, fminf, __float_as_int, __int_as_float)
LPX_DPP_ROW_REDUCE(float, lpx_row_max15_f32, fmaxf, __float_as_int, __int_as_float)
#undef LPX_DPP_ROW_REDUCE
LPX_DPP_REDUCE(float, lpx_wave_min63_f32, fminf, __float_as_int, __int_as_float)
LPX_DPP_REDUCE(float, lpx_wave_max63_f32, fmaxf, __float_as_int, __int_as_float)
LPX_DPP_REDUCE(uint32_t, lpx_wave_min63_u32, lpx_umin, (int), (uint32_t))
#undef LPX_DPP_REDUCE

#endif  // __HIPCC__
