// lpx_api.hip -- C-ABI (include/lpx.h), context / workspace management, host staging, profiling.
#include "lpx_internal.h"

#include <atomic>
#include <mutex>
#include <time.h>
#include "lpx_debug.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>

int lpx_kd_layout_copy(lpx_ctx *ctx, uint32_t m, uint32_t *d_out);

// ------------------------------------------------------------------------------------------------
// errors, buffers
// ------------------------------------------------------------------------------------------------
int lpx_fail(lpx_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx)
    {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

const char *lpx_last_error(const lpx_ctx *ctx)
{
    return ctx ? ctx->err : "no context";
}

int lpx_ensure(lpx_ctx *ctx, Buf &b, size_t bytes)
{
    if (b.bytes >= bytes && b.p)
        return LPX_OK;
    if (b.p)
    {
        LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        LPX_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    LPX_HIP(ctx, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return LPX_OK;
}

static int sync_all(lpx_ctx *ctx);  // host wait for everything a context enqueued (overlapped tails included)

static inline size_t align256(size_t v)
{
    return (v + 255) & ~(size_t)255;
}

// Which kernel families (lpx_internal.h: lpx_block) keep a frame's workgroups on one XCD.  Measured on MI355X:
//   * 1M-point frames, 8 chains of 8 in flight: every family re-read +11 % throughput (a frame's hash cells, kd nodes
//     and chunk tables are fetched into one L2 instead of eight; the replay alone 13.2 -> 10.0 ms per chain);
//   * 120k-point frames, 20 chains of 64 in flight (rocprofv3 FETCH_SIZE / WRITE_SIZE per frame, stream throughput):
//     none 173 MB; every family 111 MB (the far linking pass of the component grid 22 -> 2.2 MB, the replay 43 -> 22,
//     the chunk tables 13 -> 6, the x gather 7.4 -> 3.5, the sorts -36 %) but -3.5 % throughput -- and all of that
//     loss belongs to family 6, the component grid's per-point and per-cell kernels (insert with its atomics, clear,
//     alloc, scatter, flatten) and the small label kernels, which save nothing: every OTHER family on (0xbf) gives
//     115 MB per frame at the throughput of none (2055-2063 against 2032-2062 Mpts/s).
// Hence family 6 only for large frames.  LPX_REMAP=<hex mask> overrides (development build only: LPX_KNOB).
static uint32_t lpx_remap_mask(uint32_t points_per_slot)
{
    static const char *env = LPX_KNOB("LPX_REMAP");
    if (env)
        return (uint32_t)strtoul(env, nullptr, 16) & 0xffu;
    return points_per_slot >= 400000u ? 0xffu : 0xbfu;
}

// One arena per frame slot: every internal buffer is a fixed sub-range of it, so slot b of any buffer is
// b * fstride bytes behind slot 0 (what the kernels add for their frame index, lpx_block().z).  Growing reallocates the arena
// and drops its contents, which only happens before a call enqueues work.  The neighbour lists have their
// own arena so that the capacity retry of the host entry points keeps the frame data.
int lpx_ensure_capacity(lpx_ctx *ctx, uint32_t n, uint64_t nb)
{
    // 2^30 points per frame is the limit of the 32-bit table sizes below (the cell table has 2^k >= 2 n slots); a
    // larger -- or garbage -- count is an argument error, not a hang
    if (n >= (1u << 30))
        return lpx_fail(ctx, LPX_ERR_ARG, "%u points in a frame: the limit is 2^30 - 1", n);
    // The tables of the expansion-driven search (chunk tables 256 B per point, the component grid's cell tables up to
    // ~144 B per point, the point -> group / cell maps) are part of a slot only for contexts that search: a context in
    // LPX_NEIGHBOURS_LISTS mode -- every single-frame context by default -- never touches them (44 MB less per 123k-point
    // slot, 1.8 GB per 5M-point slot).  Switching a context to the search mode rebuilds the arena (before any work is
    // enqueued, like every growth).
    const bool want_search = !ctx->use_lists;
    if (n > ctx->cap_n || !ctx->arena || (want_search && !ctx->arena_has_search))
    {
        if (n < ctx->cap_n)
            n = ctx->cap_n;
        const bool with_search = want_search || ctx->arena_has_search;
        const size_t n4 = sizeof(uint32_t) * ((size_t)n + 16);
        const size_t sort_blocks = ((size_t)n + LPX_SORT_TILE - 1) / LPX_SORT_TILE + 1;
        size_t hist_bytes = 64 + 256 * sizeof(uint32_t) * sort_blocks + 64;
        if (hist_bytes < (1u << 16))
            hist_bytes = 1u << 16;
        const size_t blk_bytes = sizeof(uint32_t) * (2 * LPX_SEG_MAX_BLOCKS(n) + 2);
        // expansion-driven search: at most n / 16 + 2 kd groups (2^(D+1) with n >> D <= 64), a cell table of the
        // next power of two >= 2 n slots
        const size_t chunk_bytes = sizeof(ChunkRec) * LPX_GROUP_CHUNKS * ((size_t)n / 8 + 64);  // groups of >= 32 nodes: fewer than n / 8
        uint32_t cell_cap = 64;
        while ((size_t)cell_cap < 2 * (size_t)n)
            cell_cap <<= 1;
        struct Item
        {
            Buf *b;
            size_t bytes;
        };
        const Item items[] = {
            {&ctx->frame, sizeof(FrameState)},
            {&ctx->seg_state, 2 * sizeof(SegState) * LPX_MAX_PARTITIONS},
            {&ctx->seg_part, 2 * sizeof(long long) * LPX_ACC_WORDS * LPX_SEG_MAX_BLOCKS(n)},
            {&ctx->seg_far, 3 * sizeof(long long) * LPX_FAR_WORDS * LPX_MAX_PARTITIONS},
            {&ctx->d_planes, sizeof(float) * 4 * LPX_MAX_PARTITIONS},
            {&ctx->d_counts, 64},
            {&ctx->hist, hist_bytes},
            {&ctx->blk_counts, blk_bytes},
            {&ctx->pts4, 4 * n4}, {&ctx->XS, n4},      {&ctx->YS, n4},
            {&ctx->ZS, n4},       {&ctx->OX, n4},       {&ctx->OY, n4},      {&ctx->OZ, n4},      {&ctx->key_a, n4},
            {&ctx->key_b, n4},    {&ctx->val_a, n4},    {&ctx->val_b, n4},   {&ctx->lpos, n4},    {&ctx->rpos, n4},
            {&ctx->nb_len, n4},   {&ctx->nb_off, n4},   {&ctx->parent, n4},  {&ctx->cc_lo, n4},   {&ctx->cc_hi, n4},
            {&ctx->seed_of, n4},  {&ctx->queue, n4},    {&ctx->valid, n4},   {&ctx->d_labels, n4}, {&ctx->d_gidx, n4},
            {&ctx->d_oidx, n4},   {&ctx->d_clabels, n4}, {&ctx->key64_a, 2 * n4}, {&ctx->key64_b, 2 * n4},
            {&ctx->nodes, 4 * n4}, {&ctx->nodes_pre, 4 * n4}, {&ctx->flags, (size_t)n + 64}, {&ctx->state, (size_t)n + 64},
            {&ctx->kd_state, 48 * 1024},  // 48-byte states of up to 1024 ranges (ten top levels)
            // expansion-driven search only (with_search)
            {&ctx->grp_of, with_search ? 4 * n4 : 0},   {&ctx->cell_of, with_search ? n4 : 0},
            {&ctx->chunks, with_search ? chunk_bytes : 0},
            {&ctx->cell_key, with_search ? sizeof(uint64_t) * cell_cap + LPX_CELL_BITS_BYTES : 0},  // + the occupancy bitmap
            {&ctx->cell_rep, with_search ? sizeof(uint32_t) * cell_cap : 0},
            {&ctx->cell_parent, with_search ? sizeof(uint32_t) * cell_cap : 0},
            {&ctx->cell_xyz, with_search ? 3 * sizeof(float4) * cell_cap : 0},  // representative, box low, box high
            {&ctx->cell_start, with_search ? sizeof(uint32_t) * cell_cap : 0},
            {&ctx->cell_pts, with_search ? 4 * n4 : 0},
            {&ctx->cell_list, with_search ? n4 : 0},
        };
        size_t total = 0;
        for (const Item &it : items)
            total += align256(it.bytes);
        if (ctx->arena)
        {
            LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            LPX_HIP(ctx, hipFree(ctx->arena));
            ctx->arena = nullptr;
            ctx->cap_n = 0;
        }
        // whatever an earlier call left in the slots is gone with the old arena: every caller that grows the workspace
        // (begin_call paths, lpx_reserve, lpx_reserve_single_pass, the switch to the search tables) passes through here
        ctx->seg_valid = false;
        ctx->clu_valid = false;
        ctx->last_n = 0;
        LPX_HIP(ctx, hipMalloc(&ctx->arena, total * ctx->batch));
        // zero once: frame states and scratch heads
        LPX_HIP(ctx, hipMemsetAsync(ctx->arena, 0, total * ctx->batch, ctx->stream));
        size_t off = 0;
        for (const Item &it : items)
        {
            it.b->p = (char *)ctx->arena + off;
            it.b->bytes = it.bytes;
            off += align256(it.bytes);
        }
        ctx->fstride = total;
        ctx->cap_n = n;
        ctx->cell_cap = cell_cap;
        ctx->arena_has_search = with_search;
        ctx->fs_tag = total | lpx_remap_mask(n);
    }
    if (!ctx->use_lists && !ctx->nb_arena)
        return LPX_OK;  // the neighbour-list workspace below belongs to the list path only (allocated on first use)
    // Neighbour workspace of a slot: cap_nb words for lists of exact length (what lpx_reserve promises), then
    // cap_rs words the neighbour kernel may use for single-pass lists reserved by an upper bound; a group that
    // finds no room there falls back to counting, so the second region only ever buys speed.
    uint64_t rs = (uint64_t)ctx->cap_n * ctx->rs_per_point;
    if (nb < ctx->cap_nb)
        nb = ctx->cap_nb;
    if (nb + rs > 0xfffffff0ull)
        rs = 0xfffffff0ull - nb;  // offsets are 32-bit
    if (nb != ctx->cap_nb || rs != ctx->cap_rs || !ctx->nb_arena)
    {
        const size_t one = align256(sizeof(uint32_t) * (nb + rs + 64));  // one word per neighbour
        if (ctx->nb_arena)
        {
            LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            LPX_HIP(ctx, hipFree(ctx->nb_arena));
            ctx->nb_arena = nullptr;
            ctx->cap_nb = 0;
            ctx->cap_rs = 0;
        }
        LPX_HIP(ctx, hipMalloc(&ctx->nb_arena, one * ctx->batch));
        ctx->nb_idx.p = ctx->nb_arena;
        ctx->nb_idx.bytes = one;
        ctx->nb_fstride = one;
        ctx->cap_nb = nb;
        ctx->cap_rs = rs;
    }
    return LPX_OK;
}

// every entry point states how many frame slots its launches cover and the pitch of the caller arrays
static int begin_call(lpx_ctx *ctx, uint32_t frames, uint32_t upitch)
{
    if (frames == 0 || frames > ctx->batch)
        return lpx_fail(ctx, LPX_ERR_ARG, "%u frames in a call, the context has %u frame slots", frames, ctx->batch);
    if (ctx->tail_pending)
    {
        // the tail of an earlier overlapped call still reads (and writes) this slot set
        if (hipStreamWaitEvent(ctx->stream, ctx->ev_tail, 0) != hipSuccess)
            return lpx_fail(ctx, LPX_ERR_HIP, "hipStreamWaitEvent on the overlapped tail failed");
        ctx->tail_pending = false;
    }
    ctx->cur_b = frames;
    ctx->upitch = upitch;
    ctx->seg_valid = false;  // set again at the end of a host segmentation call (what lpx_coloured_clouds serves)
    ctx->seg_fresh = false;
    ctx->clu_valid = false;  // set again at the end of a host clustering (what lpx_cluster_groups / _hulls serve)
    if (ctx->la_pending)
    {
        // a look-ahead clustering nobody asked for (lpx_cluster takes its flag down before it gets here): stop guessing
        // until an lpx_cluster call is served from a resident cloud again
        ctx->la_pending = false;
        ctx->la_armed = false;
    }
    ctx->in_off[0] = 0;  // PCL records: x, y, z lead the record; the *_fields entry points overwrite this
    ctx->in_off[1] = 4;
    ctx->in_off[2] = 8;
    return LPX_OK;
}

// PointCloud2 layout of the call being enqueued (after begin_call)
static int set_fields(lpx_ctx *ctx, size_t point_step, uint32_t off_x, uint32_t off_y, uint32_t off_z)
{
    const uint32_t o[3] = {off_x, off_y, off_z};
    for (int a = 0; a < 3; ++a)
    {
        if (point_step < 4 || o[a] > point_step - 4)
            return lpx_fail(ctx, LPX_ERR_ARG, "field offset %u does not fit a point_step of %zu bytes", o[a], point_step);
        ctx->in_off[a] = o[a];
    }
    return LPX_OK;
}

// The list workspace grows on evidence (lpx_internal.h: nb_per_point): the records the list-mode clusterings of this
// context have left in pinned memory since the last look -- only those whose sequence number says they are complete --
// raise the words per point of the two regions to 1.25 x the largest demand.  The reallocation itself happens in
// lpx_ensure_capacity right after (it waits for the stream: rare, and never for a scene the workspace has seen).
static void lists_grow_on_evidence(lpx_ctx *ctx)
{
    if (!ctx->use_lists || !ctx->h_liststat || !ctx->cap_n || ctx->list_seq == ctx->list_seq_seen)
        return;
    uint32_t newest = ctx->list_seq_seen;
    for (uint32_t b = 0; b < ctx->batch; ++b)
    {
        const uint32_t seq = __atomic_load_n(&ctx->h_liststat[b].seq, __ATOMIC_ACQUIRE);
        if ((int32_t)(seq - ctx->list_seq_seen) <= 0 || (int32_t)(seq - ctx->list_seq) > 0)
            continue;  // looked at already, or not a record of this context's calls
        const LpxListStat e = ctx->h_liststat[b];
        if (e.seq != seq)
            continue;  // (being rewritten by a later call: that call's record will be looked at next time)
        newest = (int32_t)(seq - newest) > 0 ? seq : newest;
        if (b == 0 && e.n_obstacle)
            ctx->list_short = e.entries < 48ull * e.n_obstacle ? 1 : 0;  // (CC_FLAT_BELOW of lpx_lists.hip)
        const uint64_t stripe_cap = ctx->cap_rs / LPX_RS_STRIPES;
        // (a context with more than one frame slot keeps its single-pass region as it is: an overflowing stripe costs a
        // counting pass, growing costs a reallocation of tens of GB -- more than a second during which the chain waits)
        if (e.stripe_max > stripe_cap && ctx->cap_rs && ctx->batch == 1)
        {
            const uint64_t want = (e.stripe_max + e.stripe_max / 4) * LPX_RS_STRIPES;
            const uint64_t per = (want + ctx->cap_n - 1) / ctx->cap_n;
            if (per > ctx->rs_per_point)
                ctx->rs_per_point = (uint32_t)(per < 4096 ? per : 4096);
        }
        if (e.nb_total > ctx->cap_nb)
        {
            const uint64_t want = e.nb_total + e.nb_total / 4;
            const uint64_t per = (want + ctx->cap_n - 1) / ctx->cap_n;
            if (per > ctx->nb_per_point)
                ctx->nb_per_point = (uint32_t)(per < 8192 ? per : 8192);
        }
    }
    ctx->list_seq_seen = newest;
    if (ctx->twin)
    {
        ctx->twin->nb_per_point = ctx->nb_per_point > ctx->twin->nb_per_point ? ctx->nb_per_point : ctx->twin->nb_per_point;
        ctx->twin->rs_per_point = ctx->rs_per_point > ctx->twin->rs_per_point ? ctx->rs_per_point : ctx->twin->rs_per_point;
    }
}

static int ensure_for(lpx_ctx *ctx, uint32_t n)
{
    lists_grow_on_evidence(ctx);
    // sized by the points the context is RESERVED for, not by this call's largest frame: a chain whose largest frame has a
    // few points more than the last chain's must not move a 16 GB arena (it did: lpx_reserve comes before the first
    // configuration, and the radius prior of check_clu after it)
    uint64_t nb = (uint64_t)(n > ctx->cap_n ? n : ctx->cap_n) * ctx->nb_per_point;
    if (nb > 0xfffffff0ull)
        nb = 0xfffffff0ull;  // offsets are 32-bit
    const int rc = lpx_ensure_capacity(ctx, n, nb);
    lpx_note_enqueue(ctx);  // a new frame (or chain of frames) is being enqueued on this context
    // LPX_POISON=<byte> (development build only): before every new frame the per-point workspace (everything a call
    // must write before it reads) and the neighbour lists are filled with that byte -- a test that passes with 0, 0xff
    // and 0xa5 does not depend on what an earlier frame left behind
    static const char *poison = LPX_KNOB("LPX_POISON");
    if (rc == LPX_OK && poison)
    {
        const int byte = (int)strtol(poison, nullptr, 0) & 0xff;
        const size_t span = (size_t)((char *)ctx->kd_state.p - (char *)ctx->pts4.p);
        for (uint32_t b = 0; b < ctx->batch; ++b)
            LPX_HIP(ctx, hipMemsetAsync((char *)ctx->pts4.p + b * ctx->fstride, byte, span, ctx->stream));
        if (ctx->nb_arena)
            LPX_HIP(ctx, hipMemsetAsync(ctx->nb_arena, byte, ctx->nb_fstride * ctx->batch, ctx->stream));
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// profiling
// ------------------------------------------------------------------------------------------------
static const char *k_stage_names[ST_COUNT] = {"ingest",   "xsort",   "gather",  "zsort", "seeds",  "plane_passes", "compact",
                                              "kd_build", "cc_hook", "neighbours", "components", "replay", "labels",
                                              "groups"};

StageTimer::StageTimer(lpx_ctx *c, int s) : ctx(c), stage(s)
{
    if (!ctx->profiling)
        return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess)
    {
        a = b = nullptr;
        return;
    }
    hipEventRecord(a, ctx->stream);
}

StageTimer::~StageTimer()
{
    if (!a)
        return;
    hipEventRecord(b, ctx->stream);
    if (ctx->n_pending == ctx->cap_pending)
    {
        const int nc = ctx->cap_pending ? ctx->cap_pending * 2 : 256;
        ctx->pending = (lpx_ctx::Pending *)realloc(ctx->pending, sizeof(lpx_ctx::Pending) * nc);
        ctx->cap_pending = nc;
    }
    ctx->pending[ctx->n_pending++] = {stage, a, b};
}

extern "C" int lpx_profile_enable(lpx_ctx *ctx, int enable)
{
    if (!ctx)
        return LPX_ERR_ARG;
    ctx->profiling = enable != 0;
    return LPX_OK;
}

extern "C" int lpx_profile_stage_count(void)
{
    return ST_COUNT;
}

extern "C" const char *lpx_profile_stage_name(int stage)
{
    return (stage >= 0 && stage < ST_COUNT) ? k_stage_names[stage] : "";
}

extern "C" int lpx_profile_read(lpx_ctx *ctx, float *ms, uint32_t *launches, int reset)
{
    if (!ctx)
        return LPX_ERR_ARG;
    {
        const int rc = sync_all(ctx);
        if (rc)
            return rc;
    }
    lpx_ctx *sets[2] = {ctx, ctx->twin};
    for (lpx_ctx *c : sets)
    {
        if (!c)
            continue;
        for (int i = 0; i < c->n_pending; ++i)
        {
            float t = 0.0f;
            if (hipEventElapsedTime(&t, c->pending[i].a, c->pending[i].b) == hipSuccess)
            {
                ctx->st_ms[c->pending[i].stage] += t;
                ctx->st_launches[c->pending[i].stage] += 1;
            }
            hipEventDestroy(c->pending[i].a);
            hipEventDestroy(c->pending[i].b);
        }
        c->n_pending = 0;
    }
    for (int s = 0; s < ST_COUNT; ++s)
    {
        if (ms)
            ms[s] = ctx->st_ms[s];
        if (launches)
            launches[s] = ctx->st_launches[s];
        if (reset)
        {
            ctx->st_ms[s] = 0.0f;
            ctx->st_launches[s] = 0;
        }
    }
    return LPX_OK;
}

// ------------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------------
// Frames in flight on a device, as far as this process can tell: the frame slots of the contexts that enqueued a
// frame within the last 100 ms (the replay sizes its resident footprint by it, lpx_cluster.hip).  A small registry
// of (device, slots, time of the last enqueue); contexts that sit idle do not count.
namespace
{
constexpr int REG_MAX = 1024;
struct RegEntry
{
    std::atomic<int> device{-1};  // -1: free
    std::atomic<uint32_t> slots{0};
    std::atomic<long long> last_ns{0};
};
RegEntry g_reg[REG_MAX];

long long now_ns()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
}  // namespace

static int reg_claim(int device, uint32_t slots)
{
    for (int i = 0; i < REG_MAX; ++i)
    {
        int expected = -1;
        if (g_reg[i].device.compare_exchange_strong(expected, device))
        {
            g_reg[i].slots = slots;
            g_reg[i].last_ns = 0;
            return i;
        }
    }
    return -1;  // more than REG_MAX live contexts: this one is not counted
}

void lpx_note_enqueue(lpx_ctx *ctx)
{
    if (ctx->reg_index >= 0)
        g_reg[ctx->reg_index].last_ns.store(now_ns(), std::memory_order_relaxed);
}

uint32_t lpx_active_frame_slots(int device)
{
    const long long t = now_ns();
    uint32_t sum = 0;
    for (int i = 0; i < REG_MAX; ++i)
    {
        const int d = g_reg[i].device.load(std::memory_order_relaxed);
        if (d == device && t - g_reg[i].last_ns.load(std::memory_order_relaxed) < 100000000ll)
            sum += g_reg[i].slots.load(std::memory_order_relaxed);
    }
    return sum;
}

static int create_common(int device, hipStream_t stream, bool own, uint32_t batch, lpx_ctx **out)
{
    if (!out)
        return LPX_ERR_ARG;
    *out = nullptr;
    if (batch == 0 || batch > LPX_MAX_BATCH)
        return LPX_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return LPX_ERR_NO_DEVICE;  // no CPU fallback: fail loudly
    if (device < 0 || device >= count)
        return LPX_ERR_ARG;
    lpx_ctx *ctx = new (std::nothrow) lpx_ctx();
    if (!ctx)
        return LPX_ERR_INTERNAL;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess)
    {
        delete ctx;
        return LPX_ERR_HIP;
    }
    if (own)
    {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
        {
            delete ctx;
            return LPX_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    else
        ctx->stream = stream;
    ctx->batch = batch;
    if (hipHostMalloc((void **)&ctx->h_search, 6 * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess)
        memset(ctx->h_search, 0, 6 * sizeof(uint64_t));
    else
        ctx->h_search = nullptr;
    if (hipHostMalloc((void **)&ctx->h_liststat, sizeof(LpxListStat) * batch, hipHostMallocDefault) == hipSuccess)
        memset(ctx->h_liststat, 0, sizeof(LpxListStat) * batch);
    else
        ctx->h_liststat = nullptr;  // (no evidence, no growth: the workspace stays what lpx_reserve made it)
    ctx->reg_index = reg_claim(device, batch);
    ctx->use_lists = batch == 1;  // LPX_NEIGHBOURS_AUTO
    int rc;
    if ((rc = lpx_ensure_capacity(ctx, 1024, 1024)))
    {
        lpx_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return LPX_OK;
}

// HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue run
// their kernels one after the other: the fifth context of a process -- or a feeder's copy stream that lands on a
// context's queue -- waits for whole chains (measured: the feeder moved 5.6 k frames/s on four lanes with the default
// and 10.7 k with 32 queues).  That variable belongs to the PROCESS (HIP reads it once, on its first call): the host
// program's launcher sets it (bench.py, tests/conftest.py and the tools do; INTEGRATION.md for a ROS 2 launch file).
// The library never touches the environment; lpx_build_info() reports what it found.
// The hash of the library's sources (csrc/*.hip, lpx_internal.h, include/*.h), made by the Makefile: what ties a
// committed profile (profiles/*: `library_source_hash`) to the library that produced it -- bench.py marks
// `roofline.traffic` stale when the two differ.
#if __has_include("lpx_source_hash.h")
#include "lpx_source_hash.h"
#endif
#ifndef LPX_SOURCE_HASH
#define LPX_SOURCE_HASH "unknown"
#endif
extern "C" const char *lpx_build_info(void)
{
    static char info[224];
    static std::once_flag once;
    std::call_once(once, [] {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
#ifdef LPX_DEV_KNOBS
        const char *flavour = "development build: LPX_* environment knobs are read";
#else
        const char *flavour = "release build: no LPX_* environment knobs";
#endif
        snprintf(info, sizeof info, "liblpx gfx950, %s; GPU_MAX_HW_QUEUES=%s; src %s", flavour,
                 q ? q : "unset (HIP default 4)", LPX_SOURCE_HASH);
    });
    return info;
}

// device memory the context holds: out[0] = the slot arenas (every per-point buffer of every frame slot), out[1] = the
// neighbour-list arena of the list mode (0 until a call has used it)
extern "C" int lpx_workspace_bytes(const lpx_ctx *ctx, uint64_t *out2)
{
    if (!ctx || !out2)
        return LPX_ERR_ARG;
    out2[0] = ctx->arena ? (uint64_t)ctx->fstride * ctx->batch : 0;
    out2[1] = ctx->nb_arena ? (uint64_t)ctx->nb_fstride * ctx->batch : 0;
    return LPX_OK;
}

extern "C" int lpx_create(int device, lpx_ctx **out)
{
    return create_common(device, nullptr, true, 1, out);
}

extern "C" int lpx_create_batch(int device, uint32_t max_frames, lpx_ctx **out)
{
    return create_common(device, nullptr, true, max_frames, out);
}

extern "C" int lpx_create_on_stream(int device, void *hip_stream, lpx_ctx **out)
{
    return create_common(device, (hipStream_t)hip_stream, false, 1, out);
}

extern "C" void lpx_destroy(lpx_ctx *ctx)
{
    if (!ctx)
        return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->tail_pending)
        hipEventSynchronize(ctx->ev_tail);
    if (ctx->twin)
    {
        lpx_destroy(ctx->twin);
        ctx->twin = nullptr;
    }
    if (ctx->ev_seg)
        hipEventDestroy(ctx->ev_seg);
    if (ctx->copy_stream)
        hipStreamDestroy(ctx->copy_stream);
    if (ctx->h_frame)
        hipHostFree(ctx->h_frame);
    if (ctx->ev_fork)
        hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join)
        hipEventDestroy(ctx->ev_join);
    if (ctx->ev_front)
        hipEventDestroy(ctx->ev_front);
    if (ctx->ev_tail)
        hipEventDestroy(ctx->ev_tail);
    if (ctx->arena)
        hipFree(ctx->arena);
    if (ctx->nb_arena)
        hipFree(ctx->nb_arena);
    if (ctx->in_aos.p)
        hipFree(ctx->in_aos.p);
    if (ctx->rec_out.p)
        hipFree(ctx->rec_out.p);
    if (ctx->dbg_store.p)
        hipFree(ctx->dbg_store.p);
    for (int i = 0; i < ctx->n_pending; ++i)
    {
        hipEventDestroy(ctx->pending[i].a);
        hipEventDestroy(ctx->pending[i].b);
    }
    free(ctx->pending);
    if (ctx->own_stream)
        hipStreamDestroy(ctx->stream);
    if (ctx->h_search)
        hipHostFree(ctx->h_search);
    if (ctx->h_liststat)
        hipHostFree(ctx->h_liststat);
    if (ctx->reg_index >= 0)
        g_reg[ctx->reg_index].device.store(-1);
    delete ctx;
}

extern "C" int lpx_reserve(lpx_ctx *ctx, uint32_t n_points, uint32_t neighbours_per_point)
{
    if (!ctx)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if (neighbours_per_point)
        ctx->nb_per_point = neighbours_per_point;
    int rc = ensure_for(ctx, n_points);
    if (rc == LPX_OK && ctx->twin)  // the second slot set of an overlapped context is sized with the first
    {
        ctx->twin->use_lists = ctx->use_lists;
        ctx->twin->nb_per_point = ctx->nb_per_point;
        ctx->twin->rs_per_point = ctx->rs_per_point;
        if ((rc = ensure_for(ctx->twin, n_points)))
            lpx_fail(ctx, rc, "%s", ctx->twin->err);
    }
    return rc;
}

extern "C" int lpx_reserve_single_pass(lpx_ctx *ctx, uint32_t words_per_point)
{
    if (!ctx)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    ctx->rs_per_point = words_per_point;
    int rc = lpx_ensure_capacity(ctx, ctx->cap_n, ctx->cap_nb);
    if (rc == LPX_OK && ctx->twin)
    {
        ctx->twin->rs_per_point = words_per_point;
        if ((rc = lpx_ensure_capacity(ctx->twin, ctx->twin->cap_n, ctx->twin->cap_nb)))
            lpx_fail(ctx, rc, "%s", ctx->twin->err);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Overlapped tail (lpx_set_overlap).  In a closed loop of N chains the replay is the one stage whose duration does not
// shrink with the load it shares the device with: the wide kernels of a chain take N x their share of the device, the
// replay -- a few latency-bound wavefronts per frame -- takes its ~10 ms whatever else runs, and while a context's
// stream sits in it the context contributes nothing to the wide work.  With the tail on a second stream and a second
// set of frame slots, the context's stream goes straight on to the front end of its next chain: the delay leaves the
// cycle.  The tail streams are a small pool per device shared by all contexts (a tail stream is idle most of the
// time; the pool holds 12: 10 overlapped contexts + their 10 tail streams are 20 hardware queues, inside the ~24 the
// device serves at full speed -- a process with more contexts than that should not overlap).
// ------------------------------------------------------------------------------------------------
constexpr int TAIL_POOL_MAX = 64;
static hipStream_t g_tail_pool[16][TAIL_POOL_MAX];
static std::atomic<int> g_tail_made[16];
static std::atomic<uint32_t> g_tail_next[16];
static std::mutex g_tail_mutex;

static hipStream_t tail_stream_for(int device)
{
    static const int pool = [] {
        const char *e = LPX_KNOB("LPX_TAIL_STREAMS");
        const int v = e ? atoi(e) : 12;
        return v < 1 ? 1 : (v > TAIL_POOL_MAX ? TAIL_POOL_MAX : v);
    }();
    if (device < 0 || device >= 16)
        return nullptr;
    std::lock_guard<std::mutex> lock(g_tail_mutex);
    const uint32_t k = g_tail_next[device].fetch_add(1) % (uint32_t)pool;
    while (g_tail_made[device].load() <= (int)k)
    {
        const int i = g_tail_made[device].load();
        if (hipStreamCreateWithFlags(&g_tail_pool[device][i], hipStreamNonBlocking) != hipSuccess)
            return nullptr;
        g_tail_made[device].store(i + 1);
    }
    return g_tail_pool[device][k];
}

// Side streams of the forked front end (lpx_set_fork): a small pool per device, shared by all contexts -- a side
// stream carries eight short launches per chain and is idle most of the time.
static hipStream_t g_fork_pool[16][TAIL_POOL_MAX];
static std::atomic<int> g_fork_made[16];
static std::atomic<uint32_t> g_fork_next[16];

static hipStream_t fork_stream_for(int device)
{
    static const int pool = [] {
        const char *e = LPX_KNOB("LPX_FORK_STREAMS");
        const int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > TAIL_POOL_MAX ? TAIL_POOL_MAX : v);
    }();
    if (device < 0 || device >= 16)
        return nullptr;
    std::lock_guard<std::mutex> lock(g_tail_mutex);
    const uint32_t k = g_fork_next[device].fetch_add(1) % (uint32_t)pool;
    while (g_fork_made[device].load() <= (int)k)
    {
        const int i = g_fork_made[device].load();
        if (hipStreamCreateWithFlags(&g_fork_pool[device][i], hipStreamNonBlocking) != hipSuccess)
            return nullptr;
        g_fork_made[device].store(i + 1);
    }
    return g_fork_pool[device][k];
}

extern "C" int lpx_set_fork(lpx_ctx *ctx, int on)
{
    if (!ctx)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_all(ctx);
    if (rc)
        return rc;
    lpx_ctx *sets[2] = {ctx, ctx->twin};
    for (lpx_ctx *c : sets)
    {
        if (!c)
            continue;
        if (on && !c->fork_stream)
        {
            c->fork_stream = fork_stream_for(c->device);
            if (!c->fork_stream || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
                return lpx_fail(ctx, LPX_ERR_HIP, "the forked front end could not get its stream and events");
        }
        c->fork = on != 0;
    }
    return LPX_OK;
}

extern "C" int lpx_set_lookahead(lpx_ctx *ctx, int on)
{
    if (!ctx)
        return LPX_ERR_ARG;
    ctx->lookahead = on ? 1 : 0;
    if (!on)
        ctx->la_armed = false;
    return LPX_OK;
}

extern "C" int lpx_set_record_copy(lpx_ctx *ctx, int on)
{
    if (!ctx)
        return LPX_ERR_ARG;
    ctx->keep_copy = on != 0;
    if (ctx->twin)
        ctx->twin->keep_copy = ctx->keep_copy;
    return LPX_OK;
}

extern "C" uint64_t lpx_dbg_lookahead_hits(lpx_ctx *ctx)
{
    return ctx ? ctx->la_hits : 0;
}

static int overlap_arm(lpx_ctx *c)
{
    if (c->tail_stream)
        return LPX_OK;
    c->tail_stream = tail_stream_for(c->device);
    if (!c->tail_stream || hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess)
        return lpx_fail(c, LPX_ERR_HIP, "the overlapped tail could not get its stream and events");
    return LPX_OK;
}

// waits (on the host) for everything a context has enqueued, tails of both slot sets included
static int sync_all(lpx_ctx *ctx)
{
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lpx_ctx *sets[2] = {ctx, ctx->twin};
    for (lpx_ctx *c : sets)
        if (c && c->tail_pending)
            LPX_HIP(ctx, hipEventSynchronize(c->ev_tail));  // (stays pending for the STREAM until begin_call waits)
    return LPX_OK;
}

extern "C" int lpx_set_overlap(lpx_ctx *ctx, int on)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (ctx->batch < 2 && on)
        return lpx_fail(ctx, LPX_ERR_ARG, "lpx_set_overlap serves batch contexts (lpx_create_batch)");
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = sync_all(ctx);
    if (rc)
        return rc;
    if (on && !ctx->twin)
    {
        if ((rc = overlap_arm(ctx)))
            return rc;
        lpx_ctx *t = nullptr;
        if ((rc = create_common(ctx->device, ctx->stream, false, ctx->batch, &t)))
            return lpx_fail(ctx, rc, "the second slot set of the overlapped context could not be created");
        // both slot sets hand their tails to ONE stream: consecutive tails of a context may as well queue behind one
        // another, and hardware queues are scarce (the device serves about 24 at full speed)
        t->tail_stream = ctx->tail_stream;
        if (hipEventCreateWithFlags(&t->ev_front, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&t->ev_tail, hipEventDisableTiming) != hipSuccess)
        {
            lpx_destroy(t);
            return lpx_fail(ctx, LPX_ERR_HIP, "the overlapped tail could not get its events");
        }
        // the second slot set starts with the capacity and settings of the first: its first call must not allocate
        // (hipMalloc / hipFree synchronise the whole device and would stall every other context's chains)
        t->use_lists = ctx->use_lists;
        t->nb_per_point = ctx->nb_per_point;
        t->rs_per_point = ctx->rs_per_point;
        if ((rc = ensure_for(t, ctx->cap_n)))
        {
            lpx_fail(ctx, rc, "%s", t->err);
            lpx_destroy(t);
            return rc;
        }
        ctx->twin = t;
    }
    ctx->overlap = on != 0;
    ctx->last = nullptr;
    return LPX_OK;
}

// the slot set that would serve the next overlapped batch call, with the primary's settings; the alternation only
// advances (overlap_commit) once that call has passed its argument checks and begun to enqueue
static lpx_ctx *overlap_pick(lpx_ctx *ctx)
{
    if (!ctx->overlap || !ctx->twin)
        return ctx;
    lpx_ctx *t = (ctx->flip & 1u) ? ctx->twin : ctx;
    if (t != ctx)
    {
        t->use_lists = ctx->use_lists;
        t->nb_per_point = ctx->nb_per_point;
        t->rs_per_point = ctx->rs_per_point;
        t->profiling = ctx->profiling;
        t->dbg_buf = ctx->dbg_buf;
    }
    return t;
}

static void overlap_commit(lpx_ctx *primary, lpx_ctx *used)
{
    if (primary->overlap && primary->twin)
        primary->flip++;
    primary->last = used;  // statistics and coloured clouds of "the last batch call" read this slot set
}

// one batch call through the slot set whose turn it is (the primary itself without lpx_set_overlap)
static int batch_call(lpx_ctx *ctx, uint32_t n_frames, const void *d_pts, size_t stride, const uint32_t *offs,
                      uint32_t frame_pitch, const uint32_t *n_points, const lpx_seg_cfg *seg_cfg,
                      const lpx_clu_cfg *clu_cfg, uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes,
                      int32_t *d_clabels, uint32_t *d_counts)
{
    if (!ctx)
        return LPX_ERR_ARG;
    lpx_ctx *t = overlap_pick(ctx);
    t->split_tail = t->tail_stream != nullptr && ctx->overlap;
    t->enqueued = false;
    const int rc = lpx_batch_impl(t, n_frames, d_pts, stride, offs, frame_pitch, n_points, seg_cfg, clu_cfg, d_labels,
                                  d_gidx, d_oidx, d_planes, d_clabels, d_counts);
    t->split_tail = false;
    if (t->enqueued)  // (a call that failed its checks has touched no slot set: the alternation stays where it was)
        overlap_commit(ctx, t);
    if (rc && t != ctx)
        lpx_fail(ctx, rc, "%s", t->err);
    return rc;
}

extern "C" int lpx_synchronize(lpx_ctx *ctx)
{
    if (!ctx)
        return LPX_ERR_ARG;
    return sync_all(ctx);
}

// Overlapped contexts keep two batch calls in flight: this waits (on the host) until every call but the LAST one is
// complete, i.e. for the tail of the slot set the last call did not use.  Without overlap it is lpx_synchronize.
extern "C" int lpx_wait_previous(lpx_ctx *ctx)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (!ctx->overlap || !ctx->twin || !ctx->last)
        return sync_all(ctx);
    lpx_ctx *other = ctx->last == ctx ? ctx->twin : ctx;
    if (other->tail_pending)
        LPX_HIP(ctx, hipEventSynchronize(other->ev_tail));
    return LPX_OK;
}

// ------------------------------------------------------------------------------------------------
// argument checks
// ------------------------------------------------------------------------------------------------
static int check_seg(lpx_ctx *ctx, const lpx_seg_cfg *c, size_t stride, bool pcl_records = true)
{
    if (!c || (pcl_records && (stride < 12 || (stride & 3))))
        return lpx_fail(ctx, LPX_ERR_ARG, "stride must be a multiple of 4 and at least 12 bytes");
    if (c->number_of_planar_partitions == 0 || c->number_of_planar_partitions > LPX_MAX_PARTITIONS ||
        c->number_of_iterations > LPX_MAX_ITERATIONS)
        return lpx_fail(ctx, LPX_ERR_ARG, "partitions must be 1..%u and iterations 0..%u", LPX_MAX_PARTITIONS,
                        LPX_MAX_ITERATIONS);
    return LPX_OK;
}

static int check_clu(lpx_ctx *ctx, const lpx_clu_cfg *c, size_t stride, bool pcl_records = true)
{
    if (!c || (pcl_records && (stride < 12 || (stride & 3))))
        return lpx_fail(ctx, LPX_ERR_ARG, "stride must be a multiple of 4 and at least 12 bytes");
    if (!(c->distance_squared >= 0.0f) || !(c->distance_squared < 3.0e38f))
        return lpx_fail(ctx, LPX_ERR_ARG, "distance_squared must be finite and >= 0");
    // The list workspace's prior: a point's list grows with the surface its ball cuts out of the scene, i.e. with d^2.  The
    // defaults (64 + 192 words per point) are sized for the reference's d = 0.5 m; a larger radius scales them before the
    // first frame is seen -- up to 4 x, the 256 + 768 words of rounds 1-5, which carried every d = 1 m scene of
    // tools/fuzz.py -- and the evidence of the frames themselves takes over from there (lists_grow_on_evidence).
    // A context with MORE THAN ONE frame slot starts at the 4 x: its chains are device calls, which the library cannot
    // repeat -- a frame that outgrows the workspace comes back with LPX_ERR_CAPACITY in its status word -- and the small
    // start cost the reference's own frames exactly that (chains of 32 of the 154 data/ frames, d = 0.5 m: 17 frames
    // refused over the first ten chains while two contexts grew, tools/r6_lists_chain.py).  With the 4 x none is.
    if (ctx->use_lists)
    {
        const float scale = c->distance_squared / 0.25f;
        const uint32_t k = ctx->batch > 1 ? 4u : (scale <= 1.0f ? 1u : (scale >= 4.0f ? 4u : (uint32_t)ceilf(scale)));
        for (lpx_ctx *t = ctx; t; t = (t == ctx ? ctx->twin : nullptr))
        {
            if (64u * k > t->nb_per_point)
                t->nb_per_point = 64u * k;
            if (192u * k > t->rs_per_point)
                t->rs_per_point = 192u * k;
        }
    }
    return LPX_OK;
}

static int status_to_rc(lpx_ctx *ctx, uint32_t status)
{
    if (status == 0)
        return LPX_OK;
    const int rc = -(int)status;
    if (rc == LPX_ERR_RANGE)
        return lpx_fail(ctx, rc, "a coordinate is NaN or infinite");
    if (rc == LPX_ERR_CAPACITY)
        return lpx_fail(ctx, rc, "neighbour workspace too small");
    return lpx_fail(ctx, rc, "device status %u", status);
}

// ------------------------------------------------------------------------------------------------
// device entry points
// ------------------------------------------------------------------------------------------------
extern "C" int lpx_segment_device(lpx_ctx *ctx, const void *d_pts, size_t stride, uint32_t n, const lpx_seg_cfg *cfg,
                                  uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes,
                                  uint32_t *d_counts)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = check_seg(ctx, cfg, stride);
    if (rc)
        return rc;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_for(ctx, n)) || (rc = begin_call(ctx, 1, 0)))
        return rc;
    ctx->last_n = n;
    if ((rc = lpx_run_segment(ctx, d_pts, stride, &n, cfg, d_labels, d_gidx, d_oidx, d_planes)))
        return rc;
    return d_counts ? lpx_write_counts(ctx, d_counts) : LPX_OK;
}

extern "C" int lpx_cluster_device(lpx_ctx *ctx, const void *d_pts, size_t stride, uint32_t m, const lpx_clu_cfg *cfg,
                                  int32_t *d_labels, uint32_t *d_counts)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = check_clu(ctx, cfg, stride);
    if (rc)
        return rc;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_for(ctx, m)) || (rc = begin_call(ctx, 1, 0)))
        return rc;
    if ((rc = lpx_ingest_obstacles(ctx, d_pts, stride, m)))
        return rc;
    return lpx_run_cluster(ctx, m, cfg, d_labels, d_counts, false);
}

// offs: byte offsets of x, y, z in a record (PointCloud2 fields), or null for PCL records (0, 4, 8)
static int segment_cluster_device_impl(lpx_ctx *ctx, const void *d_pts, size_t stride, const uint32_t *offs, uint32_t n,
                                       const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg, uint32_t *d_labels,
                                       uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes, int32_t *d_clabels,
                                       uint32_t *d_counts)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = check_seg(ctx, seg_cfg, stride, !offs);
    if (rc || (rc = check_clu(ctx, clu_cfg, stride, !offs)))
        return rc;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_for(ctx, n)) || (rc = begin_call(ctx, 1, 0)) ||
        (offs && (rc = set_fields(ctx, stride, offs[0], offs[1], offs[2]))))
        return rc;
    ctx->last_n = n;
    if ((rc = lpx_run_segment(ctx, d_pts, stride, &n, seg_cfg, d_labels, d_gidx, d_oidx, d_planes)))
        return rc;
    if (!d_clabels)
        d_clabels = (int32_t *)ctx->d_clabels.p;
    return lpx_run_cluster(ctx, n, clu_cfg, d_clabels, d_counts, false);  // n bounds the obstacle count
}

extern "C" int lpx_segment_cluster_device(lpx_ctx *ctx, const void *d_pts, size_t stride, uint32_t n,
                                          const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg, uint32_t *d_labels,
                                          uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes, int32_t *d_clabels,
                                          uint32_t *d_counts)
{
    return segment_cluster_device_impl(ctx, d_pts, stride, nullptr, n, seg_cfg, clu_cfg, d_labels, d_gidx, d_oidx,
                                       d_planes, d_clabels, d_counts);
}

// The data[] buffer of a sensor_msgs/PointCloud2 message ingested as it is: point_step and the offsets of its
// float32 x / y / z fields replace the host-side decode of reference src/conversions.cpp:62-85.
extern "C" int lpx_segment_cluster_fields_device(lpx_ctx *ctx, const void *d_data, uint32_t point_step, uint32_t off_x,
                                                 uint32_t off_y, uint32_t off_z, uint32_t n, const lpx_seg_cfg *seg_cfg,
                                                 const lpx_clu_cfg *clu_cfg, uint32_t *d_labels, uint32_t *d_gidx,
                                                 uint32_t *d_oidx, float *d_planes, int32_t *d_clabels,
                                                 uint32_t *d_counts)
{
    const uint32_t offs[3] = {off_x, off_y, off_z};
    return segment_cluster_device_impl(ctx, d_data, point_step, offs, n, seg_cfg, clu_cfg, d_labels, d_gidx, d_oidx,
                                       d_planes, d_clabels, d_counts);
}

// B frames per launch chain: every kernel covers all frames (gridDim.z), so the launch count of the chain
// is paid once per batch and each launch has B times the workgroups of a single frame.
int lpx_batch_impl(lpx_ctx *ctx, uint32_t n_frames, const void *d_pts, size_t stride, const uint32_t *offs,
                   uint32_t frame_pitch, const uint32_t *n_points, const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                   uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes, int32_t *d_clabels,
                   uint32_t *d_counts)
{
    if (!ctx || !n_points)
        return LPX_ERR_ARG;
    int rc = check_seg(ctx, seg_cfg, stride, !offs);
    if (rc || (rc = check_clu(ctx, clu_cfg, stride, !offs)))
        return rc;
    if (!d_labels || !d_gidx || !d_oidx || !d_clabels || !d_counts)
        return lpx_fail(ctx, LPX_ERR_ARG, "the batch entry point needs every output array except planes");
    if ((rc = begin_call(ctx, n_frames, frame_pitch)) || (offs && (rc = set_fields(ctx, stride, offs[0], offs[1], offs[2]))))
        return rc;
    uint32_t n = 0;
    for (uint32_t b = 0; b < n_frames; ++b)
    {
        if (n_points[b] > frame_pitch)
            return lpx_fail(ctx, LPX_ERR_ARG, "frame %u holds %u points, the pitch is %u", b, n_points[b], frame_pitch);
        n = n_points[b] > n ? n_points[b] : n;
    }
    if (n && !d_pts)
        return lpx_fail(ctx, LPX_ERR_ARG, "null points");
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_for(ctx, n)))
        return rc;
    ctx->enqueued = true;  // from here on the slot set holds this call's frames
    ctx->last = ctx;       // (a primary used directly -- the feeder's lanes -- is its own "last" slot set)
    if ((rc = lpx_run_segment(ctx, d_pts, stride, n_points, seg_cfg, d_labels, d_gidx, d_oidx, d_planes)))
        return rc;
    return lpx_run_cluster(ctx, n, clu_cfg, d_clabels, d_counts, false);
}

extern "C" int lpx_segment_cluster_batch_device(lpx_ctx *ctx, uint32_t n_frames, const void *d_pts, size_t stride,
                                                uint32_t frame_pitch, const uint32_t *n_points,
                                                const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                                                uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes,
                                                int32_t *d_clabels, uint32_t *d_counts)
{
    return batch_call(ctx, n_frames, d_pts, stride, nullptr, frame_pitch, n_points, seg_cfg, clu_cfg, d_labels, d_gidx,
                      d_oidx, d_planes, d_clabels, d_counts);
}

// the same for n_frames PointCloud2-style buffers (records of point_step bytes, x / y / z at the given offsets)
extern "C" int lpx_segment_cluster_batch_fields_device(lpx_ctx *ctx, uint32_t n_frames, const void *d_data,
                                                       uint32_t point_step, uint32_t off_x, uint32_t off_y,
                                                       uint32_t off_z, uint32_t frame_pitch, const uint32_t *n_points,
                                                       const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg,
                                                       uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx,
                                                       float *d_planes, int32_t *d_clabels, uint32_t *d_counts)
{
    const uint32_t offs[3] = {off_x, off_y, off_z};
    return batch_call(ctx, n_frames, d_data, point_step, offs, frame_pitch, n_points, seg_cfg, clu_cfg, d_labels, d_gidx,
                      d_oidx, d_planes, d_clabels, d_counts);
}

// ------------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------------
static int upload(lpx_ctx *ctx, const void *pts, size_t stride, uint32_t n)
{
    int rc = lpx_ensure(ctx, ctx->in_aos, stride * (size_t)n + 64);
    if (rc)
        return rc;
    if (n)
        LPX_HIP(ctx, hipMemcpyAsync(ctx->in_aos.p, pts, stride * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    return LPX_OK;
}

// Before the neighbour stage runs again with a larger workspace (frame slot 0): clears status and every
// counter behind it, keeps the point counts.
static int reset_neighbour_state(lpx_ctx *ctx)
{
    char *f = (char *)ctx->frame.p;
    const size_t st = offsetof(FrameState, status), cnt = offsetof(FrameState, nb_total);
    static_assert(offsetof(FrameState, nb_total) > offsetof(FrameState, status) &&
                      offsetof(FrameState, n_in) < offsetof(FrameState, status),
                  "FrameState layout: sizes, then status, then the counters");
    LPX_HIP(ctx, hipMemsetAsync(f + st, 0, sizeof(uint32_t), ctx->stream));
    LPX_HIP(ctx, hipMemsetAsync(f + cnt, 0, sizeof(FrameState) - cnt, ctx->stream));
    return LPX_OK;
}

static int read_frame(lpx_ctx *ctx, FrameState *fs)
{
    LPX_HIP(ctx, hipMemcpyAsync(fs, ctx->frame.p, sizeof(FrameState), hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LPX_OK;
}

// The resources of the look-ahead (lpx_set_lookahead), made on first use
static bool lookahead_ready(lpx_ctx *ctx)
{
    if (ctx->copy_stream && ctx->ev_seg && ctx->h_frame)
        return true;
    if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
        ctx->copy_stream = nullptr;
    if (!ctx->ev_seg && hipEventCreateWithFlags(&ctx->ev_seg, hipEventDisableTiming) != hipSuccess)
        ctx->ev_seg = nullptr;
    if (!ctx->h_frame && hipHostMalloc(&ctx->h_frame, sizeof(FrameState), hipHostMallocDefault) != hipSuccess)
        ctx->h_frame = nullptr;
    return ctx->copy_stream && ctx->ev_seg && ctx->h_frame;  // (not ready: the call goes the plain way)
}

// snapshot: the frame state is already on its way to pinned memory (ctx->ev_seg follows it); otherwise read here
static int download_segment(lpx_ctx *ctx, uint32_t n, uint32_t P, uint32_t *labels, uint32_t *gidx, uint32_t *n_ground,
                            uint32_t *oidx, uint32_t *n_obstacle, float *planes, FrameState *fs,
                            const FrameState *snapshot = nullptr)
{
    if (labels && n)
        LPX_HIP(ctx, hipMemcpyAsync(labels, ctx->d_labels.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (planes)
        LPX_HIP(ctx, hipMemcpyAsync(planes, ctx->d_planes.p, sizeof(float) * 4 * P, hipMemcpyDeviceToHost, ctx->stream));
    int rc = LPX_OK;
    if (snapshot)
    {
        LPX_HIP(ctx, hipEventSynchronize(ctx->ev_seg));
        *fs = *snapshot;
    }
    else
        rc = read_frame(ctx, fs);
    if (rc)
        return rc;
    if ((rc = status_to_rc(ctx, fs->status)) && rc != LPX_ERR_CAPACITY)
        return rc;
    if (gidx && fs->n_ground)
        LPX_HIP(ctx, hipMemcpyAsync(gidx, ctx->d_gidx.p, sizeof(uint32_t) * fs->n_ground, hipMemcpyDeviceToHost,
                                    ctx->stream));
    if (oidx && fs->n_obstacle)
        LPX_HIP(ctx, hipMemcpyAsync(oidx, ctx->d_oidx.p, sizeof(uint32_t) * fs->n_obstacle, hipMemcpyDeviceToHost,
                                    ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_ground)
        *n_ground = fs->n_ground;
    if (n_obstacle)
        *n_obstacle = fs->n_obstacle;
    return LPX_OK;
}

static int segment_impl(lpx_ctx *ctx, const void *pts, size_t stride, const uint32_t *offs, uint32_t n,
                        const lpx_seg_cfg *cfg, uint32_t *labels, uint32_t *gidx, uint32_t *n_ground, uint32_t *oidx,
                        uint32_t *n_obstacle, float *planes)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (n_ground)
        *n_ground = 0;
    if (n_obstacle)
        *n_obstacle = 0;
    int rc = check_seg(ctx, cfg, stride, !offs);
    if (rc)
        return rc;
    if (n && !pts)
        return lpx_fail(ctx, LPX_ERR_ARG, "null points");
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_for(ctx, n)) || (rc = begin_call(ctx, 1, 0)) ||
        (offs && (rc = set_fields(ctx, stride, offs[0], offs[1], offs[2]))) || (rc = upload(ctx, pts, stride, n)))
        return rc;
    ctx->last_n = n;
    if ((rc = lpx_run_segment(ctx, ctx->in_aos.p, stride, &n, cfg, (uint32_t *)ctx->d_labels.p,
                              (uint32_t *)ctx->d_gidx.p, (uint32_t *)ctx->d_oidx.p, (float *)ctx->d_planes.p)))
        return rc;
    FrameState fs;
    bool ahead = ctx->lookahead && ctx->la_armed && n > 0 && lookahead_ready(ctx);
    if (ahead)
    {
        // the frame state as the segmentation leaves it (the clustering writes into the same record), then the
        // clustering right behind; the downloads wait for the segmentation only, on a stream of their own
        LPX_HIP(ctx, hipMemcpyAsync(ctx->h_frame, ctx->frame.p, sizeof(FrameState), hipMemcpyDeviceToHost, ctx->stream));
        LPX_HIP(ctx, hipEventRecord(ctx->ev_seg, ctx->stream));
        ctx->la_failed = lpx_run_cluster(ctx, n, &ctx->la_cfg, (int32_t *)ctx->d_clabels.p, nullptr, false) != LPX_OK;
        // (a clustering nobody asked for must not fail the segmentation: whatever of it was enqueued runs out harmlessly,
        // the resident cloud counts as consumed, and lpx_cluster takes the plain path)
        hipStream_t const main_stream = ctx->stream;
        LPX_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->ev_seg, 0));
        ctx->stream = ctx->copy_stream;
        rc = download_segment(ctx, n, cfg->number_of_planar_partitions, labels, gidx, n_ground, oidx, n_obstacle, planes,
                              &fs, (const FrameState *)ctx->h_frame);
        ctx->stream = main_stream;
        if (rc)
            return rc;
    }
    else if ((rc = download_segment(ctx, n, cfg->number_of_planar_partitions, labels, gidx, n_ground, oidx, n_obstacle,
                                    planes, &fs)))
        return rc;
    ctx->seg_valid = true;
    ctx->seg_ground = fs.n_ground;
    ctx->seg_obstacle = fs.n_obstacle;
    ctx->seg_fresh = !ahead;                     // the obstacle SoA and the kd input are as the compaction wrote them ...
    ctx->la_pending = ahead && !ctx->la_failed;  // ... or the clustering that consumes them is on its way
    if (ahead && ctx->la_failed)
        ctx->la_armed = false;
    ctx->seg_hash = fs.obs_hash;
    ctx->seg_hash2 = fs.obs_hash2;
    return LPX_OK;
}

extern "C" int lpx_segment(lpx_ctx *ctx, const void *pts, size_t stride, uint32_t n, const lpx_seg_cfg *cfg,
                           uint32_t *labels, uint32_t *gidx, uint32_t *n_ground, uint32_t *oidx, uint32_t *n_obstacle,
                           float *planes)
{
    return segment_impl(ctx, pts, stride, nullptr, n, cfg, labels, gidx, n_ground, oidx, n_obstacle, planes);
}

extern "C" int lpx_segment_fields(lpx_ctx *ctx, const void *data, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                                  uint32_t off_z, uint32_t n, const lpx_seg_cfg *cfg, uint32_t *labels, uint32_t *gidx,
                                  uint32_t *n_ground, uint32_t *oidx, uint32_t *n_obstacle, float *planes)
{
    const uint32_t offs[3] = {off_x, off_y, off_z};
    return segment_impl(ctx, data, point_step, offs, n, cfg, labels, gidx, n_ground, oidx, n_obstacle, planes);
}

// Runs the clustering of the obstacle SoA resident in ctx (count on the device, at most m_bound points) and
// reads the frame state back; grows the neighbour workspace and retries when the frame needs more.
static int cluster_resident(lpx_ctx *ctx, uint32_t m_bound, const lpx_clu_cfg *cfg, FrameState *fs,
                            bool first_enqueued = false)
{
    int rc;
    for (int attempt = 0; attempt < 3; ++attempt)
    {
        // the kd-tree of a first attempt stays valid (and must not be rebuilt from the permuted node array:
        // the layout depends on the input order, src/kdtree.hpp:174-225)
        // A retry counts every list (no single-pass reservations): which groups win a reservation depends on
        // scheduling, so only then is nb_total the exact requirement, and the attempt after it always fits.
        if (attempt > 0 || !first_enqueued)  // (first_enqueued: the look-ahead of lpx_segment has enqueued attempt 0)
        {
            ctx->exact_lists_only = attempt > 0;
            rc = lpx_run_cluster(ctx, m_bound, cfg, (int32_t *)ctx->d_clabels.p, nullptr, attempt > 0);
            ctx->exact_lists_only = false;
            if (rc)
                return rc;
        }
        if ((rc = read_frame(ctx, fs)))
            return rc;
        if (fs->status != (uint32_t)(-LPX_ERR_CAPACITY) || attempt == 2)
            break;
        if (fs->nb_total > 0xfffffff0ull)
            return lpx_fail(ctx, LPX_ERR_CAPACITY, "neighbour lists need %llu entries (> 2^32)",
                            (unsigned long long)fs->nb_total);
        uint64_t want = fs->nb_total + fs->nb_total / 8 + 1024;
        if (want > 0xfffffff0ull)
            want = 0xfffffff0ull;
        if ((rc = lpx_ensure_capacity(ctx, ctx->cap_n, want)))
            return rc;
        if ((rc = reset_neighbour_state(ctx)))
            return rc;
    }
    return status_to_rc(ctx, fs->status);
}

static int download_clusters(lpx_ctx *ctx, const FrameState &fs, int32_t *labels, uint32_t *n_clusters)
{
    if (labels && fs.n_obstacle)
        LPX_HIP(ctx, hipMemcpyAsync(labels, ctx->d_clabels.p, sizeof(int32_t) * fs.n_obstacle, hipMemcpyDeviceToHost,
                                    ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_clusters)
        *n_clusters = fs.n_clusters;
    return LPX_OK;
}

// a host clustering is complete: d_clabels holds the labels of m points in n_clusters clusters
static void clustering_done(lpx_ctx *ctx, uint32_t m, uint32_t n_clusters)
{
    ctx->clu_valid = true;
    ctx->clu_m = m;
    ctx->clu_clusters = n_clusters;
    ++ctx->clu_epoch;
}

extern "C" uint64_t lpx_cluster_epoch(const lpx_ctx *ctx)
{
    return ctx && ctx->clu_valid ? ctx->clu_epoch : 0;
}

// lpx_cluster_groups / lpx_cluster_hulls: the caller's m / n_clusters must be those of the clustering whose labels are
// resident (0 = fine)
static int check_resident_labels(lpx_ctx *ctx, const char *who, uint32_t m, uint32_t n_clusters)
{
    if (!ctx->clu_valid)
        return lpx_fail(ctx, LPX_ERR_ARG, "%s: the last call on this context was not a host clustering (lpx_cluster / "
                                          "lpx_segment_cluster*); its labels are gone", who);
    if (m != ctx->clu_m || n_clusters != ctx->clu_clusters)
        return lpx_fail(ctx, LPX_ERR_ARG, "%s: %u points / %u clusters, but the resident labels are those of %u points / "
                                          "%u clusters", who, m, n_clusters, ctx->clu_m, ctx->clu_clusters);
    return LPX_OK;
}

extern "C" int lpx_cluster(lpx_ctx *ctx, const void *pts, size_t stride, uint32_t m, const lpx_clu_cfg *cfg,
                           int32_t *labels, uint32_t *n_clusters)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (n_clusters)
        *n_clusters = 0;
    int rc = check_clu(ctx, cfg, stride);
    if (rc)
        return rc;
    if (m == 0)  // src/clustering.cpp:51-54
    {
        ctx->clu_valid = false;
        return LPX_OK;
    }
    if (!pts)
        return lpx_fail(ctx, LPX_ERR_ARG, "null points");
    if (const char *e = LPX_KNOB("LPX_FAIL_CLUSTER"))  // development build: the first k calls of the process fail
    {                                                   // (what the degrade path of Clusterer::cluster is tested with)
        static int failed = 0;
        if (failed < atoi(e))
        {
            ++failed;
            ctx->clu_valid = false;
            return lpx_fail(ctx, LPX_ERR_CAPACITY, "forced failure %d of lpx_cluster (LPX_FAIL_CLUSTER)", failed);
        }
    }
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    // The unchanged node hands Clusterer::cluster the obstacle cloud Segmenter::segment has just built from this very
    // context's index list (src/processor.cpp:150-178).  That cloud is still resident -- obstacle SoA, kd input, frame
    // state, exactly as the fused lpx_segment_cluster finds them -- so when the caller's records have its size and its
    // position-bound checksum (lpx_obstacle_mix over all m points, ~30 us on the host) the upload and the ingest are
    // skipped.  Anything else (another cloud, a second clustering of the same one: the kd build consumes its input)
    // takes the upload.
    // Look-ahead: when the lpx_segment call has already enqueued this very clustering (same cloud, same configuration --
    // what the call before this one asked for), there is nothing to enqueue, only to wait for.  A look-ahead with
    // another configuration has consumed the kd input like any clustering: upload.
    const bool ahead = ctx->la_pending && memcmp(cfg, &ctx->la_cfg, sizeof(*cfg)) == 0;
    if (ctx->seg_valid && (ctx->seg_fresh || ahead) && m == ctx->seg_obstacle && m <= ctx->cap_n)
    {
        // two independent position-bound sums over ALL m points (lpx_obstacle_mix, lpx_obstacle_mix2): both must agree
        uint64_t h = 0, h2 = 0;
        const char *p = (const char *)pts;
        for (uint32_t i = 0; i < m; ++i, p += stride)
        {
            uint32_t w[3];
            memcpy(w, p, 12);  // x, y, z are the first three floats of a record (lpx.h: lpx_cluster)
            h += lpx_obstacle_mix(i, w[0], w[1], w[2]);
            h2 += lpx_obstacle_mix2(i, w[0], w[1], w[2]);
        }
        if (h == ctx->seg_hash && h2 == ctx->seg_hash2)
        {
            const uint32_t ng = ctx->seg_ground, last_n = ctx->last_n;
            ctx->la_pending = false;  // (used, not mispredicted: begin_call must not take the guess down)
            if ((rc = begin_call(ctx, 1, 0)))
                return rc;
            FrameState fs;
            if ((rc = cluster_resident(ctx, m, cfg, &fs, ahead)))
                return rc;
            if ((rc = download_clusters(ctx, fs, labels, n_clusters)))
                return rc;
            // the segmentation's points and index lists are still in place (as after the fused call)
            ctx->seg_valid = true;
            ctx->seg_ground = ng;
            ctx->seg_obstacle = m;
            ctx->last_n = last_n;
            ctx->la_cfg = *cfg;  // the caller runs segment() then cluster() on one context: look ahead next time
            ctx->la_armed = true;
            ctx->la_hits += ahead ? 1u : 0u;
            clustering_done(ctx, m, fs.n_clusters);
            return LPX_OK;
        }
    }
    if ((rc = ensure_for(ctx, m)) || (rc = begin_call(ctx, 1, 0)) || (rc = upload(ctx, pts, stride, m)))
        return rc;
    if ((rc = lpx_ingest_obstacles(ctx, ctx->in_aos.p, stride, m)))
        return rc;
    FrameState fs;
    if ((rc = cluster_resident(ctx, m, cfg, &fs)) || (rc = download_clusters(ctx, fs, labels, n_clusters)))
        return rc;
    clustering_done(ctx, m, fs.n_clusters);
    return LPX_OK;
}

static int segment_cluster_impl(lpx_ctx *ctx, const void *pts, size_t stride, const uint32_t *offs, uint32_t n,
                                const lpx_seg_cfg *seg_cfg, const lpx_clu_cfg *clu_cfg, uint32_t *labels, uint32_t *gidx,
                                uint32_t *n_ground, uint32_t *oidx, uint32_t *n_obstacle, float *planes,
                                int32_t *cluster_labels, uint32_t *n_clusters)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (n_ground)
        *n_ground = 0;
    if (n_obstacle)
        *n_obstacle = 0;
    if (n_clusters)
        *n_clusters = 0;
    int rc = check_seg(ctx, seg_cfg, stride, !offs);
    if (rc || (rc = check_clu(ctx, clu_cfg, stride, !offs)))
        return rc;
    if (n && !pts)
        return lpx_fail(ctx, LPX_ERR_ARG, "null points");
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_for(ctx, n)) || (rc = begin_call(ctx, 1, 0)) ||
        (offs && (rc = set_fields(ctx, stride, offs[0], offs[1], offs[2]))) || (rc = upload(ctx, pts, stride, n)))
        return rc;
    ctx->last_n = n;
    if ((rc = lpx_run_segment(ctx, ctx->in_aos.p, stride, &n, seg_cfg, (uint32_t *)ctx->d_labels.p,
                              (uint32_t *)ctx->d_gidx.p, (uint32_t *)ctx->d_oidx.p, (float *)ctx->d_planes.p)))
        return rc;
    // The clustering is enqueued right behind the segmentation (n bounds the obstacle count, the kernels take
    // the real one from the device): no synchronisation or copy in the middle of the chain.  Everything is
    // downloaded at the end.
    FrameState fs;
    if ((rc = cluster_resident(ctx, n, clu_cfg, &fs)))
        return rc;
    const uint32_t P = seg_cfg->number_of_planar_partitions;
    if (labels && n)
        LPX_HIP(ctx, hipMemcpyAsync(labels, ctx->d_labels.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (planes)
        LPX_HIP(ctx, hipMemcpyAsync(planes, ctx->d_planes.p, sizeof(float) * 4 * P, hipMemcpyDeviceToHost, ctx->stream));
    if (gidx && fs.n_ground)
        LPX_HIP(ctx, hipMemcpyAsync(gidx, ctx->d_gidx.p, sizeof(uint32_t) * fs.n_ground, hipMemcpyDeviceToHost,
                                    ctx->stream));
    if (oidx && fs.n_obstacle)
        LPX_HIP(ctx, hipMemcpyAsync(oidx, ctx->d_oidx.p, sizeof(uint32_t) * fs.n_obstacle, hipMemcpyDeviceToHost,
                                    ctx->stream));
    if (n_ground)
        *n_ground = fs.n_ground;
    if (n_obstacle)
        *n_obstacle = fs.n_obstacle;
    if ((rc = download_clusters(ctx, fs, cluster_labels, n_clusters)))  // one synchronisation for all copies
        return rc;
    // the clustering of the resident obstacle cloud leaves the segmentation's points and index lists in place
    ctx->seg_valid = true;
    ctx->seg_ground = fs.n_ground;
    ctx->seg_obstacle = fs.n_obstacle;
    clustering_done(ctx, fs.n_obstacle, fs.n_clusters);
    return LPX_OK;
}

extern "C" int lpx_segment_cluster(lpx_ctx *ctx, const void *pts, size_t stride, uint32_t n, const lpx_seg_cfg *seg_cfg,
                                   const lpx_clu_cfg *clu_cfg, uint32_t *labels, uint32_t *gidx, uint32_t *n_ground,
                                   uint32_t *oidx, uint32_t *n_obstacle, float *planes, int32_t *cluster_labels,
                                   uint32_t *n_clusters)
{
    return segment_cluster_impl(ctx, pts, stride, nullptr, n, seg_cfg, clu_cfg, labels, gidx, n_ground, oidx, n_obstacle,
                                planes, cluster_labels, n_clusters);
}

extern "C" int lpx_segment_cluster_fields(lpx_ctx *ctx, const void *data, uint32_t point_step, uint32_t off_x,
                                          uint32_t off_y, uint32_t off_z, uint32_t n, const lpx_seg_cfg *seg_cfg,
                                          const lpx_clu_cfg *clu_cfg, uint32_t *labels, uint32_t *gidx,
                                          uint32_t *n_ground, uint32_t *oidx, uint32_t *n_obstacle, float *planes,
                                          int32_t *cluster_labels, uint32_t *n_clusters)
{
    const uint32_t offs[3] = {off_x, off_y, off_z};
    return segment_cluster_impl(ctx, data, point_step, offs, n, seg_cfg, clu_cfg, labels, gidx, n_ground, oidx,
                                n_obstacle, planes, cluster_labels, n_clusters);
}

// ------------------------------------------------------------------------------------------------
// egress: the PointXYZRGBL clouds of reference src/processor.cpp:152-163 / src/conversions.cpp:164-193
// ------------------------------------------------------------------------------------------------
extern "C" int lpx_coloured_clouds_device(lpx_ctx *ctx, const uint32_t *d_gidx, const uint32_t *d_oidx,
                                          void *d_ground_records, void *d_obstacle_records)
{
    if (!ctx || !d_gidx || !d_oidx || !d_ground_records || !d_obstacle_records)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    ctx->cur_b = 1;
    ctx->upitch = 0;
    return lpx_run_colour(ctx, ctx->last_n, d_gidx, d_oidx, d_ground_records, d_obstacle_records);
}

extern "C" int lpx_coloured_clouds_batch_device(lpx_ctx *ctx, uint32_t n_frames, uint32_t frame_pitch,
                                                const uint32_t *d_gidx, const uint32_t *d_oidx, void *d_ground_records,
                                                void *d_obstacle_records)
{
    if (!ctx || !d_gidx || !d_oidx || !d_ground_records || !d_obstacle_records)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    if (n_frames == 0 || n_frames > ctx->batch)
        return lpx_fail(ctx, LPX_ERR_ARG, "%u frames, the context has %u frame slots", n_frames, ctx->batch);
    if (ctx->last)
        ctx = ctx->last;  // overlapped context: the slot set of the last batch call holds the frames
    ctx->cur_b = n_frames;
    ctx->upitch = frame_pitch;
    return lpx_run_colour(ctx, frame_pitch, d_gidx, d_oidx, d_ground_records, d_obstacle_records);
}

// host form: records of the LAST lpx_segment* / lpx_segment_cluster* HOST call of this context.  Any other call on the
// context in between (lpx_cluster of another cloud, a device or batch entry point) re-initialises the frame state and
// may move the arena: the call then fails with LPX_ERR_ARG instead of copying records of the wrong cloud.  Never
// copies more than the counts that segmentation call returned to the caller.
extern "C" int lpx_coloured_clouds(lpx_ctx *ctx, void *ground_records, void *obstacle_records, uint32_t *n_ground,
                                   uint32_t *n_obstacle)
{
    if (!ctx || !ground_records || !obstacle_records)
        return LPX_ERR_ARG;
    if (n_ground)
        *n_ground = 0;
    if (n_obstacle)
        *n_obstacle = 0;
    if (!ctx->seg_valid)
        return lpx_fail(ctx, LPX_ERR_ARG, "lpx_coloured_clouds: the last call on this context was not a host "
                                         "segmentation (lpx_segment* / lpx_segment_cluster*); its clouds are gone");
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->last_n, ng = ctx->seg_ground, no = ctx->seg_obstacle;
    if ((uint64_t)ng + no > n)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "remembered cloud sizes %u + %u exceed the %u points segmented", ng, no, n);
    if (n == 0)
        return LPX_OK;
    int rc = lpx_ensure(ctx, ctx->rec_out, 64 * (size_t)n + 64);
    if (rc)
        return rc;
    char *grec = (char *)ctx->rec_out.p, *orec = grec + 32 * (size_t)n;
    ctx->cur_b = 1;
    ctx->upitch = 0;
    // While a look-ahead clustering runs on the context's stream the records are made beside it, on the stream that
    // carried the segmentation's downloads: they need the points, the labels and the index lists only, none of which
    // the clustering writes.
    hipStream_t const main_stream = ctx->stream;
    if (ctx->la_pending && ctx->copy_stream)
        ctx->stream = ctx->copy_stream;
    rc = lpx_run_colour(ctx, n, (const uint32_t *)ctx->d_gidx.p, (const uint32_t *)ctx->d_oidx.p, grec, orec);
    hipError_t he = hipSuccess;
    if (!rc && ng)
        he = hipMemcpyAsync(ground_records, grec, 32 * (size_t)ng, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && he == hipSuccess && no)
        he = hipMemcpyAsync(obstacle_records, orec, 32 * (size_t)no, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && he == hipSuccess)
        he = hipStreamSynchronize(ctx->stream);
    ctx->stream = main_stream;
    if (rc)
        return rc;
    LPX_HIP(ctx, he);
    if (n_ground)
        *n_ground = ng;
    if (n_obstacle)
        *n_obstacle = no;
    return LPX_OK;
}

// Cluster regrouping (reference src/processor.cpp:180-200) of the labels of the LAST clustering call on
// this context, which are still resident on the device.
extern "C" int lpx_cluster_groups(lpx_ctx *ctx, uint32_t m, uint32_t n_clusters, uint32_t *offsets, uint32_t *indices,
                                  uint32_t *n_valid)
{
    if (!ctx || !offsets)
        return LPX_ERR_ARG;
    if (n_valid)
        *n_valid = 0;
    offsets[0] = 0;
    if (m == 0)
    {
        for (uint32_t c = 0; c <= n_clusters; ++c)
            offsets[c] = 0;
        return LPX_OK;
    }
    int rc = check_resident_labels(ctx, "lpx_cluster_groups", m, n_clusters);
    if (rc)
        return rc;
    if (n_clusters == 0)
        return LPX_OK;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    // offsets / indices go to kd-build scratch (free once the clustering is done): the index lists of the
    // segmentation stay resident for lpx_coloured_clouds
    uint32_t *d_off = (uint32_t *)ctx->lpos.p, *d_ind = (uint32_t *)ctx->rpos.p;
    const bool seg_valid = ctx->seg_valid;  // regrouping touches neither the frame state nor the segmentation's buffers
    rc = begin_call(ctx, 1, 0);
    if (rc)
        return rc;  // (a failed begin_call leaves the labels invalidated: ADVICE round 5)
    ctx->seg_valid = seg_valid;
    ctx->clu_valid = true;                  // ... nor the labels
    if ((rc = lpx_run_groups(ctx, (const int32_t *)ctx->d_clabels.p, m, d_off, d_ind)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(offsets, d_off, sizeof(uint32_t) * ((size_t)n_clusters + 1),
                                hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t nv = offsets[n_clusters];
    if (nv > m)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "group offsets out of range");
    if (indices && nv)
        LPX_HIP(ctx, hipMemcpy(indices, d_ind, sizeof(uint32_t) * nv, hipMemcpyDeviceToHost));
    if (n_valid)
        *n_valid = nv;
    return LPX_OK;
}

// N3: convex hulls (Andrew monotone chain, counter-clockwise) of the valid clusters with fewer than max_points
// points -- the convex branch of reference src/polygon_simplification.cpp:96-115 (max_points = 20) -- for the
// labels of the LAST clustering call of this context.
extern "C" int lpx_cluster_hulls(lpx_ctx *ctx, uint32_t m, uint32_t n_clusters, uint32_t max_points,
                                 uint32_t *hull_offsets, uint32_t *hull_indices, float *hull_xy,
                                 uint32_t *n_hull_points)
{
    if (!ctx || !hull_offsets)
        return LPX_ERR_ARG;
    if (n_hull_points)
        *n_hull_points = 0;
    hull_offsets[0] = 0;
    if (m == 0)
    {
        for (uint32_t c = 0; c <= n_clusters; ++c)
            hull_offsets[c] = 0;
        return LPX_OK;
    }
    int rc = check_resident_labels(ctx, "lpx_cluster_hulls", m, n_clusters);  // (before n_clusters sizes a write)
    if (rc)
        return rc;
    for (uint32_t c = 0; c <= n_clusters; ++c)
        hull_offsets[c] = 0;
    if (n_clusters == 0)
        return LPX_OK;
    if (m > ctx->cap_n || n_clusters > m)
        return lpx_fail(ctx, LPX_ERR_ARG, "%u points / %u clusters do not match the last clustering call", m, n_clusters);
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *d_off = (uint32_t *)ctx->lpos.p, *d_ind = (uint32_t *)ctx->rpos.p;
    uint32_t *d_hoff = (uint32_t *)ctx->nb_off.p, *d_hidx = (uint32_t *)ctx->nb_len.p;
    float *d_hxy = (float *)ctx->key64_a.p;
    const bool seg_valid = ctx->seg_valid;  // (as lpx_cluster_groups)
    rc = begin_call(ctx, 1, 0);
    if (rc)
        return rc;
    ctx->seg_valid = seg_valid;
    ctx->clu_valid = true;
    if ((rc = lpx_run_groups(ctx, (const int32_t *)ctx->d_clabels.p, m, d_off, d_ind)) ||
        (rc = lpx_run_hulls(ctx, (const int32_t *)ctx->d_clabels.p, m, d_off, d_ind, max_points, d_hoff, d_hidx, d_hxy)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(hull_offsets, d_hoff, sizeof(uint32_t) * ((size_t)n_clusters + 1), hipMemcpyDeviceToHost,
                                ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t total = hull_offsets[n_clusters];
    if (total > m)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "hull offsets out of range");
    if (hull_indices && total)
        LPX_HIP(ctx, hipMemcpyAsync(hull_indices, d_hidx, sizeof(uint32_t) * total, hipMemcpyDeviceToHost, ctx->stream));
    if (hull_xy && total)
        LPX_HIP(ctx, hipMemcpyAsync(hull_xy, d_hxy, sizeof(float) * 2 * total, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_hull_points)
        *n_hull_points = total;
    return LPX_OK;
}

// device form: d_offsets / d_indices are the CSR lpx_cluster_groups_device wrote for d_labels (m points, the cloud
// of the last clustering call of this context); d_hull_offsets needs n_clusters + 1 (at most m + 1) entries,
// d_hull_indices m, d_hull_xy 2 * m floats.
extern "C" int lpx_cluster_hulls_device(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, const uint32_t *d_offsets,
                                        const uint32_t *d_indices, uint32_t max_points, uint32_t *d_hull_offsets,
                                        uint32_t *d_hull_indices, float *d_hull_xy)
{
    if (!ctx || !d_labels || !d_offsets || !d_indices || !d_hull_offsets || !d_hull_indices || !d_hull_xy)
        return LPX_ERR_ARG;
    if (m > ctx->cap_n)
        return lpx_fail(ctx, LPX_ERR_ARG, "%u points do not match the last clustering call", m);
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = begin_call(ctx, 1, 0);
    if (rc)
        return rc;
    return lpx_run_hulls(ctx, d_labels, m, d_offsets, d_indices, max_points, d_hull_offsets, d_hull_indices, d_hull_xy);
}

extern "C" int lpx_cluster_groups_device(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, uint32_t *d_offsets,
                                         uint32_t *d_indices)
{
    if (!ctx || !d_labels || !d_offsets || !d_indices)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_for(ctx, m);
    if (rc || (rc = begin_call(ctx, 1, 0)))
        return rc;
    return lpx_run_groups(ctx, d_labels, m, d_offsets, d_indices);
}

// frame statistics of the last call on this context: {n_ground, n_obstacle, n_clusters, status,
// neighbour entries (lo, hi), components, expansions, entries read by the replay (lo, hi)}
extern "C" int lpx_dbg_search_stats_slot(lpx_ctx *ctx, uint32_t slot, uint32_t *out4)
{
    if (!ctx || !out4 || slot >= ctx->batch)
        return LPX_ERR_ARG;
    {
        const int rc = sync_all(ctx);  // (an overlapped tail writes the statistics)
        if (rc)
            return rc;
        if (ctx->last)
            ctx = ctx->last;  // the slot set of the last batch call
    }
    FrameState fs;
    LPX_HIP(ctx, hipMemcpyAsync(&fs, (const char *)ctx->frame.p + (size_t)slot * ctx->fstride, sizeof fs,
                                hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out4[0] = (uint32_t)fs.cand_total;
    out4[1] = (uint32_t)(fs.cand_total >> 32);
    out4[2] = fs.n_windows;
    out4[3] = fs.n_overflow;
    return LPX_OK;
}

extern "C" int lpx_dbg_frame_stats_slot(lpx_ctx *ctx, uint32_t slot, uint32_t *out12)
{
    if (!ctx || !out12 || slot >= ctx->batch)
        return LPX_ERR_ARG;
    {
        const int rc = sync_all(ctx);  // (an overlapped tail writes the statistics)
        if (rc)
            return rc;
        if (ctx->last)
            ctx = ctx->last;  // the slot set of the last batch call
    }
    FrameState fs;
    LPX_HIP(ctx, hipMemcpyAsync(&fs, (const char *)ctx->frame.p + (size_t)slot * ctx->fstride, sizeof fs,
                                hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out12[0] = fs.n_ground;
    out12[1] = fs.n_obstacle;
    out12[2] = fs.n_clusters;
    out12[3] = fs.status;
    out12[4] = (uint32_t)lpx_entries_written(fs);
    out12[5] = (uint32_t)(lpx_entries_written(fs) >> 32);
    out12[6] = fs.n_roots + fs.n_single;  // sets: the replay's work list plus the single-point sets
    out12[7] = fs.n_expansions;
    out12[8] = (uint32_t)fs.replay_entries;
    out12[9] = (uint32_t)(fs.replay_entries >> 32);
    uint64_t words = fs.nb_total;
    for (uint32_t i = 0; i < LPX_RS_STRIPES; ++i)
        words += fs.rs_stripe[i].v < ctx->cap_rs / LPX_RS_STRIPES ? fs.rs_stripe[i].v : ctx->cap_rs / LPX_RS_STRIPES;
    out12[10] = (uint32_t)words;
    out12[11] = (uint32_t)(words >> 32);
    return LPX_OK;
}

extern "C" int lpx_dbg_frame_stats(lpx_ctx *ctx, uint32_t *out12)
{
    return lpx_dbg_frame_stats_slot(ctx, 0, out12);
}

int lpx_ensure_lists(lpx_ctx *ctx)
{
    const bool was = ctx->use_lists;
    ctx->use_lists = true;
    uint64_t nb = (uint64_t)ctx->cap_n * ctx->nb_per_point;
    if (nb > 0xfffffff0ull)
        nb = 0xfffffff0ull;
    const int rc = lpx_ensure_capacity(ctx, ctx->cap_n, nb > ctx->cap_nb ? nb : ctx->cap_nb);
    ctx->use_lists = was;
    return rc;
}

extern "C" int lpx_set_neighbour_mode(lpx_ctx *ctx, int mode)
{
    if (!ctx || mode < LPX_NEIGHBOURS_AUTO || mode > LPX_NEIGHBOURS_SEARCH)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    ctx->use_lists = mode == LPX_NEIGHBOURS_LISTS || (mode == LPX_NEIGHBOURS_AUTO && ctx->batch == 1);
    return ctx->use_lists ? lpx_ensure_lists(ctx) : LPX_OK;
}

// tools only: per-group statistics of the neighbour kernel ({T, intervals, queries, hits, cycles to
// allocation, cycles total, -, -} per group); pass n_groups = 0 to switch it off again
extern "C" int lpx_dbg_group_stats(lpx_ctx *ctx, uint32_t n_groups, uint32_t *out)
{
    if (!ctx)
        return LPX_ERR_ARG;
    Buf &buf = ctx->dbg_store;
    if (out && ctx->dbg_buf)
    {
        LPX_HIP(ctx, hipMemcpy(out, ctx->dbg_buf, 32 * (size_t)n_groups, hipMemcpyDeviceToHost));
        return LPX_OK;
    }
    if (n_groups == 0)
    {
        ctx->dbg_buf = nullptr;
        return LPX_OK;
    }
    int rc = lpx_ensure(ctx, buf, 32 * (size_t)n_groups);
    if (rc)
        return rc;
    LPX_HIP(ctx, hipMemset(buf.p, 0, 32 * (size_t)n_groups));
    ctx->dbg_buf = buf.p;
    return LPX_OK;
}

// ------------------------------------------------------------------------------------------------
// stage-level entry points for the parity tests
// ------------------------------------------------------------------------------------------------
extern "C" int lpx_dbg_sort_pairs(lpx_ctx *ctx, uint32_t *keys, uint32_t *values, uint32_t n, uint32_t bits)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = ensure_for(ctx, n);
    if (rc || n == 0 || (rc = begin_call(ctx, 1, 0)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(ctx->key_a.p, keys, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    LPX_HIP(ctx, hipMemcpyAsync(ctx->val_a.p, values, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    uint32_t *ko, *vo;
    if ((rc = lpx_sort_pairs(ctx, (uint32_t *)ctx->key_a.p, (uint32_t *)ctx->key_b.p, (uint32_t *)ctx->val_a.p,
                             (uint32_t *)ctx->val_b.p, n, nullptr, bits, &ko, &vo)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(keys, ko, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipMemcpyAsync(values, vo, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LPX_OK;
}

extern "C" int lpx_dbg_sort_keys64(lpx_ctx *ctx, uint64_t *keys, uint32_t n, uint32_t bits)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = ensure_for(ctx, n);
    if (rc || n == 0 || (rc = begin_call(ctx, 1, 0)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(ctx->key64_a.p, keys, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t *ko;
    if ((rc = lpx_sort_keys64(ctx, (uint64_t *)ctx->key64_a.p, (uint64_t *)ctx->key64_b.p, n, nullptr, bits, &ko)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(keys, ko, 8 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LPX_OK;
}

extern "C" int lpx_dbg_scan(lpx_ctx *ctx, uint32_t *data, uint32_t n, uint64_t *total)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = ensure_for(ctx, n);
    if (rc || (rc = begin_call(ctx, 1, 0)))
        return rc;
    if (n)
        LPX_HIP(ctx, hipMemcpyAsync(ctx->key_a.p, data, 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t *d_total = (uint64_t *)ctx->d_counts.p;
    if ((rc = lpx_exclusive_scan(ctx, (uint32_t *)ctx->key_a.p, (uint32_t *)ctx->key_a.p, n, nullptr, d_total)))
        return rc;
    if (n)
        LPX_HIP(ctx, hipMemcpyAsync(data, ctx->key_a.p, 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipMemcpyAsync(total, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LPX_OK;
}

static int upload_xyz_as_obstacles(lpx_ctx *ctx, const float *xyz, uint32_t m)
{
    int rc = ensure_for(ctx, m);
    if (rc || (rc = begin_call(ctx, 1, 0)) || (rc = upload(ctx, xyz, 12, m)))
        return rc;
    return lpx_ingest_obstacles(ctx, ctx->in_aos.p, 12, m);
}

extern "C" int lpx_dbg_kd_layout(lpx_ctx *ctx, const float *xyz, uint32_t m, uint32_t *layout_idx)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (m == 0)
        return LPX_OK;
    int rc = upload_xyz_as_obstacles(ctx, xyz, m);
    if (rc || (rc = lpx_kd_build(ctx, m)) || (rc = lpx_kd_layout_copy(ctx, m, (uint32_t *)ctx->key_a.p)))
        return rc;
    LPX_HIP(ctx, hipMemcpyAsync(layout_idx, ctx->key_a.p, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LPX_OK;
}

extern "C" int lpx_dbg_neighbours(lpx_ctx *ctx, const float *xyz, uint32_t m, float r2, uint64_t *offsets,
                                  uint32_t *idx, float *dist, uint64_t capacity)
{
    if (!ctx)
        return LPX_ERR_ARG;
    offsets[0] = 0;
    if (m == 0)
        return LPX_OK;
    int rc = upload_xyz_as_obstacles(ctx, xyz, m);
    if (rc || (rc = lpx_ensure_lists(ctx)) || (rc = lpx_kd_build(ctx, m)))
        return rc;
    FrameState fs;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        if ((rc = lpx_neighbours(ctx, m, r2, r2, false)) || (rc = read_frame(ctx, &fs)))
            return rc;
        if (fs.status == (uint32_t)(-LPX_ERR_CAPACITY) && attempt == 0 && fs.nb_total <= 0xfffffff0ull)
        {
            if ((rc = lpx_ensure_capacity(ctx, ctx->cap_n, fs.nb_total + 1024)))
                return rc;
            if ((rc = reset_neighbour_state(ctx)))
                return rc;
            continue;
        }
        break;
    }
    if ((rc = status_to_rc(ctx, fs.status)))
        return rc;
    // device lists are grouped by kd bucket and hold index | absorb << 31; hand them back as a CSR ordered by
    // point index with the distance KDTree::radius_search reports (src/kdtree.hpp:145-157, :315)
    uint32_t *off32 = (uint32_t *)malloc(4 * (size_t)m), *len32 = (uint32_t *)malloc(4 * (size_t)m);
    bool reserved = false;  // single-pass lists lie anywhere in the striped region behind the exact one
    for (uint32_t i = 0; i < LPX_RS_STRIPES; ++i)
        reserved = reserved || fs.rs_stripe[i].v != 0;
    const uint64_t span = reserved ? ctx->cap_nb + ctx->cap_rs : fs.nb_total;
    uint32_t *didx = (uint32_t *)malloc(4 * (size_t)span + 4);
    LPX_HIP(ctx, hipMemcpyAsync(off32, ctx->nb_off.p, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipMemcpyAsync(len32, ctx->nb_len.p, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    if (span)
        LPX_HIP(ctx, hipMemcpyAsync(didx, ctx->nb_idx.p, 4 * span, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t run = 0;
    for (uint32_t i = 0; i < m; ++i)
    {
        offsets[i] = run;
        run += len32[i];
    }
    offsets[m] = run;
    rc = LPX_OK;
    if (run != lpx_entries_written(fs) || run > span)
        rc = lpx_fail(ctx, LPX_ERR_INTERNAL, "list lengths sum to %llu, %llu written, %llu words in use",
                      (unsigned long long)run, (unsigned long long)lpx_entries_written(fs), (unsigned long long)span);
    else if (run > capacity)
        rc = lpx_fail(ctx, LPX_ERR_CAPACITY, "caller buffers hold %llu entries, %llu needed",
                      (unsigned long long)capacity, (unsigned long long)run);
    else
        for (uint32_t i = 0; i < m && rc == LPX_OK; ++i)
        {
            const float qx = xyz[3 * (size_t)i], qy = xyz[3 * (size_t)i + 1], qz = xyz[3 * (size_t)i + 2];
            for (uint32_t t = 0; t < len32[i]; ++t)
            {
                const uint32_t w = didx[(size_t)off32[i] + t], k = w & 0x7fffffffu;
                if (k >= m)
                {
                    rc = lpx_fail(ctx, LPX_ERR_INTERNAL, "list of point %u names point %u", i, k);
                    break;
                }
                const float d0 = qx - xyz[3 * (size_t)k], d1 = qy - xyz[3 * (size_t)k + 1];
                const float d2 = qz - xyz[3 * (size_t)k + 2];
                const float d = d0 * d0 + (d1 * d1 + d2 * d2);  // no contraction (Makefile): the device's value
                // with thr_f == r2 every listed neighbour carries the absorb bit
                if (!(w >> 31) || !(d <= r2))
                {
                    rc = lpx_fail(ctx, LPX_ERR_INTERNAL, "list of point %u: entry %u has distance %g, flag %u", i, k,
                                  (double)d, w >> 31);
                    break;
                }
                idx[offsets[i] + t] = k;
                dist[offsets[i] + t] = d;
            }
        }
    free(off32);
    free(len32);
    free(didx);
    return rc;
}

__global__ void dbg_roots_kernel(uint32_t *parent, uint32_t m, uint32_t *root)
{
    const LpxBlock lpx_blk = lpx_block<6>(0);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    uint32_t x = i;
    while (parent[x] != x)
        x = parent[x];
    root[i] = x;
}

extern "C" int lpx_dbg_components(lpx_ctx *ctx, const float *xyz, uint32_t m, float r2, uint32_t *root)
{
    if (!ctx)
        return LPX_ERR_ARG;
    if (m == 0)
        return LPX_OK;
    int rc = upload_xyz_as_obstacles(ctx, xyz, m);
    if (rc || (rc = lpx_ensure_lists(ctx)) || (rc = lpx_kd_build(ctx, m)))
        return rc;
    FrameState fs;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        if ((rc = lpx_neighbours(ctx, m, r2, r2, true)) || (rc = read_frame(ctx, &fs)))
            return rc;
        if (fs.status == (uint32_t)(-LPX_ERR_CAPACITY) && attempt == 0 && fs.nb_total <= 0xfffffff0ull)
        {
            if ((rc = lpx_ensure_capacity(ctx, ctx->cap_n, fs.nb_total + 1024)))
                return rc;
            if ((rc = reset_neighbour_state(ctx)))
                return rc;
            continue;
        }
        break;
    }
    if ((rc = status_to_rc(ctx, fs.status)))
        return rc;
    hipLaunchKernelGGL(dbg_roots_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, (uint32_t *)ctx->parent.p, m,
                       (uint32_t *)ctx->key_a.p);
    LPX_HIP(ctx, hipMemcpyAsync(root, ctx->key_a.p, 4 * (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    LPX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LPX_OK;
}

// ------------------------------------------------------------------------------------------------
// achievable HBM bandwidth: a plain streaming copy, timed with HIP events on the context stream.  Reported next to the
// 8 TB/s nominal peak.  Shape: ONE 16-byte element per lane, one workgroup per 4 KiB, non-temporal load
// and store -- measured on MI355X (tools/probe/copy_bw.hip, 1 GiB): 6.56 TB/s (6.24 with plain accesses), the
// guide's 6.29 TB/s float4 copy.  A grid-stride loop over 2048-16384 resident workgroups, the usual shape elsewhere,
// reaches only 4.4-5.2 TB/s here, and hipMemcpyAsync device-to-device 4.8.
// ------------------------------------------------------------------------------------------------
typedef float lpx_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const lpx_v4f *__restrict__ src, lpx_v4f *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

extern "C" int lpx_dbg_copy_bandwidth(lpx_ctx *ctx, size_t bytes, uint32_t reps, double *gb_per_s)
{
    if (!ctx || !gb_per_s || bytes < 4096 || reps == 0)
        return LPX_ERR_ARG;
    LPX_HIP(ctx, hipSetDevice(ctx->device));
    void *a = nullptr, *b = nullptr;
    LPX_HIP(ctx, hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess)
    {
        hipFree(a);
        return lpx_fail(ctx, LPX_ERR_HIP, "hipMalloc of the copy destination failed");
    }
    hipMemsetAsync(a, 1, bytes, ctx->stream);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n16 = bytes / 16;
    const dim3 grid((unsigned)((n16 + 255) / 256)), blk(256);
    hipLaunchKernelGGL(copy_kernel, grid, blk, 0, ctx->stream, (const lpx_v4f *)a, (lpx_v4f *)b, n16);  // warm-up
    hipEventRecord(e0, ctx->stream);
    for (uint32_t r = 0; r < reps; ++r)
        hipLaunchKernelGGL(copy_kernel, grid, blk, 0, ctx->stream, (const lpx_v4f *)a, (lpx_v4f *)b, n16);
    hipEventRecord(e1, ctx->stream);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(a);
    hipFree(b);
    *gb_per_s = ms > 0.0f ? 2.0 * (double)(n16 * 16) * reps / (ms * 1e-3) / 1e9 : 0.0;  // read + write
    return LPX_OK;
}

int lpx_dbg_plane_run(lpx_ctx *ctx, const void *d_pts, uint32_t n, float *d_out);

// plane of all n points through the device moment + Jacobi path; returns 1 if the fit failed (n < 3)
extern "C" int lpx_dbg_plane(lpx_ctx *ctx, const float *xyz, uint32_t n, float *plane)
{
    if (!ctx)
        return LPX_ERR_ARG;
    int rc = ensure_for(ctx, n);
    if (rc || (rc = begin_call(ctx, 1, 0)) || (rc = upload(ctx, xyz, 12, n)))
        return rc;
    if ((rc = lpx_dbg_plane_run(ctx, ctx->in_aos.p, n, (float *)ctx->d_planes.p)))
        return rc;
    float out[5];
    LPX_HIP(ctx, hipMemcpyAsync(out, ctx->d_planes.p, sizeof out, hipMemcpyDeviceToHost, ctx->stream));
    FrameState fs;
    if ((rc = read_frame(ctx, &fs)))
        return rc;
    if ((rc = status_to_rc(ctx, fs.status)))
        return rc;
    memcpy(plane, out, 16);
    return out[4] != 0.0f ? 1 : 0;
}
