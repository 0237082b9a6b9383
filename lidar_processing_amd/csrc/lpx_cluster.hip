// lpx_cluster.hip -- Fast Euclidean Clustering with the reference's exact, order-dependent semantics.
//
// Replaces Clusterer::cluster (reference src/clustering.cpp:47-125).
//
// The reference is a greedy seeded BFS: seeds in input order (:70), FIFO queue (:80-88), neighbours
// in kd-tree pre-order (:90-92), a touched point within (1-q)*d of the expanded point is absorbed
// (removed, never expanded), otherwise queued (:102-109); groups whose TOUCH COUNT (duplicates
// included, :99-100) is outside [min,max] are relabelled INVALID (:113-119).  The partition depends
// on that order, so it cannot be produced by a union-find alone.  What is parallel:
//   * every radius-neighbour list, already in reference emission order   (lpx_kdtree.hip)
//   * the connected components of the d-graph (union-find while filling the lists).  A BFS never
//     leaves its component and components do not interact, so replaying the greedy loop per
//     component -- seeds ascending inside the component -- gives the reference's partition.
//   * the replay of different components: one wavefront each, 64 neighbours per step.
// Queue duplicates: the reference may queue a point several times; every pop after the first finds
// it removed and does nothing, so the replay queues a point once (state QUEUED) -- same sequence of
// expansions, bounded queue (one slot per member).
// Labels: dense 0..L-1 in seed order (:120-123) = exclusive scan over the valid-seed flags.
#include "lpx_internal.h"

#include <math.h>

namespace
{
enum : uint8_t
{
    PT_FRESH = 0,
    PT_QUEUED = 1,
    PT_REMOVED = 2
};

__global__ void flatten_kernel(uint32_t *parent, const FrameState *__restrict__ frame, uint32_t *__restrict__ root,
                               uint32_t *__restrict__ iota, uint8_t *__restrict__ state,
                               uint32_t *__restrict__ valid, uint32_t *__restrict__ cc_lo,
                               uint32_t *__restrict__ cc_hi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame->n_obstacle)
        return;
    uint32_t x = i;
    for (;;)
    {
        const uint32_t p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x)
            break;
        x = p;
    }
    root[i] = x;
    iota[i] = i;
    state[i] = PT_FRESH;
    valid[i] = 0;
    cc_lo[i] = 0;
    cc_hi[i] = 0;
}

// sorted by root (stable): members of a component are contiguous, ascending original index
__global__ void cc_ranges_kernel(const uint32_t *__restrict__ sroot, FrameState *frame,
                                 uint32_t *__restrict__ cc_lo, uint32_t *__restrict__ cc_hi,
                                 uint32_t *__restrict__ roots)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = frame->n_obstacle;
    if (p >= M)
        return;
    const uint32_t r = sroot[p];
    if (p == 0 || sroot[p - 1] != r)
    {
        cc_lo[r] = p;
        roots[atomicAdd(&frame->n_roots, 1u)] = r;  // work list of the replay
    }
    if (p + 1 == M || sroot[p + 1] != r)
        cc_hi[r] = p + 1;
}

struct ReplayParams
{
    double thr;   // (1-q)^2 * d^2 in double, src/clustering.cpp:66-67
    float thr_f;  // largest float <= thr: for a float d, (double)d <= thr  <=>  d <= thr_f
    uint32_t min_size, max_size;
};

constexpr int RP_WAVES = 4;

// one wavefront per component (root r == smallest member == first seed)
__global__ __launch_bounds__(RP_WAVES *WAVE) void replay_kernel(const FrameState *__restrict__ frame,
                                                                 const uint32_t *__restrict__ cc_lo,
                                                                 const uint32_t *__restrict__ cc_hi,
                                                                 const uint32_t *__restrict__ members,
                                                                 const uint32_t *__restrict__ nb_off,
                                                                 const uint32_t *__restrict__ nb_len,
                                                                 const uint32_t *__restrict__ nb_idx,
                                                                 const float *__restrict__ nb_dist, uint8_t *state,
                                                                 int32_t *seed_of, uint32_t *queue, uint32_t *valid,
                                                                 ReplayParams prm, uint64_t cap)
{
    const uint32_t r = blockIdx.x * RP_WAVES + threadIdx.x / WAVE;
    const uint32_t lane = threadIdx.x % WAVE;
    const uint32_t M = frame->n_obstacle;
    if (r >= M || frame->nb_total > cap)
        return;
    const uint32_t lo = cc_lo[r], hi = cc_hi[r];
    if (hi <= lo)
        return;  // r is not a root
    const unsigned long long lt = lpx_lanemask_lt();
    uint32_t *q = queue + lo;  // one slot per member
    uint32_t cursor = lo;
    for (;;)
    {
        // next seed: first member (ascending index) that is not removed (:70-75)
        uint32_t seed = 0xffffffffu;
        while (cursor < hi)
        {
            const uint32_t p = cursor + lane;
            const uint32_t cand = (p < hi) ? members[p] : 0u;
            const bool ok = (p < hi) && state[cand] != PT_REMOVED;
            const unsigned long long m = __ballot(ok);
            if (m)
            {
                const int f = __ffsll((long long)m) - 1;
                seed = __shfl(cand, f, 64);
                cursor += f + 1;
                break;
            }
            cursor += WAVE;
        }
        if (seed == 0xffffffffu)
            break;
        uint32_t qh = 0, qt = 0;
        unsigned long long touches = 0;  // indices_.size(), duplicates included (:99-100)
        if (lane == 0)
        {
            q[0] = seed;
            state[seed] = PT_QUEUED;
        }
        qt = 1;
        __threadfence_block();
        while (qh < qt)
        {
            // pop up to 64 entries, skip the removed ones (:82-88)
            const uint32_t qi = qh + lane;
            const uint32_t cand = (qi < qt) ? q[qi] : 0u;
            const bool ok = (qi < qt) && state[cand] != PT_REMOVED;
            const unsigned long long m = __ballot(ok);
            if (!m)
            {
                qh = min(qh + WAVE, qt);
                continue;
            }
            const int f = __ffsll((long long)m) - 1;
            const uint32_t j = __shfl(cand, f, 64);
            qh += f + 1;
            // expand j: its neighbours in reference order (:90-110)
            const uint32_t o0 = nb_off[j], cnt = nb_len[j];
            for (uint32_t base = 0; base < cnt; base += WAVE)
            {
                const uint32_t t = base + lane;
                const bool in = t < cnt;
                const uint32_t k = in ? nb_idx[o0 + t] : 0u;
                const float d = in ? nb_dist[o0 + t] : 0.0f;
                const uint8_t sk = in ? state[k] : (uint8_t)PT_REMOVED;
                const bool vis = in && sk != PT_REMOVED;
                touches += __popcll(__ballot(vis));
                const bool absorb = vis && ((double)d <= prm.thr);
                const bool push = vis && !absorb && sk == PT_FRESH;
                const unsigned long long pm = __ballot(push);
                if (vis)
                    seed_of[k] = (int32_t)seed;
                if (absorb)
                    state[k] = PT_REMOVED;
                if (push)
                {
                    q[qt + __popcll(pm & lt)] = k;
                    state[k] = PT_QUEUED;
                }
                qt += __popcll(pm);
                __threadfence_block();  // the next step reads state[] / q[] written by other lanes
            }
        }
        if (lane == 0)
            valid[seed] = (touches >= prm.min_size && touches <= prm.max_size) ? 1u : 0u;  // :113
    }
}

// Same replay with the point states in LDS: 2 bits per point over the whole index range (bit 0 queued,
// bit 1 removed), one wavefront per workgroup.  Removes the global round trip from the dependent chain
// of every step; list chunks are loaded four at a time; offsets/lengths are fetched with the queue
// window.  Used when the bitmap fits (M <= 393 216 points), which covers every real frame.
__global__ __launch_bounds__(WAVE) void replay_lds_kernel(const FrameState *__restrict__ frame,
                                                           const uint32_t *__restrict__ cc_lo,
                                                           const uint32_t *__restrict__ cc_hi,
                                                           const uint32_t *__restrict__ members,
                                                           const uint32_t *__restrict__ nb_off,
                                                           const uint32_t *__restrict__ nb_len,
                                                           const uint32_t *__restrict__ nb_idx,
                                                           const float *__restrict__ nb_dist, int32_t *seed_of,
                                                           uint32_t *queue, uint32_t *valid, ReplayParams prm,
                                                           uint64_t cap, FrameState *fstate,
                                                           const uint32_t *__restrict__ roots)
{
    extern __shared__ uint32_t sbits[];
    const uint32_t lane = threadIdx.x;
    const uint32_t M = frame->n_obstacle;
    if (frame->nb_total > cap)
        return;
    const uint32_t n_roots = frame->n_roots;
    if (blockIdx.x >= n_roots)
        return;
    // the bitmap is zeroed once: components own disjoint points, so the 2-bit states of one component
    // are never read by another
    const uint32_t words = (M + 15) / 16;
    for (uint32_t i = lane; i < words; i += WAVE)
        sbits[i] = 0;
    __builtin_amdgcn_wave_barrier();
    unsigned long long st_entries = 0;
    uint32_t st_exp = 0;
  for (;;)
  {
    uint32_t ticket = 0;
    if (lane == 0)
        ticket = atomicAdd(&fstate->root_cursor, 1u);
    ticket = __shfl(ticket, 0, 64);
    if (ticket >= n_roots)
        break;
    const uint32_t r = roots[ticket];
    const uint32_t lo = cc_lo[r], hi = cc_hi[r];
#define ST_GET(k) ((sbits[(k) >> 4] >> (((k) & 15u) * 2u)) & 3u)
#define ST_OR(k, v) atomicOr(&sbits[(k) >> 4], (uint32_t)(v) << (((k) & 15u) * 2u))
    const unsigned long long lt = lpx_lanemask_lt();
    uint32_t *q = queue + lo;
    uint32_t cursor = lo;
    for (;;)
    {
        uint32_t seed = 0xffffffffu;
        while (cursor < hi)
        {
            const uint32_t p = cursor + lane;
            const uint32_t cand = (p < hi) ? members[p] : 0u;
            const bool ok = (p < hi) && !(ST_GET(cand) & 2u);
            const unsigned long long m = __ballot(ok);
            if (m)
            {
                const int f = __ffsll((long long)m) - 1;
                seed = __shfl(cand, f, 64);
                cursor += f + 1;
                break;
            }
            cursor += WAVE;
        }
        if (seed == 0xffffffffu)
            break;
        uint32_t qh = 0, qt = 1;
        unsigned long long touches = 0;
        if (lane == 0)
        {
            q[0] = seed;
            ST_OR(seed, 1u);
            seed_of[seed] = (int32_t)seed;  // queued before it is ever touched
        }
        uint32_t wb = 0, wn = 0;  // queue window [wb, wb + wn) held in registers
        uint32_t wcand = 0, woff = 0, wlen = 0;
        // first four list chunks of the NEXT unremoved window candidate, loaded while the current one is
        // processed (it is the next expansion unless the current one absorbs it)
        bool pf_valid = false;
        uint32_t pf_q = 0;
        uint32_t pk[4] = {0, 0, 0, 0};
        float pd[4] = {0, 0, 0, 0};
        while (qh < qt)
        {
            if (qh >= wb + wn)
            {
                __threadfence_block();  // queue entries pushed by other lanes
                wb = qh;
                wn = min((uint32_t)WAVE, qt - qh);
                const bool in = lane < wn;
                wcand = in ? q[wb + lane] : 0u;
                woff = in ? nb_off[wcand] : 0u;
                wlen = in ? nb_len[wcand] : 0u;
                pf_valid = false;
            }
            const bool ok = (lane < wn) && (wb + lane >= qh) && !(ST_GET(wcand) & 2u);
            const unsigned long long m = __ballot(ok);
            if (!m)
            {
                qh = wb + wn;
                continue;
            }
            const int f = __ffsll((long long)m) - 1;
            const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)woff, f);
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)wlen, f);
            qh = wb + f + 1;
            st_entries += cnt;
            ++st_exp;
            uint32_t kk[4];
            float dd[4];
            if (pf_valid && pf_q == wb + (uint32_t)f)
            {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    kk[c] = pk[c];
                    dd[c] = pd[c];
                }
            }
            else
            {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    const uint32_t t = c * WAVE + lane;
                    const bool in = t < cnt;
                    kk[c] = in ? nb_idx[o0 + t] : 0xffffffffu;
                    dd[c] = in ? nb_dist[o0 + t] : 0.0f;
                }
            }
            {
                const unsigned long long m2 = (f == 63) ? 0ull : (m & ~((2ull << f) - 1ull));
                pf_valid = m2 != 0;
                if (pf_valid)
                {
                    const int f2 = __ffsll((long long)m2) - 1;
                    pf_q = wb + (uint32_t)f2;
                    const uint32_t o2 = (uint32_t)__builtin_amdgcn_readlane((int)woff, f2);
                    const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)wlen, f2);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                    {
                        const uint32_t t = c * WAVE + lane;
                        const bool in = t < c2;
                        pk[c] = in ? nb_idx[o2 + t] : 0xffffffffu;
                        pd[c] = in ? nb_dist[o2 + t] : 0.0f;
                    }
                }
            }
            for (uint32_t base = 0; base < cnt; base += 4 * WAVE)
            {
                if (base)
                {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                    {
                        const uint32_t t = base + c * WAVE + lane;
                        const bool in = t < cnt;
                        kk[c] = in ? nb_idx[o0 + t] : 0xffffffffu;
                        dd[c] = in ? nb_dist[o0 + t] : 0.0f;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    if (base + c * WAVE >= cnt)
                        break;
                    const bool in = kk[c] != 0xffffffffu;
                    const uint32_t k = in ? kk[c] : 0u;
                    const uint32_t sk = in ? ST_GET(k) : 2u;
                    const bool vis = in && !(sk & 2u);
                    touches += __popcll(__ballot(vis));
                    const bool absorb = vis && (dd[c] <= prm.thr_f);
                    const bool push = vis && !absorb && sk == 0u;
                    const unsigned long long pm = __ballot(push);
                    if (vis && sk == 0u)
                        seed_of[k] = (int32_t)seed;  // first touch; later touches in this BFS carry the same seed
                    if (absorb)
                        ST_OR(k, 2u);
                    if (push)
                    {
                        q[qt + __popcll(pm & lt)] = k;
                        ST_OR(k, 1u);
                    }
                    qt += __popcll(pm);
                }
            }
        }
        if (lane == 0)
            valid[seed] = (touches >= prm.min_size && touches <= prm.max_size) ? 1u : 0u;
    }
  }
    if (lane == 0 && st_exp)
    {
        atomicAdd((unsigned long long *)&fstate->replay_entries, st_entries);
        atomicAdd(&fstate->n_expansions, st_exp);
    }
#undef ST_GET
#undef ST_OR
}

__global__ void relabel_kernel(const FrameState *__restrict__ frame, const int32_t *__restrict__ seed_of,
                               const uint32_t *__restrict__ valid, const uint32_t *__restrict__ dense,
                               int32_t *__restrict__ labels, uint64_t cap)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame->n_obstacle)
        return;
    if (frame->nb_total > cap)
    {
        labels[i] = LPX_CLUSTER_UNDEFINED;
        return;
    }
    const uint32_t s = (uint32_t)seed_of[i];
    labels[i] = valid[s] ? (int32_t)dense[s] : LPX_CLUSTER_INVALID;
}

__global__ void set_clusters_kernel(FrameState *frame, const uint64_t *total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        frame->n_clusters = (uint32_t)*total;
}
}  // namespace

static uint32_t bits_for_count(uint32_t n)  // bits to hold values 0..n-1
{
    uint32_t b = 1;
    while (b < 32 && (1ull << b) < n)
        ++b;
    return b;
}

int lpx_run_cluster(lpx_ctx *ctx, uint32_t m_max, const lpx_clu_cfg *cfg, int32_t *d_labels)
{
    FrameState *frame = (FrameState *)ctx->frame.p;
    hipStream_t st = ctx->stream;
    if (m_max == 0)
        return LPX_OK;
    int rc = lpx_kd_build(ctx, m_max);
    if (rc)
        return rc;
    rc = lpx_neighbours(ctx, m_max, cfg->distance_squared, true);
    if (rc)
        return rc;

    const dim3 blk(256), grd((m_max + 255) / 256);
    uint32_t *root = (uint32_t *)ctx->key_a.p, *iota = (uint32_t *)ctx->val_a.p;
    uint32_t *sroot = nullptr, *members = nullptr;
    uint32_t *cc_lo = (uint32_t *)ctx->cc_lo.p, *cc_hi = (uint32_t *)ctx->cc_hi.p;
    uint32_t *valid = (uint32_t *)ctx->valid.p;
    {
        StageTimer tm(ctx, ST_CC);
        hipLaunchKernelGGL(flatten_kernel, grd, blk, 0, st, (uint32_t *)ctx->parent.p, frame, root, iota,
                           (uint8_t *)ctx->state.p, valid, cc_lo, cc_hi);
        rc = lpx_sort_pairs(ctx, root, (uint32_t *)ctx->key_b.p, iota, (uint32_t *)ctx->val_b.p, m_max,
                            &frame->n_obstacle, bits_for_count(m_max), &sroot, &members);
        if (rc)
            return rc;
        hipLaunchKernelGGL(cc_ranges_kernel, grd, blk, 0, st, sroot, frame, cc_lo, cc_hi, (uint32_t *)ctx->rpos.p);
    }
    {
        StageTimer tm(ctx, ST_REPLAY);
        ReplayParams prm;
        const double one_minus_q = 1.0 - (double)cfg->cluster_quality;
        prm.thr = (one_minus_q * one_minus_q) * (double)cfg->distance_squared;  // std::pow(x, 2) == x*x exactly
        prm.thr_f = (float)prm.thr;
        if ((double)prm.thr_f > prm.thr)
            prm.thr_f = nextafterf(prm.thr_f, -INFINITY);
        prm.min_size = cfg->min_cluster_size;
        prm.max_size = cfg->max_cluster_size;
        const size_t lds = sizeof(uint32_t) * (((size_t)m_max + 15) / 16);
        if (lds <= 96 * 1024)
        {
            if (!ctx->attr_replay)
            {
                LPX_HIP(ctx, hipFuncSetAttribute((const void *)replay_lds_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                ctx->attr_replay = true;
            }
            const uint32_t rgrid = m_max < 512u ? m_max : 512u;  // persistent: blocks pull components from a list
            hipLaunchKernelGGL(replay_lds_kernel, dim3(rgrid), dim3(WAVE), lds, st, frame, cc_lo, cc_hi, members,
                               (const uint32_t *)ctx->nb_off.p, (const uint32_t *)ctx->nb_len.p,
                               (const uint32_t *)ctx->nb_idx.p, (const float *)ctx->nb_dist.p,
                               (int32_t *)ctx->seed_of.p, (uint32_t *)ctx->queue.p, valid, prm, ctx->cap_nb, frame,
                               (const uint32_t *)ctx->rpos.p);
        }
        else
            hipLaunchKernelGGL(replay_kernel, dim3((m_max + RP_WAVES - 1) / RP_WAVES), dim3(RP_WAVES * WAVE), 0, st,
                               frame, cc_lo, cc_hi, members, (const uint32_t *)ctx->nb_off.p,
                               (const uint32_t *)ctx->nb_len.p, (const uint32_t *)ctx->nb_idx.p,
                               (const float *)ctx->nb_dist.p, (uint8_t *)ctx->state.p, (int32_t *)ctx->seed_of.p,
                               (uint32_t *)ctx->queue.p, valid, prm, ctx->cap_nb);
    }
    {
        StageTimer tm(ctx, ST_LABELS);
        uint32_t *dense = (uint32_t *)ctx->nb_len.p;  // neighbour lengths are no longer needed
        uint64_t *total = (uint64_t *)((char *)ctx->hist.p);  // 8-byte scratch at the head of hist
        rc = lpx_exclusive_scan(ctx, valid, dense, m_max, &frame->n_obstacle, total);
        if (rc)
            return rc;
        hipLaunchKernelGGL(set_clusters_kernel, dim3(1), dim3(64), 0, st, frame, total);
        hipLaunchKernelGGL(relabel_kernel, grd, blk, 0, st, frame, (const int32_t *)ctx->seed_of.p, valid, dense,
                           d_labels, ctx->cap_nb);
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
