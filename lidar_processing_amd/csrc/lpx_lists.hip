// lpx_lists.hip -- every radius-neighbour list of the obstacle cloud, in the reference's emission order, and the
// connected components of the d-graph from those lists (the LIST neighbour mode: LPX_NEIGHBOURS_LISTS).
//
// Replaces KDTree<float,3>::radius_search (reference src/kdtree.hpp:292-341) as Clusterer::cluster calls it for the
// points it expands (src/clustering.cpp:90); the tree itself comes from lpx_kdbuild.hip.
#include "lpx_kd_shared.h"

#include <string.h>
#include <stdlib.h>

namespace
{
// ------------------------------------------------------------------------------------------------
// pre-order layout.  PR[rank] = node with pre-order rank `rank`; a subtree is a contiguous rank
// interval [rank(root), rank(root) + size), so "emit in pre-order" becomes "emit in array order".
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// radius-neighbour lists of every point, in the emission order of KDTree::radius_search
// (src/kdtree.hpp:292-341: pre-order, left before right, inclusive dist <= r2).
//
// The reference prunes a child only when no point below it can be in range, so the list of a query
// is exactly {nodes with dist_sqr <= r2} in pre-order.  One wavefront serves a GROUP of queries:
// the <= 64 nodes of one bucket subtree (level D, where subtrees hold <= 64 nodes), or one node above
// that level.  It walks the top D levels once for the group's bounding box (+r), breadth-first but
// order-preserving (each unexpanded subtree is replaced in place by [node, left?, right?]), which
// yields the candidate set as a short sequence of rank intervals already in pre-order; candidates are
// then distance-tested 64 at a time with the reference's float expression, and accepted ones are
// appended in order -- no sort.  Stopping the expansion early (sequence full) only widens the
// candidate intervals, it never changes the result.
// ------------------------------------------------------------------------------------------------
// BLOCK = true : one workgroup per bucket subtree (<= 64 queries); the four wavefronts share the
//                candidate tile and split the queries (query j -> wavefront j % 4)
// BLOCK = false: one wavefront per node above the bucket level (a single query each)
// Both count, allocate (64-bit atomic bump of frame->nb_total, one block of list storage per group)
// and fill in the same launch; off[i] / len[i] locate the list of point i.
#ifdef LPX_NB_WPE
#define NB_GROUP_BOUNDS __launch_bounds__(NB_THREADS) __attribute__((amdgpu_waves_per_eu(LPX_NB_WPE, LPX_NB_WPE)))
#else
#define NB_GROUP_BOUNDS __launch_bounds__(NB_THREADS)
#endif
__global__ NB_GROUP_BOUNDS void nb_group_kernel(const Node *__restrict__ PR, FrameState *frame,
                                                               float r2, float rr, float thr_f,
                                                               uint32_t *__restrict__ len,
                                                               uint32_t *__restrict__ off,
                                                               uint32_t *__restrict__ nb_idx, uint64_t cap,
                                                               uint64_t cap_rs,
                                                               uint32_t *__restrict__ parent,
                                                               uint32_t *__restrict__ dbg, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<4>(fv.fs);
    __shared__ Item s_seq[NB_SEQ];
    PR = lpx_slot(PR, fv.fs);
    frame = lpx_slot(frame, fv.fs);
    len = lpx_slot(len, fv.fs);
    off = lpx_slot(off, fv.fs);
    parent = lpx_slot(parent, fv.fs);
    nb_idx = lpx_slot(nb_idx, fv.fs_nb);
    __shared__ uint32_t s_pre[NB_SEQ / 2 + 8 * NB_WAVES];
    __shared__ Node s_tile[NB_NODES + NB_WAVES * NB_GRAN];  // + one granule of far-away nodes per wavefront
    __shared__ float s_cbox[NB_NODES / NB_GRAN][6];
    __shared__ uint32_t s_q[2][WAVE];  // per-query counts / write cursors (BLOCK mode)
    __shared__ uint32_t s_n[4];        // n_cur, T, cur offset, abort
    __shared__ uint32_t s_phase;       // the group tries a single-pass reservation
    const uint32_t w = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const uint32_t M = frame->n_obstacle;
    if (M == 0)
        return;
    const float r2c = r2 * 1.0001f + 1.0e-6f;  // conservative radius^2 for the chunk cull
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t D = 0;
    while ((M >> D) > (uint32_t)NB_BUCKET)
        ++D;
    const uint32_t nbk = 1u << D;
    // blocks [0, nbk): one bucket group each; blocks [nbk, ...): four single-node groups each
    const bool BLOCK = lpx_blk.x < nbk;
    uint32_t level, path, gid;
    if (BLOCK)
    {
        gid = lpx_blk.x;
        level = D;
        path = gid;
    }
    else
    {
        const uint32_t u = (lpx_blk.x - nbk) * NB_WAVES + w;
        if (u >= nbk - 1)
            return;
        gid = nbk + u;
        level = 31 - __clz(u + 1);
        path = u + 1 - (1u << level);
    }
    // A split node above the bucket level is a query of its own -- 65 535 one-query groups beside the 61 440 buckets of a
    // 5M-point frame, each a wavefront that walks the whole tree for ONE list (18 % of the kernel).  The nodes of the last
    // NB_ADOPT levels above the buckets (15/16 of them) are adopted by a bucket instead: node u rides with its in-order
    // predecessor bucket -- the rightmost bucket of its left subtree, which lies in the same small cell -- as that
    // group's last query.  Both sides evaluate the same predicate: the bucket adopts its ancestor at level D - 1 - t (t =
    // trailing ones of its path, the bit above them a zero) iff t < NB_ADOPT and it has a lane to spare; the node's own
    // group leaves iff that bucket exists and adopts it.
#ifndef LPX_NB_ADOPT
#define LPX_NB_ADOPT 4
#endif
    constexpr uint32_t NB_ADOPT = LPX_NB_ADOPT;
    const uint32_t tz = BLOCK ? (uint32_t)__builtin_ctz(~path) : 0u;  // (path == 2^D - 1: tz >= D, nothing above)
    uint32_t gb = 0, ge = M, grank = 0, rank_up = 0xffffffffu;
    for (int d = (int)level - 1; d >= 0; --d)
    {
        if (gb >= ge)
            break;
        if (BLOCK && (uint32_t)d == tz)
            rank_up = grank;  // the node whose left subtree this bucket closes on the right
        const uint32_t mid = gb + (ge - gb) / 2;
        if ((path >> d) & 1u)
        {
            grank += 1 + (mid - gb);
            gb = mid + 1;
        }
        else
        {
            grank += 1;
            ge = mid;
        }
    }
    if (gb >= ge)
        return;
    if (!BLOCK && level < D && D - 1 - level < NB_ADOPT)
    {
        // this node's in-order predecessor bucket: left child, then right children down to the bucket level
        uint32_t b1 = gb, e1 = gb + (ge - gb) / 2;
        for (uint32_t l = level + 1; l < D && b1 < e1; ++l)
            b1 = b1 + (e1 - b1) / 2 + 1;
        if (b1 < e1 && e1 - b1 < (uint32_t)WAVE)
            return;  // adopted: that bucket's group builds this node's list
    }
    const uint32_t nqb = BLOCK ? (ge - gb) : 1u;
    const bool adopt = BLOCK && tz < NB_ADOPT && tz < D && rank_up != 0xffffffffu && nqb < (uint32_t)WAVE;
    const uint32_t nq = __builtin_amdgcn_readfirstlane(nqb + (adopt ? 1u : 0u));
    const bool active = lane < nq;
    const Node q = PR[(adopt && lane == nqb) ? rank_up : grank + (active ? lane : 0u)];
    const uint32_t qi = __float_as_uint(q.w);

    // LDS partition: BLOCK mode uses everything, wave mode a quarter each
    const uint32_t caps = BLOCK ? NB_SEQ / 2 : NB_SEQ / 2 / NB_WAVES;
    const uint32_t tile_cap = BLOCK ? NB_NODES : NB_NODES / NB_WAVES;
    Item *seqbuf = BLOCK ? s_seq : s_seq + w * (NB_SEQ / NB_WAVES);
    uint32_t *pre = BLOCK ? s_pre : s_pre + w * (NB_SEQ / 2 / NB_WAVES + 8);
    Node *tile = BLOCK ? s_tile : s_tile + w * (NB_NODES / NB_WAVES);
    float(*cbox)[6] = BLOCK ? s_cbox : s_cbox + w * (NB_NODES / NB_GRAN / NB_WAVES);
    // granule index (relative to `tile`) of this wavefront's far-away granule: what a distance step reads
    // in the lane rows it has no surviving granule for
    const uint32_t pad_g = BLOCK ? (uint32_t)(NB_NODES / NB_GRAN) + w
                                 : (uint32_t)(NB_NODES / NB_GRAN) + w - w * (NB_NODES / NB_WAVES / NB_GRAN);
    if (lane < (uint32_t)NB_GRAN)
        s_tile[NB_NODES + w * NB_GRAN + lane] = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __uint_as_float(0xffffffffu));
    const uint32_t nthr = BLOCK ? NB_THREADS : WAVE;
    const uint32_t tix = BLOCK ? threadIdx.x : lane;
    const uint32_t nwav = BLOCK ? NB_WAVES : 1;
    const uint32_t wix = BLOCK ? w : 0;

    Item *cur = nullptr;
    uint32_t n_cur = 0, T = 0;
    if (!BLOCK || w == 0)
    {
        // bounding box of the group's queries, widened by a conservative radius
        float blo[3] = {q.x, q.y, q.z}, bhi[3] = {q.x, q.y, q.z};
#pragma unroll
        for (int a = 0; a < 3; ++a)
        {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
            {
                blo[a] = fminf(blo[a], __shfl_xor(blo[a], o, 64));
                bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], o, 64));
            }
            // widened by the radius plus two ulps of the result: far from the origin (map / UTM frames) the
            // rounding of this subtraction is larger than any fixed margin
            blo[a] -= rr;
            bhi[a] += rr;
            blo[a] -= fabsf(blo[a]) * 2.4e-7f;
            bhi[a] += fabsf(bhi[a]) * 2.4e-7f;
        }
        n_cur = nb_traverse(PR, M, D, blo, bhi, seqbuf, caps, pre, lane, &cur, &T);
        if (BLOCK && lane == 0)
        {
            s_n[0] = n_cur;
            s_n[1] = T;
            s_n[2] = (uint32_t)(cur - seqbuf);
            s_n[3] = 0;
            // whether the group tries a single-pass reservation: decided ONCE per group, here (see `phase` below)
            s_phase = frame->rs_stripe[gid % LPX_RS_STRIPES].v < cap_rs / LPX_RS_STRIPES ? 1u : 0u;
        }
    }
    if (BLOCK)
    {
        __syncthreads();
        n_cur = s_n[0];
        T = s_n[1];
        cur = seqbuf + s_n[2];
    }
    if (dbg && lane == 0 && (!BLOCK || w == 0))
        dbg[gid * 8 + 6] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);  // the traversal

    const unsigned long long lt = lpx_lanemask_lt();
    uint32_t my_cnt = 0;     // lane j (of the wavefront that owns query j): room asked for the list of query j
    uint32_t my_cursor = 0;  // ... its write position
    uint32_t my_len = 0;     // ... the number of neighbours written
    uint32_t my_min = qi;    // ... and the smallest neighbour index (first union-find link)
    // Phases.  RESERVE asks for room without computing a distance: every candidate of every chunk that
    // survives the cull of a query could be a neighbour, so that sum bounds the list length.  If the
    // single-pass region [cap, cap + cap_rs) of the workspace has that much room the distances are evaluated
    // ONCE (FILL) and the lists keep a gap at the end; otherwise COUNT evaluates them to get exact lengths
    // first and the lists go to the exact region [0, cap).  FILL writes the lists in place.
    enum
    {
        PH_RESERVE,
        PH_COUNT,
        PH_FILL
    };
    // (sub-region exhausted: do not try.)  The four wavefronts of a bucket group MUST agree -- they meet at the barriers of
    // the phase loop and pool their per-query sizes for ONE allocation.  Rounds 1-5 let every thread read the stripe's
    // cursor for itself: the cursor moves while other groups allocate, and close to the end of a sub-region one wavefront
    // of a group could still see room where another no longer did -- one reserved by upper bounds while the other counted
    // exact lengths, and the group's lists came out with the wrong sizes.  Never seen with 512 words per point of
    // single-pass room (1000 fresh contexts: 0 wrong partitions); 2-5 per 1000 with the 192 words of round 6, which is
    // how it was found (tools/r6_flaky.py).  Now lane 0 of wavefront 0 decides (s_phase, behind the barrier above); a
    // single-node group is one wavefront and takes lane 0's reading.
    const uint32_t try_rs = BLOCK ? s_phase
                                  : (uint32_t)__builtin_amdgcn_readfirstlane(
                                        frame->rs_stripe[gid % LPX_RS_STRIPES].v < cap_rs / LPX_RS_STRIPES ? 1 : 0);
    int phase = try_rs ? PH_RESERVE : PH_COUNT;
    bool staged_once = false;
    for (;;)
    {
        for (uint32_t t0 = 0; t0 < T; t0 += tile_cap)
        {
            const uint32_t tn = min(tile_cap, T - t0);
            // a group whose candidates fit one tile keeps tile and chunk boxes from its first phase
            const bool stage = !staged_once || T > tile_cap;
            // the last granule is padded with nodes infinitely far away, so the distance loop needs no bounds test
            const uint32_t tn_pad = (tn + NB_GRAN - 1) & ~(uint32_t)(NB_GRAN - 1);
            for (uint32_t c = tix; stage && c < tn_pad; c += nthr)
            {
                Node nd = make_float4(3.0e38f, 3.0e38f, 3.0e38f, __uint_as_float(0xffffffffu));
                if (c < tn)
                {
                    const uint32_t ci = t0 + c;
                    uint32_t lo = 0, hi = n_cur - 1;  // last interval with pre <= ci
                    while (lo < hi)
                    {
                        const uint32_t m2 = (lo + hi + 1) / 2;
                        if (pre[m2] <= ci)
                            lo = m2;
                        else
                            hi = m2 - 1;
                    }
                    nd = PR[cur[lo].rank + (ci - pre[lo])];
                }
                tile[c] = nd;
            }
            if (BLOCK)
                __syncthreads();
            else
                Coop<WAVE>::sync();
            // bounding box of each granule of 16 consecutive candidates (rank order keeps them compact): a
            // wavefront reduces four granules at a time, one per row of 16 lanes, with DPP row shifts
            const uint32_t ngran = tn_pad / NB_GRAN;
            for (uint32_t s4 = wix * 4; stage && s4 < ngran; s4 += nwav * 4)
            {
                const uint32_t c = s4 * NB_GRAN + lane;
                const bool valid = c < tn;
                // lanes past the end repeat the first node of their granule (which is always a real one)
                const Node nd = tile[valid ? c : min(c & ~(uint32_t)(NB_GRAN - 1), tn - 1)];
                const float lo0 = lpx_row_min15_f32(nd.x), lo1 = lpx_row_min15_f32(nd.y);
                const float lo2 = lpx_row_min15_f32(nd.z), hi0 = lpx_row_max15_f32(nd.x);
                const float hi1 = lpx_row_max15_f32(nd.y), hi2 = lpx_row_max15_f32(nd.z);
                const uint32_t g = s4 + lane / NB_GRAN;
                if ((lane % NB_GRAN) == NB_GRAN - 1 && g < ngran)
                {
                    cbox[g][0] = lo0;
                    cbox[g][1] = lo1;
                    cbox[g][2] = lo2;
                    cbox[g][3] = hi0;
                    cbox[g][4] = hi1;
                    cbox[g][5] = hi2;
                }
            }
            if (BLOCK)
                __syncthreads();
            else
                Coop<WAVE>::sync();
            for (uint32_t j = wix; j < nq; j += nwav)
            {
                const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), j));
                const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), j));
                const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), j));
                // cull: lane g tests granule g's box against the query sphere (conservatively)
                bool keep = false;
                if (lane < ngran)
                {
                    const float ex = fmaxf(fmaxf(cbox[lane][0] - qx, qx - cbox[lane][3]), 0.0f);
                    const float ey = fmaxf(fmaxf(cbox[lane][1] - qy, qy - cbox[lane][4]), 0.0f);
                    const float ez = fmaxf(fmaxf(cbox[lane][2] - qz, qz - cbox[lane][5]), 0.0f);
                    keep = (ex * ex + ey * ey + ez * ez) <= r2c;
                }
                unsigned long long km = __ballot(keep);
                if (phase == PH_RESERVE)
                {
                    uint32_t ub = (uint32_t)__popcll(km) * NB_GRAN;
                    if (ngran && ((km >> (ngran - 1)) & 1ull))
                        ub -= ngran * NB_GRAN - tn;  // the last granule of the tile may be partial
                    if (lane == j)
                        my_cnt += ub;
                    continue;
                }
                uint32_t run = (phase == PH_COUNT) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)my_cursor, j);
                const uint32_t run0 = run;
                uint32_t mn = 0xffffffffu;
                const uint32_t row = lane / NB_GRAN, col = lane % NB_GRAN;
                while (km)
                {
                    // four surviving granules per step, one per row of 16 lanes, in candidate order; rows
                    // without a granule read the far-away one
                    uint32_t g0, g1 = pad_g, g2 = pad_g, g3 = pad_g;
                    g0 = (uint32_t)(__ffsll((long long)km) - 1);
                    km &= km - 1;
                    if (km)
                    {
                        g1 = (uint32_t)(__ffsll((long long)km) - 1);
                        km &= km - 1;
                    }
                    if (km)
                    {
                        g2 = (uint32_t)(__ffsll((long long)km) - 1);
                        km &= km - 1;
                    }
                    if (km)
                    {
                        g3 = (uint32_t)(__ffsll((long long)km) - 1);
                        km &= km - 1;
                    }
                    const uint32_t gs = row == 0 ? g0 : (row == 1 ? g1 : (row == 2 ? g2 : g3));
                    const Node n0 = tile[gs * NB_GRAN + col];
                    const float a0 = qx - n0.x, a1 = qy - n0.y, a2 = qz - n0.z;
                    // src/kdtree.hpp:145-157 sums d^2 from the last axis into 0.0f; a square is never -0, so the
                    // "+ 0.0f" of the reference is the identity and is not issued
                    const float da = a0 * a0 + (a1 * a1 + a2 * a2);
                    const bool ia = da <= r2;  // :315 inclusive; padding is never in range
                    const unsigned long long ma = __ballot(ia);
                    if (phase == PH_FILL && ia)
                    {
                        mn = min(mn, __float_as_uint(n0.w));
                        // one word per neighbour: index | (within the absorb radius) << 31.  For a float d,
                        // (double)d <= thr of src/clustering.cpp:102 <=> d <= thr_f
                        nb_idx[run + __popcll(ma & lt)] = __float_as_uint(n0.w) | (da <= thr_f ? 0x80000000u : 0u);
                    }
                    run += (uint32_t)__popcll(ma);
                }
                if (phase == PH_FILL)
                    mn = (uint32_t)__builtin_amdgcn_readlane((int)lpx_wave_min63_u32(mn), WAVE - 1);
                if (lane == j)
                {
                    if (phase == PH_COUNT)
                        my_cnt += run;
                    else
                    {
                        my_min = min(my_min, mn);
                        my_cursor += run - run0;
                        my_len += run - run0;
                    }
                }
            }
            if (BLOCK)
                __syncthreads();
            else
                Coop<WAVE>::sync();
        }
        staged_once = true;
        const bool mine = active && (!BLOCK || (lane % NB_WAVES) == w);
        if (phase == PH_FILL)
        {
            if (mine)
            {
                len[qi] = my_len;
                // first link of the union-find forest: every point under its smallest neighbour
                if (parent)
                    parent[qi] = my_min;
            }
            const uint32_t wrote = lpx_wave_sum_u32(mine ? my_len : 0u);
            if (lane == 0 && wrote)
                atomicAdd((unsigned long long *)&frame->ent_stripe[gid % LPX_RS_STRIPES].v, (unsigned long long)wrote);
            break;
        }
        // allocate the group's list storage: one 64-bit atomic bump of frame->nb_total per group
        bool ok;
        if (BLOCK)
        {
            // gather the per-query sizes (query j lives in lane j of wavefront j % 4)
            if (mine)
                s_q[0][lane] = my_cnt;
            __syncthreads();
            if (w == 0)
            {
                const uint32_t c = active ? s_q[0][lane] : 0u;
                const uint32_t incl = lpx_wave_incl_scan_u32(c);
                const uint32_t total = __shfl(incl, WAVE - 1, 64);
                unsigned long long base = 0;
                const uint32_t stripe = gid % LPX_RS_STRIPES;
                const unsigned long long stripe_cap = cap_rs / LPX_RS_STRIPES;
                unsigned long long *counter =
                    (unsigned long long *)(phase == PH_RESERVE ? &frame->rs_stripe[stripe].v : &frame->nb_total);
                if (lane == 0)
                    base = atomicAdd(counter, (unsigned long long)total);
                base = __shfl(base, 0, 64);
                const bool fits = base + total <= (phase == PH_RESERVE ? stripe_cap : cap);
                if (phase == PH_RESERVE)
                    base += cap + stripe * stripe_cap;  // the single-pass region lies behind the exact one
                if (fits && active)
                {
                    const uint32_t o = (uint32_t)base + incl - c;
                    s_q[1][lane] = o;
                    off[qi] = o;
                }
                if (lane == 0)
                {
                    s_n[3] = fits ? 0u : 1u;
                    if (dbg)
                    {
                        dbg[gid * 8 + 0] = T;
                        dbg[gid * 8 + 1] = n_cur;
                        dbg[gid * 8 + 2] = nq;
                        dbg[gid * 8 + 3] = total;
                        dbg[gid * 8 + 4] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
                    }
                }
            }
            __syncthreads();
            ok = s_n[3] == 0;
            if (ok && mine)
                my_cursor = s_q[1][lane];
            __syncthreads();  // s_q / s_n are reused if the group has to count
        }
        else
        {
            const uint32_t total = __builtin_amdgcn_readlane((int)my_cnt, 0);
            unsigned long long base = 0;
            const uint32_t stripe = gid % LPX_RS_STRIPES;
            const unsigned long long stripe_cap = cap_rs / LPX_RS_STRIPES;
            unsigned long long *counter =
                (unsigned long long *)(phase == PH_RESERVE ? &frame->rs_stripe[stripe].v : &frame->nb_total);
            if (lane == 0)
                base = atomicAdd(counter, (unsigned long long)total);
            base = __shfl(base, 0, 64);
            ok = base + total <= (phase == PH_RESERVE ? stripe_cap : cap);
            if (phase == PH_RESERVE)
                base += cap + stripe * stripe_cap;
            if (ok)
            {
                my_cursor = (uint32_t)base;
                if (lane == 0)
                {
                    off[qi] = my_cursor;
                    if (dbg)
                    {
                        dbg[gid * 8 + 0] = T;
                        dbg[gid * 8 + 1] = n_cur;
                        dbg[gid * 8 + 2] = nq;
                        dbg[gid * 8 + 3] = total;
                        dbg[gid * 8 + 4] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
                    }
                }
            }
        }
        if (ok)
            phase = PH_FILL;
        else if (phase == PH_RESERVE)
        {
            phase = PH_COUNT;
            my_cnt = 0;
        }
        else
        {
            // exact lengths do not fit: nb_total keeps growing to (at least) the required size
            if (lane == 0 && (!BLOCK || w == 0))
                atomicCAS(&frame->status, 0u, (uint32_t)(-LPX_ERR_CAPACITY));  // an earlier error code stays
            return;
        }
    }
    if (dbg && lane == 0 && (!BLOCK || w == 0))
        dbg[gid * 8 + 5] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
}

// connected components of the d-graph.  The neighbour kernel has already put every point under its
// smallest neighbour; cc_flatten_kernel points everybody at the current root, then one wavefront per
// list checks every edge: equal roots (the common case) cost one cached load, the rest are united.
__global__ void cc_flatten_kernel(const FrameState *__restrict__ frame, uint32_t *parent, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<4>(fs);
    frame = lpx_slot(frame, fs);
    parent = lpx_slot(parent, fs);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i >= frame->n_obstacle)
        return;
    if (frame->status)  // the lists did not fit: groups returned before they wrote their parents (caller retries)
    {
        uf_st(parent + i, i);
        return;
    }
    uint32_t x = i, p = uf_ld(parent + x);
    while (p != x)
    {
        x = p;
        p = uf_ld(parent + x);
    }
    uf_st(parent + i, x);
}

// Which of the two edge checks serves a frame is a property of its lists: one wavefront per list (cc_hook_kernel) when
// they are long -- the reference's frames at d = 0.5 m: ~140 entries -- the lists of 64 points end to end
// (cc_hook_flat_kernel) when they are short -- BASELINE's 5M-point cloud: 19.  Both are launched; each leaves at once
// when the frame's average is not its own (the average is on the device only, and an empty launch costs nothing).
constexpr uint32_t CC_FLAT_BELOW = 48;  // entries per point
__device__ __forceinline__ bool cc_lists_are_short(const FrameState *frame)
{
    unsigned long long e = frame->nb_entries;
    for (uint32_t i = 0; i < LPX_RS_STRIPES; ++i)
        e += frame->ent_stripe[i].v;
    return e < (unsigned long long)CC_FLAT_BELOW * frame->n_obstacle;
}

__global__ __launch_bounds__(256) void cc_hook_kernel(const FrameState *__restrict__ frame,
                                                       const uint32_t *__restrict__ off,
                                                       const uint32_t *__restrict__ len,
                                                       const uint32_t *__restrict__ nb_idx, uint32_t *parent,
                                                       uint64_t cap, uint32_t roots_only, uint32_t regime, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<4>(fv.fs);
    frame = lpx_slot(frame, fv.fs);
    off = lpx_slot(off, fv.fs);
    len = lpx_slot(len, fv.fs);
    parent = lpx_slot(parent, fv.fs);
    nb_idx = lpx_slot(nb_idx, fv.fs_nb);
    const uint32_t lane = threadIdx.x % WAVE;
    const uint32_t M = frame->n_obstacle;
    if (frame->nb_total > cap || (regime && cc_lists_are_short(frame)))
        return;
    const uint32_t stride = gridDim.x * (blockDim.x / WAVE);
    for (uint32_t i = (lpx_blk.x * blockDim.x + threadIdx.x) / WAVE; i < M; i += stride)
    {
        // roots_only: a first, cheap round over the lists of the forest's roots alone.  A root has no
        // smaller neighbour; any neighbour that hangs under another tree merges the two, which removes
        // most stale-root mismatches from the full round that follows (after another flatten).
        if (roots_only && uf_ld(parent + i) != i)
            continue;
        const uint32_t lim = roots_only ? 0xffffffffu : i;  // a root's neighbours all have larger indices
        const uint32_t o = off[i], n = len[i];
        // values known to lie in i's component: its cached root and up to three (possibly stale) roots met
        // in this list.  Stale roots repeat all over a list, so each distinct one costs ONE union attempt
        // by one lane instead of a divergent find per entry.
        uint32_t a0 = uf_ld(parent + i), a1 = a0, a2 = a0, a3 = a0;
        // four chunks of the list per trip: the index loads, then the parent gathers, are issued together
        for (uint32_t t0 = 0; t0 < n; t0 += 4 * WAVE)
        {
            uint32_t k[4], pk[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const uint32_t t = t0 + c * WAVE + lane;
                k[c] = (t < n) ? (nb_idx[o + t] & 0x7fffffffu) : 0xffffffffu;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
                pk[c] = uf_ld(parent + (k[c] < lim ? k[c] : i));  // other entries (and padding) read parent[i]
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                bool bad = k[c] < lim && pk[c] != a0 && pk[c] != a1 && pk[c] != a2 && pk[c] != a3;
                unsigned long long bm = __ballot(bad);
                while (bm)
                {
                    const int f = __ffsll((long long)bm) - 1;
                    const uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)pk[c], f);
                    if (lane == 0)
                        uf_unite(parent, i, cand);  // cand is an ancestor of a neighbour: same component as i
                    a3 = a2;
                    a2 = a1;
                    a1 = cand;
                    bad = bad && pk[c] != cand;
                    bm = __ballot(bad);
                }
            }
        }
    }
}

// The same check over the lists of 64 consecutive points AT ONCE (round 6).  cc_hook_kernel gives every list a whole
// wavefront and three dependent round trips (offset / length / root, the list, the parents of its entries): on a
// 5M-point frame a list holds 19 entries -- a third of the lanes -- and the 2.3 M trips of 16 384 wavefronts were the
// kernel's 0.79 ms.  Here a wavefront fetches offset, length and root of 64 points with one coalesced trip, lays their
// entries end to end (an inclusive scan of the lengths, kept in LDS) and walks that stream 64 entries per step, four steps
// in flight: every lane finds the point its entry belongs to by a binary search over the scan, every load is independent
// of the one before.  An entry whose parent differs from its point's root is united right there, lane by lane
// (uf_unite: stale reads only cost a retry).  (The cheap first round over the roots' lists alone stays: without it the
// full round costs more than both together -- 5M: 0.465 against 0.397 ms.)
constexpr int CCF_WAVES = 4;
constexpr int CCF_UNROLL = 4;
__global__ __launch_bounds__(CCF_WAVES *WAVE) void cc_hook_flat_kernel(const FrameState *__restrict__ frame,
                                                                         const uint32_t *__restrict__ off,
                                                                         const uint32_t *__restrict__ len,
                                                                         const uint32_t *__restrict__ nb_idx,
                                                                         uint32_t *parent, uint64_t cap,
                                                                         uint32_t roots_only, uint32_t regime, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<4>(fv.fs);
    __shared__ uint32_t s_incl[CCF_WAVES][WAVE];
    // values already known to lie in a point's set besides its root: the (possibly stale) roots it has been united with.
    // Stale roots repeat all over a dense list; each distinct one costs ONE union instead of a find per entry.
    __shared__ uint32_t s_acc[CCF_WAVES][2][WAVE];
    frame = lpx_slot(frame, fv.fs);
    off = lpx_slot(off, fv.fs);
    len = lpx_slot(len, fv.fs);
    parent = lpx_slot(parent, fv.fs);
    nb_idx = lpx_slot(nb_idx, fv.fs_nb);
    const uint32_t lane = threadIdx.x % WAVE, w = threadIdx.x / WAVE;
    const uint32_t M = frame->n_obstacle;
    if (frame->nb_total > cap || (regime && !cc_lists_are_short(frame)))
        return;
    uint32_t *const incl_w = s_incl[w];
    const uint32_t stride = gridDim.x * CCF_WAVES * WAVE;
    for (uint32_t i0 = (lpx_blk.x * CCF_WAVES + w) * WAVE; i0 < M; i0 += stride)
    {
        const uint32_t i = i0 + lane;
        const bool have = i < M;
        const uint32_t o = have ? off[i] : 0u;
        const uint32_t ri = have ? uf_ld(parent + i) : 0u;
        // roots_only: a first, cheap round over the lists of the forest's roots alone (see cc_hook_kernel)
        const uint32_t n = (have && !(roots_only && ri != i)) ? len[i] : 0u;
        const uint32_t incl = lpx_wave_incl_scan_u32(n);
        const uint32_t T = __shfl(incl, WAVE - 1, WAVE);
        __builtin_amdgcn_wave_barrier();
        incl_w[lane] = incl;
        s_acc[w][0][lane] = ri;
        s_acc[w][1][lane] = ri;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e0 = 0; e0 < T; e0 += CCF_UNROLL * WAVE)
        {
            uint32_t jj[CCF_UNROLL], kk[CCF_UNROLL];
#pragma unroll
            for (int u = 0; u < CCF_UNROLL; ++u)
            {
                const uint32_t e = e0 + u * WAVE + lane;
                // the point whose list holds entry e: the first j with incl[j] > e (64 sorted values in LDS)
                uint32_t lo = 0;
#pragma unroll
                for (uint32_t step = WAVE / 2; step > 0; step >>= 1)
                    lo += (incl_w[lo + step - 1] <= e) ? step : 0u;
                lo = lo < (uint32_t)WAVE ? lo : WAVE - 1u;
                jj[u] = lo;
                const uint32_t oj = __shfl(o, lo, WAVE);
                const uint32_t ej = e - (__shfl(incl, lo, WAVE) - __shfl(n, lo, WAVE));
                kk[u] = nb_idx[e < T ? oj + ej : 0u];
            }
            uint32_t pk[CCF_UNROLL];
#pragma unroll
            for (int u = 0; u < CCF_UNROLL; ++u)
            {
                const uint32_t e = e0 + u * WAVE + lane;
                const uint32_t k = kk[u] & 0x7fffffffu;
                const uint32_t ij = i0 + jj[u];
                // a root's neighbours all have larger indices; otherwise only the smaller end of an edge checks it
                const bool ask = e < T && (roots_only || k < ij);
                kk[u] = ask ? k : 0xffffffffu;
                pk[u] = uf_ld(parent + (ask ? k : 0u));
            }
#pragma unroll
            for (int u = 0; u < CCF_UNROLL; ++u)
            {
                const uint32_t rj = __shfl(ri, jj[u], WAVE);
                if (kk[u] != 0xffffffffu && pk[u] != rj && pk[u] != s_acc[w][0][jj[u]] && pk[u] != s_acc[w][1][jj[u]])
                {
                    uf_unite(parent, i0 + jj[u], pk[u]);  // pk is an ancestor of a neighbour: same component
                    // (lanes of one point may race here: any of their values is one the point has been united with)
                    s_acc[w][1][jj[u]] = s_acc[w][0][jj[u]];
                    s_acc[w][0][jj[u]] = pk[u];
                }
            }
        }
    }
}
}  // namespace

int lpx_neighbours(lpx_ctx *ctx, uint32_t m_max, float r2, float thr_f, bool hook)
{
    if (m_max == 0)
        return LPX_OK;
    FrameState *frame = (FrameState *)ctx->frame.p;
    Node *PR = (Node *)ctx->nodes_pre.p;
    uint32_t *len = (uint32_t *)ctx->nb_len.p, *off = (uint32_t *)ctx->nb_off.p;
    // conservative radius for the group traversal (superset of every query's own traversal)
    const float rr = sqrtf(r2) * 1.0001f + 1.0e-3f;
    uint32_t dmax = 0;
    while ((m_max >> dmax) > (uint32_t)NB_BUCKET)
        ++dmax;
    const uint32_t groups = 2u << dmax;  // 2^D bucket groups + (2^D - 1) upper nodes
    {
        StageTimer tm(ctx, ST_NB_FILL);
        // bucket groups (one workgroup each) and the single-node groups (four per workgroup) in ONE launch.
        // The device derives the bucket level from the real point count, which may be lower than the
        // host's bound; surplus blocks return at once.
        const uint32_t nbk = groups / 2;
        hipLaunchKernelGGL(nb_group_kernel, dim3(nbk + (nbk + NB_WAVES - 1) / NB_WAVES, 1, ctx->cur_b),
                           dim3(NB_THREADS), 0, ctx->stream, (const Node *)PR, frame, r2, rr, thr_f, len, off,
                           (uint32_t *)ctx->nb_idx.p, ctx->cap_nb, ctx->exact_lists_only ? 0ull : ctx->cap_rs,
                           hook ? (uint32_t *)ctx->parent.p : (uint32_t *)nullptr, (uint32_t *)ctx->dbg_buf,
                           lpx_fv(ctx));
    }
    if (hook)
    {
        StageTimer tm(ctx, ST_NB_SCAN);
        hipLaunchKernelGGL(cc_flatten_kernel, dim3((m_max + 255) / 256, 1, ctx->cur_b), dim3(256), 0, ctx->stream, frame,
                           (uint32_t *)ctx->parent.p, ctx->fs_tag);
        const uint32_t hgrid = (m_max + 3) / 4 < 4096u ? (m_max + 3) / 4 : 4096u;
        // LPX_CC_HOOK=list / flat (development build) forces one form for every frame
        static const char *hook_env = LPX_KNOB("LPX_CC_HOOK");
        // A single frame launches only the form the context's LAST frame called for (two empty launches are ~15 us of a
        // 1.7 ms frame); its first frame, and every launch chain, launches both and lets the device decide per frame.
        const bool known = !hook_env && ctx->cur_b == 1 && ctx->list_short >= 0;
        const bool only_list = hook_env ? strcmp(hook_env, "list") == 0 : (known && ctx->list_short == 0);
        const bool only_flat = hook_env ? strcmp(hook_env, "flat") == 0 : (known && ctx->list_short == 1);
        const uint32_t regime = (only_list || only_flat) ? 0u : 1u;
        const uint32_t fgrid = (m_max + 255) / 256 < 4096u ? (m_max + 255) / 256 : 4096u;
        for (uint32_t roots_only = 1;; roots_only = 0)
        {
            if (!only_flat)
                hipLaunchKernelGGL(cc_hook_kernel, dim3(hgrid, 1, ctx->cur_b), dim3(256), 0, ctx->stream, frame,
                                   (const uint32_t *)off, (const uint32_t *)len, (const uint32_t *)ctx->nb_idx.p,
                                   (uint32_t *)ctx->parent.p, ctx->cap_nb, roots_only, regime, lpx_fv(ctx));
            if (!only_list)
                hipLaunchKernelGGL(cc_hook_flat_kernel, dim3(fgrid, 1, ctx->cur_b), dim3(CCF_WAVES * WAVE), 0, ctx->stream,
                                   frame, (const uint32_t *)off, (const uint32_t *)len, (const uint32_t *)ctx->nb_idx.p,
                                   (uint32_t *)ctx->parent.p, ctx->cap_nb, roots_only, regime, lpx_fv(ctx));
            if (!roots_only)
                break;
            hipLaunchKernelGGL(cc_flatten_kernel, dim3((m_max + 255) / 256, 1, ctx->cur_b), dim3(256), 0, ctx->stream,
                               frame, (uint32_t *)ctx->parent.p, ctx->fs_tag);
        }
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
