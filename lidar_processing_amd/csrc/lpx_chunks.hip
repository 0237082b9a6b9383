// lpx_chunks.hip -- the candidate chunks of every kd group (the SEARCH neighbour mode: LPX_NEIGHBOURS_SEARCH), the
// block boxes they are culled with, and the exact components of large frames from the same tables (kd_link_queries).
//
// Serves KDTree<float,3>::radius_search (reference src/kdtree.hpp:292-341) for the replay's expansion-driven searches
// (lpx_cluster.hip: replay_search_kernel); the tree comes from lpx_kdbuild.hip.
#include "lpx_kd_shared.h"

#include <string.h>
#include <stdlib.h>

namespace
{
// ------------------------------------------------------------------------------------------------
// Expansion-driven search, part 1: the candidate chunks of every kd group.
//
// The greedy loop of the reference expands (calls radius_search on) only ~15-20 % of the points; the rest are
// absorbed.  Instead of materialising every radius list, the replay (lpx_cluster.hip) searches for a point when
// it expands it.  What CAN be prepared for all points at once is the traversal: one wavefront per kd group (a
// bucket subtree of <= 64 nodes, or one node above the bucket level) walks the top levels for the group's box
// (+ radius) exactly like the list kernel does and leaves the candidate set as <= 64 CHUNKS of consecutive
// pre-order ranks, <= 64 nodes each, in pre-order, each with the exact bounding box of its nodes:
// chunks[gid][lane] = (rank, count, box).  A search then costs one 2 KiB load of the chunk table, a cull of the
// chunks against its query ball and one 16-byte load per surviving candidate, all independent.  If a group has more
// chunks than lanes the last one is long (covers the rest of the rank range, gaps included: nodes the traversal
// pruned fail the distance test anyway) and is never culled.  grp_of[point] = gid.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Exact connected components of the d-graph from the chunk tables (replaces the clique-cell grid: eight launches of
// hash inserts, probes and pointer chases that held 37 % of a chain's resident wavefront time while waiting for memory).
// The wavefront that has just built the chunk table of a kd group holds the group's <= 64 queries in registers and
// knows every chunk that can contain a neighbour of any of them; it tests its queries against those candidates --
// all pairs, 64 queries at once, one candidate per step broadcast from the lane that loaded it -- with the
// reference's float expression (src/kdtree.hpp:145-157, inclusive).  Every unordered pair is tested once, by the group
// of its HIGHER pre-order rank (a query only looks at candidates of lower rank).
// A query does not unite with every neighbour: with a neighbour c only if c is farther than d from the neighbour it
// linked LAST.  (Induction on the higher rank of a pair: if c is within d of an earlier linked neighbour c', the pair
// (c', c) -- both of lower rank than the query -- is connected by the time every group has run, and the query is
// linked to c'.)  That leaves one to three unions per point instead of ~90; they are kept in four registers and done
// after the scan, all lanes at once (uf_unite: hooks by CAS, stale reads only cost a retry).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void kd_link_queries(const Node *__restrict__ PR, uint32_t *parent, const ChunkRec &rec,
                                                unsigned long long cmask, float qx, float qy, float qz, uint32_t qidx,
                                                uint32_t qrank, bool act, uint32_t rank_end, const float *blo,
                                                const float *bhi, float r2, uint32_t lane)
{
    float lx = 0.0f, ly = 0.0f, lz = 0.0f;
    bool have_last = false;
    uint32_t l0 = 0xffffffffu, l1 = 0xffffffffu, l2 = 0xffffffffu, l3 = 0xffffffffu;  // pending unions, newest first
    while (cmask)
    {
        const int c = __ffsll((long long)cmask) - 1;
        cmask &= cmask - 1;
        const uint32_t crank = (uint32_t)__builtin_amdgcn_readlane((int)rec.rank, c);
        const uint32_t ccnt = (uint32_t)__builtin_amdgcn_readlane((int)rec.count, c);
        for (uint32_t o = 0; o < ccnt; o += WAVE)
        {
            const uint32_t r0 = crank + o;
            if (r0 >= rank_end)
                break;  // ranks ascend inside a chunk: nothing below the group's last rank is left
            const uint32_t cnt = min((uint32_t)WAVE, min(ccnt - o, rank_end - r0));
            const Node nd = PR[lane < cnt ? r0 + lane : 0u];
            // candidates outside the group's box (widened by the radius) cannot be a neighbour of any query
            const bool near = lane < cnt && nd.x >= blo[0] && nd.x <= bhi[0] && nd.y >= blo[1] && nd.y <= bhi[1] &&
                              nd.z >= blo[2] && nd.z <= bhi[2];
            unsigned long long km = __ballot(near);
            while (km)
            {
                const int k = __ffsll((long long)km) - 1;
                km &= km - 1;
                const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd.x), k));
                const float cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd.y), k));
                const float cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd.z), k));
                const float a0 = qx - cx, a1 = qy - cy, a2 = qz - cz;
                const float da = a0 * a0 + (a1 * a1 + a2 * a2);
                const bool hit = act && (r0 + (uint32_t)k) < qrank && da <= r2;
                if (__ballot(hit) == 0ull)
                    continue;
                const float b0 = lx - cx, b1 = ly - cy, b2 = lz - cz;
                const float db = b0 * b0 + (b1 * b1 + b2 * b2);
                if (hit && !(have_last && db <= r2))
                {
                    if (l3 != 0xffffffffu)
                        uf_unite(parent, qidx, l3);  // (more than four mutually distant neighbours: rare)
                    l3 = l2;
                    l2 = l1;
                    l1 = l0;
                    l0 = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(nd.w), k);
                    lx = cx;
                    ly = cy;
                    lz = cz;
                    have_last = true;
                }
            }
        }
    }
    if (l0 != 0xffffffffu)
        uf_unite(parent, qidx, l0);
    if (l1 != 0xffffffffu)
        uf_unite(parent, qidx, l1);
    if (l2 != 0xffffffffu)
        uf_unite(parent, qidx, l2);
    if (l3 != 0xffffffffu)
        uf_unite(parent, qidx, l3);
}


// Bounding boxes of the pre-order layout in aligned blocks of IX_SUB ranks: {lo, hi} as two float4 per block.  The chunk
// tables need the box of every candidate chunk of every group, and a node is a candidate of many groups (a KITTI frame:
// ~20, BASELINE's dense box clouds: ~64): folding the chunk's 64 nodes for every table read every node that many times
// (64 KiB of L2 reads and sixteen dependent trips per group).  The blocks are folded ONCE per frame here; a chunk's box
// is then the union of the at most IX_SUB_SPAN blocks it overlaps -- a superset of its exact box by what the two end
// blocks hold beyond the chunk (up to IX_SUB - 1 ranks each), which only makes the replay's cull a little more
// permissive, never wrong.
#ifndef LPX_IX_SUB
#define LPX_IX_SUB 16
#endif
constexpr uint32_t IX_SUB = LPX_IX_SUB;
constexpr uint32_t IX_SUB_SPAN = 64 / IX_SUB + 1;
__global__ __launch_bounds__(256) void sub_box_kernel(const Node *__restrict__ PR, const FrameState *__restrict__ frame,
                                                      float4 *__restrict__ SB, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<4>(fs);
    PR = lpx_slot(PR, fs);
    frame = lpx_slot(frame, fs);
    SB = lpx_slot(SB, fs);
    const uint32_t M = frame->n_obstacle;
    const uint32_t j = lpx_blk.x * blockDim.x + threadIdx.x;
    if (j * IX_SUB >= M)
        return;
    const uint32_t last = M - 1 - j * IX_SUB;  // (a clamped index repeats the block's last node: no minimum changes)
    Node nd[IX_SUB];
#pragma unroll
    for (uint32_t i = 0; i < IX_SUB; ++i)
        nd[i] = PR[j * IX_SUB + (i < last ? i : last)];
    float4 lo = make_float4(nd[0].x, nd[0].y, nd[0].z, 0.0f), hi = lo;
#pragma unroll
    for (uint32_t i = 1; i < IX_SUB; ++i)
    {
        lo.x = fminf(lo.x, nd[i].x), lo.y = fminf(lo.y, nd[i].y), lo.z = fminf(lo.z, nd[i].z);
        hi.x = fmaxf(hi.x, nd[i].x), hi.y = fmaxf(hi.y, nd[i].y), hi.z = fmaxf(hi.z, nd[i].z);
    }
    SB[2 * j] = lo;
    SB[2 * j + 1] = hi;
}

constexpr int IX_CAPS = 160;  // traversal items per wavefront (2 x 160 x 12 B + prefix = 4.6 KiB)
#ifndef LPX_IX_BOX_UNROLL
#define LPX_IX_BOX_UNROLL 4
#endif

// (frame and wframe name the SAME record -- read-only view and the two words the table clear resets -- so neither is
// __restrict__: aliased restrict pointers with a write through one of them would be undefined behaviour)
__global__ __launch_bounds__(NB_THREADS) void nb_index_kernel(const Node *__restrict__ PR,
                                                               const FrameState *frame, float rr,
                                                               ChunkRec *__restrict__ chunks,
                                                               float4 *__restrict__ grp_of, uint32_t spine_max,
                                                               uint32_t bucket, uint32_t *parent, float r2,
                                                               unsigned long long *__restrict__ tkey,
                                                               uint32_t *__restrict__ tparent, uint32_t *__restrict__ thead,
                                                               uint32_t cap_max, FrameState *wframe,
                                                               const float4 *__restrict__ SB, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<4>(fv.fs);
    parent = lpx_slot(parent, fv.fs);
    SB = lpx_slot(SB, fv.fs);
    if (tkey)
    {
        // The cell table of the component grid, which runs right behind this kernel, is emptied here (what
        // grid_clear_kernel did in a launch of its own: one launch less per chain).  Nothing in this kernel reads it.
        tkey = lpx_slot(tkey, fv.fs);
        tparent = lpx_slot(tparent, fv.fs);
        thead = lpx_slot(thead, fv.fs);
        wframe = lpx_slot(wframe, fv.fs);
        const uint32_t cap = cell_cap_for(wframe->n_obstacle, cap_max);
        for (uint32_t sl = lpx_blk.x * NB_THREADS + threadIdx.x; sl < cap; sl += gridDim.x * NB_THREADS)
        {
            tkey[sl] = CELL_EMPTY;
            tparent[sl] = sl;
            thead[sl] = 0;
        }
        uint32_t *const bits = (uint32_t *)(tkey + cap_max);  // the occupancy bitmap behind the table
        for (uint32_t i = lpx_blk.x * NB_THREADS + threadIdx.x; i < LPX_CELL_BITS_WORDS; i += gridDim.x * NB_THREADS)
            bits[i] = 0;
        if (lpx_blk.x == 0 && threadIdx.x == 0)
        {
            wframe->n_cells = 0;
            wframe->cell_cursor = 0;
        }
    }
    __shared__ Item s_seq[NB_WAVES][2 * IX_CAPS];
    __shared__ uint32_t s_pre[NB_WAVES][IX_CAPS + 8];
    __shared__ uint32_t s_mrank[NB_WAVES][IX_CAPS + 8], s_mpre[NB_WAVES][IX_CAPS + 8];
    __shared__ uint2 s_out[NB_WAVES][LPX_GROUP_CHUNKS];
    PR = lpx_slot(PR, fv.fs);
    frame = lpx_slot(frame, fv.fs);
    chunks = lpx_slot(chunks, fv.fs);
    grp_of = lpx_slot(grp_of, fv.fs);
    const uint32_t w = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const uint32_t M = frame->n_obstacle;
    if (M == 0)
        return;
    uint32_t D = 0;
    while ((M >> D) > bucket)
        ++D;
    const uint32_t nbk = 1u << D;
    const uint32_t gid = lpx_blk.x * NB_WAVES + w;  // [0, nbk): buckets; [nbk, 2 nbk - 1): upper nodes
    if (gid >= 2 * nbk - 1)
        return;
    uint32_t level, path;
    if (gid < nbk)
    {
        level = D;
        path = gid;
    }
    else
    {
        const uint32_t u = gid - nbk;
        level = 31 - __clz(u + 1);
        path = u + 1 - (1u << level);
    }
    uint32_t gb = 0, ge = M, grank = 0;
    for (int d = (int)level - 1; d >= 0; --d)
    {
        if (gb >= ge)
            break;
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
    // A bucket also serves the upper nodes directly above it on its left spine (at most two: its parent when the
    // bucket is a left child, and the grandparent when the parent is one too): in pre-order they are the ranks just
    // before the bucket, and they lie on the boundary of its region, so the group's box barely grows -- while a group
    // of their own would cost a whole traversal and a 2 KiB chunk table for ONE point each (they were 3/8 of all
    // groups).  Upper nodes further up keep their single-node groups.
    uint32_t spine = 0;
    if (gid < nbk)
    {
        spine = path ? (uint32_t)__ffs(path) - 1u : D;
        spine = spine < spine_max ? spine : spine_max;
    }
    else if (level + spine_max >= D)
        return;  // served by the leftmost bucket below it
    const uint32_t g0 = grank - spine;
    const uint32_t nq = __builtin_amdgcn_readfirstlane(gid < nbk ? (ge - gb) + spine : 1u);  // <= 64 + 2
    const bool active = lane < nq;
    const Node q = PR[g0 + (active ? lane : 0u)];
    if (active)
        grp_of[__float_as_uint(q.w)] = make_float4(q.x, q.y, q.z, __uint_as_float(gid));
    float blo[3] = {q.x, q.y, q.z}, bhi[3] = {q.x, q.y, q.z};
    if (nq > (uint32_t)WAVE)
    {
        const bool more = lane + WAVE < nq;
        const Node q2 = PR[g0 + (more ? lane + WAVE : 0u)];
        if (more)
        {
            grp_of[__float_as_uint(q2.w)] = make_float4(q2.x, q2.y, q2.z, __uint_as_float(gid));
            blo[0] = fminf(blo[0], q2.x), blo[1] = fminf(blo[1], q2.y), blo[2] = fminf(blo[2], q2.z);
            bhi[0] = fmaxf(bhi[0], q2.x), bhi[1] = fmaxf(bhi[1], q2.y), bhi[2] = fmaxf(bhi[2], q2.z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
        {
            blo[a] = fminf(blo[a], __shfl_xor(blo[a], o, 64));
            bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], o, 64));
        }
        blo[a] -= rr;
        bhi[a] += rr;
        blo[a] -= fabsf(blo[a]) * 2.4e-7f;  // two ulps: far from the origin the rounding beats any fixed margin
        bhi[a] += fabsf(bhi[a]) * 2.4e-7f;
    }
    Item *cur = nullptr;
    uint32_t T = 0;
    uint32_t *pre = s_pre[w];
    const uint32_t n_cur = nb_traverse(PR, M, D, blo, bhi, s_seq[w], IX_CAPS, pre, lane, &cur, &T);
    // merge items with consecutive ranks into runs: run starts where the rank does not continue the previous item
    uint32_t n_runs = 0;
    for (uint32_t c0 = 0; c0 < n_cur; c0 += WAVE)
    {
        const uint32_t i = c0 + lane;
        const bool valid = i < n_cur;
        bool start = false;
        uint32_t rank = 0;
        if (valid)
        {
            rank = cur[i].rank;
            const uint32_t cnt_prev = i ? pre[i] - pre[i - 1] : 0u;
            start = (i == 0) || (cur[i - 1].rank + cnt_prev != rank);
        }
        const unsigned long long sm = __ballot(start);
        if (start)
        {
            const uint32_t m = n_runs + (uint32_t)__popcll(sm & lpx_lanemask_lt());
            s_mrank[w][m] = rank;
            s_mpre[w][m] = pre[i];
        }
        n_runs += (uint32_t)__popcll(sm);
    }
    if (lane == 0)
        s_mpre[w][n_runs] = T;
    Coop<WAVE>::sync();
    // cut every run into chunks of <= 64 ranks; chunk c of run m starts at rank + 64 c.  Chunks 0 .. 62 are stored as
    // they are; everything from chunk 63 on becomes ONE tail chunk that starts at the lowest of their ranks
    s_out[w][lane] = make_uint2(lane == LPX_GROUP_CHUNKS - 1 ? 0xffffffffu : 0u, 0u);
    Coop<WAVE>::sync();
    uint32_t n_chunks = 0;
    for (uint32_t c0 = 0; c0 < n_runs; c0 += WAVE)
    {
        const uint32_t m = c0 + lane;
        const bool valid = m < n_runs;
        const uint32_t len = valid ? s_mpre[w][m + 1] - s_mpre[w][m] : 0u;
        const uint32_t nc = (len + 63) / 64;
        const uint32_t incl = lpx_wave_incl_scan_u32(nc);
        uint32_t pos = n_chunks + incl - nc;
        const uint32_t rank = valid ? s_mrank[w][m] : 0u;
        for (uint32_t c = 0; c < nc; ++c, ++pos)
        {
            const uint32_t cr = rank + 64 * c, cc = min(64u, len - 64 * c);
            if (pos < LPX_GROUP_CHUNKS - 1)
                s_out[w][pos] = make_uint2(cr, cc);
            else
                atomicMin(&s_out[w][LPX_GROUP_CHUNKS - 1].x, cr);
        }
        n_chunks += __builtin_amdgcn_readfirstlane(__shfl(incl, WAVE - 1, 64));
    }
    Coop<WAVE>::sync();
    if (lane == 0)
    {
        if (n_chunks >= LPX_GROUP_CHUNKS)
        {
            // [first rank of chunk 63, end of the last run): may be longer than 64 and may span pruned subtrees
            // (their nodes fail the distance test), the search loops over it
            const uint32_t first = s_out[w][LPX_GROUP_CHUNKS - 1].x;
            const uint32_t last_end = s_mrank[w][n_runs - 1] + (s_mpre[w][n_runs] - s_mpre[w][n_runs - 1]);
            s_out[w][LPX_GROUP_CHUNKS - 1] = make_uint2(first, last_end - first);
        }
        else
            s_out[w][LPX_GROUP_CHUNKS - 1] = make_uint2(0u, 0u);
    }
    Coop<WAVE>::sync();
    // Bounding box of every chunk (a search culls chunks against its query ball before it loads a candidate): lane c
    // answers for chunk c and unites the boxes of the aligned blocks of IX_SUB ranks the chunk overlaps (sub_box_kernel) --
    // ten 16-byte loads per lane, all in flight together.
    const uint2 mine = s_out[w][lane];
    const uint32_t stored = min(n_chunks, (uint32_t)LPX_GROUP_CHUNKS);
    float lo0 = 0.0f, lo1 = 0.0f, lo2 = 0.0f, hi0 = 0.0f, hi1 = 0.0f, hi2 = 0.0f;
    {
        const uint32_t span = mine.y > 64u ? 64u : mine.y;
        const bool has = lane < stored && span != 0u;
        const uint32_t j0 = mine.x / IX_SUB, j1 = has ? (mine.x + span - 1u) / IX_SUB : j0;
        float4 bl[IX_SUB_SPAN], bh[IX_SUB_SPAN];
#pragma unroll
        for (uint32_t u = 0; u < IX_SUB_SPAN; ++u)
        {
            const uint32_t j = has ? (j0 + u < j1 ? j0 + u : j1) : 0u;
            bl[u] = SB[2 * j];
            bh[u] = SB[2 * j + 1];
        }
        if (has)
        {
            lo0 = bl[0].x, lo1 = bl[0].y, lo2 = bl[0].z, hi0 = bh[0].x, hi1 = bh[0].y, hi2 = bh[0].z;
#pragma unroll
            for (uint32_t u = 1; u < IX_SUB_SPAN; ++u)
            {
                lo0 = fminf(lo0, bl[u].x), lo1 = fminf(lo1, bl[u].y), lo2 = fminf(lo2, bl[u].z);
                hi0 = fmaxf(hi0, bh[u].x), hi1 = fmaxf(hi1, bh[u].y), hi2 = fmaxf(hi2, bh[u].z);
            }
        }
    }
    if (mine.y > 64u)
    {
        lo0 = lo1 = lo2 = -3.0e38f;  // long tail chunk: never culled
        hi0 = hi1 = hi2 = 3.0e38f;
    }
    ChunkRec rec;
    rec.rank = mine.x;
    rec.count = mine.y;
    rec.lo[0] = lo0;
    rec.lo[1] = lo1;
    rec.lo[2] = lo2;
    rec.hi[0] = hi0;
    rec.hi[1] = hi1;
    rec.hi[2] = hi2;
    chunks[(size_t)gid * LPX_GROUP_CHUNKS + lane] = rec;
    if (!parent)
        return;
    // ---- the group's share of the connected components (kd_link_queries) ----
    {
        const uint32_t rank_end = g0 + nq;
        // chunks that begin below the group's last rank and whose exact box meets the group's widened box
        const bool wanted = lane < stored && rec.count != 0u && rec.rank < rank_end && rec.lo[0] <= bhi[0] &&
                            rec.hi[0] >= blo[0] && rec.lo[1] <= bhi[1] && rec.hi[1] >= blo[1] && rec.lo[2] <= bhi[2] &&
                            rec.hi[2] >= blo[2];
        const unsigned long long cmask = __ballot(wanted);
        kd_link_queries(PR, parent, rec, cmask, q.x, q.y, q.z, __float_as_uint(q.w), g0 + lane, active, rank_end, blo, bhi,
                        r2, lane);
        if (nq > (uint32_t)WAVE)
        {
            // the one or two queries beyond the 64th (a full bucket with its spine nodes): a second scan for them
            const bool more = lane + WAVE < nq;
            const Node q2 = PR[g0 + (more ? lane + WAVE : 0u)];
            kd_link_queries(PR, parent, rec, cmask, q2.x, q2.y, q2.z, __float_as_uint(q2.w), g0 + WAVE + lane, more,
                            rank_end, blo, bhi, r2, lane);
        }
    }
}
}  // namespace

// Components of the search path: from the chunk tables (kd_link_queries, inside nb_index_kernel) or from the
// clique-cell grid (lpx_grid_components).  Both are exact; which is cheaper depends on the cloud.  Measured on MI355X
// (alone on the device, per launch chain): 64 KITTI frames (53k obstacle points each, ~20k occupied cells) grid 1.77 ms,
// chunk tables 2.45 ms -- a group's table covers its whole box widened by the radius, ~3000 candidates for 52 queries,
// and all of them are tested -- and 1990 against 1615 Mpts/s with twenty chains in flight; 32 frames of BASELINE's
// 1M-point box cloud (318k obstacle points on dense surfaces: many cells, many cell pairs) grid 8.3 ms, chunk tables
// 5.4 ms, 1134 against 1379 Mpts/s.  Hence by frame size.  LPX_CC=grid / chunks forces one (development build).
bool lpx_cc_from_chunks(uint32_t m_max)
{
    static const char *e = LPX_KNOB("LPX_CC");
    if (e)
        return strcmp(e, "chunks") == 0;
    return m_max >= 400000u;
}


#ifdef LPX_DEV_KNOBS
// LPX_DUMMY_LAUNCHES=N (development build): N empty launches of one wavefront per frame in every chain -- what a launch
// costs a loaded device apart from its work (docs/experiments.md, round 5)
__global__ void noop_kernel(uint32_t *sink)
{
    if (sink && threadIdx.x == 0xffffffffu)
        *sink = 0;
}
#endif

int lpx_group_index(lpx_ctx *ctx, uint32_t m_max, float r2, bool clear_grid)
{
    if (m_max == 0)
        return LPX_OK;
#ifdef LPX_DEV_KNOBS
    {
        static const int dummies = LPX_KNOB("LPX_DUMMY_LAUNCHES") ? atoi(LPX_KNOB("LPX_DUMMY_LAUNCHES")) : 0;
        for (int i = 0; i < dummies; ++i)
            hipLaunchKernelGGL(noop_kernel, dim3(1, 1, ctx->cur_b), dim3(64), 0, ctx->stream, (uint32_t *)nullptr);
    }
#endif
    StageTimer tm(ctx, ST_NB_FILL);
    const float rr = sqrtf(r2) * 1.0001f + 1.0e-3f;
    // Group size: 64 nodes, or 32 when the searches of the previous call on this context tested many candidates per
    // hit (dense surfaces: BASELINE's synthetic box clouds test 200 candidates per neighbour with 64-node groups and
    // run 36 % faster with 32; KITTI frames test 12 and lose 4 %).  Hysteresis between 20 and 40 candidates per hit.
    // The choice changes the work, never a result.  LPX_IX_BUCKET fixes it.
    static const uint32_t env_bucket = LPX_KNOB("LPX_IX_BUCKET") ? (uint32_t)atoi(LPX_KNOB("LPX_IX_BUCKET")) : 0u;
    uint32_t bucket = ctx->ix_bucket;
    if (env_bucket >= 32 && env_bucket <= 64)
        bucket = env_bucket;
    else if (ctx->h_search)
    {
        const uint64_t hits = ctx->h_search[0], cand = ctx->h_search[2], exps = ctx->h_search[3] & 0xffffffffull;
        if (hits > 0 && exps >= 1000)
        {
            const uint64_t per_hit = cand / hits;
            if (per_hit > 40)
                bucket = 32;
            else if (per_hit < 20)
                bucket = 64;
        }
    }
    ctx->ix_bucket = bucket;
    uint32_t dmax = 0;
    while ((m_max >> dmax) > bucket)
        ++dmax;
    const uint32_t groups = (2u << dmax) - 1;
    if (sizeof(ChunkRec) * LPX_GROUP_CHUNKS * (size_t)groups > ctx->chunks.bytes)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "chunk table of %u groups does not fit the workspace", groups);
    static const uint32_t ix_spine = LPX_KNOB("LPX_IX_SPINE") ? (uint32_t)atoi(LPX_KNOB("LPX_IX_SPINE")) : 2u;
    // (the block boxes live in the kd build's stop-list scratch, which is free from here to the next build of this slot)
    float4 *const sub_boxes = (float4 *)ctx->lpos.p;
    if (2 * sizeof(float4) * ((size_t)m_max / IX_SUB + 1) > ctx->lpos.bytes)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "block boxes of %u nodes do not fit their scratch", m_max);
    hipLaunchKernelGGL(sub_box_kernel, dim3((m_max / IX_SUB + 256) / 256, 1, ctx->cur_b), dim3(256), 0, ctx->stream,
                       (const Node *)ctx->nodes_pre.p, (const FrameState *)ctx->frame.p, sub_boxes, ctx->fs_tag);
    hipLaunchKernelGGL(nb_index_kernel, dim3((groups + NB_WAVES - 1) / NB_WAVES, 1, ctx->cur_b), dim3(NB_THREADS), 0,
                       ctx->stream, (const Node *)ctx->nodes_pre.p, (const FrameState *)ctx->frame.p, rr,
                       (ChunkRec *)ctx->chunks.p, (float4 *)ctx->grp_of.p, ix_spine, bucket,
                       lpx_cc_from_chunks(m_max) ? (uint32_t *)ctx->parent.p : (uint32_t *)nullptr, r2,
                       clear_grid ? (unsigned long long *)ctx->cell_key.p : (unsigned long long *)nullptr,
                       (uint32_t *)ctx->cell_parent.p, (uint32_t *)ctx->cell_rep.p, ctx->cell_cap,
                       (FrameState *)ctx->frame.p, (const float4 *)sub_boxes, lpx_fv(ctx));
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
