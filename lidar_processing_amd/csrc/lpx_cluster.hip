// lpx_cluster.hip -- Fast Euclidean Clustering with the reference's exact, order-dependent semantics.
//
// Replaces Clusterer::cluster (reference src/clustering.cpp:47-125).
//
// The reference is a greedy seeded BFS: seeds in input order (:70), FIFO queue (:80-88), neighbours
// in kd-tree pre-order (:90-92), a touched point within (1-q)*d of the expanded point is absorbed
// (removed, never expanded), otherwise queued (:102-109); groups whose TOUCH COUNT (duplicates
// included, :99-100) is outside [min,max] are relabelled INVALID (:113-119).  The partition depends
// on that order, so it cannot be produced by a union-find alone.  What is parallel:
//   * every radius-neighbour list, already in reference emission order   (lpx_lists.hip)
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
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace
{
enum : uint8_t
{
    PT_FRESH = 0,
    PT_QUEUED = 1,
    PT_REMOVED = 2
};

// One workgroup per radix-sort tile (LPX_SORT_TILE points, eight per thread): the roots are the keys of the component
// sort that follows, and the tile's histogram of their lowest byte is left where its first pass expects it (hist,
// block-major; null: not wanted) -- that pass needs no histogram launch of its own.
__global__ __launch_bounds__(256) void flatten_kernel(uint32_t *parent, const FrameState *__restrict__ frame,
                                                      uint32_t *__restrict__ root, uint32_t *__restrict__ iota,
                                                      uint8_t *__restrict__ state, uint32_t *__restrict__ valid,
                                                      uint32_t *__restrict__ cc_lo, uint32_t *__restrict__ cc_hi,
                                                      uint32_t *__restrict__ hist, LpxListStat *hstat, uint32_t seq,
                                                      size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    __shared__ uint32_t h[256];
    parent = lpx_slot(parent, fs);
    frame = lpx_slot(frame, fs);
    root = lpx_slot(root, fs);
    iota = lpx_slot(iota, fs);
    state = lpx_slot(state, fs);
    valid = lpx_slot(valid, fs);
    cc_lo = lpx_slot(cc_lo, fs);
    cc_hi = lpx_slot(cc_hi, fs);
    hist = lpx_slot(hist, fs);
    const uint32_t tid = threadIdx.x, M = frame->n_obstacle;
    const bool forest = !frame->status;  // (a frame whose lists did not fit has no forest: every point its own root)
    if (hstat && lpx_blk.x == 0 && tid == 0)
    {
        // list path: what this frame asked of the list workspace, for the host's sizing of the NEXT call (pinned memory)
        LpxListStat *o = hstat + lpx_blk.z;
        uint64_t mx = 0;
        for (uint32_t i = 0; i < LPX_RS_STRIPES; ++i)
            mx = frame->rs_stripe[i].v > mx ? frame->rs_stripe[i].v : mx;
        o->nb_total = frame->nb_total;
        o->stripe_max = mx;
        unsigned long long ent = frame->nb_entries;
        for (uint32_t i = 0; i < LPX_RS_STRIPES; ++i)
            ent += frame->ent_stripe[i].v;
        o->entries = ent;
        o->status = frame->status;
        o->n_obstacle = M;
        __threadfence_system();
        __hip_atomic_store(&o->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (hist)
    {
        h[tid] = 0;
        __syncthreads();
    }
#pragma unroll 2
    for (uint32_t r = 0; r < LPX_SORT_TILE / 256u; ++r)
    {
        const uint32_t i = lpx_blk.x * LPX_SORT_TILE + r * 256u + tid;
        if (i >= M)
            continue;
        uint32_t x = i;
        while (forest)
        {
            const uint32_t p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p == x)
                break;
            x = p;
        }
        root[i] = x;
        if (iota)
            iota[i] = i;
        state[i] = PT_FRESH;
        valid[i] = 0;
        cc_lo[i] = 0;
        cc_hi[i] = 0;
        if (hist)
            atomicAdd(&h[x & 255u], 1u);
    }
    if (hist)
    {
        __syncthreads();
        hist[lpx_blk.x * 256u + tid] = h[tid];
    }
}

// sorted by root (stable): members of a component are contiguous, ascending original index.
// A set of ONE point needs no sequencer: the greedy loop (src/clustering.cpp:69-124) seeds it, its radius search
// returns the point itself and nothing else, the point is touched once and absorbed (distance 0), so the group has one
// touch -- decided right here.  KITTI frames hold several hundred such sets; each used to cost a sequencer a ticket,
// three dependent loads and a search.
__global__ void cc_ranges_kernel(const uint32_t *__restrict__ sroot, const uint32_t *__restrict__ members,
                                 FrameState *frame, uint32_t *__restrict__ cc_lo, uint32_t *__restrict__ cc_hi,
                                 uint32_t *__restrict__ roots, int32_t *__restrict__ seed_of,
                                 uint32_t *__restrict__ valid, uint32_t min_size, uint32_t max_size, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    sroot = lpx_slot(sroot, fs);
    members = lpx_slot(members, fs);
    frame = lpx_slot(frame, fs);
    cc_lo = lpx_slot(cc_lo, fs);
    cc_hi = lpx_slot(cc_hi, fs);
    roots = lpx_slot(roots, fs);
    seed_of = lpx_slot(seed_of, fs);
    valid = lpx_slot(valid, fs);
    const uint32_t p = lpx_blk.x * blockDim.x + threadIdx.x;
    const uint32_t M = frame->n_obstacle;
    bool single = false, opens = false;
    uint32_t r = 0;
    if (p < M)
    {
        r = sroot[p];
        const bool first = p == 0 || sroot[p - 1] != r, last = p + 1 == M || sroot[p + 1] != r;
        single = first && last;
        if (single)
        {
            const uint32_t i = members[p];
            seed_of[i] = (int32_t)i;
            valid[i] = (1u >= min_size && 1u <= max_size) ? 1u : 0u;  // :113
        }
        else
        {
            opens = first;
            if (first)
                cc_lo[r] = p;
            if (last)
                cc_hi[r] = p + 1;
        }
    }
    // work list of the replay: one bump of the frame's counter per WAVEFRONT (a 5M-point frame has 30 000 sets with a
    // sequencer: one same-address atomic each was 170 us of serialisation at L2; the order of the list is free)
    const unsigned long long om = __ballot(opens);
    if (om)
    {
        uint32_t base = 0;
        if ((threadIdx.x % WAVE) == (uint32_t)(__ffsll((long long)om) - 1))
            base = atomicAdd(&frame->n_roots, (uint32_t)__popcll(om));
        base = __shfl(base, __ffsll((long long)om) - 1, WAVE);
        if (opens)
            roots[base + __popcll(om & lpx_lanemask_lt())] = r;
    }
    // the statistics keep counting the radius searches the reference makes: one per single-point set, one neighbour each
    // (ONE bump per wavefront that holds a single-point set: n_single; relabel_kernel adds it to n_expansions and
    // replay_entries at the end of the call -- three same-line atomics per wavefront were most of this kernel's 163 us on a
    // 5M-point frame)
    const uint32_t ns = (uint32_t)__popcll(__ballot(single));
    if (ns && (threadIdx.x % WAVE) == 0)
        atomicAdd(&frame->n_single, ns);
}

struct ReplayParams
{
    double thr;   // (1-q)^2 * d^2 in double, src/clustering.cpp:66-67
    float thr_f;  // largest float <= thr: for a float d, (double)d <= thr  <=>  d <= thr_f
    float r2;     // distance_squared (list membership)
    uint32_t min_size, max_size;
    uint32_t dbg_rows;  // rows of the tools-only statistics buffer (0: none)
};

constexpr int RP_DEPTH = 4;    // neighbour lists kept in flight ahead of the expansion being processed
constexpr int RP_RING = 4096;  // queue entries mirrored in LDS by replay_lds_kernel
constexpr int RP_STAGE = 512;  // first touches of a queue window held in LDS until their seed_of stores go out
typedef __attribute__((address_space(3))) uint32_t lds_u32;  // LDS words by their own address space: ds_* instructions, not flat_*

// The list replay: one sequencer wavefront per workgroup, components from a work list.  Where the 2-bit point states
// (bit 0 queued, bit 1 removed) live is the template argument:
//   RP_STATE_LDS    a bitmap over the WHOLE index range in LDS (M <= 393 216 points: every real frame).  No global round
//                   trip in the dependent chain of a step; list chunks are loaded four at a time; offsets / lengths are
//                   fetched with the queue window.
//   RP_STATE_HBM    one byte per point in HBM (`gstate`, zeroed by flatten_kernel): a dependent global round trip and a
//                   block-level fence per list step.  Rounds 2-5 ran every larger cloud this way: a 5M-point frame's
//                   replay lasted as long as its largest component's chain -- 3 964 expansions at ~2 500 cycles each, 4.3
//                   of the frame's 13.9 ms (tools/r6_replay5m.py).
//   RP_STATE_LOCAL  (round 6) a bitmap over the COMPONENT's points in LDS, indexed by a point's position in the member
//                   list (`pos_of`, written by pos_of_kernel).  A component owns its points, so nothing else is ever
//                   asked; 65 536 members fit 16 KiB.  The position of a list entry is one more gather, but an
//                   INDEPENDENT one: it is requested two expansions ahead, right behind the list that is requested four
//                   ahead, and the chain of a step is LDS only -- as in RP_STATE_LDS.  Components beyond the capacity
//                   take the HBM form inside the same kernel.
enum
{
    RP_STATE_LDS = 0,
    RP_STATE_HBM = 1,
    RP_STATE_LOCAL = 2
};
constexpr uint32_t RP_LOCAL_CAP = 65536;  // members of a component whose states fit the local bitmap

struct ReplayArrays
{
    const uint32_t *members, *nb_off, *nb_len, *nb_idx, *pos_of;
    const float *OX, *OY, *OZ;
    int32_t *seed_of;
    uint32_t *queue, *valid;
    uint8_t *gstate;
};

template <int MODE>
__device__ __forceinline__ void replay_component(const ReplayArrays &A, const ReplayParams &prm, uint32_t lo, uint32_t hi,
                                                 lds_u32 *sbits, lds_u32 *ring, lds_u32 *stg, uint32_t lane,
                                                 unsigned long long &st_entries, uint32_t &st_exp, uint32_t &cc_windows,
                                                 uint32_t &cc_seeds, unsigned long long *ph)
{
    // GLOBAL STORES AND THE LOAD PIPELINE (round 6).  vmcnt counts loads and stores alike, and the two complete out of
    // order with respect to each other: while a store is pending, a wait for ANY load is `s_waitcnt vmcnt(0)` -- the
    // compiler has no other choice, and it decides per loop, not per trip.  Rounds 1-5 stored seed_of[k] and queue[qi]
    // right where a list step produced them; every wait of the sequencer was therefore a wait for everything, the lists
    // "in flight ahead" included: one full memory round trip per expansion, whatever the pipeline depth (replay_lds_kernel:
    // 26 of 33 waits were vmcnt(0)).  Now the loop that walks the expansions of a window issues NO global store (LDS and
    // LOCAL forms; the HBM form's state bytes are what it is the fallback for): first touches are staged in LDS (`stg`),
    // pushes go to the ring only, and both reach memory between windows (stores_out), where the next window's gathers
    // wait for everything anyway.  A stage or ring that fills up inside a window drains at once, with an explicit wait
    // behind it, so that the loop's other trips stay store-free for the compiler's counting.
    // tools only (ph != nullptr): cycles of this component's sequencer by phase -- 0 seed search, 1 window set-up (ring,
    // gathers, selection), 2 the wait for the first lists / positions of a window, 3 applying the lists
#define RP_LAP(i)                                                                                                     \
    do                                                                                                                \
    {                                                                                                                 \
        if (ph)                                                                                                       \
        {                                                                                                             \
            const unsigned long long n_ = __builtin_amdgcn_s_memtime();                                               \
            ph[i] += n_ - ph_t;                                                                                       \
            ph_t = n_;                                                                                                \
        }                                                                                                             \
    } while (0)
    unsigned long long ph_t = ph ? __builtin_amdgcn_s_memtime() : 0ull;
    // state id of a point: its index (LDS / HBM forms) or its position in the component (LOCAL)
#define ST_GET(k) (MODE == RP_STATE_HBM ? (uint32_t)A.gstate[k] : ((sbits[(k) >> 4] >> (((k) & 15u) * 2u)) & 3u))
#define ST_OR(k, v)                                                                                                   \
    do                                                                                                                \
    {                                                                                                                 \
        if (MODE == RP_STATE_HBM)                                                                                     \
            A.gstate[k] = (uint8_t)(A.gstate[k] | (v));                                                               \
        else                                                                                                          \
            __hip_atomic_fetch_or(&sbits[(k) >> 4], (uint32_t)(v) << (((k) & 15u) * 2u), __ATOMIC_RELAXED,                 \
                                  __HIP_MEMORY_SCOPE_WORKGROUP);                                         \
    } while (0)
    // lists in flight ahead of the expansion being processed (LOCAL: the positions half as far ahead).  The large clouds
    // that take the LOCAL / HBM forms read their lists from HBM or the Infinity Cache (~1 us): eight ahead
    constexpr int DEPTH = MODE == RP_STATE_LDS ? RP_DEPTH : 2 * RP_DEPTH;
    // chunks of 64 list words a slot of the pipeline holds: lists of up to 64 CH entries are fetched ahead in full (the
    // reference's frames at d = 0.5 m: ~140 entries on average, several hundred where the scene is dense), longer ones
    // take their tail straight from memory (tail loop below)
    constexpr int CH = MODE == RP_STATE_LDS ? 4 : 2;
    // UNCOND: every load of the pipeline is issued whether or not there is something to fetch (below).  The LDS form keeps
    // its loads conditional: its lists come from the L2 / Infinity Cache of a frame that is alone on the device, an
    // expansion costs ~650 cycles there -- the apply itself -- and the dummy loads of the unconditional form only add
    // instructions (a 123k-point frame alone: 1.70 -> 1.77 ms with them)
    constexpr bool UNCOND = MODE != RP_STATE_LDS;
    // ... and its stores where a list step produces them (DEFER off): there the staging below is one ballot, one LDS
    // write, one LDS read and a loop more per step of an instruction-bound sequencer (the densest of the three committed
    // frames alone: 2.73 -> 3.11 ms with it)
    constexpr bool DEFER = MODE != RP_STATE_LDS;
    const unsigned long long lt = lpx_lanemask_lt();
    if (MODE == RP_STATE_LOCAL)
    {
        // this component's bitmap (the work list hands a sequencer one component after the other)
        for (uint32_t i = lane; i < (hi - lo + 15u) / 16u; i += WAVE)
            sbits[i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t *q = A.queue + lo;
    uint32_t cursor = lo;
    for (;;)
    {
        uint32_t seed = 0xffffffffu, seed_sid = 0;
        uint32_t sc = 0;      // first touches staged (wave-uniform)
        uint32_t q_done = 1;  // queue entries [0, q_done) are in the global queue (the seed is)
        while (cursor < hi)
        {
            const uint32_t p = cursor + lane;
            const uint32_t cand = (p < hi) ? A.members[p] : 0u;
            const uint32_t sid = MODE == RP_STATE_LOCAL ? ((p < hi) ? p - lo : 0u) : cand;
            const bool ok = (p < hi) && !(ST_GET(sid) & 2u);
            const unsigned long long m = __ballot(ok);
            if (m)
            {
                const int f = __ffsll((long long)m) - 1;
                seed = __shfl(cand, f, 64);
                seed_sid = __shfl(sid, f, 64);
                cursor += f + 1;
                break;
            }
            cursor += WAVE;
        }
        RP_LAP(0);
        if (seed == 0xffffffffu)
            break;
        uint32_t qh = 0, qt = 1;
        unsigned long long touches = 0;
        if (lane == 0)
        {
            q[0] = seed;
            ring[0] = seed;
            ST_OR(seed_sid, 1u);
            A.seed_of[seed] = (int32_t)seed;  // queued before it is ever touched
        }
        if (MODE == RP_STATE_HBM)
            __threadfence_block();
        auto stores_out = [&]() {
            if (!DEFER)
                return;
            for (uint32_t b0 = 0; b0 < sc; b0 += WAVE)
                if (b0 + lane < sc)
                    A.seed_of[stg[b0 + lane]] = (int32_t)seed;  // first touch; later touches carry the same seed
            sc = 0;
            for (uint32_t j0 = q_done; j0 < qt; j0 += WAVE)
                if (j0 + lane < qt)
                    q[j0 + lane] = ring[(j0 + lane) % RP_RING];
            q_done = qt;
        };
        // The queue is consumed in windows of up to 64 pops.  Which of a window's candidates the reference
        // expands is decided inside the window: candidate c is skipped iff it is already removed, or an
        // EXPANDED earlier candidate h of the window holds it within the absorb radius (c is then in h's
        // list with dist <= thr and gets removed before its turn).  That is a greedy pass over at most 64
        // points in registers, after which the exact sequence of expansions of the window is known and
        // the next list is always in flight while the current one is processed.
        while (qh < qt)
        {
            const uint32_t wb = qh;
            const uint32_t wn = min((uint32_t)WAVE, qt - qh);
            const bool inw = lane < wn;
            uint32_t wcand;
            if (qt - qh <= (uint32_t)RP_RING)
                wcand = inw ? ring[(wb + lane) % RP_RING] : 0u;  // LDS: in order with the pushes of this wave
            else
            {
                // (more entries pending than the ring holds: the window comes from the global queue, which stores_out
                // keeps complete before a ring slot is reused)
                if (DEFER)
                {
                    stores_out();
                    __builtin_amdgcn_s_waitcnt(0);
                }
                __threadfence_block();  // queue entries pushed by other lanes
                wcand = inw ? q[wb + lane] : 0u;
            }
            // LOCAL: the candidates' positions in the component -- a gather like the five below, but the states are
            // asked first, so this one is waited for
            uint32_t wsid, woff, wlen;
            float wx, wy, wz;
            bool alive;
            if (MODE == RP_STATE_LDS)
            {
                wsid = wcand;
                alive = inw && !(ST_GET(wsid) & 2u);
                // unconditional loads (index 0 for idle lanes): the five gathers are issued back to back
                const uint32_t ci = alive ? wcand : 0u;
                woff = A.nb_off[ci];
                wlen = alive ? A.nb_len[ci] : 0u;
                wx = A.OX[ci], wy = A.OY[ci], wz = A.OZ[ci];
            }
            else
            {
                // the states are a global round trip away (HBM) or behind one (LOCAL: the position): the five gathers go
                // out WITH it, for every candidate of the window, instead of waiting to know which are alive -- one
                // dependent trip less per window (a window holds ~9 expansions on a 5M-point cloud: four trips were 45 %
                // of the largest component's chain)
                const uint32_t ci = wcand;  // (0 for idle lanes)
                const uint32_t wpos = MODE == RP_STATE_LOCAL ? A.pos_of[ci] : 0u;
                const uint32_t wst = MODE == RP_STATE_HBM ? (uint32_t)A.gstate[ci] : 0u;
                woff = A.nb_off[ci];
                const uint32_t wlen_raw = A.nb_len[ci];
                wx = A.OX[ci], wy = A.OY[ci], wz = A.OZ[ci];
                wsid = MODE == RP_STATE_LOCAL ? (inw ? wpos - lo : 0u) : wcand;
                alive = inw && !((MODE == RP_STATE_HBM ? wst : ST_GET(wsid)) & 2u);
                wlen = alive ? wlen_raw : 0u;
            }
            ++cc_windows;
            unsigned long long am = __ballot(alive), em = 0;
            while (am)
            {
                const int h = __ffsll((long long)am) - 1;
                em |= 1ull << h;
                const float hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), h));
                const float hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), h));
                const float hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), h));
                const float d0 = hx - wx, d1 = hy - wy, d2 = hz - wz;
                const float dist = d0 * d0 + (d1 * d1 + (d2 * d2 + 0.0f));  // dist_sqr(points_[h], node)
                const bool conflict = alive && dist <= prm.r2 && dist <= prm.thr_f;
                am &= ~__ballot(conflict);
                am &= ~(1ull << h);
            }
            qh = wb + wn;
            RP_LAP(1);
            if (!em)
                continue;
            // software pipeline, DEPTH lists in flight: slot s holds the first four chunks of the
            // expansion that will be processed DEPTH steps after the one that last used the slot.
            // LOCAL: a second stage DEPTH / 2 steps ahead gathers the positions of the entries of the list
            // that has arrived by then (P4: state ids; 0xffffffff past the end of the list).
            // (K4 / P4 hold what the loads return, untouched: a select on a loaded value is a use, and a use is a wait --
            // lanes past the end of a list are masked where the words are applied, from the list's length)
            uint32_t K4[DEPTH][CH];  // words index | absorb << 31
            uint32_t P4[DEPTH / 2][CH];
            uint32_t KC[DEPTH];      // entries of the list in the slot (wave-uniform)
            unsigned long long lm = em;  // expansions whose list still has to be requested
            unsigned long long pm = em;  // ... whose positions still have to be requested (LOCAL)
            // EVERY load of the pipeline is issued unconditionally (an address of 0 where there is nothing to fetch, the
            // value discarded): a load behind a branch -- `if (lm)`, or the per-lane `t < cg` the compiler turns into one --
            // leaves the number of loads YOUNGER than the one being waited for unknown, and the only safe wait is then
            // vmcnt(0) again.
#pragma unroll
            for (int sl = 0; sl < DEPTH; ++sl)
            {
                const int g = lm ? __ffsll((long long)lm) - 1 : 0;
                const uint32_t og = (uint32_t)__builtin_amdgcn_readlane((int)woff, g);
                const uint32_t cg = lm ? (uint32_t)__builtin_amdgcn_readlane((int)wlen, g) : 0u;
                lm &= lm - 1;  // (0 stays 0)
                KC[sl] = cg;
#pragma unroll
                for (int c = 0; c < CH; ++c)
                {
                    const uint32_t t = c * WAVE + lane;
                    if (UNCOND)
                        K4[sl][c] = A.nb_idx[t < cg ? og + t : 0u];
                    else if (t < cg)
                        K4[sl][c] = A.nb_idx[og + t];
                }
            }
            if (MODE == RP_STATE_LOCAL)
            {
#pragma unroll
                for (int sp = 0; sp < DEPTH / 2; ++sp)
                {
                    pm &= pm - 1;
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        P4[sp][c] = A.pos_of[(uint32_t)(c * WAVE) + lane < KC[sp] ? K4[sp][c] & 0x7fffffffu : 0u];
                }
            }
            if (ph)
            {
                // (tools: make the wait for the first list -- and its positions -- visible as a phase of its own)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RP_LAP(2);
            }
            while (em)
            {
#pragma unroll
                for (int sl = 0; sl < DEPTH; ++sl)
                {
                    if (!em)
                        break;
                    const int f = __ffsll((long long)em) - 1;
                    em &= em - 1;
                    const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)woff, f);
                    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)wlen, f);
                    st_entries += cnt;
                    ++st_exp;
                    uint32_t kk[CH], pp[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                    {
                        const bool in = (uint32_t)(c * WAVE) + lane < cnt;
                        kk[c] = in ? K4[sl][c] : 0xffffffffu;
                        pp[c] = MODE == RP_STATE_LOCAL ? P4[sl % (DEPTH / 2)][c] - lo : 0u;
                    }
                    if (MODE == RP_STATE_LOCAL)
                    {
                        // the positions of the expansion DEPTH / 2 steps ahead: its list is in slot sl + DEPTH / 2
                        // (past the window's last expansion the slot holds 0xffffffff: dummy loads of word 0)
                        pm &= pm - 1;
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                        {
                            const uint32_t kn = K4[(sl + DEPTH / 2) % DEPTH][c];
                            P4[sl % (DEPTH / 2)][c] =
                                A.pos_of[(uint32_t)(c * WAVE) + lane < KC[(sl + DEPTH / 2) % DEPTH] ? kn & 0x7fffffffu : 0u];
                        }
                    }
                    {
                        // refill the slot with the list of the expansion DEPTH steps ahead (none left: dummy loads)
                        const int g = lm ? __ffsll((long long)lm) - 1 : 0;
                        const uint32_t og = (uint32_t)__builtin_amdgcn_readlane((int)woff, g);
                        const uint32_t cg = lm ? (uint32_t)__builtin_amdgcn_readlane((int)wlen, g) : 0u;
                        lm &= lm - 1;
                        KC[sl] = cg;
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                        {
                            const uint32_t t = c * WAVE + lane;
                            if (UNCOND)
                                K4[sl][c] = A.nb_idx[t < cg ? og + t : 0u];
                            else if (t < cg)
                                K4[sl][c] = A.nb_idx[og + t];
                        }
                    }
                    // one step of 64 list words in reference order (src/clustering.cpp:92-110): touch, absorb or queue.
                    // LDS only, but for the rare drain of a full stage / ring.
                    auto apply_step = [&](uint32_t kw, uint32_t pw) {
                        const bool in = kw != 0xffffffffu;
                        const uint32_t k = in ? (kw & 0x7fffffffu) : 0u;
                        const uint32_t sid = MODE == RP_STATE_LOCAL ? (in ? pw : 0u) : k;
                        const uint32_t sk = in ? ST_GET(sid) : 2u;
                        const bool vis = in && !(sk & 2u);
                        touches += __popcll(__ballot(vis));  // (an early way out for steps without a visible entry: 1.72 -> 1.75 ms)
                        const bool absorb = vis && (kw >> 31);
                        const bool push = vis && !absorb && sk == 0u;
                        const unsigned long long pmask = __ballot(push);
                        const bool first = vis && sk == 0u;
                        if (DEFER)
                        {
                            const unsigned long long fmask = __ballot(first);
                            if (first)
                                stg[sc + __popcll(fmask & lt)] = k;  // its seed_of store goes out with the window's
                            sc += __popcll(fmask);
                        }
                        else if (first)
                            A.seed_of[k] = (int32_t)seed;  // first touch; later touches carry the same seed
                        if (absorb)
                            ST_OR(sid, 2u);
                        if (push)
                        {
                            const uint32_t qi = qt + __popcll(pmask & lt);
                            if (!DEFER)
                                q[qi] = k;
                            ring[qi % RP_RING] = k;  // (DEFER: the global queue follows in stores_out)
                            ST_OR(sid, 1u);
                        }
                        qt += __popcll(pmask);
                        if (MODE == RP_STATE_HBM)
                            __threadfence_block();  // the next step reads states written by other lanes
                        if (DEFER && (sc > (uint32_t)(RP_STAGE - WAVE) || qt - q_done > (uint32_t)(RP_RING - WAVE)))
                        {
                            stores_out();                    // rare: drained on the spot, so that the other trips of
                            __builtin_amdgcn_s_waitcnt(0);   // this loop have no store pending
                        }
                    };
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                    {
                        if ((uint32_t)(c * WAVE) >= cnt)
                            break;
                        apply_step(kk[c], pp[c]);
                    }
                    // the tail of a list beyond what a slot holds: CH chunks at a time, waited for in place (LOCAL: and
                    // their positions behind them)
                    for (uint32_t t0 = (uint32_t)(CH * WAVE); t0 < cnt; t0 += (uint32_t)(CH * WAVE))
                    {
                        uint32_t kw[CH], pw[CH];
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                        {
                            const uint32_t t = t0 + c * WAVE + lane;
                            kw[c] = A.nb_idx[t < cnt ? o0 + t : 0u];
                        }
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                        {
                            const bool in = t0 + c * WAVE + lane < cnt;
                            pw[c] = MODE == RP_STATE_LOCAL ? A.pos_of[in ? kw[c] & 0x7fffffffu : 0u] - lo : 0u;
                            kw[c] = in ? kw[c] : 0xffffffffu;
                        }
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                        {
                            if (t0 + (uint32_t)(c * WAVE) >= cnt)
                                break;
                            apply_step(kw[c], pw[c]);
                        }
                    }
                }
            }
            stores_out();  // between windows: the next window's gathers wait for everything anyway
            RP_LAP(3);
        }
        if (lane == 0)
            A.valid[seed] = (touches >= prm.min_size && touches <= prm.max_size) ? 1u : 0u;
        ++cc_seeds;
    }
#undef ST_GET
#undef ST_OR
#undef RP_LAP
}

template <int MODE>
__global__ __launch_bounds__(WAVE) void replay_lds_kernel(const FrameState *__restrict__ frame,
                                                           const uint32_t *__restrict__ cc_lo,
                                                           const uint32_t *__restrict__ cc_hi,
                                                           const uint32_t *__restrict__ members,
                                                           const uint32_t *__restrict__ nb_off,
                                                           const uint32_t *__restrict__ nb_len,
                                                           const uint32_t *__restrict__ nb_idx,
                                                           const float *__restrict__ OX, const float *__restrict__ OY,
                                                           const float *__restrict__ OZ, int32_t *seed_of,
                                                           uint32_t *queue, uint32_t *valid, ReplayParams prm,
                                                           uint64_t cap, FrameState *fstate,
                                                           const uint32_t *__restrict__ roots,
                                                           uint32_t *__restrict__ dbg, uint8_t *gstate,
                                                           const uint32_t *__restrict__ pos_of, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<7>(fv.fs);
    extern __shared__ uint32_t sbits[];
    frame = lpx_slot(frame, fv.fs);
    fstate = lpx_slot(fstate, fv.fs);
    cc_lo = lpx_slot(cc_lo, fv.fs);
    cc_hi = lpx_slot(cc_hi, fv.fs);
    roots = lpx_slot(roots, fv.fs);
    ReplayArrays A;
    A.members = lpx_slot(members, fv.fs);
    A.nb_off = lpx_slot(nb_off, fv.fs);
    A.nb_len = lpx_slot(nb_len, fv.fs);
    A.nb_idx = lpx_slot(nb_idx, fv.fs_nb);
    A.pos_of = lpx_slot(pos_of, fv.fs);
    A.OX = lpx_slot(OX, fv.fs);
    A.OY = lpx_slot(OY, fv.fs);
    A.OZ = lpx_slot(OZ, fv.fs);
    A.seed_of = lpx_slot(seed_of, fv.fs);
    A.queue = lpx_slot(queue, fv.fs);
    A.valid = lpx_slot(valid, fv.fs);
    A.gstate = lpx_slot(gstate, fv.fs);
    const uint32_t lane = threadIdx.x;
    const uint32_t M = frame->n_obstacle;
    if (frame->nb_total > cap)
        return;
    const uint32_t n_roots = frame->n_roots;
    if (lpx_blk.x >= n_roots)
        return;
    // RP_STATE_LDS: the bitmap is zeroed once -- components own disjoint points, so the 2-bit states of one component
    // are never read by another.  RP_STATE_LOCAL: per component (replay_component).
    const uint32_t words = MODE == RP_STATE_LDS ? (M + 15) / 16 : (MODE == RP_STATE_LOCAL ? RP_LOCAL_CAP / 16u : 0u);
    for (uint32_t i = lane; MODE == RP_STATE_LDS && i < words; i += WAVE)
        sbits[i] = 0;
    // the most recent RP_RING queue entries are mirrored in LDS: a window of pops is then read without
    // waiting for the global queue stores (the global queue remains the fallback for long queues)
    lds_u32 *const lbits = (lds_u32 *)sbits;
    lds_u32 *const ring = lbits + ((words + 3) & ~3u);
    lds_u32 *const stg = ring + RP_RING;
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
        const unsigned long long cc_t0 = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
        const unsigned long long cc_e0 = st_entries;
        const uint32_t cc_x0 = st_exp;
        uint32_t cc_windows = 0, cc_seeds = 0;
        unsigned long long phase[4] = {0, 0, 0, 0};
        unsigned long long *const ph = (dbg && prm.dbg_rows == 65536u) ? phase : (unsigned long long *)nullptr;
        if (MODE == RP_STATE_LOCAL && hi - lo > RP_LOCAL_CAP)
            replay_component<RP_STATE_HBM>(A, prm, lo, hi, lbits, ring, stg, lane, st_entries, st_exp, cc_windows, cc_seeds,
                                           ph);
        else
            replay_component<MODE>(A, prm, lo, hi, lbits, ring, stg, lane, st_entries, st_exp, cc_windows, cc_seeds, ph);
        if (ph && lane == 0 && ticket < prm.dbg_rows)
            for (int i = 0; i < 4; ++i)  // (tools/r6_replay5m.py: a second block of rows behind the first 65536)
                dbg[(65536u + ticket) * 8 + i] = (uint32_t)(phase[i] >> 4);
        if (dbg && lane == 0 && ticket < prm.dbg_rows)
        {
            dbg[ticket * 8 + 0] = hi - lo;
            dbg[ticket * 8 + 1] = st_exp - cc_x0;
            dbg[ticket * 8 + 2] = (uint32_t)(st_entries - cc_e0);
            dbg[ticket * 8 + 3] = cc_windows;
            dbg[ticket * 8 + 4] = cc_seeds;
            dbg[ticket * 8 + 5] = (uint32_t)(__builtin_amdgcn_s_memtime() - cc_t0);
            dbg[ticket * 8 + 6] = 0;
        }
    }
    if (lane == 0 && st_exp)
    {
        atomicAdd((unsigned long long *)&fstate->replay_entries, st_entries);
        atomicAdd(&fstate->n_expansions, st_exp);
    }
}

// position of every point in the sorted member list (RP_STATE_LOCAL): pos_of[members[p]] = p
__global__ void pos_of_kernel(const uint32_t *__restrict__ members, const FrameState *__restrict__ frame,
                              uint32_t *__restrict__ pos_of, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    members = lpx_slot(members, fs);
    frame = lpx_slot(frame, fs);
    pos_of = lpx_slot(pos_of, fs);
    const uint32_t p = lpx_blk.x * blockDim.x + threadIdx.x;
    if (p < frame->n_obstacle)
        pos_of[members[p]] = p;
}

// ------------------------------------------------------------------------------------------------
// Expansion-driven replay: the same greedy loop, but the radius search of a point happens WHEN the loop expands
// it, against the candidate chunks of the point's kd group (lpx_chunks.hip: nb_index_kernel).  Nothing is
// materialised for the ~80 % of the points the reference never expands.
//
// This is the THROUGHPUT path (batch contexts): a single frame alone on the device is served faster by the list
// path, whose whole-device list build shortens the critical path.  So the kernel is organised for the least work
// and the most independent wavefronts, not for the latency of one component: every wavefront is a complete
// sequencer for its own component (work list, seeds, queue windows, in-window expansion selection -- see
// replay_lds_kernel -- search, ordered application); the eight wavefronts of a workgroup only share the LDS bitmap
// of point states (components own disjoint points).  A search is one load of the group's chunk table (prefetched one
// expansion ahead), a cull of the chunks against the query ball, and one independent 16-byte load per surviving
// candidate (eight chunks in flight); the reference's float distance expression is applied per candidate and the
// hits go straight into the apply step in candidate (= pre-order) order -- no list is ever stored, not even in LDS.
// (A cooperative variant -- four wavefronts searching the expansions of one window in parallel, hit lists in LDS --
// was built and measured: it halves the latency of one component but costs 40 % more LDS and three mostly idle
// wavefronts per workgroup; and splitting ONE search over several wavefronts is slower than not splitting it,
// because the per-search setup, not the candidate tests, dominates.)
// Point states: STATE_LDS keeps them as a 2-bit LDS bitmap (up to ~440k points); larger frames keep one byte per
// point in HBM (a second launch; each launch checks on the device whether a frame is its own).
// ------------------------------------------------------------------------------------------------
#ifndef LPX_RS_WAVES
#define LPX_RS_WAVES 8
#endif
constexpr int RS_WAVES = LPX_RS_WAVES;
constexpr int RS_THREADS = RS_WAVES * WAVE;
#ifndef LPX_RS_RING
#define LPX_RS_RING 256
#endif
constexpr int RS_RING = LPX_RS_RING;         // queue entries mirrored in LDS, per wavefront
#ifndef LPX_RS_BATCH
#define LPX_RS_BATCH 8
#endif
constexpr int RS_BATCH = LPX_RS_BATCH;                  // candidate chunks in flight per wavefront

constexpr int RS_HITS = 128;  // list words of one expansion gathered per wavefront before they are applied

struct RsShared  // fixed part of the LDS of replay_search_kernel (the bitmap follows)
{
    uint32_t ring[RS_WAVES][RS_RING];
    uint32_t hits[RS_WAVES][RS_HITS];
    uint32_t stage[RS_WAVES][RS_HITS];  // first touches of the expansion applied last, until their global stores go out
};
// (lds_u32: above, at the list replay)

typedef float4 KdNode;

// distance-tests one chunk of candidates against (qx, qy, qz): per lane the list word of its candidate
// (index | absorb << 31) or 0xffffffff
__device__ __forceinline__ uint32_t rs_test(const KdNode &nd, bool valid, float qx, float qy, float qz, float r2,
                                            float thr_f)
{
    const float a0 = qx - nd.x, a1 = qy - nd.y, a2 = qz - nd.z;
    const float da = a0 * a0 + (a1 * a1 + a2 * a2);  // src/kdtree.hpp:145-157 (the + 0.0f is the identity)
    const bool in = valid && da <= r2;               // :315 inclusive
    return in ? (__float_as_uint(nd.w) | (da <= thr_f ? 0x80000000u : 0u)) : 0xffffffffu;
}

// chunks of a kd group that can hold a neighbour of q (conservative box test): lane c answers for chunk c
__device__ __forceinline__ unsigned long long rs_cull(const ChunkRec &ch, float qx, float qy, float qz, float r2)
{
    const float ex = fmaxf(fmaxf(ch.lo[0] - qx, qx - ch.hi[0]), 0.0f);
    const float ey = fmaxf(fmaxf(ch.lo[1] - qy, qy - ch.hi[1]), 0.0f);
    const float ez = fmaxf(fmaxf(ch.lo[2] - qz, qz - ch.hi[2]), 0.0f);
    return __ballot(ch.count != 0u && (ex * ex + ey * ey + ez * ez) <= r2 * 1.0001f + 1.0e-6f);
}

// One batch of candidate chunks in flight: RS_BATCH independent 16-byte loads per lane.
struct RsBatch
{
    KdNode nd[RS_BATCH];
    uint32_t cnt[RS_BATCH], rk[RS_BATCH];
};

// requests the next (up to RS_BATCH) chunks named by the bits of `km` (lane c of `ch` describes chunk c), in order; a
// LONG chunk (the tail of a group with more chunks than lanes: more than 64 ranks) closes its batch, so that the rest of
// it is always the last thing rs_consume tests
__device__ __forceinline__ void rs_issue(RsBatch &b, const KdNode *__restrict__ PR, const ChunkRec &ch,
                                         unsigned long long &km, uint32_t lane)
{
    bool closed = false;
#pragma unroll
    for (int u = 0; u < RS_BATCH; ++u)
    {
        b.cnt[u] = 0u;
        b.rk[u] = 0u;
        if (km && !closed)
        {
            const int c = __ffsll((long long)km) - 1;
            km &= km - 1;
            b.rk[u] = (uint32_t)__builtin_amdgcn_readlane((int)ch.rank, c);
            b.cnt[u] = (uint32_t)__builtin_amdgcn_readlane((int)ch.count, c);
            closed = b.cnt[u] > 64u;
        }
        b.nd[u] = PR[lane < b.cnt[u] ? b.rk[u] + lane : 0u];  // unconditional: the loads of a batch go out back to back
    }
}

// distance-tests the chunks of a batch in order; SINK(word) gets the per-lane list words of every chunk step.
// TAIL8: the rest of a long chunk is fetched up to eight steps at a time -- the dense-scene kernels, where a sequencer's
// slowest expansions were thirty-odd DEPENDENT round trips through one tail (synth1m: 85 k cycles per expansion on the
// slowest wavefront, 92 % of them here).  After the batch itself, so that the eight rows take the registers of the batch
// (inside the loop over the batch they were 31 registers more: 129, one wavefront per SIMD less).
template <bool TAIL8, class Sink>
__device__ __forceinline__ void rs_consume(RsBatch &b, const KdNode *__restrict__ PR, float qx, float qy, float qz,
                                           float r2, float thr_f, uint32_t lane, unsigned long long &cand, Sink &&sink)
{
    uint32_t t_cnt = 0u, t_rk = 0u;
#pragma unroll
    for (int u = 0; u < RS_BATCH; ++u)
    {
        if (b.cnt[u] == 0u)
            break;
        cand += min(b.cnt[u], 64u);
        sink(rs_test(b.nd[u], lane < b.cnt[u], qx, qy, qz, r2, thr_f));
        if (b.cnt[u] > 64u)
        {
            t_cnt = b.cnt[u];
            t_rk = b.rk[u];
        }
    }
    // the long tail chunk of a group with more than 64 chunks: the rest of its ranks
    if (TAIL8)
    {
        for (uint32_t o = 64; o < t_cnt; o += RS_BATCH * 64)
        {
#pragma unroll
            for (int k = 0; k < RS_BATCH; ++k)  // (unconditional: independent loads, see the vmcnt rules)
                b.nd[k] = PR[o + k * 64 + lane < t_cnt ? t_rk + o + k * 64 + lane : 0u];
#pragma unroll
            for (int k = 0; k < RS_BATCH; ++k)
            {
                if (o + k * 64 >= t_cnt)
                    break;
                cand += min(t_cnt - (o + k * 64), 64u);
                sink(rs_test(b.nd[k], o + k * 64 + lane < t_cnt, qx, qy, qz, r2, thr_f));
            }
        }
    }
    else
        for (uint32_t o = 64; o < t_cnt; o += 64)
        {
            const bool v = o + lane < t_cnt;
            const KdNode n2 = PR[v ? t_rk + o + lane : 0u];
            cand += min(t_cnt - o, 64u);
            sink(rs_test(n2, v, qx, qy, qz, r2, thr_f));
        }
}

// LPX_RS_PROF (tools/replay_prof.py, a variant build): cycles of every sequencer by phase, one record per wavefront
// in the lpx_dbg_group_stats buffer: {total, seed scan, window set-up, chunk table + cull + issue, candidates (wait +
// test + gather), apply, expansions, windows}
#ifdef LPX_RS_PROF
#define RS_NOW() clock64()
#define RS_LAP(acc, t)                                                                                                \
    do                                                                                                                \
    {                                                                                                                 \
        const unsigned long long n_ = clock64();                                                                      \
        acc += n_ - (t);                                                                                              \
        (t) = n_;                                                                                                     \
    } while (0)
#else
#define RS_NOW() 0ull
#define RS_LAP(acc, t) ((void)0)
#endif

#ifdef LPX_RS_MINWAVES
#define RS_BOUNDS __launch_bounds__(RS_THREADS, LPX_RS_MINWAVES)
#else
#define RS_BOUNDS __launch_bounds__(RS_THREADS)
#endif
// REUSE: a chunk table already in registers is not requested again when the next expansion belongs to the same kd group.
// On dense surfaces (BASELINE's box clouds, 32-node groups: the host's ix_bucket) consecutive expansions share their
// group most of the time: +2.6 % on 1M-point frames; on KITTI frames they rarely do and the test only costs (-0.6 %).
template <bool STATE_LDS, bool REUSE>
__global__ RS_BOUNDS void replay_search_kernel(
    const FrameState *__restrict__ frame, const uint32_t *__restrict__ cc_lo, const uint32_t *__restrict__ cc_hi,
    const uint32_t *__restrict__ members, const KdNode *__restrict__ PR, const ChunkRec *__restrict__ chunks,
    const float4 *__restrict__ grp_of, const float *__restrict__ OX, const float *__restrict__ OY,
    const float *__restrict__ OZ, uint8_t *gstate, int32_t *seed_of, uint32_t *queue, uint32_t *valid,
    ReplayParams prm, FrameState *fstate, const uint32_t *__restrict__ roots, uint32_t m_lo, uint32_t m_hi,
    unsigned long long *prof, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<7>(fv.fs);
    extern __shared__ uint32_t smem[];
    RsShared &sh = *(RsShared *)smem;
    uint32_t *sbits = smem + sizeof(RsShared) / sizeof(uint32_t);
    FrameState *const fstate0 = fstate;  // slot 0 collects what the host reads back after the launch
    frame = lpx_slot(frame, fv.fs);
    fstate = lpx_slot(fstate, fv.fs);
    cc_lo = lpx_slot(cc_lo, fv.fs);
    cc_hi = lpx_slot(cc_hi, fv.fs);
    members = lpx_slot(members, fv.fs);
    PR = lpx_slot(PR, fv.fs);
    chunks = lpx_slot(chunks, fv.fs);
    grp_of = lpx_slot(grp_of, fv.fs);
    OX = lpx_slot(OX, fv.fs);
    OY = lpx_slot(OY, fv.fs);
    OZ = lpx_slot(OZ, fv.fs);
    gstate = lpx_slot(gstate, fv.fs);
    seed_of = lpx_slot(seed_of, fv.fs);
    queue = lpx_slot(queue, fv.fs);
    valid = lpx_slot(valid, fv.fs);
    roots = lpx_slot(roots, fv.fs);
    const uint32_t tid = threadIdx.x, w = tid / WAVE, lane = tid % WAVE;
    const uint32_t M = frame->n_obstacle;
    const uint32_t n_roots = frame->n_roots;
    if (STATE_LDS && lpx_blk.x == 0 && tid == 0)
        atomicMax(&fstate0->max_obstacle, M);
    // frames outside (m_lo, m_hi] obstacle points belong to the other launch (an empty frame to none)
    if (M <= m_lo || M > m_hi || lpx_blk.x * RS_WAVES >= n_roots)
        return;
    if (STATE_LDS)
    {
        // zeroed once: components own disjoint points, so one never reads the states of another
        const uint32_t words = (M + 15) / 16;
        for (uint32_t i = tid; i < words; i += RS_THREADS)
            sbits[i] = 0;
    }
    __syncthreads();  // the only workgroup barrier: from here on the wavefronts are independent sequencers
    const unsigned long long lt = lpx_lanemask_lt();
    const float r2 = prm.r2, thr_f = prm.thr_f;
    lds_u32 *const ring = (lds_u32 *)&sh.ring[w][0];
#define ST_GET(k) (STATE_LDS ? ((sbits[(k) >> 4] >> (((k) & 15u) * 2u)) & 3u) : (uint32_t)gstate[k])
#define ST_OR(k, v)                                                                                                   \
    do                                                                                                                \
    {                                                                                                                 \
        if (STATE_LDS)                                                                                                \
            atomicOr(&sbits[(k) >> 4], (uint32_t)(v) << (((k) & 15u) * 2u));                                         \
        else                                                                                                          \
            gstate[k] = (uint8_t)(gstate[k] | (v));                                                                   \
    } while (0)
    unsigned long long st_entries = 0, st_cand = 0;
    uint32_t st_exp = 0, st_win = 0;
    ChunkRec ch_held = {};            // the chunk table this wavefront read last ...
    uint32_t g_held = 0xffffffffu;    // ... and the kd group it belongs to
    [[maybe_unused]] unsigned long long pf_seed = 0, pf_win = 0, pf_tab = 0, pf_cand = 0, pf_apply = 0;
    [[maybe_unused]] unsigned long long pf_t = RS_NOW();
    [[maybe_unused]] const unsigned long long pf_t0 = pf_t;
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
        uint32_t *q = queue + lo;
        uint32_t cursor = lo;
        for (;;)
        {
            // next seed: first member (ascending index) that is not removed (src/clustering.cpp:70-75)
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
            RS_LAP(pf_seed, pf_t);
            if (seed == 0xffffffffu)
                break;
            uint32_t qh = 0, qt = 1;
            unsigned long long touches = 0;  // indices_.size(), duplicates included (:99-100)
            if (lane == 0)
            {
                q[0] = seed;
                ring[0] = seed;
                ST_OR(seed, 1u);
                seed_of[seed] = (int32_t)seed;  // queued before it is ever touched
            }
            if (!STATE_LDS)
                __threadfence_block();
            // Global stores and the load pipeline (round 6).  vmcnt counts loads AND stores, and the two complete out of
            // order with respect to each other: with a store pending, a wait for ANY load is a wait for everything
            // (s_waitcnt vmcnt(0) -- the compiler has no choice).  Rounds 2-5 applied the hits of expansion i, seed_of and
            // queue stores included, under the candidate loads of expansion i + 1: every state read of the apply then
            // waited for those loads first, which is where "56-59 % of a sequencer's time in apply" came from; and the
            // hit buffer was reached through a generic pointer (64 flat_load + 63 flat_store per kernel: flat operations
            // count on vmcnt AND lgkmcnt).  Now: the buffers are LDS-address-space pointers; an apply that runs under
            // prefetched loads (DEFER) touches LDS only and leaves its first touches in `stage`; their seed_of stores, and
            // the global copy of the queue entries pushed meanwhile (from the ring), go out once the prefetched candidates
            // have been consumed (stores_out).
            lds_u32 *const hb = (lds_u32 *)&sh.hits[w][0];
            lds_u32 *const stg = (lds_u32 *)&sh.stage[w][0];
            uint32_t hc = 0;        // words gathered (wave-uniform)
            uint32_t staged = 0;    // words of `stage` whose stores are still owed
            uint32_t q_done = 1;    // queue entries [0, q_done) are in the global queue (the seed is)
            auto stores_out = [&]() {
                for (uint32_t b0 = 0; b0 < staged; b0 += WAVE)
                {
                    const uint32_t wd = b0 + lane < staged ? stg[b0 + lane] : 0u;
                    if (wd >> 31)
                        seed_of[wd & 0x7fffffffu] = (int32_t)seed;  // first touch; later touches carry the same seed
                }
                staged = 0;
                for (uint32_t j0 = q_done; j0 < qt; j0 += WAVE)
                    if (j0 + lane < qt)
                        q[j0 + lane] = ring[(j0 + lane) % RS_RING];
                q_done = qt;
            };
            // one chunk step of list words in order (:92-110): touch, absorb or queue.  DEFER: no global store (see above);
            // returns the word to stage: index | first touch << 31
            auto apply = [&](uint32_t word, bool defer) -> uint32_t {
                const bool in = word != 0xffffffffu;
                const unsigned long long im = __ballot(in);
                if (!im)
                    return 0u;  // no candidate of this chunk is a neighbour
                st_entries += __popcll(im);
                const uint32_t k = in ? (word & 0x7fffffffu) : 0u;
                const uint32_t sk = in ? ST_GET(k) : 2u;
                const bool vis = in && !(sk & 2u);
                touches += __popcll(__ballot(vis));
                const bool absorb = vis && (word >> 31);
                const bool push = vis && !absorb && sk == 0u;
                const unsigned long long pm = __ballot(push);
                const bool first = vis && sk == 0u;
                if (first && !defer)
                    seed_of[k] = (int32_t)seed;  // first touch; later touches carry the same seed
                if (absorb)
                    ST_OR(k, 2u);
                if (push)
                {
                    const uint32_t qi = qt + __popcll(pm & lt);
                    if (!defer)
                        q[qi] = k;
                    ring[qi % RS_RING] = k;
                    ST_OR(k, 1u);
                }
                qt += __popcll(pm);
                if (!defer)
                    q_done = qt;
                if (!STATE_LDS)
                    __threadfence_block();  // the next step reads states written by other lanes
                return k | (first ? 0x80000000u : 0u);
            };
            // The list words of an expansion (the hits of its candidate chunks, in candidate = pre-order order) are
            // gathered in LDS first and applied 64 at a time: a chunk of 64 candidates holds ~5 hits on a KITTI frame,
            // and applying chunk by chunk paid one state read, three ballots and the LDS atomics for every chunk with a
            // hit; gathering costs one ballot and one LDS store per chunk.
            auto flush64 = [&]() {  // (while the expansion's own candidates are consumed: nothing is prefetched, stores go out)
                __builtin_amdgcn_wave_barrier();  // one wavefront, in-order LDS: the stores above are visible
                if (staged)
                    stores_out();  // (keeps the global queue in order with the direct stores below)
                apply(lane < hc ? hb[lane] : 0xffffffffu, false);
                if (hc > (uint32_t)WAVE)
                {
                    const uint32_t mv = lane < hc - WAVE ? hb[WAVE + lane] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < hc - WAVE)
                        hb[lane] = mv;
                    hc -= WAVE;
                }
                else
                    hc = 0;
            };
            auto flush_rest = [&](bool defer) {  // the (fewer than 128) hits left when an expansion has been searched
                __builtin_amdgcn_wave_barrier();
                if (staged)
                    stores_out();
                for (uint32_t b0 = 0; b0 < hc; b0 += WAVE)
                {
                    const uint32_t o = apply(b0 + lane < hc ? hb[b0 + lane] : 0xffffffffu, defer);
                    if (defer && b0 + lane < hc)
                        stg[b0 + lane] = o;
                }
                staged = defer ? hc : 0u;
                hc = 0;
            };
            auto collect = [&](uint32_t word) {
                const bool in = word != 0xffffffffu;
                const unsigned long long im = __ballot(in);
                if (!im)
                    return;  // no candidate of this chunk is a neighbour
                if (in)
                    hb[hc + __popcll(im & lt)] = word;
                hc += __popcll(im);
                if (hc >= (uint32_t)WAVE)
                    flush64();
            };
            while (qh < qt)
            {
                // a window of up to 64 pops; which of them the reference expands is decided in registers (see
                // replay_lds_kernel): a candidate is skipped iff it is removed already or an EXPANDED earlier
                // candidate of the window holds it within the absorb radius
                const uint32_t wb = qh;
                const uint32_t wn = min((uint32_t)WAVE, qt - qh);
                const bool inw = lane < wn;
                uint32_t wcand;
                if (qt - qh <= (uint32_t)RS_RING)
                    wcand = inw ? ring[(wb + lane) % RS_RING] : 0u;
                else
                {
                    __threadfence_block();
                    wcand = inw ? q[wb + lane] : 0u;
                }
                const bool alive = inw && !(ST_GET(wcand) & 2u);
                const uint32_t ci = alive ? wcand : 0u;
                // one 16-byte record per point {x, y, z, kd group}: a window of 64 scattered points is 64 memory requests
                // instead of 256 (four arrays), and no extra round trip per search
                const float4 wrec = grp_of[ci];
                const float wx = wrec.x, wy = wrec.y, wz = wrec.z;
                const uint32_t wg = __float_as_uint(wrec.w);
                unsigned long long am = __ballot(alive), em = 0;
                // The first alive candidate of a window is always expanded (nothing before it can absorb it): its chunk
                // table is requested BEFORE the selection loop below, which only needs the coordinates and then runs
                // under that load instead of in front of it.
                // (a table already in registers is not requested again: the breadth-first front moves through space, and
                // expansions that follow one another often belong to the same kd group -- 2 KiB less per such expansion)
                ChunkRec ch_first = ch_held;
                if (am)
                {
                    const uint32_t g_first = (uint32_t)__builtin_amdgcn_readlane((int)wg, __ffsll((long long)am) - 1);
                    if (!REUSE || g_first != g_held)
                        ch_first = chunks[(size_t)g_first * LPX_GROUP_CHUNKS + lane];
                    g_held = g_first;
                }
                while (am)
                {
                    const int h = __ffsll((long long)am) - 1;
                    em |= 1ull << h;
                    const float hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), h));
                    const float hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), h));
                    const float hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), h));
                    const float d0 = hx - wx, d1 = hy - wy, d2 = hz - wz;
                    const float dist = d0 * d0 + (d1 * d1 + (d2 * d2 + 0.0f));  // dist_sqr(points_[h], node)
                    const bool conflict = alive && dist <= r2 && dist <= thr_f;
                    am &= ~__ballot(conflict);
                    am &= ~(1ull << h);
                }
                qh = wb + wn;
                RS_LAP(pf_win, pf_t);
                if (!em)
                    continue;
                ++st_win;
                // The expansions of the window, in order.  Which points a window expands is known up front and a search
                // does not depend on the point states, so the searches are software-pipelined: the chunk table of
                // expansion i + 1 is requested before expansion i is searched, and the first candidate batch of i + 1
                // goes out before the hits of i are applied (the LDS work of the apply then runs under those loads).
                // Order of the loads (round 6): a wait for one load is a wait for every load issued before it AND, as soon
                // as a load behind a branch may lie in between, for everything (s_waitcnt vmcnt(0)).  So nothing is requested
                // shortly before something older is waited for: the candidate batch of expansion i + 1 and the chunk table
                // of expansion i + 2 go out TOGETHER, right after the candidates of expansion i have been consumed, and
                // the one drain per expansion -- at the head of the next consume -- finds both at least an apply old.
                // (Rounds 2-5 requested table i + 1 at the head of the loop, immediately in front of the consume of batch
                // i: every expansion waited a full round trip for a table it needed much later.)
                int e = __ffsll((long long)em) - 1;
                em &= em - 1;
                ChunkRec ch = ch_first;
                uint32_t g_cur = g_held;  // the group ch belongs to
                float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), e));
                float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), e));
                float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), e));
                unsigned long long km = rs_cull(ch, qx, qy, qz, r2);
                RsBatch bt;
                rs_issue(bt, PR, ch, km, lane);
                // the table of the window's second expansion, behind the first batch
                int e_next = em ? __ffsll((long long)em) - 1 : -1;
                ChunkRec ch_next = ch;
                uint32_t g_nxt = g_cur;
                if (e_next >= 0)
                {
                    em &= em - 1;
                    g_nxt = (uint32_t)__builtin_amdgcn_readlane((int)wg, e_next);
                    if (!REUSE || g_nxt != g_cur)
                        ch_next = chunks[(size_t)g_nxt * LPX_GROUP_CHUNKS + lane];
                }
                RS_LAP(pf_tab, pf_t);
                for (;;)
                {
                    ++st_exp;
                    rs_consume<REUSE>(bt, PR, qx, qy, qz, r2, thr_f, lane, st_cand, collect);
                    if (staged)
                        stores_out();  // the stores the apply of the expansion before this one owes (its hits are all applied)
                    while (km)
                    {
                        rs_issue(bt, PR, ch, km, lane);
                        rs_consume<REUSE>(bt, PR, qx, qy, qz, r2, thr_f, lane, st_cand, collect);
                    }
                    RS_LAP(pf_cand, pf_t);
                    const bool more = e_next >= 0;
                    if (more)
                    {
                        e = e_next;
                        ch = ch_next;
                        g_cur = g_nxt;
                        qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), e));
                        qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), e));
                        qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), e));
                        km = rs_cull(ch, qx, qy, qz, r2);
                        rs_issue(bt, PR, ch, km, lane);
                        // ... and the table of the expansion after it
                        e_next = em ? __ffsll((long long)em) - 1 : -1;
                        if (e_next >= 0)
                        {
                            em &= em - 1;
                            g_nxt = (uint32_t)__builtin_amdgcn_readlane((int)wg, e_next);
                            if (!REUSE || g_nxt != g_cur)
                                ch_next = chunks[(size_t)g_nxt * LPX_GROUP_CHUNKS + lane];
                        }
                    }
                    RS_LAP(pf_tab, pf_t);
                    // the hits of the expansion just searched, before any hit of the next one -- under the first candidate
                    // batch of the next expansion when there is one (then without global stores: DEFER)
                    flush_rest(STATE_LDS && more);
                    RS_LAP(pf_apply, pf_t);
                    if (!more)
                        break;
                }
                g_held = g_cur;
                if (REUSE)
                    ch_held = ch;  // (the table of group g_held)
            }
            if (lane == 0)
                valid[seed] = (touches >= prm.min_size && touches <= prm.max_size) ? 1u : 0u;  // :113
        }
    }
    if (lane == 0 && st_exp)
    {
        atomicAdd((unsigned long long *)&fstate->replay_entries, st_entries);
        atomicAdd(&fstate->n_expansions, st_exp);
        atomicAdd(&fstate->n_windows, st_win);
        atomicAdd((unsigned long long *)&fstate->cand_total, st_cand);
    }
#ifdef LPX_RS_PROF
    if (prof && lane == 0)
    {
        const unsigned long long slot = atomicAdd(prof, 1ull);
        if (slot < 4000)
        {
            unsigned long long *r = prof + 8 + 8 * slot;
            r[0] = clock64() - pf_t0;
            r[1] = pf_seed;
            r[2] = pf_win;
            r[3] = pf_tab;
            r[4] = pf_cand;
            r[5] = pf_apply;
            r[6] = st_exp;
            r[7] = ((unsigned long long)st_win << 32) | (uint32_t)(st_entries > 0xffffffffull ? 0xffffffffull : st_entries);
        }
    }
#endif
#undef ST_GET
#undef ST_OR
}

__global__ void relabel_kernel(FrameState *frame, const int32_t *__restrict__ seed_of,
                               const uint32_t *__restrict__ valid, const uint32_t *__restrict__ dense,
                               int32_t *__restrict__ labels, uint64_t cap, const uint64_t *__restrict__ total,
                               uint32_t *__restrict__ counts, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<6>(fv.fs);
    frame = lpx_slot(frame, fv.fs);
    seed_of = lpx_slot(seed_of, fv.fs);
    valid = lpx_slot(valid, fv.fs);
    dense = lpx_slot(dense, fv.fs);
    total = lpx_slot(total, fv.fs);
    labels = lpx_user(labels, fv.upitch);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i == 0)
    {
        const uint32_t nc = (uint32_t)*total;  // number of valid seeds = number of clusters
        frame->n_clusters = nc;
        // the single-point sets cc_ranges_kernel settled: one radius search and one list entry each in the reference
        frame->n_expansions += frame->n_single;
        frame->replay_entries += frame->n_single;
        if (counts)  // last kernel of the call: hand the frame counts to the caller
        {
            counts += 4 * (size_t)lpx_blk.z;
            counts[0] = frame->n_ground;
            counts[1] = frame->n_obstacle;
            counts[2] = nc;
            counts[3] = frame->status;
        }
    }
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

}  // namespace

// ------------------------------------------------------------------------------------------------
// cluster regrouping (what Processor::process does on the host at reference src/processor.cpp:180-200):
// points of every valid cluster, clusters in label order, points in ascending index order, INVALID dropped.
// A stable radix sort of (label, index) with INVALID mapped behind every label gives the order; the
// cluster boundaries are where the sorted key changes.
// ------------------------------------------------------------------------------------------------
__global__ void group_keys_kernel(const int32_t *__restrict__ labels, uint32_t m, uint32_t *__restrict__ key,
                                  uint32_t *__restrict__ val)
{
    const LpxBlock lpx_blk = lpx_block<6>(0);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i >= m)
        return;
    const int32_t l = labels[i];
    key[i] = l < 0 ? m : (uint32_t)l;  // labels are < m; INVALID (and UNDEFINED) sort last
    val[i] = i;
}

__global__ void group_offsets_kernel(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval, uint32_t m,
                                     uint32_t *__restrict__ offsets, uint32_t *__restrict__ indices)
{
    const LpxBlock lpx_blk = lpx_block<6>(0);
    const uint32_t p = lpx_blk.x * blockDim.x + threadIdx.x;
    if (p >= m)
        return;
    const uint32_t k = skey[p];
    const uint32_t prev = p ? skey[p - 1] : 0xffffffffu;
    if (k < m)
    {
        indices[p] = sval[p];
        if (p == 0 || prev != k)
            offsets[k] = p;
        if (p == m - 1)
            offsets[k + 1] = m;  // no rejected point: the last cluster ends the array
    }
    else if (p == 0 || prev != k)
        offsets[p ? prev + 1 : 0] = p;  // first rejected point: end of the last valid cluster
}

// ------------------------------------------------------------------------------------------------
// N3: per-cluster 2-D convex hull -- the counterpart of the convex branch of findOrderedConcaveOutlines (reference
// src/polygon_simplification.cpp:96-115: clusters with fewer than 20 points).  The reference delegates to
// geom::constructConvexHull(ANDREW_MONOTONE_CHAIN, COUNTERCLOCKWISE) of its Convex-Hull submodule, which is not
// vendored: the published algorithm is RESTATED, not verified against the submodule, with the conventions stated in
// DESIGN.md (points sorted by (x, y, index), duplicates skipped, collinear points are not vertices, float32 cross
// product without contraction, CCW from the lowest (x, y) point).  No parity is claimed beyond that branch
// (findOrderedConvexOutlines, :32-80, switches to Chan's algorithm above 1000 points in the same absent submodule).
// The concave branch (:116-131, Concave-Hull submodule) is out of scope: larger clusters get an empty hull.
//
// Input: the CSR of lpx_run_groups (clusters in label order).  Three stable radix sorts of the CSR's point list
// -- by y key, by x key, by label -- leave every cluster's points in (x, y, index) order inside the cluster's
// own CSR range; then one LANE per cluster runs the monotone chain with its stack in global scratch (the two
// top entries stay in registers, so only a pop loads), and a second kernel packs the hulls.
// ------------------------------------------------------------------------------------------------
__global__ void hull_key_kernel(const uint32_t *__restrict__ vals, const float *__restrict__ coord,
                                const int32_t *__restrict__ labels, const uint32_t *__restrict__ d_n,
                                uint32_t *__restrict__ key)
{
    const LpxBlock lpx_blk = lpx_block<6>(0);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i >= *d_n)
        return;
    const uint32_t v = vals[i];
    key[i] = labels ? (uint32_t)labels[v] : lpx_float_key(coord[v]);  // lpx_float_key maps -0 to +0
}

__device__ __forceinline__ float hull_cross(float ax, float ay, float bx, float by, float cx, float cy)
{
    const float l = (bx - ax) * (cy - ay);
    const float r = (by - ay) * (cx - ax);
    return l - r;
}

__global__ void hull_chain_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ sorted,
                                  const float *__restrict__ OX, const float *__restrict__ OY,
                                  const FrameState *__restrict__ frame, uint32_t max_points,
                                  uint32_t *__restrict__ st_idx, float2 *__restrict__ st_xy,
                                  uint32_t *__restrict__ hull_len)
{
    const LpxBlock lpx_blk = lpx_block<6>(0);
    const uint32_t c = lpx_blk.x * blockDim.x + threadIdx.x;
    const uint32_t nc = frame->n_clusters;
    if (c > nc)
        return;
    if (c == nc)
    {
        hull_len[c] = 0;  // sentinel: the exclusive scan leaves the total here
        return;
    }
    const uint32_t b = off[c], n = off[c + 1] - b;
    if (n == 0 || n >= max_points)
    {
        hull_len[c] = 0;
        return;
    }
    const uint32_t *sp = sorted + b;
    // the stack holds the lower hull plus the upper chain of the points seen so far from the end: at most 2 n
    // entries (a lower-hull vertex can sit on the upper hull of a suffix), so cluster c owns [2 b, 2 b + 2 n)
    uint32_t *S = st_idx + 2 * (size_t)b;
    float2 *SXY = st_xy + 2 * (size_t)b;
    // -0 and +0 are one value (the sort keys say so too): compare and compute on x + 0
#define HP_LOAD(i, id, x, y)                                                                                          \
    const uint32_t id = sp[i];                                                                                        \
    const float x = OX[id] + 0.0f, y = OY[id] + 0.0f;
    // distinct points (duplicates are consecutive in sorted order)
    uint32_t u = 0;
    {
        float px = 0.0f, py = 0.0f;
        for (uint32_t i = 0; i < n; ++i)
        {
            HP_LOAD(i, id, x, y)
            if (i == 0 || x != px || y != py)
            {
                if (u < 2)
                {
                    S[u] = id;
                    SXY[u] = make_float2(x, y);
                }
                ++u;
            }
            px = x;
            py = y;
        }
    }
    if (u <= 2)
    {
        hull_len[c] = u;
        return;
    }
    uint32_t k = 0;
    float ax = 0.0f, ay = 0.0f, bx = 0.0f, by = 0.0f;  // S[k-2], S[k-1]
#define HP_POP_WHILE(limit)                                                                                           \
    while (k >= (limit) && hull_cross(ax, ay, bx, by, x, y) <= 0.0f)                                                  \
    {                                                                                                                 \
        --k;                                                                                                          \
        bx = ax;                                                                                                      \
        by = ay;                                                                                                      \
        if (k >= 2)                                                                                                   \
        {                                                                                                             \
            const float2 t = SXY[k - 2];                                                                              \
            ax = t.x;                                                                                                 \
            ay = t.y;                                                                                                 \
        }                                                                                                             \
    }
#define HP_PUSH()                                                                                                     \
    S[k] = id;                                                                                                        \
    SXY[k] = make_float2(x, y);                                                                                       \
    ++k;                                                                                                              \
    ax = bx;                                                                                                          \
    ay = by;                                                                                                          \
    bx = x;                                                                                                           \
    by = y;
    {
        float px = 0.0f, py = 0.0f;
        for (uint32_t i = 0; i < n; ++i)  // lower hull
        {
            HP_LOAD(i, id, x, y)
            const bool dup = i > 0 && x == px && y == py;
            px = x;
            py = y;
            if (dup)
                continue;
            HP_POP_WHILE(2u)
            HP_PUSH()
        }
    }
    const uint32_t lower = k + 1;
    {
        bool skipped_last = false;
        for (uint32_t i = n; i-- > 0;)  // upper hull
        {
            HP_LOAD(i, id, x, y)
            if (i > 0)
            {
                const uint32_t pid = sp[i - 1];
                if (x == OX[pid] + 0.0f && y == OY[pid] + 0.0f)
                    continue;  // not the first of its run of duplicates
            }
            if (!skipped_last)
            {
                skipped_last = true;  // the last distinct point is the top of the stack already
                continue;
            }
            HP_POP_WHILE(lower)
            if (i == 0)
                break;  // the first point would close the polygon: it is not stored twice
            HP_PUSH()
        }
    }
    hull_len[c] = k;
#undef HP_LOAD
#undef HP_POP_WHILE
#undef HP_PUSH
}

__global__ void hull_count_kernel(const uint32_t *__restrict__ off, const FrameState *__restrict__ frame,
                                  uint32_t *__restrict__ d_nv)
{
    if (threadIdx.x == 0)
        *d_nv = off[frame->n_clusters];  // points in valid clusters (0 clusters: offsets[0] = 0)
}

// exclusive scan of hull_len[0 .. n_clusters] (the sentinel entry receives the total) by one workgroup
__global__ __launch_bounds__(1024) void hull_scan_kernel(const uint32_t *__restrict__ len,
                                                         const FrameState *__restrict__ frame,
                                                         uint32_t *__restrict__ out)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t n = frame->n_clusters + 1;
    const uint32_t tid = threadIdx.x, lane = tid % WAVE, w = tid / WAVE;
    if (tid == 0)
        s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024)
    {
        const uint32_t i = base + tid;
        const uint32_t v = i < n ? len[i] : 0u;
        const uint32_t incl = lpx_wave_incl_scan_u32(v);
        if (lane == WAVE - 1)
            s_wave[w] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t k = 0; k < w; ++k)
            before += s_wave[k];
        if (i < n)
            out[i] = before + incl - v;
        __syncthreads();
        if (tid == 1023)
            s_carry = before + incl;
        __syncthreads();
    }
}

__global__ void hull_pack_kernel(const uint32_t *__restrict__ off, const uint32_t *__restrict__ slabel,
                                 const uint32_t *__restrict__ d_n, const uint32_t *__restrict__ st_idx,
                                 const float2 *__restrict__ st_xy, const uint32_t *__restrict__ hull_len,
                                 const uint32_t *__restrict__ hull_off, uint32_t *__restrict__ out_idx,
                                 float2 *__restrict__ out_xy)
{
    const LpxBlock lpx_blk = lpx_block<6>(0);
    const uint32_t p = lpx_blk.x * blockDim.x + threadIdx.x;
    if (p >= *d_n)
        return;
    const uint32_t c = slabel[p];  // position p of the sorted point list belongs to cluster c
    const uint32_t j = p - off[c];
    if (j < hull_len[c])
    {
        const size_t q = 2 * (size_t)off[c] + j;  // stack entry j of cluster c
        out_idx[hull_off[c] + j] = st_idx[q];
        out_xy[hull_off[c] + j] = st_xy[q];
    }
}

#ifdef LPX_DEV_KNOBS
#include "../../experiments/burners.inc"  // resource burners (measurement only)
#endif

static uint32_t bits_for_count(uint32_t n)  // bits to hold values 0..n-1
{
    uint32_t b = 1;
    while (b < 32 && (1ull << b) < n)
        ++b;
    return b;
}

int lpx_run_cluster(lpx_ctx *ctx, uint32_t m_max, const lpx_clu_cfg *cfg, int32_t *d_labels, uint32_t *d_counts,
                    bool kd_ready)
{
    FrameState *frame = (FrameState *)ctx->frame.p;
    hipStream_t st = ctx->stream;
    const FV fv = lpx_fv(ctx);
    if (m_max == 0)
        return d_counts ? lpx_write_counts(ctx, d_counts) : LPX_OK;
    // diagnostics only (results are wrong when a stage is skipped): LPX_SKIP=kd,index,grid,sort,replay
    static const char *skip_env = LPX_KNOB("LPX_SKIP");
    const bool skip_kd = skip_env && strstr(skip_env, "kd"), skip_index = skip_env && strstr(skip_env, "index");
    const bool skip_grid = skip_env && strstr(skip_env, "grid"), skip_sort = skip_env && strstr(skip_env, "sort");
    const bool skip_replay = skip_env && strstr(skip_env, "replay");
    const dim3 blk(256), grd((m_max + 255) / 256, 1, ctx->cur_b);
    uint32_t *root = (uint32_t *)ctx->key_a.p, *iota = (uint32_t *)ctx->val_a.p;
    // Forked front end (lpx_set_fork): the component grid goes to a side stream BEFORE the kd build is enqueued here --
    // both start from the obstacle cloud alone and touch disjoint buffers; the join sits where the first consumer of
    // the components (the sort by root) is enqueued.
#ifdef LPX_DEV_KNOBS
    const bool sweep_cc = !ctx->use_lists && !skip_grid && !lpx_cc_from_chunks(m_max) && lpx_cc_from_sweep(m_max);
#else
    const bool sweep_cc = false;
#endif
    const bool grid_cc = !ctx->use_lists && !skip_grid && !lpx_cc_from_chunks(m_max) && !sweep_cc;
    const bool forked = grid_cc && ctx->fork && ctx->fork_stream && ctx->ev_fork && ctx->ev_join;
    int rc = LPX_OK;
#ifdef LPX_DEV_KNOBS
    burn_resources(ctx);
#endif
    // While the fork is open the side stream may hold grid kernels of this call: EVERY way out of this function -- an
    // error of the grid itself, of the kd build or of the chunk tables included -- joins it to the context's stream first,
    // so that the next call on the slot set cannot overtake leftover grid kernels.
    struct ForkJoin
    {
        lpx_ctx *ctx;
        hipStream_t st;
        bool open = false;
        int join()
        {
            if (!open)
                return LPX_OK;
            open = false;
            if (hipEventRecord(ctx->ev_join, ctx->fork_stream) != hipSuccess ||
                hipStreamWaitEvent(st, ctx->ev_join, 0) != hipSuccess)
            {
                (void)hipStreamSynchronize(ctx->fork_stream);  // last resort: the host waits
                return lpx_fail(ctx, LPX_ERR_HIP, "joining the forked component grid failed");
            }
            return LPX_OK;
        }
        ~ForkJoin() { (void)join(); }
    } fork_join{ctx, st};
    if (forked)
    {
        LPX_HIP(ctx, hipEventRecord(ctx->ev_fork, st));
        LPX_HIP(ctx, hipStreamWaitEvent(ctx->fork_stream, ctx->ev_fork, 0));
        fork_join.open = true;
        ctx->stream = ctx->fork_stream;
        rc = lpx_grid_components(ctx, m_max, cfg->distance_squared, root, iota, false);
        ctx->stream = st;
        if (rc)
            return rc;  // (fork_join joins)
    }
    rc = (kd_ready || skip_kd) ? LPX_OK : lpx_kd_build(ctx, m_max);
    if (rc)
        return rc;
    ReplayParams prm;
    const double one_minus_q = 1.0 - (double)cfg->cluster_quality;
    prm.thr = (one_minus_q * one_minus_q) * (double)cfg->distance_squared;  // std::pow(x, 2) == x*x exactly
    prm.thr_f = (float)prm.thr;
    if ((double)prm.thr_f > prm.thr)
        prm.thr_f = nextafterf(prm.thr_f, -INFINITY);
    prm.r2 = cfg->distance_squared;
    prm.min_size = cfg->min_cluster_size;
    prm.max_size = cfg->max_cluster_size;
    prm.dbg_rows = ctx->dbg_buf ? (uint32_t)(ctx->dbg_store.bytes / 32 < 65536 ? ctx->dbg_store.bytes / 32 : 65536) : 0u;
    uint32_t *sroot = nullptr, *members = nullptr;
    uint32_t *cc_lo = (uint32_t *)ctx->cc_lo.p, *cc_hi = (uint32_t *)ctx->cc_hi.p;
    uint32_t *valid = (uint32_t *)ctx->valid.p;
    if (ctx->use_lists)
    {
        // round-1 path: every radius list materialised, components by union-find over the lists
        rc = lpx_neighbours(ctx, m_max, cfg->distance_squared, prm.thr_f, true);
        if (rc)
            return rc;
    }
    else
    {
        // expansion-driven path: candidate chunks per kd group; the components come out of the same kernel
        // (kd_link_queries) -- or, development build with LPX_CC=grid, from the clique-cell grid
        const bool clear_in_index = grid_cc && !forked && !skip_index;
        if ((!skip_index && (rc = lpx_group_index(ctx, m_max, cfg->distance_squared, clear_in_index))) ||
            (grid_cc && !forked &&
             (rc = lpx_grid_components(ctx, m_max, cfg->distance_squared, root, iota, clear_in_index))))
            return rc;
#ifdef LPX_DEV_KNOBS
        if (sweep_cc && (rc = lpx_sweep_components(ctx, m_max, cfg->distance_squared)))
            return rc;
#endif
        if ((rc = fork_join.join()))
            return rc;
    }
    {
        StageTimer tm(ctx, ST_CC);
        // (the kernel that writes the roots also counts their lowest byte per sort tile: lpx_sort_first_hist)
        uint32_t *first_hist = skip_sort ? nullptr : lpx_sort_first_hist(ctx, m_max);
        const dim3 gtile((m_max + LPX_SORT_TILE - 1) / LPX_SORT_TILE, 1, ctx->cur_b);
        // (the members going into the sort are the positions 0, 1, 2, ...: nobody writes them, the first pass makes
        // them up -- lpx_sort_pairs(iota_vals))
        uint32_t *const iota_out = skip_sort ? iota : (uint32_t *)nullptr;
        if (ctx->use_lists || lpx_cc_from_chunks(m_max) || sweep_cc)
            hipLaunchKernelGGL(flatten_kernel, gtile, blk, 0, st, (uint32_t *)ctx->parent.p, frame, root, iota_out,
                               (uint8_t *)ctx->state.p, valid, cc_lo, cc_hi, first_hist,
                               ctx->use_lists ? ctx->h_liststat : (LpxListStat *)nullptr,
                               ctx->use_lists ? ++ctx->list_seq : 0u, fv.fs);
        else if (grid_cc && !skip_grid)
            rc = lpx_grid_flatten(ctx, m_max, root, iota_out, first_hist);
        if (rc)
            return rc;
        if (skip_sort)
        {
            sroot = root;
            members = iota;
        }
        else
            rc = lpx_sort_pairs(ctx, root, (uint32_t *)ctx->key_b.p, iota, (uint32_t *)ctx->val_b.p, m_max,
                                &frame->n_obstacle, bits_for_count(m_max), &sroot, &members, first_hist != nullptr, nullptr,
                                true, true);  // (every root is a position of the obstacle cloud: below its count)
        if (rc)
            return rc;
        hipLaunchKernelGGL(cc_ranges_kernel, grd, blk, 0, st, (const uint32_t *)sroot, (const uint32_t *)members, frame,
                           cc_lo, cc_hi, (uint32_t *)ctx->rpos.p, (int32_t *)ctx->seed_of.p, valid, prm.min_size,
                           prm.max_size, fv.fs);
    }
    // Overlapped tail: everything from here on (the replay, the label scan, the relabel kernel and the counts) depends
    // only on what the kernels above left in THIS slot set, and the caller's stream is free for the front end of its
    // next chain (on the twin slot set).  The tail stream waits for the front end; the next call on this slot set waits
    // for the tail (begin_call).  ctx->stream is swapped for the duration so that the stage timers and the scan helper
    // follow.
    hipStream_t const front_stream = ctx->stream;
    struct TailScope
    {
        lpx_ctx *c;
        hipStream_t front;
        bool on;
        ~TailScope()
        {
            if (on)
            {
                hipEventRecord(c->ev_tail, c->stream);
                c->stream = front;
                c->tail_pending = true;
            }
        }
    } tail_scope{ctx, front_stream, false};
    if (ctx->split_tail && ctx->tail_stream && ctx->ev_front && ctx->ev_tail)
    {
        LPX_HIP(ctx, hipEventRecord(ctx->ev_front, front_stream));
        LPX_HIP(ctx, hipStreamWaitEvent(ctx->tail_stream, ctx->ev_front, 0));
        ctx->stream = ctx->tail_stream;
        st = ctx->tail_stream;
        tail_scope.on = true;
    }
    if (!ctx->use_lists && !skip_replay)
    {
        StageTimer tm(ctx, ST_REPLAY);
        const size_t fixed = sizeof(RsShared);
        auto bitmap_bytes = [](size_t pts) { return sizeof(uint32_t) * ((pts + 15) / 16 + 4); };
        if (!ctx->attr_search)
        {
            LPX_HIP(ctx, hipFuncSetAttribute((const void *)replay_search_kernel<true, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            LPX_HIP(ctx, hipFuncSetAttribute((const void *)replay_search_kernel<true, true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            LPX_HIP(ctx, hipFuncSetAttribute((const void *)replay_search_kernel<false, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            LPX_HIP(ctx, hipFuncSetAttribute((const void *)replay_search_kernel<false, true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            ctx->attr_search = true;
        }
        // Persistent sequencers: every wavefront pulls components from the frame's work list.  A sequencer needs
        // little of a CU but its workgroup holds LDS and wave slots for milliseconds (the largest component of the
        // frame), so what counts is how many replay workgroups are resident on the DEVICE, over all chains in
        // flight: about 1024 is the measured optimum (kitti, 16 contexts x 32 frames: 2 per frame 1413 Mpts/s,
        // 8 per frame 1338, 16 per frame 1317; stream, 5 x 32: 8 per frame; synth1m, 4 x 4: 32-64 per frame).
        // The frame slots of the contexts of this device that enqueued a frame in the last 100 ms stand for "frames in
        // flight".  LPX_RS_GRID overrides the per-frame count.
        static const uint32_t rg_env = LPX_KNOB("LPX_RS_GRID") ? (uint32_t)atoi(LPX_KNOB("LPX_RS_GRID")) : 0u;
        uint32_t slots = lpx_active_frame_slots(ctx->device);
        slots = slots < ctx->cur_b ? ctx->cur_b : slots;
        uint32_t rg_cap = rg_env ? rg_env : 1024u / slots;
        rg_cap = rg_cap < 1u ? 1u : (rg_cap > 64u && !rg_env ? 64u : rg_cap);
        const uint32_t rgrid = (m_max + RS_WAVES - 1) / RS_WAVES < rg_cap ? (m_max + RS_WAVES - 1) / RS_WAVES : rg_cap;
#define RS_ARGS(lo_, hi_)                                                                                             \
    (const FrameState *)frame, (const uint32_t *)cc_lo, (const uint32_t *)cc_hi, (const uint32_t *)members,            \
        (const KdNode *)ctx->nodes_pre.p, (const ChunkRec *)ctx->chunks.p, (const float4 *)ctx->grp_of.p,              \
        (const float *)ctx->OX.p, (const float *)ctx->OY.p, (const float *)ctx->OZ.p, (uint8_t *)ctx->state.p,          \
        (int32_t *)ctx->seed_of.p, (uint32_t *)ctx->queue.p, valid, prm, frame, (const uint32_t *)ctx->rpos.p,         \
        (uint32_t)(lo_), (uint32_t)(hi_),                                                                             \
        (unsigned long long *)(ctx->dbg_buf && ctx->dbg_store.bytes >= (512u << 10) ? (char *)ctx->dbg_buf + (256u << 10) \
                                                                                   : nullptr),                       \
        fv
        // Point states as a 2-bit LDS bitmap sized for the host's bound (31 KiB for a 123k-point frame; the resident
        // footprint is set by rgrid, not by this), capped at what LDS holds (~440k points).  When the bound exceeds
        // the cap a second launch with one byte per point in HBM serves the frames that really are that large; each
        // launch checks the frame's obstacle count on the device (a 1M-point cloud usually has < 440k obstacles).
        static const int rs_state = LPX_KNOB("LPX_RS_STATE") ? atoi(LPX_KNOB("LPX_RS_STATE")) : 0;  // 1: states in HBM
        // (LPX_RS_STATE: 1 states in HBM, 2 in LDS up to its capacity.)  Default: the LDS bitmap while it is at most 64 KiB
        // (262 144 points): a larger one leaves room for one replay workgroup per CU and costs more than it saves --
        // 1M-point frames (100 KiB bitmaps): 1060 Mpts/s with the states in LDS, 1125 with one byte per point in HBM;
        // 120k-point frames (31 KiB): 1587 against 1505.
        const uint32_t lds_fit = (uint32_t)(((152 * 1024 - fixed) / sizeof(uint32_t) - 4) * 16);
        const uint32_t lds_pts = rs_state == 1 ? 0u : (rs_state == 2 ? lds_fit : (lds_fit < 262144u ? lds_fit : 262144u));
        uint32_t m_lds = m_max < lds_pts ? m_max : lds_pts;
        // The bound m_max is the INPUT size of the largest frame; the obstacle cloud is about half of it.  With many
        // chains in flight the replay workgroups of all of them are resident together for milliseconds, and what their
        // bitmaps hold of every CU's LDS is what the LDS-staged kernels of the other chains (kd subtrees, chunk tables,
        // seed selection) cannot get.  So the bitmap is sized by the largest obstacle count the context's previous call
        // saw (+25 %); a frame that exceeds it is served by the second launch.
        static const int rs_fit = LPX_KNOB("LPX_RS_FIT") ? atoi(LPX_KNOB("LPX_RS_FIT")) : 0;  // (measured: no gain, and a second launch per chain)
        if (rs_fit && ctx->cur_b > 1 && ctx->h_search && m_lds)
        {
            const uint32_t seen = (uint32_t)(ctx->h_search[5] & 0xffffffffull);
            const uint32_t want = seen + seen / 4 + 1024;
            if (seen && want < m_lds)
                m_lds = want;
        }
        const bool reuse = ctx->ix_bucket == 32;  // dense surfaces (lpx_group_index chose the small groups)
        if (m_lds)
        {
            if (reuse)
                hipLaunchKernelGGL((replay_search_kernel<true, true>), dim3(rgrid, 1, ctx->cur_b), dim3(RS_THREADS),
                                   fixed + bitmap_bytes(m_lds), st, RS_ARGS(0u, m_lds));
            else
                hipLaunchKernelGGL((replay_search_kernel<true, false>), dim3(rgrid, 1, ctx->cur_b), dim3(RS_THREADS),
                                   fixed + bitmap_bytes(m_lds), st, RS_ARGS(0u, m_lds));
        }
        if (m_max > m_lds)
        {
            if (reuse)
                hipLaunchKernelGGL((replay_search_kernel<false, true>), dim3(rgrid, 1, ctx->cur_b), dim3(RS_THREADS), fixed,
                                   st, RS_ARGS(m_lds, m_max));
            else
                hipLaunchKernelGGL((replay_search_kernel<false, false>), dim3(rgrid, 1, ctx->cur_b), dim3(RS_THREADS), fixed,
                                   st, RS_ARGS(m_lds, m_max));
        }
#undef RS_ARGS
        // what the searches of this call cost per hit, for the group size of the next call (read when it is there)
        if (ctx->h_search)
            LPX_HIP(ctx, hipMemcpyAsync(ctx->h_search, (const char *)frame + offsetof(FrameState, replay_entries),
                                        6 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    else if (ctx->use_lists)
    {
        StageTimer tm(ctx, ST_REPLAY);
        const size_t lds = sizeof(uint32_t) * (((((size_t)m_max + 15) / 16 + 3) & ~(size_t)3) + RP_RING + RP_STAGE);
        static const int rp_grid = LPX_KNOB("LPX_RP_GRID") ? atoi(LPX_KNOB("LPX_RP_GRID")) : 2048;
        // (tests) 1 / 2: component-local LDS states for every frame, 3: states in HBM for every frame
        static const int rp_state = LPX_KNOB("LPX_RP_STATE") ? atoi(LPX_KNOB("LPX_RP_STATE")) : 0;
#define RP_ARGS                                                                                                       \
    frame, cc_lo, cc_hi, members, (const uint32_t *)ctx->nb_off.p, (const uint32_t *)ctx->nb_len.p,                   \
        (const uint32_t *)ctx->nb_idx.p, (const float *)ctx->OX.p, (const float *)ctx->OY.p,                          \
        (const float *)ctx->OZ.p, (int32_t *)ctx->seed_of.p, (uint32_t *)ctx->queue.p, valid, prm, ctx->cap_nb,       \
        frame, (const uint32_t *)ctx->rpos.p, (uint32_t *)ctx->dbg_buf, (uint8_t *)ctx->state.p,                      \
        (const uint32_t *)ctx->lpos.p, fv
        if (lds <= 112 * 1024 && rp_state == 0)
        {
            if (!ctx->attr_replay)
            {
                LPX_HIP(ctx, hipFuncSetAttribute((const void *)replay_lds_kernel<RP_STATE_LDS>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
                ctx->attr_replay = true;
            }
            const uint32_t rgrid = m_max < 512u ? m_max : 512u;  // persistent: blocks pull components from a list
            hipLaunchKernelGGL(replay_lds_kernel<RP_STATE_LDS>, dim3(rgrid, 1, ctx->cur_b), dim3(WAVE), lds, st, RP_ARGS);
        }
        else if (rp_state != 3)
        {
            // clouds beyond the LDS bitmap (a 5M-point frame) and LPX_RP_STATE=2 (tests: every frame): the states of a
            // component in LDS by member position -- pos_of in the kd build's stop-list scratch, free since the build
            if (sizeof(uint32_t) * (size_t)m_max > ctx->lpos.bytes)
                return lpx_fail(ctx, LPX_ERR_INTERNAL, "member positions of %u points do not fit their scratch", m_max);
            hipLaunchKernelGGL(pos_of_kernel, grd, blk, 0, st, (const uint32_t *)members, (const FrameState *)frame,
                               (uint32_t *)ctx->lpos.p, fv.fs);
            const uint32_t rgrid = m_max < (uint32_t)rp_grid ? m_max : (uint32_t)rp_grid;
            hipLaunchKernelGGL(replay_lds_kernel<RP_STATE_LOCAL>, dim3(rgrid, 1, ctx->cur_b), dim3(WAVE),
                               sizeof(uint32_t) * (RP_LOCAL_CAP / 16u + RP_RING + RP_STAGE), st, RP_ARGS);
        }
        else
        {
            // LPX_RP_STATE=3 (tests): states in HBM for every component, the form a component beyond RP_LOCAL_CAP takes
            const uint32_t rgrid = m_max < (uint32_t)rp_grid ? m_max : (uint32_t)rp_grid;
            hipLaunchKernelGGL(replay_lds_kernel<RP_STATE_HBM>, dim3(rgrid, 1, ctx->cur_b), dim3(WAVE),
                               sizeof(uint32_t) * (RP_RING + RP_STAGE), st, RP_ARGS);
        }
#undef RP_ARGS
    }
    {
        StageTimer tm(ctx, ST_LABELS);
        uint32_t *dense = (uint32_t *)ctx->nb_len.p;  // neighbour lengths are no longer needed
        uint64_t *total = (uint64_t *)((char *)ctx->hist.p);  // 8-byte scratch at the head of hist
        rc = lpx_exclusive_scan(ctx, valid, dense, m_max, &frame->n_obstacle, total);
        if (rc)
            return rc;
        hipLaunchKernelGGL(relabel_kernel, grd, blk, 0, st, frame, (const int32_t *)ctx->seed_of.p, valid, dense,
                           d_labels, ctx->cap_nb, (const uint64_t *)total, d_counts, fv);
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

// Hulls of the clusters (labels d_labels over the m points resident in OX / OY) with fewer than max_points points.
// d_offsets / d_indices: the CSR of lpx_run_groups for the same labels (already computed).  Results:
// d_hull_off[n_clusters + 1], d_hull_idx / d_hull_xy[<= m].
int lpx_run_hulls(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, const uint32_t *d_offsets,
                  const uint32_t *d_indices, uint32_t max_points, uint32_t *d_hull_off, uint32_t *d_hull_idx,
                  float *d_hull_xy)
{
    if (m == 0)
    {
        LPX_HIP(ctx, hipMemsetAsync(d_hull_off, 0, sizeof(uint32_t), ctx->stream));
        return LPX_OK;
    }
    StageTimer tm(ctx, ST_GROUPS);
    hipStream_t st = ctx->stream;
    const FrameState *frame = (const FrameState *)ctx->frame.p;
    const dim3 blk(256), grd((m + 255) / 256);
    uint32_t *ka = (uint32_t *)ctx->key_a.p, *kb = (uint32_t *)ctx->key_b.p;
    uint32_t *va = (uint32_t *)ctx->val_a.p, *vb = (uint32_t *)ctx->val_b.p;
    // number of points in valid clusters = offsets[n_clusters]; as a device count for the sorts it is read from a
    // one-word copy (n_clusters is only known on the device here)
    uint32_t *d_nv = (uint32_t *)((char *)ctx->d_counts.p + 32);
    hipLaunchKernelGGL(hull_count_kernel, dim3(1), dim3(64), 0, st, d_offsets, frame, d_nv);
    LPX_HIP(ctx, hipMemcpyAsync(va, d_indices, sizeof(uint32_t) * m, hipMemcpyDeviceToDevice, st));
    const float *coords[2] = {(const float *)ctx->OY.p, (const float *)ctx->OX.p};
    uint32_t *ko = ka, *vo = va;
    int rc;
    for (int pass = 0; pass < 3; ++pass)
    {
        uint32_t *kin = (vo == va) ? ka : kb;  // keys travel with the buffer pair the values are in
        hipLaunchKernelGGL(hull_key_kernel, grd, blk, 0, st, (const uint32_t *)vo, pass < 2 ? coords[pass] : nullptr,
                           pass == 2 ? d_labels : (const int32_t *)nullptr, (const uint32_t *)d_nv, kin);
        const uint32_t bits = pass == 2 ? bits_for_count(m + 1) : 32;
        if (vo == va)
            rc = lpx_sort_pairs(ctx, ka, kb, va, vb, m, d_nv, bits, &ko, &vo);
        else
            rc = lpx_sort_pairs(ctx, kb, ka, vb, va, m, d_nv, bits, &ko, &vo);
        if (rc)
            return rc;
    }
    // scratch that is free once the clustering is done: 2 stack words and 2 float2 per point, one length per cluster
    uint32_t *st_idx = (uint32_t *)ctx->key64_b.p, *hull_len = (uint32_t *)ctx->cc_hi.p;
    float2 *st_xy = (float2 *)ctx->nodes_pre.p;
    hipLaunchKernelGGL(hull_chain_kernel, dim3((m + 1 + 63) / 64), dim3(64), 0, st, d_offsets, (const uint32_t *)vo,
                       (const float *)ctx->OX.p, (const float *)ctx->OY.p, frame, max_points, st_idx, st_xy, hull_len);
    hipLaunchKernelGGL(hull_scan_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t *)hull_len, frame, d_hull_off);
    hipLaunchKernelGGL(hull_pack_kernel, grd, blk, 0, st, d_offsets, (const uint32_t *)ko, (const uint32_t *)d_nv,
                       (const uint32_t *)st_idx, (const float2 *)st_xy, (const uint32_t *)hull_len,
                       (const uint32_t *)d_hull_off, d_hull_idx, (float2 *)d_hull_xy);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_run_groups(lpx_ctx *ctx, const int32_t *d_labels, uint32_t m, uint32_t *d_offsets, uint32_t *d_indices)
{
    if (m == 0)
    {
        LPX_HIP(ctx, hipMemsetAsync(d_offsets, 0, sizeof(uint32_t), ctx->stream));
        return LPX_OK;
    }
    StageTimer tm(ctx, ST_GROUPS);
    const dim3 blk(256), grd((m + 255) / 256);
    hipLaunchKernelGGL(group_keys_kernel, grd, blk, 0, ctx->stream, d_labels, m, (uint32_t *)ctx->key_a.p,
                       (uint32_t *)ctx->val_a.p);
    uint32_t *skey = nullptr, *sval = nullptr;
    int rc = lpx_sort_pairs(ctx, (uint32_t *)ctx->key_a.p, (uint32_t *)ctx->key_b.p, (uint32_t *)ctx->val_a.p,
                            (uint32_t *)ctx->val_b.p, m, nullptr, bits_for_count(m + 1), &skey, &sval);
    if (rc)
        return rc;
    LPX_HIP(ctx, hipMemsetAsync(d_offsets, 0, sizeof(uint32_t), ctx->stream));  // no valid cluster: offsets[0] = 0
    hipLaunchKernelGGL(group_offsets_kernel, grd, blk, 0, ctx->stream, skey, sval, m, d_offsets, d_indices);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
