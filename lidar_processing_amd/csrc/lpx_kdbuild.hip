// lpx_kdbuild.hip -- the reference kd-tree ORDER on gfx950.
//
// Replaces KDTree<float,3>::rebuild (reference src/kdtree.hpp:174-225) as used by Clusterer::cluster
// (src/clustering.cpp:63); its radius search (:292-341) is served from this tree by lpx_lists.hip (every list) and
// lpx_chunks.hip (candidate chunks for searches on demand).  (Rounds 1-5: one file, lpx_kdtree.hip.)
//
// Why the order matters: the FEC loop consumes neighbours in kd-tree pre-order and its result
// depends on that order.  The tree is a median split by std::nth_element, so the
// node array after rebuild IS the tree (node of range [b,e) sits at b+(e-b)/2; children are
// [b,mid) and [mid+1,e)), and where tied coordinates land is decided by libstdc++'s introselect
// (bits/stl_algo.h:1964-1986: median-of-3 to first, Hoare __unguarded_partition, heap_select at the
// depth limit, insertion sort below 4 elements).  We reproduce that permutation exactly, in parallel:
//
//   Hoare partition as a data-parallel step.  With pivot value v at position `first`, let
//   L_1<L_2<... be the positions in (first,last) holding keys >= v (where the left cursor stops) and
//   R_1>R_2>... those holding keys <= v (where the right cursor stops).  The sequential loop swaps
//   L_k <-> R_k for k = 1..K, K = #{k : L_k < R_k}, and returns cut = min(L_{K+1}, R_K).  Both lists
//   come from one flag pass with a prefix scan; the swaps are independent.
//
// One workgroup (or one wavefront for ranges <= 512 nodes, staged in LDS) owns one range.
#include "lpx_kd_shared.h"

#include <string.h>
#include <stdlib.h>

namespace
{
struct View
{
    Node *a;        // nodes, element i at a[i - off]
    uint32_t *lp;   // positions of keys >= pivot, ascending        (index i - off)
    uint32_t *ra;   // positions of keys <= pivot, ascending        (index i - off)
    int off;
};

__device__ __forceinline__ float nkey(const View &v, int i, int axis)
{
    return ((const float *)(v.a + (i - v.off)))[axis];
}
__device__ __forceinline__ Node nget(const View &v, int i)
{
    return v.a[i - v.off];
}
__device__ __forceinline__ void nset(const View &v, int i, const Node &n)
{
    v.a[i - v.off] = n;
}
__device__ __forceinline__ void nswap(const View &v, int i, int j)
{
    const Node t = v.a[i - v.off];
    v.a[i - v.off] = v.a[j - v.off];
    v.a[j - v.off] = t;
}

__device__ __forceinline__ int floor_log2(int n)
{
    return 31 - __clz(n);
}

// ------------------------------------------------------------------------------------------------
// sequential restatement (one thread): libstdc++ 11 bits/stl_algo.h / bits/stl_heap.h
// ------------------------------------------------------------------------------------------------
__device__ void seq_push_heap(const View &v, int f, int hole, int top, const Node &value, int axis)
{
    int parent = (hole - 1) / 2;
    while (hole > top && nkey(v, f + parent, axis) < akey(value, axis))
    {
        nset(v, f + hole, nget(v, f + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    nset(v, f + hole, value);
}

__device__ void seq_adjust_heap(const View &v, int f, int hole, int len, const Node &value, int axis)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2)
    {
        child = 2 * (child + 1);
        if (nkey(v, f + child, axis) < nkey(v, f + child - 1, axis))
            child--;
        nset(v, f + hole, nget(v, f + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2)
    {
        child = 2 * (child + 1);
        nset(v, f + hole, nget(v, f + child - 1));
        hole = child - 1;
    }
    seq_push_heap(v, f, hole, top, value, axis);
}

__device__ void seq_heap_select(const View &v, int first, int middle, int last, int axis)
{
    const int len = middle - first;
    if (len >= 2)
    {
        int parent = (len - 2) / 2;
        for (;;)
        {
            const Node value = nget(v, first + parent);
            seq_adjust_heap(v, first, parent, len, value, axis);
            if (parent == 0)
                break;
            parent--;
        }
    }
    for (int i = middle; i < last; ++i)
        if (nkey(v, i, axis) < nkey(v, first, axis))
        {
            const Node value = nget(v, i);
            nset(v, i, nget(v, first));
            seq_adjust_heap(v, first, 0, len, value, axis);
        }
}

__device__ void seq_insertion_sort(const View &v, int first, int last, int axis)
{
    if (first == last)
        return;
    for (int i = first + 1; i != last; ++i)
    {
        const Node val = nget(v, i);
        if (akey(val, axis) < nkey(v, first, axis))
        {
            for (int k = i; k > first; --k)
                nset(v, k, nget(v, k - 1));
            nset(v, first, val);
        }
        else
        {
            int l = i, nx = i - 1;
            while (akey(val, axis) < nkey(v, nx, axis))
            {
                nset(v, l, nget(v, nx));
                l = nx;
                --nx;
            }
            nset(v, l, val);
        }
    }
}

__device__ void seq_median_to_first(const View &v, int first, int last, int axis)
{
    const int mid = first + (last - first) / 2;
    const int A = first + 1, B = mid, C = last - 1;
    const float ka = nkey(v, A, axis), kb = nkey(v, B, axis), kc = nkey(v, C, axis);
    int pick;
    if (ka < kb)
    {
        if (kb < kc)
            pick = B;
        else if (ka < kc)
            pick = C;
        else
            pick = A;
    }
    else if (ka < kc)
        pick = A;
    else if (kb < kc)
        pick = C;
    else
        pick = B;
    nswap(v, first, pick);
}

__device__ int seq_partition_pivot(const View &v, int first, int last, int axis)
{
    seq_median_to_first(v, first, last, axis);
    const float pv = nkey(v, first, axis);
    int f = first + 1, l = last;
    for (;;)
    {
        while (nkey(v, f, axis) < pv)
            ++f;
        --l;
        while (pv < nkey(v, l, axis))
            --l;
        if (!(f < l))
            return f;
        nswap(v, f, l);
        ++f;
    }
}

// the loop of __introselect from a given state
__device__ void seq_introselect(const View &v, int first, int nth, int last, int depth_limit, int axis)
{
    while (last - first > 3)
    {
        if (depth_limit == 0)
        {
            seq_heap_select(v, first, nth + 1, last, axis);
            nswap(v, first, nth);
            return;
        }
        --depth_limit;
        const int cut = seq_partition_pivot(v, first, last, axis);
        if (cut <= nth)
            first = cut;
        else
            last = cut;
    }
    seq_insertion_sort(v, first, last, axis);
}

__device__ void seq_nth_element(const View &v, int first, int nth, int last, int axis)
{
    if (first == last || nth == last)
        return;
    seq_introselect(v, first, nth, last, 2 * floor_log2(last - first), axis);
}

// whole subtree of range [b,e) at `depth`, one thread
__device__ void seq_build_subtree(const View &v, int b, int e, int depth)
{
    int sb[24], se[24], sd[24];
    int sp = 0;
    sb[sp] = b;
    se[sp] = e;
    sd[sp] = depth;
    ++sp;
    while (sp)
    {
        --sp;
        const int rb = sb[sp], re = se[sp], rd = sd[sp];
        if (rb >= re)
            continue;
        const int mid = rb + (re - rb) / 2;
        seq_nth_element(v, rb, mid, re, rd % 3);
        if (mid > rb)
        {
            sb[sp] = rb;
            se[sp] = mid;
            sd[sp] = rd + 1;
            ++sp;
        }
        if (mid + 1 < re)
        {
            sb[sp] = mid + 1;
            se[sp] = re;
            sd[sp] = rd + 1;
            ++sp;
        }
    }
}

// __unguarded_partition_pivot(first, last) by a group of G threads; returns the cut
#ifdef LPX_KD_PROF
#define KD_LAP(acc, t)                                                                                                \
    do                                                                                                                \
    {                                                                                                                 \
        const unsigned long long n_ = clock64();                                                                      \
        (acc) += n_ - (t);                                                                                            \
        (t) = n_;                                                                                                     \
    } while (0)
__device__ unsigned long long kd_pf[8];  // median, flags, swaps, cut (cycles of thread 0 of the profiled workgroup)
#else
#define KD_LAP(acc, t) ((void)0)
#endif

template <int G>
__device__ int coop_partition_pivot(const View &v, int first, int last, int axis, int tid, uint32_t *cs, float *kb,
                                    bool wide = false)
{
#ifdef LPX_KD_PROF
    unsigned long long pt = clock64();
    const bool pf = tid == 0 && blockIdx.x == 0 && blockIdx.z == 0;
#define KD_P(i) if (pf) KD_LAP(kd_pf[i], pt)
#else
#define KD_P(i) ((void)0)
#endif
    if (tid == 0)
        seq_median_to_first(v, first, last, axis);
    Coop<G>::sync();
    const float pv = nkey(v, first, axis);
    KD_P(0);
    int cntL = 0, cntR = 0;
    // flag pass, PE consecutive positions per thread (one scan per G*PE keys).  The keys come through LDS: the group
    // reads them with consecutive lanes on consecutive nodes (a thread fetching its own four 16-byte nodes makes
    // every lane a separate 64-byte request -- or a 16-way bank conflict once the nodes are staged -- and that was
    // two thirds of this kernel's time), then every thread takes its four as one 16-byte LDS read.
    constexpr int PE = 4;
    // Ranges in GLOBAL memory (the first rounds of the upper levels: v.off == 0 and the nodes are not staged): eight
    // rows of G positions per step -- eight independent key loads per thread in flight, ranks by ballot, the row x
    // wavefront counts scanned once per step.  Per 8192 positions: one round trip and three barriers, where the
    // four-keys-per-thread form below pays two of each set; the stop lists it writes are the same.
    if (G > WAVE && wide)
    {
        constexpr int R = 8, NW = G / WAVE;
        static_assert(G == WAVE || R * NW <= 2 * WAVE, "the step's count table is scanned by two wavefronts");
        uint32_t *tab = (uint32_t *)kb;   // [R][NW] packed counts of a step: left stops | right stops << 16
        uint32_t *tab2 = tab + R * NW;    // their exclusive prefix, [R * NW] = the step's totals
        const uint32_t w = (uint32_t)tid / WAVE, lane = (uint32_t)tid % WAVE;
        const unsigned long long lt = lpx_lanemask_lt();
        for (int base = first + 1; base < last; base += G * R)
        {
            float kk[R];
#pragma unroll
            for (int j = 0; j < R; ++j)
            {
                const int p = base + j * G + tid;
                kk[j] = p < last ? nkey(v, p, axis) : pv;
            }
            unsigned long long bL[R], bR[R];
#pragma unroll
            for (int j = 0; j < R; ++j)
            {
                const bool valid = base + j * G + tid < last;
                bL[j] = __ballot(valid && !(kk[j] < pv));  // left cursor stops here
                bR[j] = __ballot(valid && !(pv < kk[j]));  // right cursor stops here
                if (lane == 0)
                    tab[j * NW + w] = (uint32_t)__popcll(bL[j]) | ((uint32_t)__popcll(bR[j]) << 16);
            }
            __syncthreads();
            uint32_t mine = 0, incl = 0;
            if (tid < R * NW)
            {
                mine = tab[tid];
                incl = lpx_wave_incl_scan_u32(mine);  // at most 8192 stops per step: the halves do not carry
                if (lane == WAVE - 1)
                    cs[32 + w] = incl;
            }
            __syncthreads();
            if (tid < R * NW)
            {
                const uint32_t add = w ? cs[32] : 0u;
                tab2[tid] = incl - mine + add;
                if (tid == R * NW - 1)
                    tab2[R * NW] = incl + add;
            }
            __syncthreads();
            const uint32_t total = tab2[R * NW];
#pragma unroll
            for (int j = 0; j < R; ++j)
            {
                const int p = base + j * G + tid;
                const uint32_t off = tab2[j * NW + w];
                if ((bL[j] >> lane) & 1ull)
                    v.lp[first + cntL + (int)(off & 0xffffu) + __popcll(bL[j] & lt) - v.off] = (uint32_t)p;
                if ((bR[j] >> lane) & 1ull)
                    v.ra[first + cntR + (int)(off >> 16) + __popcll(bR[j] & lt) - v.off] = (uint32_t)p;
            }
            cntL += (int)(total & 0xffffu);
            cntR += (int)(total >> 16);
        }
    }
    else
    for (int base = first + 1; base < last; base += G * PE)
    {
#pragma unroll
        for (int j = 0; j < PE; ++j)
        {
            const int p = base + j * G + tid;
            if (p < last)
                kb[j * G + tid] = nkey(v, p, axis);
        }
        Coop<G>::sync();
        const int p0 = base + tid * PE;
        uint32_t gem = 0, lem = 0;
        if (p0 < last)
        {
            float kk[PE];
            const float4 k4 = *(const float4 *)&kb[tid * PE];
            kk[0] = k4.x, kk[1] = k4.y, kk[2] = k4.z, kk[3] = k4.w;
#pragma unroll
            for (int e = 0; e < PE; ++e)
                kk[e] = (p0 + e < last) ? kk[e] : pv;
#pragma unroll
            for (int e = 0; e < PE; ++e)
            {
                const bool valid = p0 + e < last;
                gem |= ((valid && !(kk[e] < pv)) ? 1u : 0u) << e;  // left cursor stops here
                lem |= ((valid && !(pv < kk[e])) ? 1u : 0u) << e;  // right cursor stops here
            }
        }
        const uint32_t packed = (uint32_t)__popc(gem) + ((uint32_t)__popc(lem) << 16);
        uint32_t excl, total;
        Coop<G>::scan_packed(packed, excl, total, cs);
        int rL = first + cntL + (int)(excl & 0xffffu) - v.off;
        int rR = first + cntR + (int)(excl >> 16) - v.off;
        while (gem)
        {
            const int e = __ffs(gem) - 1;
            gem &= gem - 1;
            v.lp[rL++] = (uint32_t)(p0 + e);
        }
        while (lem)
        {
            const int e = __ffs(lem) - 1;
            lem &= lem - 1;
            v.ra[rR++] = (uint32_t)(p0 + e);
        }
        cntL += (int)(total & 0xffffu);
        cntR += (int)(total >> 16);
    }
    Coop<G>::sync();
    KD_P(1);
    const int kmax = min(cntL, cntR);
    uint32_t my = 0;
    for (int k0 = tid; k0 < kmax; k0 += 4 * G)
    {
        // four swaps per trip with batched loads so the memory latencies overlap
        int sl[4], sr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const int k = k0 + q * G;
            const bool in = k < kmax;
            sl[q] = in ? (int)v.lp[first + k - v.off] : 0;
            sr[q] = in ? (int)v.ra[first + cntR - 1 - k - v.off] : -1;
        }
        Node nl[4], nr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (sl[q] < sr[q])
            {
                nl[q] = nget(v, sl[q]);
                nr[q] = nget(v, sr[q]);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (sl[q] < sr[q])
            {
                nset(v, sl[q], nr[q]);
                nset(v, sr[q], nl[q]);
                ++my;
            }
    }
    const int K = (int)Coop<G>::sum(my, cs);
    KD_P(2);
    const int c1 = (K < cntL) ? (int)v.lp[first + K - v.off] : INT_MAX;
    const int c2 = (K > 0) ? (int)v.ra[first + cntR - K - v.off] : INT_MAX;
    Coop<G>::sync();
    KD_P(3);
    return min(c1, c2);
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// state of one range in the multi-workgroup top-level rounds (kd_top_* below)
struct KdTopState
{
    int first, last, nth, depth;  // introselect loop state of the range (bits/stl_algo.h:1964-1986)
    int cntL, cntR, K;            // stop-list sizes and swaps of the round in flight
    int active;                   // the range is still partitioned here (else kd_block_kernel finishes it)
    int pending;                  // a cut is waiting to be applied
    float pv;
    int pad[2];
};
constexpr int TOP_TILE = 8192;        // positions per workgroup in the flag / list passes
constexpr int TOP_THREADS = 256;
// A round of the multi-workgroup form is four launches (~38 us), a round inside kd_block_kernel's single workgroup ~10 us
// on 32k nodes: ranges at or below TOP_HAND are left to kd_block_kernel, and only TOP_EXTRA rounds are added to the
// expected number (a range that shrinks slower than 0.6 per round is simply handed over larger).  Measured on 5M-point
// frames: 4096 / 6 (round 2) kd build 4.46 ms, 599 Mpts/s, ~410 launches per frame; 32768 / 1: 3.78 ms, 653 Mpts/s, ~190.
constexpr int TOP_HAND = 32768;
constexpr int TOP_EXTRA = 1;
constexpr uint32_t TOP_MIN = 131072;  // levels whose ranges can exceed this take the multi-workgroup rounds

constexpr int BLK_G_MAX = 1024;
#ifndef LPX_BLK_CAP_BATCH
#define LPX_BLK_CAP_BATCH 1984
#endif
constexpr int BLK_CAP_BATCH = LPX_BLK_CAP_BATCH;  // batches: kd_lds_kernel needs 20 B x 1984 + 256 B = 39 936 B, four workgroups per CU
                                     // (2032 nodes = 40 896 B measured as three per CU: 402 against 273 us per chain)
constexpr int BLK_CAP_MAX = 4096;  // most nodes staged in LDS: 64 KiB + 2 x 16 KiB scratch (blk_cap is a launch argument)
constexpr int BLK_TAIL = 1024;  // batches, upper levels: the active range is staged in LDS once it is this small
constexpr int WAVE_TAIL = 1024;  // kd_block_kernel: a staged range this small is finished by one wavefront
// batches: levels whose ranges may exceed BLK_WIDE nodes get 1024-thread workgroups -- none does (it was 80 000): under
// load a 1024-thread workgroup waits for a whole compute unit, and 256 threads on every level are as fast for 120k-point
// chains (1 990 against 1 987 Mpts/s) and 3 % faster for 1M-point ones (1 518 against 1 450-1 488, tools/r4_probe25.sh)
constexpr uint32_t BLK_WIDE = 0xffffffffu;
constexpr uint32_t BLK_MID = 0;       // ... this many 256 threads, shorter ones a single wavefront
constexpr int SUB_LEAF = 4;    // at or below this one lane finishes a subtree on its own

// one workgroup per range of `level`: std::nth_element(b, mid, e) on axis level % 3
// BLK_CAP: the LDS capacity that decides which levels belong to kd_lds_kernel; STAGE_CAP: what THIS launch may
// stage in LDS (0 for the top levels, whose ranges are far above the capacity: their workgroups then need no LDS
// and find a CU at once even when other chains fill the device)
template <int BLK_G>  // threads of the workgroup that owns a range: 1024 while the ranges are long, 256 below
__global__ __launch_bounds__(BLK_G) void kd_block_kernel(Node *nodes, uint32_t *lpos, uint32_t *rasc,
                                                          const FrameState *__restrict__ frame, int level,
                                                          int BLK_CAP, int STAGE_CAP,
                                                          const KdTopState *__restrict__ top, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<3>(fs);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nodes = lpx_slot(nodes, fs);
    lpos = lpx_slot(lpos, fs);
    rasc = lpx_slot(rasc, fs);
    frame = lpx_slot(frame, fs);
    Node *l_nodes = (Node *)smem;
    uint32_t *l_lp = (uint32_t *)(smem + sizeof(Node) * STAGE_CAP);
    uint32_t *l_ra = l_lp + STAGE_CAP;
    uint32_t *cs = l_ra + STAGE_CAP;
    float *kb = (float *)(cs + 64);  // BLK_G x 4 keys of a flag-pass step (16-byte aligned: cs is)

    const int tid = threadIdx.x;
    int b = 0, e = (int)frame->n_obstacle;
    if ((e >> level) <= BLK_CAP)
        return;  // this level already belongs to kd_lds_kernel (the host planned with an upper bound)
    descend(b, e, lpx_blk.x, level);
    if (e - b < 2)
        return;
    const int axis = level % 3;
    const int nth = b + (e - b) / 2;
    int first = b, last = e;
    int depth_limit = 2 * floor_log2(e - b);
    if (top)
    {
        // the multi-workgroup rounds (kd_top_*) have narrowed the range: continue the same introselect loop
        const KdTopState st = lpx_slot(top, fs)[lpx_blk.x];
        first = st.first;
        last = st.last;
        depth_limit = st.depth;
    }
    View v;
    v.a = nodes;
    v.lp = lpos;
    v.ra = rasc;
    v.off = 0;
    bool staged = false;
    int sb = 0, se = 0;
    bool done = false;
#ifdef LPX_KD_PROF
    const bool pf = tid == 0 && blockIdx.x == 0 && blockIdx.z == 0;
    unsigned long long t_all = clock64(), t_g = 0, t_l = 0, t_lap = t_all;
    int r_g = 0, r_l = 0;
    if (pf)
        for (int i = 0; i < 8; ++i)
            kd_pf[i] = 0;
#endif
    while (last - first > 3)
    {
        if (!staged && last - first <= STAGE_CAP)
        {
            sb = first;
            se = last;
            for (int i = first + tid; i < last; i += BLK_G)
                l_nodes[i - first] = nodes[i];
            v.a = l_nodes;
            v.lp = l_lp;
            v.ra = l_ra;
            v.off = first;
            staged = true;
            Coop<BLK_G>::sync();
        }
        // A staged range of at most WAVE_TAIL nodes is finished by ONE wavefront: sixteen wavefronts meeting at eight
        // barriers per round cost ~6.8k cycles per round for a few hundred nodes (measured), and every nth_element
        // ends with about ten such rounds; a single wavefront needs no barrier at all.
        if (staged && last - first <= WAVE_TAIL)
        {
            if (tid < WAVE)
            {
                while (last - first > 3)
                {
                    if (depth_limit == 0)
                    {
                        if (tid == 0)
                        {
                            seq_heap_select(v, first, nth + 1, last, axis);
                            nswap(v, first, nth);
                        }
                        done = true;
                        break;
                    }
                    --depth_limit;
                    const int cut = coop_partition_pivot<WAVE>(v, first, last, axis, tid, cs, kb);
                    if (cut <= nth)
                        first = cut;
                    else
                        last = cut;
                }
            }
            else
                done = true;  // (the other wavefronts only wait for the write-back)
            break;
        }
        if (depth_limit == 0)
        {
            if (tid == 0)
            {
                seq_heap_select(v, first, nth + 1, last, axis);
                nswap(v, first, nth);
            }
            done = true;
            break;
        }
        --depth_limit;
#ifdef LPX_KD_PROF
        t_lap = clock64();
#endif
        const int cut = coop_partition_pivot<BLK_G>(v, first, last, axis, tid, cs, kb, !staged);
#ifdef LPX_KD_PROF
        if (staged)
            t_l += clock64() - t_lap, ++r_l;
        else
            t_g += clock64() - t_lap, ++r_g;
#endif
        if (cut <= nth)
            first = cut;
        else
            last = cut;
    }
    if (!done && tid == 0)
        seq_insertion_sort(v, first, last, axis);
    Coop<BLK_G>::sync();
    if (staged)
        for (int i = sb + tid; i < se; i += BLK_G)
            nodes[i] = l_nodes[i - sb];
#ifdef LPX_KD_PROF
    if (pf)
        printf("kd_block level %d range %d: total %llu cycles; %d global rounds %llu, %d LDS rounds %llu; median %llu flags %llu "
               "swaps %llu cut %llu\n", level, e - b, clock64() - t_all, r_g, t_g, r_l, t_l, kd_pf[0], kd_pf[1], kd_pf[2],
               kd_pf[3]);
#endif
}

// ------------------------------------------------------------------------------------------------
// Top levels of LARGE clouds: one std::nth_element shared by many workgroups.
//
// kd_block_kernel gives a whole range to ONE workgroup; at level 0 of a 2.3M-point obstacle cloud that single
// workgroup sweeps 37 MB per Hoare round while 255 CUs idle (15 of the 37 ms of a 5M-point frame).  The same
// data-parallel Hoare partition distributes over workgroups when its phases become launches: per round
//   kd_top_pivot   one thread per range: applies the cut of the previous round (first / last), then
//                  median-of-three to first, pivot value, counters reset            (bits/stl_algo.h:1878-1907)
//   kd_top_flags   tiles of 8192 positions: how many keys stop the left / the right cursor   -> per-tile counts
//   kd_top_lists   the same tiles, prefix over the tile counts, write the stop lists L (ascending), R (ascending)
//   kd_top_swap    swaps L_k <-> R_k for k < min(|L|, |R|) with L_k < R_k, counts them (K)
// and the cut = min(L_{K+1}, R_K) is taken by the next kd_top_pivot.  The introselect state of every range
// (first, last, depth limit) lives in a small table; after a fixed number of rounds kd_block_kernel continues
// from that state (the active range is then a few thousand nodes and fits its LDS), so the result is the same
// permutation as before -- the rounds only run on more CUs.
// ------------------------------------------------------------------------------------------------
__global__ void kd_top_pivot(Node *nodes, const uint32_t *__restrict__ lpos, const uint32_t *__restrict__ rasc,
                             const FrameState *__restrict__ frame, KdTopState *state, int level, int init, int hand,
                             size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<3>(fs);
    nodes = lpx_slot(nodes, fs);
    lpos = lpx_slot(lpos, fs);
    rasc = lpx_slot(rasc, fs);
    frame = lpx_slot(frame, fs);
    state = lpx_slot(state, fs);
    const uint32_t r = lpx_blk.x * blockDim.x + threadIdx.x;
    if (r >= (1u << level))
        return;
    KdTopState st = state[r];
    const int axis = level % 3;
    View v;
    v.a = nodes;
    v.lp = nullptr;
    v.ra = nullptr;
    v.off = 0;
    if (init == 1)
    {
        int b = 0, e = (int)frame->n_obstacle;
        descend(b, e, r, level);
        st.first = b;
        st.last = e;
        st.nth = b + (e - b) / 2;
        st.depth = (e - b >= 2) ? 2 * floor_log2(e - b) : 0;
        st.pending = 0;
        st.active = 1;
    }
    else if (st.active && st.pending)
    {
        // cut of the round that just ran: min(L_{K+1}, R_K)
        const int c1 = (st.K < st.cntL) ? (int)lpos[st.first + st.K] : INT_MAX;
        const int c2 = (st.K > 0) ? (int)rasc[st.first + st.cntR - st.K] : INT_MAX;
        const int cut = min(c1, c2);
        if (cut <= st.nth)
            st.first = cut;
        else
            st.last = cut;
        st.pending = 0;
    }
    // the range stays here while it is large and the depth limit has not run out (heap_select, rare, is left to
    // kd_block_kernel together with everything small)
    if (st.active && (st.last - st.first <= hand || st.depth == 0))
        st.active = 0;
    if (st.active && init != 2)  // init == 2: the last call only applies the last cut, it starts no round
    {
        --st.depth;
        seq_median_to_first(v, st.first, st.last, axis);
        st.pv = nkey(v, st.first, axis);
        st.cntL = st.cntR = st.K = 0;
        st.pending = 1;
    }
    state[r] = st;
}

// per tile: number of positions in (first, last) whose key stops the left cursor (>= pivot) / the right one (<= pivot)
__global__ __launch_bounds__(TOP_THREADS) void kd_top_flags(const Node *__restrict__ nodes,
                                                            const KdTopState *__restrict__ state,
                                                            uint2 *__restrict__ tile_cnt, int level, int tiles, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<3>(fs);
    __shared__ uint32_t s_l[TOP_THREADS / WAVE], s_r[TOP_THREADS / WAVE];
    nodes = lpx_slot(nodes, fs);
    state = lpx_slot(state, fs);
    tile_cnt = lpx_slot(tile_cnt, fs);
    const uint32_t r = lpx_blk.y;
    const KdTopState st = state[r];
    if (!st.active)
        return;
    const int axis = level % 3;
    const int p0 = st.first + 1 + (int)lpx_blk.x * TOP_TILE;
    uint32_t cl = 0, cr = 0;
    if (p0 < st.last)
    {
        const int p1 = min(p0 + TOP_TILE, st.last);
        for (int p = p0 + (int)threadIdx.x; p < p1; p += TOP_THREADS)
        {
            const float k = ((const float *)(nodes + p))[axis];
            cl += !(k < st.pv);
            cr += !(st.pv < k);
        }
    }
    cl = lpx_wave_sum_u32(cl);
    cr = lpx_wave_sum_u32(cr);
    if ((threadIdx.x % WAVE) == 0)
    {
        s_l[threadIdx.x / WAVE] = cl;
        s_r[threadIdx.x / WAVE] = cr;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t a = 0, b = 0;
        for (int i = 0; i < TOP_THREADS / WAVE; ++i)
        {
            a += s_l[i];
            b += s_r[i];
        }
        tile_cnt[(size_t)r * tiles + lpx_blk.x] = make_uint2(a, b);
    }
}

// the stop lists: lpos[first + i] = i-th position (ascending) with key >= pivot, rasc likewise for key <= pivot
__global__ __launch_bounds__(TOP_THREADS) void kd_top_lists(const Node *__restrict__ nodes, KdTopState *state,
                                                            const uint2 *__restrict__ tile_cnt,
                                                            uint32_t *__restrict__ lpos, uint32_t *__restrict__ rasc,
                                                            int level, int tiles, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<3>(fs);
    __shared__ uint32_t s_a[TOP_THREADS / WAVE], s_b[TOP_THREADS / WAVE];
    __shared__ uint32_t s_base[2];
    nodes = lpx_slot(nodes, fs);
    state = lpx_slot(state, fs);
    tile_cnt = lpx_slot(tile_cnt, fs);
    lpos = lpx_slot(lpos, fs);
    rasc = lpx_slot(rasc, fs);
    const uint32_t r = lpx_blk.y;
    const KdTopState st = state[r];
    if (!st.active)
        return;
    const int axis = level % 3;
    const int span = st.last - st.first - 1;
    const int used = (span + TOP_TILE - 1) / TOP_TILE;  // tiles that hold positions this round
    if ((int)lpx_blk.x >= used)
        return;
    // exclusive prefix of the tile counts before this tile (and, in the last tile, the totals)
    uint32_t bl = 0, br = 0;
    for (int t = (int)threadIdx.x; t < (int)lpx_blk.x; t += TOP_THREADS)
    {
        const uint2 c = tile_cnt[(size_t)r * tiles + t];
        bl += c.x;
        br += c.y;
    }
    bl = lpx_wave_sum_u32(bl);
    br = lpx_wave_sum_u32(br);
    if ((threadIdx.x % WAVE) == 0)
    {
        s_a[threadIdx.x / WAVE] = bl;
        s_b[threadIdx.x / WAVE] = br;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t a = 0, b = 0;
        for (int i = 0; i < TOP_THREADS / WAVE; ++i)
        {
            a += s_a[i];
            b += s_b[i];
        }
        s_base[0] = a;
        s_base[1] = b;
        if ((int)lpx_blk.x == used - 1)
        {
            const uint2 c = tile_cnt[(size_t)r * tiles + lpx_blk.x];
            state[r].cntL = (int)(a + c.x);
            state[r].cntR = (int)(b + c.y);
        }
    }
    __syncthreads();
    uint32_t runL = s_base[0], runR = s_base[1];
    const int p0 = st.first + 1 + (int)lpx_blk.x * TOP_TILE;
    const int p1 = min(p0 + TOP_TILE, st.last);
    const unsigned long long lt = lpx_lanemask_lt();
    const uint32_t w = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    // positions in ascending order: chunks of 256, wavefront w takes the w-th 64 of every chunk
    for (int c0 = p0; c0 < p1; c0 += TOP_THREADS)
    {
        const int p = c0 + (int)threadIdx.x;
        bool ge = false, le = false;
        if (p < p1)
        {
            const float k = ((const float *)(nodes + p))[axis];
            ge = !(k < st.pv);
            le = !(st.pv < k);
        }
        const unsigned long long mg = __ballot(ge), ml = __ballot(le);
        if (lane == 0)
        {
            s_a[w] = (uint32_t)__popcll(mg);
            s_b[w] = (uint32_t)__popcll(ml);
        }
        __syncthreads();
        uint32_t ol = runL, orr = runR, tl = 0, tr = 0;
        for (uint32_t i = 0; i < TOP_THREADS / WAVE; ++i)
        {
            if (i < w)
            {
                ol += s_a[i];
                orr += s_b[i];
            }
            tl += s_a[i];
            tr += s_b[i];
        }
        if (ge)
            lpos[st.first + ol + __popcll(mg & lt)] = (uint32_t)p;
        if (le)
            rasc[st.first + orr + __popcll(ml & lt)] = (uint32_t)p;
        runL += tl;
        runR += tr;
        __syncthreads();
    }
}

__global__ __launch_bounds__(TOP_THREADS) void kd_top_swap(Node *nodes, KdTopState *state,
                                                           const uint32_t *__restrict__ lpos,
                                                           const uint32_t *__restrict__ rasc, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<3>(fs);
    nodes = lpx_slot(nodes, fs);
    state = lpx_slot(state, fs);
    lpos = lpx_slot(lpos, fs);
    rasc = lpx_slot(rasc, fs);
    const uint32_t r = lpx_blk.y;
    const KdTopState st = state[r];
    if (!st.active)
        return;
    const int kmax = min(st.cntL, st.cntR);
    uint32_t my = 0;
    for (int k = (int)(lpx_blk.x * blockDim.x + threadIdx.x); k < kmax; k += (int)(gridDim.x * blockDim.x))
    {
        const int sl = (int)lpos[st.first + k], sr = (int)rasc[st.first + st.cntR - 1 - k];
        if (sl < sr)
        {
            const Node a = nodes[sl], b = nodes[sr];
            nodes[sl] = b;
            nodes[sr] = a;
            ++my;
        }
    }
    my = lpx_wave_sum_u32(my);
    if ((threadIdx.x % WAVE) == 0 && my)
        atomicAdd(&state[r].K, (int)my);
}

// ------------------------------------------------------------------------------------------------
// Whole subtree of a range that fits LDS (<= blk_cap nodes), one workgroup.
//
// Sub-level s has 2^s independent ranges; they are partitioned SIMULTANEOUSLY by 2^s groups of
// 1024 >> s consecutive threads (whole wavefronts while the group has >= 64 threads, lane segments of
// a wavefront below that), so a sub-level costs one nth_element's worth of rounds instead of 2^s.
// With PE = 4 keys per thread a group always covers its range in a single flag pass
// (group size * 4 >= range size), so every introselect round has the same fixed shape:
// median-of-3 (group leader) | flags + segmented scan | stop lists | swaps + count | cut.
// Below SUB_LEAF nodes one lane finishes a subtree sequentially.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void group_scan_packed(uint32_t v, int gs, int tid, uint32_t *cs, bool blockmode,
                                                  uint32_t &excl, uint32_t &total)
{
    const uint32_t incl = lpx_wave_incl_scan_u32(v);
    if (blockmode)
    {
        const int w = tid / WAVE;
        if ((tid % WAVE) == WAVE - 1)
            cs[w] = incl;
        __syncthreads();
        const int gwn = gs / WAVE, gw0 = (tid / gs) * gwn;
        uint32_t bsum = 0, ssum = 0;
        for (int i = 0; i < gwn; ++i)
        {
            const uint32_t c = cs[gw0 + i];
            if (gw0 + i < w)
                bsum += c;
            ssum += c;
        }
        __syncthreads();
        excl = bsum + incl - v;
        total = ssum;
    }
    else
    {
        const int lane = tid % WAVE;
        const int g0 = lane & ~(gs - 1);
        const uint32_t before_raw = __shfl(incl, g0 > 0 ? g0 - 1 : 0, 64);
        const uint32_t before = g0 > 0 ? before_raw : 0u;
        const uint32_t lastv = __shfl(incl, g0 + gs - 1, 64);
        excl = incl - v - before;
        total = lastv - before;
    }
}

#ifndef LPX_KD_LG
#define LPX_KD_LG 256
#endif
constexpr int LG = LPX_KD_LG;  // threads of kd_lds_kernel: one wavefront per SIMD, little per-round overhead

template <typename PosT>  // stop-list entries: uint16_t (batches: 20 B of LDS per node) or uint32_t (a single frame)
__global__ __launch_bounds__(LG) void kd_lds_kernel(Node *nodes, Node *__restrict__ PR,
                                                    const FrameState *__restrict__ frame,
                                                    uint32_t *__restrict__ dbg, int BLK_CAP,
                                                    uint32_t *__restrict__ parent, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<3>(fs);
    nodes = lpx_slot(nodes, fs);
    PR = lpx_slot(PR, fs);
    frame = lpx_slot(frame, fs);
    parent = lpx_slot(parent, fs);  // (search path: every point its own set before nb_index_kernel links them)
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t n_rounds = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Node *l_nodes = (Node *)smem;
    // stop lists as positions inside the staged range (< 4096); 16-bit entries make it 20 bytes of LDS per node, so
    // that four workgroups of a batch share a CU
    PosT *l_lp = (PosT *)(smem + sizeof(Node) * BLK_CAP);
    PosT *l_ra = l_lp + BLK_CAP;
    uint32_t *cs = (uint32_t *)(l_ra + BLK_CAP);
    const int tid = threadIdx.x;
    const int M = (int)frame->n_obstacle;
    int lv = 0;
    while ((M >> lv) > BLK_CAP)
        ++lv;
    if (lpx_blk.x >= (1u << lv))
        return;
    int b = 0, e = M;
    descend(b, e, lpx_blk.x, lv);
    const int n = e - b;
    // the split nodes above this kernel's level are final already: block 0 copies them to the
    // pre-order layout (every other node is copied by the block that owns its range)
    if (lpx_blk.x == 0)
        for (uint32_t h = tid; h + 1 < (1u << lv); h += LG)
        {
            const int l = 31 - __clz(h + 1);
            int tb = 0, te = M;
            descend(tb, te, h + 1 - (1u << l), l);
            if (tb < te)
            {
                const uint32_t mid = (uint32_t)(tb + (te - tb) / 2);
                const Node nm = nodes[mid];
                PR[kd_rank_of(mid, (uint32_t)M)] = nm;
                if (parent)
                    parent[__float_as_uint(nm.w)] = __float_as_uint(nm.w);
            }
        }
    if (n < 1)
        return;
    if (n == 1)
    {
        if (tid == 0)
        {
            const Node n1 = nodes[b];
            PR[kd_rank_of((uint32_t)b, (uint32_t)M)] = n1;
            if (parent)
                parent[__float_as_uint(n1.w)] = __float_as_uint(n1.w);
        }
        return;
    }
    for (int i = tid; i < n; i += LG)
        l_nodes[i] = nodes[b + i];
    View v;
    v.a = l_nodes;
    v.lp = nullptr;  // (the sequential helpers only touch the nodes)
    v.ra = nullptr;
    v.off = b;
    __syncthreads();

    int s = 0;
    for (; (n >> s) > SUB_LEAF && (LG >> s) >= 1; ++s)
    {
        const int gs = LG >> s;
        const bool blockmode = gs >= WAVE;
        const int g = tid / gs, gl = tid % gs;
        int rb = b, re = e;
        descend(rb, re, (uint32_t)g, s);
        const int axis = (lv + s) % 3;
        int first = rb, last = re;
        const int nth = rb + (re - rb) / 2;
        int depth_limit = (re - rb >= 2) ? 2 * floor_log2(re - rb) : 0;
        bool done = (re - rb) < 2;
        for (;;)
        {
            const bool act = !done && (last - first > 3);
            bool any;
            if (blockmode)
                any = __syncthreads_or(act ? 1 : 0) != 0;
            else
            {
                Coop<WAVE>::sync();
                any = __any(act ? 1 : 0) != 0;
            }
            if (!any)
                break;
            ++n_rounds;
            const bool part = act && depth_limit > 0;
            if (act && depth_limit == 0)
            {
                if (gl == 0)
                {
                    seq_heap_select(v, first, nth + 1, last, axis);
                    nswap(v, first, nth);
                }
                done = true;
            }
            if (part)
            {
                --depth_limit;
                if (gl == 0)
                    seq_median_to_first(v, first, last, axis);
            }
            if (blockmode)
                __syncthreads();
            else
                Coop<WAVE>::sync();
            // flag pass: the group's span is cut into gs contiguous runs of `pe` keys (pe <= 16)
            const float pv = part ? nkey(v, first, axis) : 0.0f;
            const int span = part ? (last - first - 1) : 0;
            const int pe = (span + gs - 1) / gs;
            const int p0 = first + 1 + gl * pe;
            uint32_t gem = 0, lem = 0;
            for (int q = 0; q < pe; ++q)
            {
                const int p = p0 + q;
                if (p < last)
                {
                    const float k = nkey(v, p, axis);
                    gem |= (!(k < pv) ? 1u : 0u) << q;
                    lem |= (!(pv < k) ? 1u : 0u) << q;
                }
            }
            const uint32_t packed = (uint32_t)__popc(gem) + ((uint32_t)__popc(lem) << 16);
            uint32_t excl, total;
            group_scan_packed(packed, gs, tid, cs, blockmode, excl, total);
            const int cntL = (int)(total & 0xffffu), cntR = (int)(total >> 16);
            {
                int rL = first + (int)(excl & 0xffffu) - v.off;
                int rR = first + (int)(excl >> 16) - v.off;
                uint32_t m = gem;
                while (m)
                {
                    const int q = __ffs(m) - 1;
                    m &= m - 1;
                    l_lp[rL++] = (PosT)(p0 + q - v.off);
                }
                m = lem;
                while (m)
                {
                    const int q = __ffs(m) - 1;
                    m &= m - 1;
                    l_ra[rR++] = (PosT)(p0 + q - v.off);
                }
            }
            if (blockmode)
                __syncthreads();
            else
                Coop<WAVE>::sync();
            const int kmax = part ? min(cntL, cntR) : 0;
            uint32_t my = 0;
            for (int k0 = gl; k0 < kmax; k0 += 4 * gs)
            {
                // four swaps per trip with batched loads so the LDS latencies overlap
                int sl[4], sr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                    const int k = k0 + q * gs;
                    const bool in = k < kmax;
                    sl[q] = in ? (int)l_lp[first + k - v.off] + v.off : 0;
                    sr[q] = in ? (int)l_ra[first + cntR - 1 - k - v.off] + v.off : -1;
                }
                Node nl[4], nr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (sl[q] < sr[q])
                    {
                        nl[q] = nget(v, sl[q]);
                        nr[q] = nget(v, sr[q]);
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (sl[q] < sr[q])
                    {
                        nset(v, sl[q], nr[q]);
                        nset(v, sr[q], nl[q]);
                        ++my;
                    }
            }
            uint32_t e2, ktot;
            group_scan_packed(my, gs, tid, cs, blockmode, e2, ktot);
            if (part)
            {
                const int K = (int)ktot;
                const int c1 = (K < cntL) ? (int)l_lp[first + K - v.off] + v.off : INT_MAX;
                const int c2 = (K > 0) ? (int)l_ra[first + cntR - K - v.off] + v.off : INT_MAX;
                const int cut = min(c1, c2);
                if (cut <= nth)
                    first = cut;
                else
                    last = cut;
            }
        }
        if (!done && gl == 0)
            seq_insertion_sort(v, first, last, axis);
        __syncthreads();
        if (dbg && tid == 0 && lpx_blk.x == 0 && s < 12)
        {
            dbg[2 * s] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
            dbg[2 * s + 1] = n_rounds;
        }
    }
    // leaf phase: one lane per remaining subtree
    for (uint32_t j = tid; j < (1u << s); j += LG)
    {
        int rb = b, re = e;
        descend(rb, re, j, s);
        if (re - rb >= 2)
            seq_build_subtree(v, rb, re, lv + s);
    }
    __syncthreads();
    if (dbg && tid == 0 && lpx_blk.x == 0)
    {
        dbg[30] = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
        dbg[31] = (uint32_t)n;
    }
    for (int i = tid; i < n; i += LG)
    {
        const Node nd = l_nodes[i];
        nodes[b + i] = nd;
        PR[kd_rank_of((uint32_t)(b + i), (uint32_t)M)] = nd;  // pre-order rank layout for the neighbour search
        if (parent)
            parent[__float_as_uint(nd.w)] = __float_as_uint(nd.w);
    }
}

__global__ void layout_idx_kernel(const Node *__restrict__ nodes, uint32_t m, uint32_t *__restrict__ out)
{
    const LpxBlock lpx_blk = lpx_block<3>(0);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i < m)
        out[i] = __float_as_uint(nodes[i].w);
}
}  // namespace

int lpx_kd_layout_copy(lpx_ctx *ctx, uint32_t m, uint32_t *d_out)
{
    if (m)
        hipLaunchKernelGGL(layout_idx_kernel, dim3((m + 255) / 256), dim3(256), 0, ctx->stream,
                           (const Node *)ctx->nodes.p, m, d_out);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_kd_build(lpx_ctx *ctx, uint32_t m_max)
{
    if (m_max == 0)
        return LPX_OK;
    if (m_max >= (1u << 30))
        return lpx_fail(ctx, LPX_ERR_ARG, "clustering supports fewer than 2^30 points");
    const FrameState *frame = (const FrameState *)ctx->frame.p;
    Node *nodes = (Node *)ctx->nodes.p;
    uint32_t *lpos = (uint32_t *)ctx->lpos.p, *rasc = (uint32_t *)ctx->rpos.p;
    StageTimer tm(ctx, ST_KD_BUILD);
    // the node array {x, y, z, index} was written by the producer of the obstacle cloud (compact / ingest)
    // A single frame stages up to 4096 nodes per workgroup (96 KiB of LDS: fewest global-memory rounds, best
    // latency).  A batch shares the device with the small-LDS workgroups of other chains' neighbour kernels,
    // next to which a 96 KiB workgroup rarely finds room; half the capacity schedules freely.
    // (a single frame of more than a million points has hundreds of LDS subtrees: with 3968 nodes and 16-bit stop lists a
    // subtree takes 79.6 KB and TWO share a compute unit -- 5M-point frame: kd_lds_kernel 0.90 -> see docs/experiments.md)
    const bool big_single = ctx->cur_b == 1 && m_max >= (1u << 20);
    const int blk_cap = ctx->cur_b > 1 ? BLK_CAP_BATCH : (big_single ? 3968 : BLK_CAP_MAX);
    const size_t key_lds = sizeof(float) * BLK_G_MAX * 4;  // key buffer of kd_block_kernel's flag pass
    const size_t blk_lds = sizeof(Node) * blk_cap + 2 * sizeof(uint32_t) * blk_cap + 64 * sizeof(uint32_t);
    const size_t lds_lds = sizeof(Node) * blk_cap + 2 * sizeof(uint16_t) * blk_cap + 64 * sizeof(uint32_t);  // batches
    if (!ctx->attr_kd)
    {
        const size_t max_lds = sizeof(Node) * BLK_CAP_MAX + 2 * sizeof(uint32_t) * BLK_CAP_MAX + 64 * sizeof(uint32_t);
        LPX_HIP(ctx, hipFuncSetAttribute((const void *)kd_block_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(max_lds + key_lds)));
        LPX_HIP(ctx, hipFuncSetAttribute((const void *)kd_block_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(max_lds + key_lds)));
        LPX_HIP(ctx, hipFuncSetAttribute((const void *)kd_block_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(max_lds + key_lds)));
        LPX_HIP(ctx, hipFuncSetAttribute((const void *)kd_lds_kernel<uint32_t>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
        LPX_HIP(ctx, hipFuncSetAttribute((const void *)kd_lds_kernel<uint16_t>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
        ctx->attr_kd = true;
    }
    // global-memory levels while a range can exceed the LDS capacity, then the whole rest in one launch
    int level = 0;
    uint32_t size = m_max;
    while (size > (uint32_t)blk_cap)
    {
        // a batch stages nothing while the ranges are far above the LDS capacity (most rounds run in global memory
        // anyway): those workgroups need 256 bytes of LDS instead of 48 KiB
        // (round 5: staging only the tail on EVERY level of a batch, 29 instead of 52 KB of LDS on the lower levels, is
        // neither faster nor slower: 2 117 against 2 124 Mpts/s)
        const bool stage = ctx->cur_b == 1 || size <= 4u * (uint32_t)blk_cap;
        const KdTopState *top = nullptr;
        // LPX_KD_TOP_MIN overrides the size from which a level takes the multi-workgroup rounds (tests)
        static const uint32_t top_min = LPX_KNOB("LPX_KD_TOP_MIN") ? (uint32_t)atoi(LPX_KNOB("LPX_KD_TOP_MIN")) : TOP_MIN;
        if (size > top_min && size > (uint32_t)TOP_HAND && (sizeof(KdTopState) << level) <= ctx->kd_state.bytes)
        {
            // large ranges: the first rounds of every nth_element of this level on many workgroups
            KdTopState *state = (KdTopState *)ctx->kd_state.p;
            uint2 *tile_cnt = (uint2 *)ctx->key64_b.p;  // 64-bit key scratch of the segmentation: free here
            const int tiles = (int)((size + TOP_TILE - 1) / TOP_TILE);
            const uint32_t ranges = 1u << level;
            if (sizeof(uint2) * (size_t)tiles * ranges <= ctx->key64_b.bytes)
            {
                // the active range shrinks by ~0.6 per round; kd_block_kernel finishes whatever is left (from any state:
                // LPX_KD_HAND / LPX_KD_EXTRA only move work between the four-launch rounds and its single workgroup)
                static const int hand_env = LPX_KNOB("LPX_KD_HAND") ? atoi(LPX_KNOB("LPX_KD_HAND")) : TOP_HAND;
                static const int extra_env = LPX_KNOB("LPX_KD_EXTRA") ? atoi(LPX_KNOB("LPX_KD_EXTRA")) : TOP_EXTRA;
                const int hand = hand_env;
                int rounds = extra_env;
                for (uint32_t sz = size; sz > (uint32_t)hand; sz = sz * 3 / 5)
                    ++rounds;
                const dim3 gp((ranges + 63) / 64, 1, ctx->cur_b), gt(tiles, ranges, ctx->cur_b);
                const dim3 gs(tiles < 64 ? tiles : 64, ranges, ctx->cur_b);
                for (int r = 0; r <= rounds; ++r)
                {
                    hipLaunchKernelGGL(kd_top_pivot, gp, dim3(64), 0, ctx->stream, nodes, (const uint32_t *)lpos,
                                       (const uint32_t *)rasc, frame, state, level, r == 0 ? 1 : (r == rounds ? 2 : 0),
                                       hand, ctx->fs_tag);
                    if (r == rounds)
                        break;  // the last call only applies the last cut
                    hipLaunchKernelGGL(kd_top_flags, gt, dim3(TOP_THREADS), 0, ctx->stream, (const Node *)nodes,
                                       (const KdTopState *)state, tile_cnt, level, tiles, ctx->fs_tag);
                    hipLaunchKernelGGL(kd_top_lists, gt, dim3(TOP_THREADS), 0, ctx->stream, (const Node *)nodes, state,
                                       (const uint2 *)tile_cnt, lpos, rasc, level, tiles, ctx->fs_tag);
                    hipLaunchKernelGGL(kd_top_swap, gs, dim3(TOP_THREADS), 0, ctx->stream, nodes, state,
                                       (const uint32_t *)lpos, (const uint32_t *)rasc, ctx->fs_tag);
                }
                top = state;
            }
        }
        // The top levels of a batch stage only the END of every nth_element: the introselect loop runs ~15 rounds per
        // range whatever its size, each a chain of dependent global round trips (~6 us); once the active range is down
        // to BLK_TAIL nodes the remaining ~10 rounds run from 24 KiB of LDS.  (Staging the full blk_cap there makes
        // these workgroups wait for a CU with 48 KiB free while other chains fill the device.)
        static const int tail_env = LPX_KNOB("LPX_KD_TAIL") ? atoi(LPX_KNOB("LPX_KD_TAIL")) : BLK_TAIL;
        const int stage_cap = stage ? blk_cap : (tail_env < blk_cap ? tail_env : blk_cap);
        const size_t stage_lds = sizeof(Node) * stage_cap + 2 * sizeof(uint32_t) * stage_cap + 64 * sizeof(uint32_t);
        // Workgroup size by range length.  An introselect round is a chain of dependent steps whatever the range holds
        // (~100 us per level from 47k nodes down to 3k), so below BLK_WIDE nodes sixteen wavefronts only wait for one
        // another: with many chains in flight what a kernel costs the device is its resident wavefronts x their
        // lifetime, and four wavefronts per range instead of sixteen give the other chains three quarters of it back.
        static const uint32_t wide_env = LPX_KNOB("LPX_KD_WIDE") ? (uint32_t)atoi(LPX_KNOB("LPX_KD_WIDE")) : BLK_WIDE;
        static const uint32_t mid_env = LPX_KNOB("LPX_KD_MID") ? (uint32_t)atoi(LPX_KNOB("LPX_KD_MID")) : BLK_MID;
        // (a single frame keeps sixteen wavefronts on every level: four per range cost it 0.58 -> 0.66 ms)
        if (size > wide_env || ctx->cur_b == 1)
            hipLaunchKernelGGL(kd_block_kernel<1024>, dim3(1u << level, 1, ctx->cur_b), dim3(1024), stage_lds + key_lds,
                               ctx->stream, nodes, lpos, rasc, frame, level, blk_cap, stage_cap, top, ctx->fs_tag);
        else if (size > mid_env)
            hipLaunchKernelGGL(kd_block_kernel<256>, dim3(1u << level, 1, ctx->cur_b), dim3(256), stage_lds + key_lds / 4,  // (4 keys x 256 threads)
                               ctx->stream, nodes, lpos, rasc, frame, level, blk_cap, stage_cap, top, ctx->fs_tag);
        else
            hipLaunchKernelGGL(kd_block_kernel<64>, dim3(1u << level, 1, ctx->cur_b), dim3(64), stage_lds + key_lds / 16,
                               ctx->stream, nodes, lpos, rasc, frame, level, blk_cap, stage_cap, top, ctx->fs_tag);
        size = size / 2;  // larger child holds at most size / 2 nodes
        ++level;
    }
    // search path: the kernel that writes every node to its final place also makes every point its own set (the
    // union-find forest nb_index_kernel links); the list path builds its forest from the lists
    uint32_t *forest = (!ctx->use_lists && lpx_cc_from_chunks(m_max)) ? (uint32_t *)ctx->parent.p : (uint32_t *)nullptr;
    if (ctx->cur_b > 1 || big_single)
        hipLaunchKernelGGL(kd_lds_kernel<uint16_t>, dim3(1u << level, 1, ctx->cur_b), dim3(LG), lds_lds, ctx->stream, nodes,
                           (Node *)ctx->nodes_pre.p, frame, (uint32_t *)ctx->dbg_buf, blk_cap, forest, ctx->fs_tag);
    else
        hipLaunchKernelGGL(kd_lds_kernel<uint32_t>, dim3(1u << level, 1, ctx->cur_b), dim3(LG), blk_lds, ctx->stream, nodes,
                           (Node *)ctx->nodes_pre.p, frame, (uint32_t *)ctx->dbg_buf, blk_cap, forest, ctx->fs_tag);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
