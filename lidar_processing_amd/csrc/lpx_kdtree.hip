// lpx_kdtree.hip -- the reference kd-tree ORDER and its radius search on gfx950.
//
// Replaces KDTree<float,3>::rebuild (reference src/kdtree.hpp:174-225) and ::radius_search
// (:292-341) as used by Clusterer::cluster (src/clustering.cpp:63,90).
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
#include "lpx_internal.h"

#include <string.h>

#include <limits.h>
#include <stdlib.h>

namespace
{
typedef float4 Node;  // x, y, z, original index (bit pattern)

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
__device__ __forceinline__ float akey(const Node &n, int axis)
{
    return axis == 0 ? n.x : (axis == 1 ? n.y : n.z);
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

// ------------------------------------------------------------------------------------------------
// cooperative group primitives: G = 64 (one wavefront) or 1024 (one workgroup)
// ------------------------------------------------------------------------------------------------
// a workgroup of G threads (G / 64 wavefronts, at most 16)
template <int G>
struct Coop
{
    static constexpr int NW = G / WAVE;
    static_assert(G % WAVE == 0 && NW >= 2 && NW <= 16, "block groups are 2..16 wavefronts");
    static __device__ __forceinline__ void sync()
    {
        __threadfence_block();
        __syncthreads();
    }
    // cs: >= 32 words of LDS
    static __device__ __forceinline__ void scan2(bool f0, bool f1, uint32_t &r0, uint32_t &r1, uint32_t &t0,
                                                  uint32_t &t1, uint32_t *cs)
    {
        const unsigned long long lt = lpx_lanemask_lt();
        const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
        const uint32_t w = threadIdx.x / WAVE;
        if ((threadIdx.x % WAVE) == 0)
            cs[w] = (uint32_t)__popcll(m0) | ((uint32_t)__popcll(m1) << 16);
        __syncthreads();
        uint32_t b0 = 0, b1 = 0, s0 = 0, s1 = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i)
        {
            const uint32_t c = cs[i];
            if (i < (int)w)
            {
                b0 += c & 0xffffu;
                b1 += c >> 16;
            }
            s0 += c & 0xffffu;
            s1 += c >> 16;
        }
        __syncthreads();
        r0 = b0 + __popcll(m0 & lt);
        r1 = b1 + __popcll(m1 & lt);
        t0 = s0;
        t1 = s1;
    }
    static __device__ __forceinline__ void scan_packed(uint32_t v, uint32_t &excl, uint32_t &total, uint32_t *cs)
    {
        const uint32_t incl = lpx_wave_incl_scan_u32(v);
        const uint32_t w = threadIdx.x / WAVE;
        if ((threadIdx.x % WAVE) == WAVE - 1)
            cs[w] = incl;
        __syncthreads();
        uint32_t b = 0, s = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i)
        {
            const uint32_t c = cs[i];
            if (i < (int)w)
                b += c;
            s += c;
        }
        __syncthreads();
        excl = b + incl - v;
        total = s;
    }
    static __device__ __forceinline__ uint32_t sum(uint32_t v, uint32_t *cs)
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_xor(v, o, 64);
        const uint32_t w = threadIdx.x / WAVE;
        if ((threadIdx.x % WAVE) == 0)
            cs[16 + w] = v;
        __syncthreads();
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            s += cs[16 + i];
        __syncthreads();
        return s;
    }
};

template <>
struct Coop<64>
{
    static __device__ __forceinline__ void sync()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    static __device__ __forceinline__ void scan2(bool f0, bool f1, uint32_t &r0, uint32_t &r1, uint32_t &t0,
                                                  uint32_t &t1, uint32_t *)
    {
        const unsigned long long lt = lpx_lanemask_lt();
        const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
        r0 = __popcll(m0 & lt);
        r1 = __popcll(m1 & lt);
        t0 = __popcll(m0);
        t1 = __popcll(m1);
    }
    static __device__ __forceinline__ uint32_t sum(uint32_t v, uint32_t *)
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v += __shfl_xor(v, o, 64);
        return v;
    }
    // exclusive scan of two 16-bit counters packed in one word
    static __device__ __forceinline__ void scan_packed(uint32_t v, uint32_t &excl, uint32_t &total, uint32_t *)
    {
        const uint32_t incl = lpx_wave_incl_scan_u32(v);
        excl = incl - v;
        total = __shfl(incl, WAVE - 1, 64);
    }
};

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

// range of node `r` (path bits, MSB first) at `level` below [b,e)
__device__ __forceinline__ void descend(int &b, int &e, uint32_t r, int level)
{
    for (int d = level - 1; d >= 0; --d)
    {
        if (b >= e)
            return;
        const int mid = b + (e - b) / 2;
        if ((r >> d) & 1u)
            b = mid + 1;
        else
            e = mid;
    }
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

// pre-order rank of array position p in the implicit median-split tree over [0, M)
__device__ __forceinline__ uint32_t kd_rank_of(uint32_t p, uint32_t M)
{
    uint32_t b = 0, e = M, rank = 0;
    for (;;)
    {
        const uint32_t mid = b + (e - b) / 2;
        if (p == mid)
            return rank;
        if (p < mid)
        {
            rank += 1;
            e = mid;
        }
        else
        {
            rank += 1 + (mid - b);
            b = mid + 1;
        }
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

// ------------------------------------------------------------------------------------------------
// union-find (roots are the smallest original index of a component = its first FEC seed)
// ------------------------------------------------------------------------------------------------
// Cacheable relaxed loads: a stale parent is still an ancestor (parents only ever move towards the
// root and roots only ever get hooked under smaller roots), and every hook is a CAS that returns the
// current value, so staleness costs a retry, never a wrong union.  Roots are read back in a later launch.
__device__ __forceinline__ uint32_t uf_ld(uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void uf_st(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
    uint32_t p = uf_ld(parent + x);
    while (p != x)
    {
        const uint32_t gp = uf_ld(parent + p);
        if (gp != p)
            uf_st(parent + x, gp);  // path halving: only ever replaces a parent by an ancestor
        x = p;
        p = gp;
    }
    return x;
}

__device__ void uf_unite(uint32_t *parent, uint32_t a, uint32_t b)
{
    for (;;)
    {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b)
            return;
        if (a < b)
        {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        const uint32_t old = atomicCAS(parent + a, a, b);  // hook the larger root under the smaller
        if (old == a)
            return;
        // a was no root any more (the cached find saw an old value): the CAS returned its current parent
        // from the coherence point, an ancestor -- continue from there instead of trusting the cache again
        a = old;
    }
}

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
constexpr int NB_WAVES = 4;
constexpr int NB_THREADS = NB_WAVES * WAVE;
constexpr int NB_SEQ = 512;     // interval items in LDS per block (6 KiB)
constexpr int NB_NODES = 1024;  // candidate nodes staged in LDS per block (16 KiB): 6 workgroups per CU
constexpr int NB_BUCKET = 64;
constexpr int NB_GRAN = 16;     // candidates per cull granule: one row of 16 lanes
constexpr uint32_t NB_FINAL = 0x80000000u;

struct Item
{
    uint32_t rank, b, e;  // unexpanded subtree: node range [b,e), root at `rank`; final: e == NB_FINAL, b = count
};

// order-preserving breadth-first walk of the top D levels for the box [blo,bhi] (already widened by
// the radius); one wavefront.  Leaves the candidate intervals in cur[0..n) and their exclusive size
// prefix in pre[0..n]; returns n and the total T.
__device__ uint32_t nb_traverse(const Node *__restrict__ PR, uint32_t M, uint32_t D, const float *blo,
                                const float *bhi, Item *buf, uint32_t caps, uint32_t *pre, uint32_t lane,
                                Item **cur_out, uint32_t *T_out)
{
    Item *cur = buf, *nxt = buf + caps;
    uint32_t n_cur = 1;
    if (lane == 0)
    {
        cur[0].rank = 0;
        cur[0].b = 0;
        cur[0].e = M;
    }
    Coop<WAVE>::sync();
    // TWO levels per round trip: with the node of an unexpanded subtree its two children are requested as well (the
    // pre-order layout knows where they are), and the item is expanded twice from registers -- every level used to be one
    // dependent global round trip for the whole wavefront, ten of them for a 53k-point cloud, and a wavefront that waits
    // holds its slot.  (An odd last level is a single step.)
    for (uint32_t lvl = 0; lvl < D;)
    {
#ifdef LPX_TRAVERSE_ONE_LEVEL
        const bool two = false;
#else
        const bool two = lvl + 1 < D;
#endif
        const int axis = (int)(lvl % 3), axis2 = (int)((lvl + 1) % 3);
        const float lo_a = axis == 0 ? blo[0] : (axis == 1 ? blo[1] : blo[2]);
        const float hi_a = axis == 0 ? bhi[0] : (axis == 1 ? bhi[1] : bhi[2]);
        const float lo_b = axis2 == 0 ? blo[0] : (axis2 == 1 ? blo[1] : blo[2]);
        const float hi_b = axis2 == 0 ? bhi[0] : (axis2 == 1 ? bhi[1] : bhi[2]);
        uint32_t out_base = 0;
        bool overflow = false;
        for (uint32_t c0 = 0; c0 < n_cur; c0 += WAVE)
        {
            const bool valid = c0 + lane < n_cur;
            Item it;
            it.rank = it.b = 0;
            it.e = NB_FINAL;
            if (valid)
                it = cur[c0 + lane];
            const bool fin = it.e == NB_FINAL;
            // children of the item's root: left [b, mid) at rank + 1, right [mid + 1, e) at rank + 1 + (mid - b)
            const uint32_t mid = fin ? 0u : it.b + (it.e - it.b) / 2;
            const bool hasL = valid && !fin && mid > it.b, hasR = valid && !fin && mid + 1 < it.e;
            const uint32_t rankL = it.rank + 1, rankR = it.rank + 1 + (mid - it.b);
            Node nd, ndL, ndR;
            nd = PR[(valid && !fin) ? it.rank : 0u];
            ndL = PR[(two && hasL) ? rankL : 0u];
            ndR = PR[(two && hasR) ? rankR : 0u];
            uint32_t cnt = 0;
            bool goL = false, goR = false, goLL = false, goLR = false, goRL = false, goRR = false;
            uint32_t midL = 0, midR = 0;
            if (valid)
            {
                if (fin)
                    cnt = 1;
                else
                {
                    const float s0 = akey(nd, axis);
                    goL = hasL && (s0 >= lo_a);
                    goR = hasR && (s0 <= hi_a);
                    cnt = 1u + (goL ? 1u : 0u) + (goR ? 1u : 0u);
                    if (two)
                    {
                        if (goL)
                        {
                            midL = it.b + (mid - it.b) / 2;
                            const float sl = akey(ndL, axis2);
                            goLL = (midL > it.b) && (sl >= lo_b);
                            goLR = (midL + 1 < mid) && (sl <= hi_b);
                            cnt += (goLL ? 1u : 0u) + (goLR ? 1u : 0u);
                        }
                        if (goR)
                        {
                            midR = (mid + 1) + (it.e - (mid + 1)) / 2;
                            const float sr = akey(ndR, axis2);
                            goRL = (midR > mid + 1) && (sr >= lo_b);
                            goRR = (midR + 1 < it.e) && (sr <= hi_b);
                            cnt += (goRL ? 1u : 0u) + (goRR ? 1u : 0u);
                        }
                    }
                }
            }
            const uint32_t incl = lpx_wave_incl_scan_u32(cnt);
            const uint32_t tot = __builtin_amdgcn_readfirstlane(__shfl(incl, WAVE - 1, 64));
            if (out_base + tot > caps)
            {
                overflow = true;
                break;
            }
            if (valid)
            {
                uint32_t pos = out_base + incl - cnt;
                if (fin)
                    nxt[pos] = it;
                else
                {
                    Item o;
                    o.rank = it.rank;  // the root itself: a single final node
                    o.b = 1;
                    o.e = NB_FINAL;
                    nxt[pos++] = o;
                    if (goL)
                    {
                        if (!two)
                        {
                            o.rank = rankL;
                            o.b = it.b;
                            o.e = mid;
                            nxt[pos++] = o;
                        }
                        else
                        {
                            o.rank = rankL;  // the left child's root, then its two subtrees
                            o.b = 1;
                            o.e = NB_FINAL;
                            nxt[pos++] = o;
                            if (goLL)
                            {
                                o.rank = rankL + 1;
                                o.b = it.b;
                                o.e = midL;
                                nxt[pos++] = o;
                            }
                            if (goLR)
                            {
                                o.rank = rankL + 1 + (midL - it.b);
                                o.b = midL + 1;
                                o.e = mid;
                                nxt[pos++] = o;
                            }
                        }
                    }
                    if (goR)
                    {
                        if (!two)
                        {
                            o.rank = rankR;
                            o.b = mid + 1;
                            o.e = it.e;
                            nxt[pos++] = o;
                        }
                        else
                        {
                            o.rank = rankR;
                            o.b = 1;
                            o.e = NB_FINAL;
                            nxt[pos++] = o;
                            if (goRL)
                            {
                                o.rank = rankR + 1;
                                o.b = mid + 1;
                                o.e = midR;
                                nxt[pos++] = o;
                            }
                            if (goRR)
                            {
                                o.rank = rankR + 1 + (midR - (mid + 1));
                                o.b = midR + 1;
                                o.e = it.e;
                                nxt[pos++] = o;
                            }
                        }
                    }
                }
            }
            out_base += tot;
        }
        if (overflow)
            break;  // stopping early only widens the candidate intervals
        Item *t = cur;
        cur = nxt;
        nxt = t;
        n_cur = out_base;
        lvl += two ? 2u : 1u;
        Coop<WAVE>::sync();
    }
    uint32_t T = 0;
    for (uint32_t c0 = 0; c0 < n_cur; c0 += WAVE)
    {
        const bool valid = c0 + lane < n_cur;
        uint32_t cnt = 0;
        if (valid)
        {
            const Item it = cur[c0 + lane];
            cnt = (it.e == NB_FINAL) ? it.b : (it.e - it.b);
        }
        const uint32_t incl = lpx_wave_incl_scan_u32(cnt);
        if (valid)
            pre[c0 + lane] = T + incl - cnt;
        T += __builtin_amdgcn_readfirstlane(__shfl(incl, WAVE - 1, 64));
    }
    if (lane == 0)
        pre[n_cur] = T;
    Coop<WAVE>::sync();
    *cur_out = cur;
    *T_out = T;
    return n_cur;
}

// BLOCK = true : one workgroup per bucket subtree (<= 64 queries); the four wavefronts share the
//                candidate tile and split the queries (query j -> wavefront j % 4)
// BLOCK = false: one wavefront per node above the bucket level (a single query each)
// Both count, allocate (64-bit atomic bump of frame->nb_total, one block of list storage per group)
// and fill in the same launch; off[i] / len[i] locate the list of point i.
__global__ __launch_bounds__(NB_THREADS) void nb_group_kernel(const Node *__restrict__ PR, FrameState *frame,
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
    const uint32_t nq = __builtin_amdgcn_readfirstlane(BLOCK ? (ge - gb) : 1u);
    const bool active = lane < nq;
    const Node q = PR[grank + (active ? lane : 0u)];
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
    int phase = (frame->rs_stripe[gid % LPX_RS_STRIPES].v < cap_rs / LPX_RS_STRIPES) ? PH_RESERVE : PH_COUNT;  // sub-region
                                                                                               // exhausted: do not try
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

__global__ __launch_bounds__(256) void cc_hook_kernel(const FrameState *__restrict__ frame,
                                                       const uint32_t *__restrict__ off,
                                                       const uint32_t *__restrict__ len,
                                                       const uint32_t *__restrict__ nb_idx, uint32_t *parent,
                                                       uint64_t cap, uint32_t roots_only, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<4>(fv.fs);
    frame = lpx_slot(frame, fv.fs);
    off = lpx_slot(off, fv.fs);
    len = lpx_slot(len, fv.fs);
    parent = lpx_slot(parent, fv.fs);
    nb_idx = lpx_slot(nb_idx, fv.fs_nb);
    const uint32_t lane = threadIdx.x % WAVE;
    const uint32_t M = frame->n_obstacle;
    if (frame->nb_total > cap)
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

// (cell table of the component grid, further down; nb_index_kernel empties it)
constexpr unsigned long long CELL_EMPTY = ~0ull;
constexpr uint32_t CELL_NONE = 0xffffffffu;
__device__ __forceinline__ uint32_t cell_cap_for(uint32_t M, uint32_t cap_max)
{
    uint32_t cap = 64;
    while (cap < 2 * M && cap < cap_max)
        cap <<= 1;
    return cap;
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

// ------------------------------------------------------------------------------------------------
// Expansion-driven search, part 2: the connected components of the d-graph WITHOUT neighbour lists.
//
// The replay needs sets of points that no BFS can leave.  A uniform grid with cell edge c = 0.99 d / sqrt(3) makes
// every cell a CLIQUE of the d-graph (its diagonal is shorter than d), so the components are those of the graph
// whose vertices are the occupied cells and whose edges are the cell pairs that hold a point pair within d.  A
// point within d of a point of cell A lies at most 2 cells away on every axis (2 c > d), so each cell has 124
// possible partners, 62 by symmetry: one wavefront per occupied cell probes them, one partner per lane, and a lane
// that finds its partner occupied and not yet in the same set walks the partner's points against the cell's own
// (staged in LDS) until the first pair within d -- the reference's float expression, inclusive -- and unites the
// two cells.  Touching cells first, the others in a second launch that skips pairs already in one set.  (A coarser grid with
// 26-adjacency and no distance test at all gives sets that are only unions of components; on the reference's
// frames they are barely coarser, but on a cluttered scene -- BASELINE's synthetic box clouds -- they collapse
// into one giant set and serialise the replay: measured 88 ms against 6 ms per 1M-point frame.)
// Cell indices are floor(v / c) in double precision; indices saturate at +-2^20 cells, which can only merge
// sets (allowed: a set may be a union of components, it must never split one).  Open-addressing table keyed by
// the packed index triple; the points of a cell hang on a linked list (head per slot, next per point); union-find
// over table slots.
// ------------------------------------------------------------------------------------------------

// Home slot of a cell: the 2 x 2 x 2 block of cells it belongs to is hashed, the cell's position inside the block
// picks one of the 8 slots of that 64-byte line -- the 124 partners a cell probes then lie in ~27 lines instead of
// ~124 (the table is far larger than L2 once 256 frames are in flight, so every line is a fabric request).
__device__ __forceinline__ uint32_t cell_hash(unsigned long long key)
{
    const unsigned long long blk = key & ~((1ull << 42) | (1ull << 21) | 1ull);  // low bit of every index cleared
    unsigned long long k = blk;
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    const uint32_t sub = (uint32_t)(((key >> 42) & 1ull) << 2 | ((key >> 21) & 1ull) << 1 | (key & 1ull));
    return ((uint32_t)k << 3) | sub;
}


__device__ __forceinline__ uint32_t cell_coord(float v, double inv_c)
{
    const double f = fmin(fmax(floor((double)v * inv_c), -1048576.0), 1048575.0);
    return (uint32_t)((int)f + 1048576);  // 21 bits
}

__host__ __device__ __forceinline__ double cell_inv_edge(float d)
{
    return 1.0 / ((double)d * 0.5716);  // edge = 0.99 d / sqrt(3): every cell is a clique
}

__global__ void grid_clear_kernel(FrameState *__restrict__ frame, unsigned long long *__restrict__ tkey,
                                  uint32_t *__restrict__ tparent, uint32_t *__restrict__ thead, uint32_t cap_max,
                                  size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    frame = lpx_slot(frame, fs);
    tkey = lpx_slot(tkey, fs);
    tparent = lpx_slot(tparent, fs);
    thead = lpx_slot(thead, fs);
    const uint32_t s = lpx_blk.x * blockDim.x + threadIdx.x;
    if (s == 0)
    {
        frame->n_cells = 0;
        frame->cell_cursor = 0;
    }
    if (s < LPX_CELL_BITS_WORDS)
        ((uint32_t *)(tkey + cap_max))[s] = 0;  // the occupancy bitmap behind the table
    if (s >= cell_cap_for(frame->n_obstacle, cap_max) || frame->n_obstacle == 0)
        return;
    tkey[s] = CELL_EMPTY;
    tparent[s] = s;
    thead[s] = 0;  // points of the cell
}

// Insert, aggregated per tile in LDS.  The obstacle cloud arrives in x order (the segmentation emits it slab by slab,
// every slab x-sorted), so the GI_TILE consecutive points of a workgroup lie in a thin x slice and share their cells:
// a KITTI frame holds 3.3 points per cell and a cell's points almost always sit in ONE tile.  Rounds 2-4 sent every
// POINT to the table in memory -- an agent-scope load of the key, one atomicAdd on the cell's counter, and on this part
// an agent-scope atomic is executed on the memory side whatever the L2 holds (5 300 cycles of latency each, one fabric
// request each: 4.0 M atomics + 5.9 M reads and writes per 64-frame chain, the largest single consumer of the chain's
// requests).  Now the tile's points are first counted per cell in an LDS table (LDS atomics), and only every DISTINCT
// (tile, cell) goes to the global table: one probe, one atomicAdd of the tile's whole count (its old value is where
// the tile's points begin inside the cell's run), and the workgroup's newly claimed cells are listed with ONE bump of
// the frame's cell counter -- so that cells claimed by one tile are neighbours in the cell list, take neighbouring
// runs in grid_alloc_kernel, and the scatter of a tile writes one compact region.  Nothing depends on the order of the
// input: an unsorted cloud (lpx_cluster of any cloud) just aggregates less.  The order of a cell's points inside its
// run and which point represents a cell differ from run to run, like before; the components do not.
constexpr int GI_THREADS = 256;
#ifndef LPX_GI_PER
#define LPX_GI_PER 1
#endif
constexpr int GI_PER = LPX_GI_PER;             // points per thread
constexpr int GI_TILE = GI_THREADS * GI_PER;   // 256 points per workgroup: 10 KiB of LDS.  (1024 / 512 / 256 points per
                                               // tile: 2 076 / 2 109 / 2 115 Mpts/s on one box -- the larger tiles aggregate better
                                               // but their 41 / 20 KiB workgroups wait for room under load)
constexpr int GI_SLOTS = 2 * GI_TILE;          // LDS table: load factor <= 1/2
__device__ __forceinline__ uint32_t gi_lds_hash(unsigned long long key)
{
    unsigned long long k = key * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(k >> 40);
}

__global__ __launch_bounds__(GI_THREADS) void grid_insert_kernel(FrameState *__restrict__ frame,
                                                                 const float *__restrict__ OX,
                                                                 const float *__restrict__ OY,
                                                                 const float *__restrict__ OZ, float d,
                                                                 unsigned long long *tkey, uint32_t *thead,
                                                                 uint32_t *__restrict__ next, uint32_t *__restrict__ cells,
                                                                 unsigned long long *__restrict__ ckeys,
                                                                 uint32_t *__restrict__ cell_of, float4 *__restrict__ trep,
                                                                 uint32_t cap_max, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    trep = lpx_slot(trep, fs);
    ckeys = lpx_slot(ckeys, fs);
    frame = lpx_slot(frame, fs);
    OX = lpx_slot(OX, fs);
    OY = lpx_slot(OY, fs);
    OZ = lpx_slot(OZ, fs);
    tkey = lpx_slot(tkey, fs);
    thead = lpx_slot(thead, fs);
    next = lpx_slot(next, fs);
    cells = lpx_slot(cells, fs);
    cell_of = lpx_slot(cell_of, fs);
    __shared__ unsigned long long lkey[GI_SLOTS];  // the tile's cells
    __shared__ uint32_t lcnt[GI_SLOTS];            // points of the tile in the cell; after phase 2: where they begin in the cell's run
    __shared__ uint32_t lslot[GI_SLOTS];           // first: a point of the tile in that cell (the representative); then: the cell's table slot
    __shared__ uint32_t lclaim[GI_TILE];           // table slots this workgroup claimed (then: their LDS slots, for the keys)
    __shared__ uint32_t llist[GI_TILE];            // the occupied LDS slots (phase 2 walks them with every lane busy)
    __shared__ uint16_t lclaim_s[GI_TILE];         // LDS slot of every claimed cell (its key goes to the cell list too)
    __shared__ uint32_t nclaim, claim_base, nlist;
    const uint32_t M = frame->n_obstacle;
    const uint32_t tile0 = lpx_blk.x * GI_TILE;
    if (tile0 >= M)
        return;
    const uint32_t tid = threadIdx.x;
    for (uint32_t s = tid; s < GI_SLOTS; s += GI_THREADS)
    {
        lkey[s] = CELL_EMPTY;
        lcnt[s] = 0;
    }
    if (tid == 0)
        nclaim = nlist = 0;
    __syncthreads();
    const uint32_t mask = cell_cap_for(M, cap_max) - 1;
    const double inv_c = cell_inv_edge(d);
    // phase 1: every point into the LDS table (consecutive lanes on consecutive points, the GI_PER loads together)
    uint32_t ls[GI_PER], lrank[GI_PER];
    float px[GI_PER], py[GI_PER], pz[GI_PER];
    bool in[GI_PER];
#pragma unroll
    for (int u = 0; u < GI_PER; ++u)
    {
        const uint32_t i = tile0 + u * GI_THREADS + tid;
        in[u] = i < M;
        px[u] = in[u] ? OX[i] : 0.0f;
        py[u] = in[u] ? OY[i] : 0.0f;
        pz[u] = in[u] ? OZ[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < GI_PER; ++u)
    {
        ls[u] = 0;
        lrank[u] = 0;
        if (!in[u])
            continue;
        const unsigned long long key = ((unsigned long long)cell_coord(px[u], inv_c) << 42) |
                                       ((unsigned long long)cell_coord(py[u], inv_c) << 21) |
                                       (unsigned long long)cell_coord(pz[u], inv_c);
        uint32_t s = gi_lds_hash(key) & (GI_SLOTS - 1);
        for (;;)
        {
            unsigned long long o = lkey[s];
            if (o == CELL_EMPTY)
                o = atomicCAS(&lkey[s], CELL_EMPTY, key);
            if (o == CELL_EMPTY)
            {
                lslot[s] = tile0 + u * GI_THREADS + tid;  // this point stands for the cell if the tile claims it
                llist[atomicAdd(&nlist, 1u)] = s;
                break;
            }
            if (o == key)
                break;
            s = (s + 1) & (GI_SLOTS - 1);
        }
        ls[u] = s;
        lrank[u] = atomicAdd(&lcnt[s], 1u);
    }
    __syncthreads();
    // phase 2: every distinct cell of the tile to the table in memory (a thread per cell: the dependent round trips of
    // all cells of the tile -- probe, claim, position -- are in flight together)
    const uint32_t ncell = nlist;
    for (uint32_t c = tid; c < ncell; c += GI_THREADS)
    {
        const uint32_t s = llist[c];
        const unsigned long long key = lkey[s];
        uint32_t h = cell_hash(key) & mask;
        // (straight to the CAS: most (tile, cell) pairs are new cells, and a look first -- an agent-scope load is a trip
        // to the memory side like the CAS itself -- made their chain three dependent trips instead of two)
        for (;;)
        {
            const unsigned long long o = atomicCAS(tkey + h, CELL_EMPTY, key);
            if (o == CELL_EMPTY)
            {
                const uint32_t rep = lslot[s];
                trep[h] = make_float4(OX[rep], OY[rep], OZ[rep], 0.0f);  // represents the cell in the quick test of the linking
                const uint32_t ci = atomicAdd(&nclaim, 1u);
                lclaim[ci] = h;
                lclaim_s[ci] = (uint16_t)s;
                // ... and shows in the occupancy bitmap (one no-return atomic per CELL: lpx_cell_bit_index)
                const uint32_t bi = lpx_cell_bit_index((uint32_t)(key >> 42), (uint32_t)(key >> 21) & 0x1fffffu,
                                                       (uint32_t)key & 0x1fffffu);
                atomicOr((uint32_t *)(tkey + cap_max) + (bi >> 5), 1u << (bi & 31u));
                break;
            }
            if (o == key)
                break;
            h = (h + 1) & mask;
        }
        lcnt[s] = atomicAdd(thead + h, lcnt[s]);  // the tile's points take consecutive positions among the cell's points
        lslot[s] = h;
    }
    __syncthreads();
    if (tid == 0 && nclaim)
        claim_base = atomicAdd(&frame->n_cells, nclaim);
    __syncthreads();
    for (uint32_t c = tid; c < nclaim; c += GI_THREADS)
    {
        // the cell list carries the cell's key beside its slot: the linking reads both with ONE trip per round instead of
        // slot -> key in two dependent ones
        cells[claim_base + c] = lclaim[c];
        ckeys[claim_base + c] = lkey[lclaim_s[c]];
    }
    // phase 3: what the scatter needs per point
#pragma unroll
    for (int u = 0; u < GI_PER; ++u)
        if (in[u])
        {
            const uint32_t i = tile0 + u * GI_THREADS + tid;
            cell_of[i] = lslot[ls[u]];
            next[i] = lcnt[ls[u]] + lrank[u];
        }
}

// The points of every cell as ONE contiguous run of {x, y, z, index} records (the linking then reads a cell's points
// with independent loads instead of walking a list): the cells take their runs in the order of the cell list, a
// wavefront's 64 cells with ONE bump of the frame's cursor (an exclusive scan of their counts: 64 x fewer atomics on
// that word, and the cells a tile claimed together get neighbouring runs) ...
__global__ __launch_bounds__(256) void grid_alloc_kernel(FrameState *__restrict__ frame, const uint32_t *__restrict__ cells,
                                                         const uint32_t *__restrict__ tcount, uint32_t *__restrict__ tstart,
                                                         size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    frame = lpx_slot(frame, fs);
    cells = lpx_slot(cells, fs);
    tcount = lpx_slot(tcount, fs);
    tstart = lpx_slot(tstart, fs);
    const uint32_t c = lpx_blk.x * blockDim.x + threadIdx.x;
    const uint32_t nc = frame->n_cells;
    if ((c & ~(uint32_t)(WAVE - 1)) >= nc)
        return;  // (whole wavefronts leave together)
    const bool on = c < nc;
    const uint32_t h = on ? cells[c] : 0u;
    const uint32_t cnt = on ? tcount[h] : 0u;
    const uint32_t incl = lpx_wave_incl_scan_u32(cnt);
    const uint32_t total = __shfl(incl, WAVE - 1, WAVE);
    uint32_t base = 0;
    if ((threadIdx.x & (WAVE - 1)) == 0)
        base = atomicAdd(&frame->cell_cursor, total);
    base = __shfl(base, 0, WAVE);
    if (on)
        tstart[h] = base + incl - cnt;
}

// ... and every point goes to its position in the run of its cell
__global__ void grid_scatter_kernel(const FrameState *__restrict__ frame, const float *__restrict__ OX,
                                    const float *__restrict__ OY, const float *__restrict__ OZ,
                                    const uint32_t *__restrict__ cell_of, const uint32_t *__restrict__ rank,
                                    const uint32_t *__restrict__ tstart, float4 *__restrict__ cpts, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    frame = lpx_slot(frame, fs);
    OX = lpx_slot(OX, fs);
    OY = lpx_slot(OY, fs);
    OZ = lpx_slot(OZ, fs);
    cell_of = lpx_slot(cell_of, fs);
    rank = lpx_slot(rank, fs);
    tstart = lpx_slot(tstart, fs);
    cpts = lpx_slot(cpts, fs);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i >= frame->n_obstacle)
        return;
    cpts[tstart[cell_of[i]] + rank[i]] = make_float4(OX[i], OY[i], OZ[i], __uint_as_float(i));
}

// The linking: one (cell, partner) PAIR per lane.  (One wavefront per cell, one partner per lane -- the first form of
// this kernel -- keeps 13 or 49 of the 64 lanes busy and walks four or five dependent loads per cell (cell list ->
// key -> probe -> representatives / roots), so it was latency-bound at full occupancy, about 80 cell iterations per
// resident wavefront; flat pairs fill every lane -- 5 x fewer wavefront iterations in the touching pass, 1.3 x in
// the far pass -- and the lanes of one cell read the same words: 1393 -> 1520 Mpts/s on the headline workload.)
// Two passes (two launches): the 13 partners that touch the cell -- almost all of them are connected and the quick
// test settles them -- then the 49 partners one cell further away, when every union of the first pass is visible:
// most of those pairs already share a set through the cells between them and are skipped by the root comparison,
// only pairs of different sets pay for a point-pair scan (every point of the partner against every point of the
// cell, both contiguous runs of `cpts`, until the first pair within d).
__constant__ uint8_t FAR_T[49] = {64,  65,  69,  70,  71,  72,  73,  74,  75,  76,  77,  78,  79,  80,  84,  85,  89,
                                  90,  94,  95,  96,  97,  98,  99,  100, 101, 102, 103, 104, 105, 106, 107, 108, 109,
                                  110, 111, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124};

#ifndef LPX_WPE_PAIRS
#define LPX_WPE_PAIRS 8
#endif
template <bool FAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, LPX_WPE_PAIRS))) void grid_pairs_kernel(const FrameState *__restrict__ frame,
                                                         const unsigned long long *__restrict__ tkey,
                                                         uint32_t *tparent, const uint32_t *__restrict__ tcount,
                                                         const uint32_t *__restrict__ tstart,
                                                         const uint32_t *__restrict__ cells,
                                                         const unsigned long long *__restrict__ ckeys,
                                                         const float4 *__restrict__ cpts,
                                                         const float4 *__restrict__ trep, float r2, uint32_t cap_max,
                                                         int dbg, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<5>(fs);
    trep = lpx_slot(trep, fs);
    ckeys = lpx_slot(ckeys, fs);
    frame = lpx_slot(frame, fs);
    tkey = lpx_slot(tkey, fs);
    tparent = lpx_slot(tparent, fs);
    tcount = lpx_slot(tcount, fs);
    tstart = lpx_slot(tstart, fs);
    cells = lpx_slot(cells, fs);
    cpts = lpx_slot(cpts, fs);
    const uint32_t M = frame->n_obstacle;
    if (M == 0)
        return;
    const uint32_t mask = cell_cap_for(M, cap_max) - 1;
    constexpr uint32_t P = FAR ? 49u : 13u;
    const uint32_t total = frame->n_cells * P;  // (fewer than 2^30 obstacle points: 32-bit item arithmetic)
    // Phase A / phase B.  Of a cell's 62 possible partners about ten exist, and finding that out used to cost every
    // (cell, partner) item a hash and a scattered probe of the table -- 47 M probes per 64-frame chain, a quarter of the
    // texture-addresser cycles and a seventh of the vector-ALU cycles a chain spends (profiles/r05_stream_cu_resources.json:
    // the compute units, not the memory system, are what sixteen chains in flight saturate).  Now the workgroup holds the
    // frame's occupancy bitmap in LDS (lpx_cell_bit_index: 8 KiB): phase A walks the
    // items, asks the bitmap, and queues the few survivors -- (cell slot, partner key) -- densely in LDS; phase B hashes,
    // probes and links only those, U per lane side by side as before.  A false positive of the bitmap (an aliased
    // position) costs one probe that finds nothing; there are no false negatives (every claimed cell set its bit in
    // grid_insert_kernel).  Which pairs are linked, and with them the components, do not change.
    constexpr int U = 4;
#ifndef LPX_PAIRS_QCAP
#define LPX_PAIRS_QCAP 1024
#endif
    constexpr uint32_t QCAP = LPX_PAIRS_QCAP;  // survivors queued per round (12 bytes each)
    __shared__ uint32_t s_bits[LPX_CELL_BITS_WORDS];
    __shared__ uint8_t s_far[FAR ? 52 : 4];
    __shared__ unsigned long long q_key[QCAP];
    __shared__ uint32_t q_slot[QCAP];
    __shared__ uint32_t q_n;
    constexpr uint32_t CCAP = 256;  // pairs queued for the slow tail (phase C)
    __shared__ uint32_t c_a[CCAP], c_b[CCAP];
    __shared__ uint32_t c_n;
    {
        const uint4 *src = (const uint4 *)(tkey + cap_max);  // behind the table (16-byte aligned: cap_max is a power of two)
        uint4 *dst = (uint4 *)s_bits;
        for (uint32_t i = threadIdx.x; i < LPX_CELL_BITS_WORDS / 4; i += blockDim.x)
            dst[i] = src[i];
        if (threadIdx.x == 0)
            q_n = c_n = 0;
        // (the far pass's offset table from LDS: indexed per lane, the constant-memory copy was a vector load -- a trip to
        // memory in front of every bitmap test)
        if (FAR && threadIdx.x < 49)
            s_far[threadIdx.x] = FAR_T[threadIdx.x];
    }
    __syncthreads();
    // The slow tail of an item -- the pair's representatives are farther apart than d: (far pass) the boxes of the two
    // cells, then every point of one against every point of the other -- is needed by about one existing pair in forty,
    // but a wavefront that holds ONE such item walks its dependent loads (boxes, run bounds, points) while the other 63
    // lanes wait, and nearly every wavefront of every round held one: the far pass spent more time there than on all
    // its probes.  Phase B now only QUEUES such pairs (c_a / c_b, a few hundred per workgroup); phase C, once after the
    // last round, walks them with every lane busy -- and finds many of them united meanwhile.
    auto slow_pair = [&](uint32_t sc, uint32_t pc) {
        if (dbg == 3)
            return;
        if (uf_find(tparent, sc) == uf_find(tparent, pc))
            return;  // united meanwhile through other pairs
        if (FAR)
        {
            // (the far pass queues a pair as soon as its two cells have different roots: the quick test is made here)
            const float4 qa = trep[sc], qb = trep[pc];
            const float e0 = qa.x - qb.x, e1 = qa.y - qb.y, e2 = qa.z - qb.z;
            if (e0 * e0 + (e1 * e1 + e2 * e2) <= r2)
            {
                if (dbg != 2)
                    uf_unite(tparent, sc, pc);
                return;
            }
        }
        if (FAR)
        {
            // the boxes of the two cells' points (grid_compress_kernel): when even the boxes are farther apart than d
            // no pair can be within d -- the gaps are differences of coordinates that occur, float subtraction, squares
            // of non-negative values and the sums below are monotonic, so the expression of EVERY pair is at least this
            // one -- and the scan, all na x nb pairs with no hit to stop it, is skipped
            const float4 la = trep[(size_t)cap_max + sc], ha = trep[2 * (size_t)cap_max + sc];
            const float4 lb = trep[(size_t)cap_max + pc], hb = trep[2 * (size_t)cap_max + pc];
            const float g0 = fmaxf(fmaxf(la.x - hb.x, lb.x - ha.x), 0.0f);
            const float g1 = fmaxf(fmaxf(la.y - hb.y, lb.y - ha.y), 0.0f);
            const float g2 = fmaxf(fmaxf(la.z - hb.z, lb.z - ha.z), 0.0f);
            if (g0 * g0 + (g1 * g1 + g2 * g2) > r2)
                return;
        }
        // every point of the partner against every point of the cell, until the first pair within d.  Cells hold
        // 2.6 points on average: PS points of either run are requested TOGETHER and the pairs are tested from
        // registers, longer runs go on in steps of PS -- the plain double loop (PS = 1) is a chain of na x nb
        // dependent 16-byte loads in one lane while the other 63 lanes of the wavefront wait.  Measured on one box,
        // 16 x 64 KITTI frames in flight / the two linking kernels of a chain alone: PS 1 2000-2014 Mpts/s / 1.65 ms;
        // PS 2 1996-2012 / 1.53; PS 3 1978 / 1.47; PS 6 1879-1899 / 1.62 -- the wider scans are faster alone and
        // SLOWER under load (every scanning lane requests 2 PS records whatever its runs hold, and with twenty chains
        // in flight the memory pipeline is what the kernels queue for), so: two.
#ifdef LPX_PAIR_SCAN_PS
        constexpr uint32_t PS = LPX_PAIR_SCAN_PS;
#else
        constexpr uint32_t PS = 2;
#endif
        bool joined = false;
        const float4 *A = cpts + tstart[sc], *B = cpts + tstart[pc];
        const uint32_t na = tcount[sc], nb = tcount[pc];
        for (uint32_t b0 = 0; b0 < nb && !joined; b0 += PS)
            for (uint32_t a0 = 0; a0 < na && !joined; a0 += PS)
            {
                float4 pa[PS], pb[PS];
#pragma unroll
                for (uint32_t i = 0; i < PS; ++i)
                {
                    pa[i] = A[min(a0 + i, na - 1)];
                    pb[i] = B[min(b0 + i, nb - 1)];
                }
#pragma unroll
                for (uint32_t j = 0; j < PS; ++j)
#pragma unroll
                    for (uint32_t i = 0; i < PS; ++i)
                    {
                        const float d0 = pa[i].x - pb[j].x, d1 = pa[i].y - pb[j].y, d2 = pa[i].z - pb[j].z;
                        // dist_sqr, src/kdtree.hpp:145-157, inclusive :315 (a clamped index repeats a point of the run)
                        joined = joined || (d0 * d0 + (d1 * d1 + d2 * d2) <= r2);
                    }
            }
        if (joined)
            uf_unite(tparent, sc, pc);
    };
    const unsigned long long lt = lpx_lanemask_lt();
    const uint32_t lane = threadIdx.x % WAVE;
    // LPX_GP_PROF (a variant build: tools/build_variant.sh gpprof -DLPX_GP_PROF): cycles of two workgroups of frame 0 by phase
#ifdef LPX_GP_PROF
    unsigned long long gp_t = clock64(), gp_a = 0, gp_b = 0, gp_c = 0, gp_nq = 0;
    const unsigned long long gp_t0 = gp_t;
    uint32_t gp_rounds = 0;
#define GP_LAP(acc) do { const unsigned long long n_ = clock64(); (acc) += n_ - gp_t; gp_t = n_; } while (0)
#else
#define GP_LAP(acc) ((void)0)
#endif
    const uint32_t per_round = blockDim.x * (QCAP / 256u);  // items one round may queue at most
    // The (slot, key) of the cells of a round's items are requested one round AHEAD, when the previous round's phase A is
    // through with its own: they travel while phase B works, and phase A itself is LDS work only.
    constexpr uint32_t RI = QCAP / 256u;  // items per thread and round
    uint32_t f_slot[RI];
    unsigned long long f_key[RI];
    auto fetch = [&](unsigned long long r64) {
#pragma unroll
        for (uint32_t r = 0; r < RI; ++r)
        {
            const unsigned long long item = r64 + r * blockDim.x + threadIdx.x;
            const uint32_t ci = item < total ? (uint32_t)item / P : 0u;  // (total > 0 here: cell 0 exists)
            f_slot[r] = cells[ci];
            f_key[r] = ckeys[ci];  // (beside the slot in the cell list: no second, dependent trip)
        }
    };
    if ((unsigned long long)lpx_blk.x * per_round < total)
        fetch((unsigned long long)lpx_blk.x * per_round);
    for (unsigned long long round64 = (unsigned long long)lpx_blk.x * per_round; round64 < total;
         round64 += (unsigned long long)gridDim.x * per_round)
    {
        const uint32_t round0 = (uint32_t)round64;
        uint32_t c_slot[RI];
        unsigned long long c_key[RI];
#pragma unroll
        for (uint32_t r = 0; r < RI; ++r)
        {
            c_slot[r] = f_slot[r];
            c_key[r] = f_key[r];
        }
        if (round64 + (unsigned long long)gridDim.x * per_round < total)
            fetch(round64 + (unsigned long long)gridDim.x * per_round);
        // ---- phase A: QCAP / 256 items per thread, consecutive lanes on consecutive items (49 / 13 items share a cell) ----
#pragma unroll
        for (uint32_t r = 0; r < RI; ++r)
        {
            const uint32_t item = round0 + r * blockDim.x + threadIdx.x;  // (below total + per_round: no wrap)
            bool keep = item < total;
            uint32_t slq = 0;
            unsigned long long nkq = 0;
            if (keep)
            {
                const uint32_t ci = item / P;
                slq = c_slot[r];
                const unsigned long long key = c_key[r];
                const uint32_t j = item - ci * P;
                int dx, dy, dz;
                if (FAR)
                {
                    const int t = s_far[j];
                    dx = t / 25 - 2, dy = (t / 5) % 5 - 2, dz = t % 5 - 2;
                }
                else
                {
                    const int t = 14 + (int)j;  // the offsets of [-1, 1]^3 that follow (0, 0, 0) lexicographically
                    dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
                }
                const int nx = (int)(key >> 42) + dx, ny = (int)((key >> 21) & 0x1fffffu) + dy,
                          nz = (int)(key & 0x1fffffu) + dz;
                keep = !((unsigned)nx > 0x1fffffu || (unsigned)ny > 0x1fffffu || (unsigned)nz > 0x1fffffu);
                const uint32_t bi = lpx_cell_bit_index((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
                keep = keep && ((s_bits[bi >> 5] >> (bi & 31u)) & 1u);
                nkq = ((unsigned long long)nx << 42) | ((unsigned long long)ny << 21) | (unsigned long long)nz;
            }
            const unsigned long long km = __ballot(keep);
            if (km)
            {
                uint32_t pos = 0;
                if (lane == 0)
                    pos = atomicAdd(&q_n, (uint32_t)__popcll(km));
                pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
                if (keep)
                {
                    const uint32_t at = pos + __popcll(km & lt);
                    q_slot[at] = slq;
                    q_key[at] = nkq;
                }
            }
        }
        __syncthreads();
        const uint32_t nq = q_n;
        GP_LAP(gp_a);
#ifdef LPX_GP_PROF
        gp_nq += nq;
        ++gp_rounds;
#endif
        // ---- phase B: the survivors, U per lane side by side ----
        for (uint32_t e0 = threadIdx.x; e0 < nq; e0 += U * blockDim.x)
        {
        uint32_t sl[U], hh[U];
        unsigned long long nk[U], k2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const uint32_t e = e0 + u * blockDim.x;
            live[u] = e < nq;
            sl[u] = live[u] ? q_slot[e] : 0u;
            nk[u] = live[u] ? q_key[e] : 0ull;
            hh[u] = cell_hash(nk[u]) & mask;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            k2[u] = live[u] ? tkey[hh[u]] : CELL_EMPTY;
        // the partner's slot (almost always the home slot or none; further probes one item at a time)
        uint32_t partner[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            partner[u] = CELL_NONE;
            uint32_t h = hh[u];
            unsigned long long k = k2[u];
            while (k != CELL_EMPTY)
            {
                if (k == nk[u])
                {
                    partner[u] = h;
                    break;
                }
                h = (h + 1) & mask;
                k = tkey[h];
            }
            if (dbg == 1)
                partner[u] = CELL_NONE;
        }
        // the far pass first skips what the touching pass already united (grid_compress_kernel ran in between: one
        // load per side answers it for all but the pairs of this pass)
        if (FAR)
        {
            uint32_t pa[U], pb[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                pa[u] = partner[u] != CELL_NONE ? uf_ld(tparent + sl[u]) : 0u;
                pb[u] = partner[u] != CELL_NONE ? uf_ld(tparent + partner[u]) : 0u;
            }
            // ... and everything else of such a pair -- one existing partner in forty -- belongs to phase C: queued
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (partner[u] != CELL_NONE && pa[u] != pb[u])
                {
                    const uint32_t at = atomicAdd(&c_n, 1u);
                    if (at < CCAP)
                    {
                        c_a[at] = sl[u];
                        c_b[at] = partner[u];
                    }
                    else
                        slow_pair(sl[u], partner[u]);  // (the queue is full: rare, done on the spot)
                }
            continue;
        }
        // quick test: the point that claimed the cell against the one that claimed the partner
        float4 ra[U], rb[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const bool on = partner[u] != CELL_NONE;
            ra[u] = trep[on ? sl[u] : 0u];
            rb[u] = trep[on ? partner[u] : 0u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (partner[u] == CELL_NONE)
                continue;
            const uint32_t sc = sl[u], pc = partner[u];
            if (FAR && uf_find(tparent, sc) == uf_find(tparent, pc))
                continue;
            {
                const float d0 = ra[u].x - rb[u].x, d1 = ra[u].y - rb[u].y, d2 = ra[u].z - rb[u].z;
                if (d0 * d0 + (d1 * d1 + d2 * d2) <= r2)
                {
                    if (dbg != 2)
                        uf_unite(tparent, sc, pc);
                    continue;
                }
            }
            if (dbg == 3)
                continue;
            {
                const uint32_t at = atomicAdd(&c_n, 1u);
                if (at < CCAP)
                {
                    c_a[at] = sc;
                    c_b[at] = pc;
                }
                else
                    slow_pair(sc, pc);  // (the queue is full: rare, done on the spot)
            }
        }
        }  // phase B
        __syncthreads();  // everybody is through with the queue
        if (threadIdx.x == 0)
            q_n = 0;
        __syncthreads();
        GP_LAP(gp_b);
    }
    // ---- phase C: the queued slow pairs ----
    __syncthreads();
    {
        const uint32_t nc = c_n < CCAP ? c_n : CCAP;
        for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x)
            slow_pair(c_a[i], c_b[i]);
    }
#ifdef LPX_GP_PROF
    __syncthreads();
    GP_LAP(gp_c);
    if (threadIdx.x == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))
        printf("grid_pairs<%d> block %u of %u: cells %u rounds %u survivors %llu slow pairs %u | cycles total %llu A %llu B %llu C %llu\n",
               (int)FAR, blockIdx.x, gridDim.x, frame->n_cells, gp_rounds, gp_nq, c_n, clock64() - gp_t0, gp_a, gp_b, gp_c);
#endif
#undef GP_LAP
}

// between the two linking passes: every cell points straight at its root, so that the far pass recognises pairs of
// one set with one load per side
__global__ void grid_compress_kernel(const FrameState *__restrict__ frame, const uint32_t *__restrict__ cells,
                                     uint32_t *tparent, const uint32_t *__restrict__ tcount,
                                     const uint32_t *__restrict__ tstart, const float4 *__restrict__ cpts,
                                     float4 *__restrict__ trep, uint32_t cap_max, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    frame = lpx_slot(frame, fs);
    cells = lpx_slot(cells, fs);
    tparent = lpx_slot(tparent, fs);
    tcount = lpx_slot(tcount, fs);
    tstart = lpx_slot(tstart, fs);
    cpts = lpx_slot(cpts, fs);
    trep = lpx_slot(trep, fs);
    const uint32_t c = lpx_blk.x * blockDim.x + threadIdx.x;
    if (c >= frame->n_cells)
        return;
    const uint32_t s = cells[c];
    // The bounding box of the cell's points (behind the representatives: trep[cap_max + slot], trep[2 cap_max + slot]):
    // the far pass scans the points of two cells only when their boxes are within d.  A cell with many points (a dense
    // surface near the sensor) gets an unbounded box -- it is never skipped -- instead of a long loop in one lane.
    {
        const uint32_t n = tcount[s];
        const float4 *P = cpts + tstart[s];
        float4 lo = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f), hi = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
        if (n <= 16u)
        {
            lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
            for (uint32_t i = 0; i < n; ++i)
            {
                const float4 p = P[i];
                lo.x = fminf(lo.x, p.x), lo.y = fminf(lo.y, p.y), lo.z = fminf(lo.z, p.z);
                hi.x = fmaxf(hi.x, p.x), hi.y = fmaxf(hi.y, p.y), hi.z = fmaxf(hi.z, p.z);
            }
        }
        trep[(size_t)cap_max + s] = lo;
        trep[2 * (size_t)cap_max + s] = hi;
    }
    uint32_t x = s, p = uf_ld(tparent + x);
    while (p != x)
    {
        x = p;
        p = uf_ld(tparent + x);
    }
    uf_st(tparent + s, x);
}

// root[i] = a point of the root cell of point i's set (the same word for all its members), iota, state reset
// (one workgroup per radix-sort tile, eight points per thread; hist: the tile's histogram of the lowest byte of the
// roots for the first pass of the component sort that follows, or null -- see flatten_kernel, lpx_cluster.hip)
__global__ __launch_bounds__(256) void grid_flatten_kernel(const FrameState *__restrict__ frame, uint32_t *tparent,
                                                           const uint32_t *__restrict__ tstart,
                                                           const uint32_t *__restrict__ cell_of,
                                                           uint32_t *__restrict__ root, uint32_t *__restrict__ iota,
                                                           uint8_t *__restrict__ state, uint32_t *__restrict__ valid,
                                                           uint32_t *__restrict__ cc_lo, uint32_t *__restrict__ cc_hi,
                                                           uint32_t *__restrict__ hist, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<6>(fs);
    __shared__ uint32_t h[256];
    frame = lpx_slot(frame, fs);
    tparent = lpx_slot(tparent, fs);
    tstart = lpx_slot(tstart, fs);
    cell_of = lpx_slot(cell_of, fs);
    root = lpx_slot(root, fs);
    iota = lpx_slot(iota, fs);
    state = lpx_slot(state, fs);
    valid = lpx_slot(valid, fs);
    cc_lo = lpx_slot(cc_lo, fs);
    cc_hi = lpx_slot(cc_hi, fs);
    hist = lpx_slot(hist, fs);
    const uint32_t tid = threadIdx.x, M = frame->n_obstacle;
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
        uint32_t x = cell_of[i];
        for (;;)
        {
            const uint32_t p = __hip_atomic_load(tparent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p == x)
                break;
            x = p;
        }
        const uint32_t rt = tstart[x];  // where the points of the root cell begin: one word per set, below M
        root[i] = rt;
        if (iota)
            iota[i] = i;
        state[i] = 0;
        valid[i] = 0;
        cc_lo[i] = 0;
        cc_hi[i] = 0;
        if (hist)
            atomicAdd(&h[rt & 255u], 1u);
    }
    if (hist)
    {
        __syncthreads();
        hist[lpx_blk.x * 256u + tid] = h[tid];
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
    const int blk_cap = ctx->cur_b > 1 ? BLK_CAP_BATCH : BLK_CAP_MAX;
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
    if (ctx->cur_b > 1)
        hipLaunchKernelGGL(kd_lds_kernel<uint16_t>, dim3(1u << level, 1, ctx->cur_b), dim3(LG), lds_lds, ctx->stream, nodes,
                           (Node *)ctx->nodes_pre.p, frame, (uint32_t *)ctx->dbg_buf, blk_cap, forest, ctx->fs_tag);
    else
        hipLaunchKernelGGL(kd_lds_kernel<uint32_t>, dim3(1u << level, 1, ctx->cur_b), dim3(LG), blk_lds, ctx->stream, nodes,
                           (Node *)ctx->nodes_pre.p, frame, (uint32_t *)ctx->dbg_buf, blk_cap, forest, ctx->fs_tag);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

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
        for (uint32_t roots_only = 1;; roots_only = 0)
        {
            hipLaunchKernelGGL(cc_hook_kernel, dim3(hgrid, 1, ctx->cur_b), dim3(256), 0, ctx->stream, frame,
                               (const uint32_t *)off, (const uint32_t *)len, (const uint32_t *)ctx->nb_idx.p,
                               (uint32_t *)ctx->parent.p, ctx->cap_nb, roots_only, lpx_fv(ctx));
            if (!roots_only)
                break;
            hipLaunchKernelGGL(cc_flatten_kernel, dim3((m_max + 255) / 256, 1, ctx->cur_b), dim3(256), 0, ctx->stream,
                               frame, (uint32_t *)ctx->parent.p, ctx->fs_tag);
        }
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

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
// development build only: the components from a sweep over the cloud's own order (measured, not adopted) live outside
// the product sources
#include "../../experiments/sweep_components.inc"
#endif

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

int lpx_grid_components(lpx_ctx *ctx, uint32_t m_max, float r2, uint32_t *d_root, uint32_t *d_iota, bool cleared)
{
    if (m_max == 0)
        return LPX_OK;
    StageTimer tm(ctx, ST_NB_SCAN);
    FrameState *frame = (FrameState *)ctx->frame.p;
    uint32_t cap = 64;
    while (cap < 2 * m_max && cap < ctx->cell_cap)
        cap <<= 1;
    const dim3 blk(256), gc(((cap > LPX_CELL_BITS_WORDS ? cap : LPX_CELL_BITS_WORDS) + 255) / 256, 1, ctx->cur_b),
        gm((m_max + 255) / 256, 1, ctx->cur_b);
    unsigned long long *tkey = (unsigned long long *)ctx->cell_key.p;
    uint32_t *tparent = (uint32_t *)ctx->cell_parent.p, *thead = (uint32_t *)ctx->cell_rep.p;
    // (nothing here touches a buffer of the kd build or of the chunk tables: a forked front end runs them side by side)
    uint32_t *next = (uint32_t *)ctx->parent.p, *cells = (uint32_t *)ctx->cell_list.p;
    // (the keys of the cell list live in the segmentation's 64-bit key scratch: nothing of the clustering uses it)
    unsigned long long *ckeys = (unsigned long long *)ctx->key64_a.p;
    if (sizeof(unsigned long long) * (size_t)m_max > ctx->key64_a.bytes)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "cell keys of %u points do not fit their scratch", m_max);
    if (!cleared)  // (nb_index_kernel has emptied the table when it ran right in front of this: one launch less)
        hipLaunchKernelGGL(grid_clear_kernel, gc, blk, 0, ctx->stream, frame, tkey, tparent, thead, ctx->cell_cap,
                           ctx->fs_tag);
    const dim3 gi((m_max + GI_TILE - 1) / GI_TILE, 1, ctx->cur_b);
    hipLaunchKernelGGL(grid_insert_kernel, gi, dim3(GI_THREADS), 0, ctx->stream, frame, (const float *)ctx->OX.p,
                       (const float *)ctx->OY.p, (const float *)ctx->OZ.p, sqrtf(r2), tkey, thead, next, cells, ckeys,
                       (uint32_t *)ctx->cell_of.p, (float4 *)ctx->cell_xyz.p, ctx->cell_cap, ctx->fs_tag);
    uint32_t *tstart = (uint32_t *)ctx->cell_start.p;
    float4 *cpts = (float4 *)ctx->cell_pts.p;
    hipLaunchKernelGGL(grid_alloc_kernel, gm, blk, 0, ctx->stream, frame, (const uint32_t *)cells, (const uint32_t *)thead,
                       tstart, ctx->fs_tag);
    hipLaunchKernelGGL(grid_scatter_kernel, gm, blk, 0, ctx->stream, (const FrameState *)frame, (const float *)ctx->OX.p,
                       (const float *)ctx->OY.p, (const float *)ctx->OZ.p, (const uint32_t *)ctx->cell_of.p,
                       (const uint32_t *)next, (const uint32_t *)tstart, cpts, ctx->fs_tag);
    // (cell, partner) pairs, four per lane and trip, grid-stride (the device knows how many cells there are).  Few
    // workgroups per frame: the kernel is latency-bound, and under load what it costs the other chains is its resident
    // wavefronts x their lifetime -- measured on 16 chains of 32 KITTI frames: 512 / 2048 workgroups per frame (near /
    // far pass) 1589 Mpts/s, 128 / 512 1686, 32 / 128 1723, 16 / 64 1723, while the kernels alone take 336 + 313,
    // 295 + 234 (64 / 256) and 484 + 255 us (16 / 64).  A single frame keeps the wide launch.
    {
        // (round 5, after the bitmap and the phase-A changes left every workgroup less to do: 16 / 64 per frame 2 272-2 283
        // against 2 241-2 257 Mpts/s with 32 / 128)
        static const uint32_t g0_env = LPX_KNOB("LPX_GP_G0") ? (uint32_t)atoi(LPX_KNOB("LPX_GP_G0")) : 16u;
        static const uint32_t g1_env = LPX_KNOB("LPX_GP_G1") ? (uint32_t)atoi(LPX_KNOB("LPX_GP_G1")) : 64u;
        // (per 128k points of the largest frame: a 1M-point frame gets eight times the workgroups of a KITTI frame)
        static const uint32_t gs_env = LPX_KNOB("LPX_GP_SCALE") ? (uint32_t)atoi(LPX_KNOB("LPX_GP_SCALE")) : 1u;
        const uint32_t scale = gs_env ? (m_max + 131071u) / 131072u : 1u;
        const uint32_t w0 = ctx->cur_b > 1 || LPX_KNOB("LPX_GP_G0") ? g0_env * scale : 512u;
        const uint32_t w1 = ctx->cur_b > 1 || LPX_KNOB("LPX_GP_G1") ? g1_env * scale : 2048u;
        const uint32_t pg0 = (m_max * 13u + 255u) / 256u < w0 ? (m_max * 13u + 255u) / 256u : w0;
        const uint32_t pg1 = (m_max * 13u + 255u) / 256u < w1 ? (m_max * 13u + 255u) / 256u : w1;
#define GP_ARGS                                                                                                        \
    (const FrameState *)frame, (const unsigned long long *)tkey, tparent, (const uint32_t *)thead,                    \
        (const uint32_t *)tstart, (const uint32_t *)cells, (const unsigned long long *)ckeys, (const float4 *)cpts,    \
        (const float4 *)ctx->cell_xyz.p, r2,                                                                           \
        ctx->cell_cap, gp_dbg, ctx->fs_tag
        static const int gp_dbg = LPX_KNOB("LPX_GP_DBG") ? atoi(LPX_KNOB("LPX_GP_DBG")) : 0;  // timing experiments only
        hipLaunchKernelGGL(grid_pairs_kernel<false>, dim3(pg0, 1, ctx->cur_b), blk, 0, ctx->stream, GP_ARGS);
        hipLaunchKernelGGL(grid_compress_kernel, gm, blk, 0, ctx->stream, (const FrameState *)frame, (const uint32_t *)cells,
                           tparent, (const uint32_t *)thead, (const uint32_t *)tstart, (const float4 *)cpts,
                           (float4 *)ctx->cell_xyz.p, ctx->cell_cap, ctx->fs_tag);
        hipLaunchKernelGGL(grid_pairs_kernel<true>, dim3(pg1, 1, ctx->cur_b), blk, 0, ctx->stream, GP_ARGS);
#undef GP_ARGS
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_grid_flatten(lpx_ctx *ctx, uint32_t m_max, uint32_t *d_root, uint32_t *d_iota, uint32_t *first_hist)
{
    if (m_max == 0)
        return LPX_OK;
    const dim3 gtile((m_max + LPX_SORT_TILE - 1) / LPX_SORT_TILE, 1, ctx->cur_b);
    hipLaunchKernelGGL(grid_flatten_kernel, gtile, dim3(256), 0, ctx->stream, (const FrameState *)ctx->frame.p,
                       (uint32_t *)ctx->cell_parent.p, (const uint32_t *)ctx->cell_start.p,
                       (const uint32_t *)ctx->cell_of.p, d_root, d_iota, (uint8_t *)ctx->state.p,
                       (uint32_t *)ctx->valid.p, (uint32_t *)ctx->cc_lo.p, (uint32_t *)ctx->cc_hi.p, first_hist, ctx->fs_tag);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

