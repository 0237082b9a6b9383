// lpx_kd_shared.h -- device helpers shared by the kd-tree files: lpx_kdbuild.hip (the reference's tree), lpx_lists.hip
// (all radius lists), lpx_chunks.hip (candidate chunks per kd group) and lpx_grid.hip (clique-cell components).
// Included once per translation unit; everything here has internal linkage (anonymous namespace).
#pragma once

#include "lpx_internal.h"

#include <limits.h>

namespace
{
typedef float4 Node;  // x, y, z, original index (bit pattern)

__device__ __forceinline__ float akey(const Node &n, int axis)
{
    return axis == 0 ? n.x : (axis == 1 ? n.y : n.z);
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
// The order-preserving traversal both neighbour kernels start with (nb_group_kernel: all lists; nb_index_kernel: the
// candidate chunks of a kd group).
// ------------------------------------------------------------------------------------------------
constexpr int NB_WAVES = 4;
constexpr int NB_THREADS = NB_WAVES * WAVE;
constexpr int NB_SEQ = 512;     // interval items in LDS per block (6 KiB)
#ifndef LPX_NB_NODES
#define LPX_NB_NODES 1024
#endif
constexpr int NB_NODES = LPX_NB_NODES;  // candidate nodes staged in LDS per block (16 KiB): 6 workgroups per CU
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
}  // namespace
