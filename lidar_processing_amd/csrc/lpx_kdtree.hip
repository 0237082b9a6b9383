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

#include <limits.h>

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
template <int G>
struct Coop;

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
};

template <>
struct Coop<1024>
{
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
        for (int i = 0; i < 16; ++i)
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
        for (int i = 0; i < 16; ++i)
            s += cs[16 + i];
        __syncthreads();
        return s;
    }
};

// __unguarded_partition_pivot(first, last) by a group of G threads; returns the cut
template <int G>
__device__ int coop_partition_pivot(const View &v, int first, int last, int axis, int tid, uint32_t *cs)
{
    if (tid == 0)
        seq_median_to_first(v, first, last, axis);
    Coop<G>::sync();
    const float pv = nkey(v, first, axis);
    int cntL = 0, cntR = 0;
    for (int base = first + 1; base < last; base += G)
    {
        const int p = base + tid;
        const bool valid = p < last;
        const float k = valid ? nkey(v, p, axis) : 0.0f;
        const bool ge = valid && !(k < pv);  // left cursor stops here
        const bool le = valid && !(pv < k);  // right cursor stops here
        uint32_t rL, rR, tL, tR;
        Coop<G>::scan2(ge, le, rL, rR, tL, tR, cs);
        if (ge)
            v.lp[first + cntL + (int)rL - v.off] = (uint32_t)p;
        if (le)
            v.ra[first + cntR + (int)rR - v.off] = (uint32_t)p;
        cntL += (int)tL;
        cntR += (int)tR;
    }
    Coop<G>::sync();
    const int kmax = min(cntL, cntR);
    uint32_t my = 0;
    for (int k = tid; k < kmax; k += G)
    {
        const int l = (int)v.lp[first + k - v.off];
        const int r = (int)v.ra[first + cntR - 1 - k - v.off];
        if (l < r)
        {
            nswap(v, l, r);
            ++my;
        }
    }
    const int K = (int)Coop<G>::sum(my, cs);
    const int c1 = (K < cntL) ? (int)v.lp[first + K - v.off] : INT_MAX;
    const int c2 = (K > 0) ? (int)v.ra[first + cntR - K - v.off] : INT_MAX;
    Coop<G>::sync();
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
__global__ void kd_init_kernel(const float *__restrict__ OX, const float *__restrict__ OY,
                               const float *__restrict__ OZ, const FrameState *__restrict__ frame,
                               Node *__restrict__ nodes, uint32_t *__restrict__ parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame->n_obstacle)
        return;
    nodes[i] = make_float4(OX[i], OY[i], OZ[i], __uint_as_float(i));
    parent[i] = i;
}

constexpr int BLK_G = 1024;
constexpr int BLK_CAP = 4096;  // nodes staged in LDS: 64 KiB + 2 x 16 KiB scratch

// one workgroup per range of `level`: std::nth_element(b, mid, e) on axis level % 3
__global__ __launch_bounds__(BLK_G) void kd_block_kernel(Node *nodes, uint32_t *lpos, uint32_t *rasc,
                                                          const FrameState *__restrict__ frame, int level)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Node *l_nodes = (Node *)smem;
    uint32_t *l_lp = (uint32_t *)(smem + sizeof(Node) * BLK_CAP);
    uint32_t *l_ra = l_lp + BLK_CAP;
    uint32_t *cs = l_ra + BLK_CAP;

    const int tid = threadIdx.x;
    int b = 0, e = (int)frame->n_obstacle;
    descend(b, e, blockIdx.x, level);
    if (e - b < 2)
        return;
    const int axis = level % 3;
    const int nth = b + (e - b) / 2;
    int first = b, last = e;
    int depth_limit = 2 * floor_log2(e - b);
    View v;
    v.a = nodes;
    v.lp = lpos;
    v.ra = rasc;
    v.off = 0;
    bool staged = false;
    int sb = 0, se = 0;
    bool done = false;
    while (last - first > 3)
    {
        if (!staged && last - first <= BLK_CAP)
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
        const int cut = coop_partition_pivot<BLK_G>(v, first, last, axis, tid, cs);
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
}

constexpr int SUB_CAP = 512;   // nodes per wavefront subtree
constexpr int SUB_LEAF = 16;   // below this one lane finishes a subtree on its own
constexpr int SUB_WAVES = 4;

// one wavefront per range of `level` (<= SUB_CAP nodes): the whole subtree below it, in LDS
__global__ __launch_bounds__(SUB_WAVES *WAVE) void kd_subtree_kernel(Node *nodes, uint32_t *lpos, uint32_t *rasc,
                                                                     const FrameState *__restrict__ frame, int level)
{
    __shared__ Node l_nodes[SUB_WAVES][SUB_CAP];
    __shared__ uint32_t l_lp[SUB_WAVES][SUB_CAP];
    __shared__ uint32_t l_ra[SUB_WAVES][SUB_CAP];
    const int w = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const uint32_t r = blockIdx.x * SUB_WAVES + w;
    if (r >= (1u << level))
        return;
    int b = 0, e = (int)frame->n_obstacle;
    descend(b, e, r, level);
    const int n = e - b;
    if (n < 2)
        return;
    View v;
    const bool staged = n <= SUB_CAP;
    if (staged)
    {
        for (int i = lane; i < n; i += WAVE)
            l_nodes[w][i] = nodes[b + i];
        v.a = l_nodes[w];
        v.lp = l_lp[w];
        v.ra = l_ra[w];
        v.off = b;
    }
    else
    {  // host bound was wrong: stay correct, in global memory
        v.a = nodes;
        v.lp = lpos;
        v.ra = rasc;
        v.off = 0;
    }
    Coop<WAVE>::sync();

    // cooperative levels while ranges are larger than SUB_LEAF
    int s = 0;
    for (;; ++s)
    {
        // largest range at sub-level s is ceil-ish n / 2^s
        if ((n >> s) <= SUB_LEAF)
            break;
        const int axis = (level + s) % 3;
        for (uint32_t j = 0; j < (1u << s); ++j)
        {
            int rb = b, re = e;
            descend(rb, re, j, s);
            if (re - rb < 2)
                continue;
            const int nth = rb + (re - rb) / 2;
            int first = rb, last = re;
            int depth_limit = 2 * floor_log2(re - rb);
            bool done = false;
            while (last - first > 3)
            {
                if (depth_limit == 0)
                {
                    if (lane == 0)
                    {
                        seq_heap_select(v, first, nth + 1, last, axis);
                        nswap(v, first, nth);
                    }
                    done = true;
                    break;
                }
                --depth_limit;
                const int cut = coop_partition_pivot<WAVE>(v, first, last, axis, lane, nullptr);
                if (cut <= nth)
                    first = cut;
                else
                    last = cut;
            }
            if (!done && lane == 0)
                seq_insertion_sort(v, first, last, axis);
            Coop<WAVE>::sync();
        }
    }
    // leaf phase: lane j finishes sub-range j of sub-level s on its own
    for (uint32_t j = lane; j < (1u << s); j += WAVE)
    {
        int rb = b, re = e;
        descend(rb, re, j, s);
        if (re - rb >= 2)
            seq_build_subtree(v, rb, re, level + s);
    }
    Coop<WAVE>::sync();
    if (staged)
        for (int i = lane; i < n; i += WAVE)
            nodes[b + i] = l_nodes[w][i];
}

// ------------------------------------------------------------------------------------------------
// union-find (roots are the smallest original index of a component = its first FEC seed)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uf_ld(uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    }
}

// ------------------------------------------------------------------------------------------------
// radius search of every point, one thread per query (src/kdtree.hpp:292-341): emission order is
// the tree's pre-order, left before right, pruning on (node[axis]-target[axis])^2 <= r2.
// ------------------------------------------------------------------------------------------------
constexpr int RS_STACK = 48;

template <bool FILL>
__global__ __launch_bounds__(256) void radius_kernel(const Node *__restrict__ nodes, const float *__restrict__ OX,
                                                      const float *__restrict__ OY, const float *__restrict__ OZ,
                                                      const FrameState *__restrict__ frame, float r2,
                                                      uint32_t *__restrict__ len, const uint32_t *__restrict__ off,
                                                      uint32_t *__restrict__ nb_idx, float *__restrict__ nb_dist,
                                                      uint64_t cap, uint32_t *parent, int hook)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t M = frame->n_obstacle;
    if (i >= M)
        return;
    if (FILL && frame->nb_total > cap)
        return;
    const float tx = OX[i], ty = OY[i], tz = OZ[i];
    uint32_t sb[RS_STACK], se[RS_STACK];
    int sp = 0;
    sb[0] = 0;
    se[0] = M;  // axis in bits 30..31
    sp = 1;
    uint32_t cnt = 0;
    const uint32_t base = FILL ? off[i] : 0u;
    while (sp)
    {
        --sp;
        const uint32_t b = sb[sp];
        const uint32_t ea = se[sp];
        const uint32_t e = ea & 0x3fffffffu, axis = ea >> 30;
        const uint32_t mid = b + (e - b) / 2;
        const Node nd = nodes[mid];
        const float d0 = tx - nd.x, d1 = ty - nd.y, d2 = tz - nd.z;
        const float dist = d0 * d0 + (d1 * d1 + (d2 * d2 + 0.0f));  // src/kdtree.hpp:145-157
        if (dist <= r2)
        {
            if (FILL)
            {
                const uint32_t k = __float_as_uint(nd.w);
                nb_idx[base + cnt] = k;
                nb_dist[base + cnt] = dist;
                if (hook && k < i)
                    uf_unite(parent, i, k);
            }
            ++cnt;
        }
        const uint32_t next = (axis + 1) % 3;
        const float delta = (axis == 0 ? nd.x - tx : (axis == 1 ? nd.y - ty : nd.z - tz));
        const float ads = delta * delta;
        const bool has_l = mid > b, has_r = mid + 1 < e;
        if (ads <= r2)
        {
            if (has_r)
            {
                sb[sp] = mid + 1;
                se[sp] = e | (next << 30);
                ++sp;
            }
            if (has_l)
            {
                sb[sp] = b;
                se[sp] = mid | (next << 30);
                ++sp;
            }
        }
        else if (delta > 0.0f)
        {
            if (has_l)
            {
                sb[sp] = b;
                se[sp] = mid | (next << 30);
                ++sp;
            }
        }
        else if (has_r)
        {
            sb[sp] = mid + 1;
            se[sp] = e | (next << 30);
            ++sp;
        }
    }
    if (!FILL)
        len[i] = cnt;
}

__global__ void nb_check_kernel(FrameState *frame, uint64_t cap)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && frame->nb_total > cap)
        frame->status = (uint32_t)(-LPX_ERR_CAPACITY);
}

__global__ void layout_idx_kernel(const Node *__restrict__ nodes, uint32_t m, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
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
    hipLaunchKernelGGL(kd_init_kernel, dim3((m_max + 255) / 256), dim3(256), 0, ctx->stream, (const float *)ctx->OX.p,
                       (const float *)ctx->OY.p, (const float *)ctx->OZ.p, frame, nodes, (uint32_t *)ctx->parent.p);
    const size_t blk_lds = sizeof(Node) * BLK_CAP + 2 * sizeof(uint32_t) * BLK_CAP + 64 * sizeof(uint32_t);
    static bool attr_set = false;
    if (!attr_set)
    {
        LPX_HIP(ctx, hipFuncSetAttribute((const void *)kd_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)blk_lds));
        attr_set = true;
    }
    // sizes at level d are within one of m_max / 2^d
    int level = 0;
    uint32_t size = m_max;
    while (size > (uint32_t)SUB_CAP)
    {
        hipLaunchKernelGGL(kd_block_kernel, dim3(1u << level), dim3(BLK_G), blk_lds, ctx->stream, nodes, lpos, rasc,
                           frame, level);
        size = size / 2;  // larger child holds (size)/2 nodes at most: (n-1) - (n-1)/2 <= n/2
        ++level;
    }
    const uint32_t ranges = 1u << level;
    hipLaunchKernelGGL(kd_subtree_kernel, dim3((ranges + SUB_WAVES - 1) / SUB_WAVES), dim3(SUB_WAVES * WAVE), 0,
                       ctx->stream, nodes, lpos, rasc, frame, level);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_neighbours(lpx_ctx *ctx, uint32_t m_max, float r2, bool hook)
{
    if (m_max == 0)
        return LPX_OK;
    FrameState *frame = (FrameState *)ctx->frame.p;
    const Node *nodes = (const Node *)ctx->nodes.p;
    const dim3 blk(256), grd((m_max + 255) / 256);
    uint32_t *len = (uint32_t *)ctx->nb_len.p, *off = (uint32_t *)ctx->nb_off.p;
    {
        StageTimer tm(ctx, ST_NB_COUNT);
        hipLaunchKernelGGL((radius_kernel<false>), grd, blk, 0, ctx->stream, nodes, (const float *)ctx->OX.p,
                           (const float *)ctx->OY.p, (const float *)ctx->OZ.p, frame, r2, len,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (float *)nullptr, (uint64_t)0,
                           (uint32_t *)nullptr, 0);
    }
    {
        StageTimer tm(ctx, ST_NB_SCAN);
        int rc = lpx_exclusive_scan(ctx, len, off, m_max, &frame->n_obstacle, &frame->nb_total);
        if (rc)
            return rc;
        hipLaunchKernelGGL(nb_check_kernel, dim3(1), dim3(64), 0, ctx->stream, frame, ctx->cap_nb);
    }
    {
        StageTimer tm(ctx, ST_NB_FILL);
        hipLaunchKernelGGL((radius_kernel<true>), grd, blk, 0, ctx->stream, nodes, (const float *)ctx->OX.p,
                           (const float *)ctx->OY.p, (const float *)ctx->OZ.p, frame, r2, len, off,
                           (uint32_t *)ctx->nb_idx.p, (float *)ctx->nb_dist.p, ctx->cap_nb, (uint32_t *)ctx->parent.p,
                           hook ? 1 : 0);
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
