// lpx_segment.hip -- ground segmentation on gfx950.
//
// Replaces Segmenter::segment (reference src/segmentation.cpp:311-345) and what it calls:
//   form_planar_partitions :104-149   -> ingest + radix sort by (x, index) + x-sorted SoA gather
//   extract_initial_seeds  :151-217   -> radix sort of (segment, z) keys + seed_kernel
//   estimate_plane_coefficients :62-102 and fit_ground_plane :219-309
//                                     -> plane_pass_kernel (inlier test fused with the moment
//                                        accumulation of the next fit; last block solves the 3x3)
//   label / cloud scatter  :327-344   -> compact_kernel
//
// Data layout in HBM: SoA float arrays in x-sorted order (XS, YS, ZS), so a segment is one
// contiguous range and every pass is a coalesced stream of 12 B/point.  Moments are exact integers
// (coordinates rounded to 2^-16 m, |q| < 2^27): int64 lanes, wave64 shuffle reduction, one set of
// 64-bit atomics per block -- the sums do not depend on the order of accumulation, so the plane
// is bit-identical to the oracle's.
#include "lpx_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace
{
constexpr int SEG_THREADS = 256;
constexpr int SEG_WAVES = SEG_THREADS / WAVE;
#ifndef LPX_PASS_CHUNK
#define LPX_PASS_CHUNK 4096
#endif
#ifndef LPX_PASS_MINWAVES
#define LPX_PASS_MINWAVES 3
#endif
constexpr uint32_t SEG_CHUNK = LPX_PASS_CHUNK;  // points per block of the plane passes / the compaction (64 per lane of a one-wavefront
                                      // pass block; the int64 moment lanes hold 256 points: |q| < 2^27, products < 2^54)
constexpr float FIX_SCALE = 65536.0f;
constexpr float FIX_LIMIT = 2048.0f;       // below: |q| < 2^27, the int32 / int64 fast path
constexpr float FIX_CLAMP = 16777216.0f;   // 2^24 m: |q| <= 2^40, the wide path of the rare far points

// SegState::pad[1] is the pass a record is the state of (what a chained plane pass waits for, plane_chain_kernel); the seed
// kernels, which write S_0 into set 0, mark the record the LAST call left in set 1 with a number no pass has
constexpr uint32_t SEG_STATE_STALE = 0xffffffffu;

struct SegParams
{
    uint32_t n;        // points in the frame      } host values are the maximum over the frames of the
    uint32_t n_per;    // points per segment (n / P) } call; kernels rebind both to their own frame
    uint32_t P;
    uint32_t I;
    uint32_t bps;      // blocks per segment
    uint32_t chunk;    // points per block
    float z_floor;     // -1.5 * sensor_height
    float seed_thr;    // initial_seed_threshold
    float odt;         // orthogonal_distance_threshold
    uint32_t n_lpr;
    uint32_t part_stride;  // int64 words between the two sets of moment partials (seg_part)
    uint32_t head_solve;   // plane passes: every block solves at its head (launches that are resident all at once) instead
                           // of the last block of a segment at its tail (plane_pass_kernel)
};

// per-frame sizes: the launch geometry (bps) comes from the largest frame of the call, the ranges
// from the frame's own point count; blocks past the end of a short segment see an empty range
__device__ __forceinline__ void seg_bind(SegParams &prm, const FrameState *frame)
{
    prm.n = frame->n_in;
    prm.n_per = prm.n / prm.P;
}

// ------------------------------------------------------------------------------------------------
// frame state reset + input sizes (first launch of every call), counts hand-over (last)
// ------------------------------------------------------------------------------------------------
__global__ void frame_init_kernel(FrameState *frame, NArr n, uint32_t as_obstacles, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<0>(fs);
    frame = lpx_slot(frame, fs);
    uint32_t *w = (uint32_t *)frame;
    for (uint32_t i = threadIdx.x; i < sizeof(FrameState) / sizeof(uint32_t); i += blockDim.x)
        w[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        frame->n_obstacle = as_obstacles ? n.v[lpx_blk.z] : 0u;
        frame->n_in = n.v[lpx_blk.z];
    }
}

__global__ void counts_kernel(const FrameState *frame, uint32_t *counts, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<0>(fs);
    frame = lpx_slot(frame, fs);
    counts += 4 * (size_t)lpx_blk.z;
    if (threadIdx.x == 0)
    {
        counts[0] = frame->n_ground;
        counts[1] = frame->n_obstacle;
        counts[2] = frame->n_clusters;
        counts[3] = frame->status;
    }
}

// ------------------------------------------------------------------------------------------------
// K0 ingest: strided AoS -> SoA, x keys, iota; range check
// ------------------------------------------------------------------------------------------------
// Record layout: float32 x, y, z at byte offsets off.x, off.y, off.z of every `stride`-byte record -- a PCL point
// array (offsets 0, 4, 8; 16- or 32-byte records) or the data[] buffer of a sensor_msgs/PointCloud2 message with
// its point_step and field offsets (what the reference decodes on the host at src/conversions.cpp:62-85).
// ALIGNED: base, stride and offsets are multiples of 4 (every PCL type, every sane message); otherwise bytes.
struct XyzOff
{
    uint32_t x, y, z;
};

template <bool ALIGNED>
__device__ __forceinline__ float ld_f32(const char *p)
{
    if (ALIGNED)
        return *(const float *)p;
    const unsigned char *b = (const unsigned char *)p;
    return __uint_as_float((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
}

// One workgroup per radix-sort tile (LPX_SORT_TILE points, eight per thread): besides the records it leaves the tile's
// histogram of the lowest key byte where the first pass of the x sort expects it (hist, block-major; null: not wanted),
// which saves that pass its histogram launch.
constexpr int INGEST_THREADS = 256;
constexpr int INGEST_ITEMS = LPX_SORT_TILE / INGEST_THREADS;

template <bool ALIGNED>
__global__ __launch_bounds__(INGEST_THREADS) void ingest_kernel(const char *__restrict__ pts, size_t stride, XyzOff off,
                                                                 float *__restrict__ X, float *__restrict__ Y,
                                                                 float *__restrict__ Z, float4 *__restrict__ P4,
                                                                 uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                                 FrameState *__restrict__ frame,
                                                                 float4 *__restrict__ nodes, uint32_t *__restrict__ hist,
                                                                 FV fv)
{
    const LpxBlock lpx_blk = lpx_block<0>(fv.fs);
    __shared__ uint32_t h[256];
    pts += (size_t)lpx_blk.z * fv.upitch * stride;
    P4 = lpx_slot(P4, fv.fs);
    X = lpx_slot(X, fv.fs);
    Y = lpx_slot(Y, fv.fs);
    Z = lpx_slot(Z, fv.fs);
    key = lpx_slot(key, fv.fs);
    val = lpx_slot(val, fv.fs);
    frame = lpx_slot(frame, fv.fs);
    nodes = lpx_slot(nodes, fv.fs);
    hist = lpx_slot(hist, fv.fs);
    const uint32_t n = frame->n_in, tid = threadIdx.x;
    if (hist)
    {
        h[tid] = 0;
        __syncthreads();
    }
    float amax = 0.0f, nonfinite = 0.0f;
    float xs[INGEST_ITEMS], ys[INGEST_ITEMS], zs[INGEST_ITEMS];
#pragma unroll
    for (int r = 0; r < INGEST_ITEMS; ++r)  // the loads of a thread's eight records go out together
    {
        const uint32_t i = lpx_blk.x * LPX_SORT_TILE + r * INGEST_THREADS + tid;
        xs[r] = ys[r] = zs[r] = 0.0f;
        if (i < n)
        {
            const char *p = pts + (size_t)i * stride;
            xs[r] = ld_f32<ALIGNED>(p + off.x);
            ys[r] = ld_f32<ALIGNED>(p + off.y);
            zs[r] = ld_f32<ALIGNED>(p + off.z);
        }
    }
#pragma unroll
    for (int r = 0; r < INGEST_ITEMS; ++r)
    {
        const uint32_t i = lpx_blk.x * LPX_SORT_TILE + r * INGEST_THREADS + tid;
        const float x = xs[r], y = ys[r], z = zs[r];
        if (i < n)
        {
            if (P4)
                P4[i] = make_float4(x, y, z, 0.0f);  // one 16-byte record per point: whoever gathers it touches one line
            else if (X)  // (neither: the consumers read the caller's records where they lie -- lpx_ctx::rec_direct)
            {
                X[i] = x;
                Y[i] = y;
                Z[i] = z;
            }
            if (key)
            {
                const uint32_t k = lpx_float_key(x);
                key[i] = k;
                if (val)
                    val[i] = i;
                if (hist)
                    atomicAdd(&h[k & 255u], 1u);
            }
            if (nodes)
                nodes[i] = make_float4(x, y, z, __uint_as_float(i));
        }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)));  // fmaxf drops a NaN operand, hence:
        nonfinite += (x - x) + (y - y) + (z - z);                        // 0 for finite input, NaN for NaN / Inf
    }
    // Any finite cloud is processed like the reference does (src/segmentation.cpp:311-345).  NaN / Inf are
    // undefined behaviour upstream (comparators) and flag the frame.  Coordinates beyond +-2048 m leave the
    // int32 fixed-point range of the moment fast path: the plane kernels give those points the wide path.
    if (!(nonfinite == 0.0f))
    {
        frame->status = (uint32_t)(-LPX_ERR_RANGE);
        frame->n_obstacle = 0;  // nothing downstream runs on non-finite coordinates
    }
    else if (!(amax < FIX_LIMIT))
        frame->has_far = 1u;
    if (hist)
    {
        __syncthreads();
        hist[lpx_blk.x * 256u + tid] = h[tid];  // block-major row of this tile, as radix_hist_kernel writes it
    }
}

// ------------------------------------------------------------------------------------------------
// gather into x-sorted SoA + composite (segment, z) keys
// ------------------------------------------------------------------------------------------------
// Eight points per thread: the eight index loads go out together, then the eight dependent 16-byte gathers -- two
// round trips per wavefront instead of two per 64 points (a wavefront that waits holds its slot: lpx_primitives.hip).
constexpr int GATHER_ITEMS = 8;
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *__restrict__ sidx, const float4 *__restrict__ P4,
                                                     float *__restrict__ XS, float *__restrict__ YS,
                                                     float *__restrict__ ZS, uint64_t *__restrict__ zkey, SegParams prm,
                                                     const FrameState *__restrict__ frame, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<0>(fs);
    sidx = lpx_slot(sidx, fs);
    P4 = lpx_slot(P4, fs);
    XS = lpx_slot(XS, fs);
    YS = lpx_slot(YS, fs);
    ZS = lpx_slot(ZS, fs);
    zkey = lpx_slot(zkey, fs);
    seg_bind(prm, lpx_slot(frame, fs));
    const uint32_t p0 = lpx_blk.x * (256u * GATHER_ITEMS) + threadIdx.x;
    uint32_t si[GATHER_ITEMS];
#pragma unroll
    for (int r = 0; r < GATHER_ITEMS; ++r)
    {
        const uint32_t p = p0 + r * 256u;
        si[r] = p < prm.n ? sidx[p] : 0u;
    }
    float4 q[GATHER_ITEMS];
#pragma unroll
    for (int r = 0; r < GATHER_ITEMS; ++r)
        q[r] = P4[si[r]];  // one random 16-byte read per point
#pragma unroll
    for (int r = 0; r < GATHER_ITEMS; ++r)
    {
        const uint32_t p = p0 + r * 256u;
        if (p >= prm.n)
            continue;
        const float z = q[r].z;
        XS[p] = q[r].x;
        YS[p] = q[r].y;
        ZS[p] = z;
        uint32_t seg = prm.n_per ? p / prm.n_per : prm.P;
        if (seg > prm.P)
            seg = prm.P;  // the N mod P tail (Q2) sorts behind every segment
        if (zkey)  // only the sort-based seed path needs the composite keys
            zkey[p] = ((uint64_t)seg << 32) | lpx_float_key(z);
    }
}

// ------------------------------------------------------------------------------------------------
// seeds: one block per segment over the z-sorted keys (extract_initial_seeds, :151-217)
// ------------------------------------------------------------------------------------------------
constexpr int SEED_LDS = 4096;

// The three sets of far-point accumulators of segment s start a frame at zero (pass t adds into set t % 3, the head
// of pass t + 1 reads it and clears set (t + 2) % 3: plane_pass_kernel).  Called by the first 3 * LPX_FAR_WORDS threads
// of the seed kernels, the launch before pass 0.
__device__ __forceinline__ void seg_far_reset(long long *facc, uint32_t s, uint32_t tid)
{
    if (tid < 3u * LPX_FAR_WORDS)
        facc[((size_t)(tid / LPX_FAR_WORDS) * LPX_MAX_PARTITIONS + s) * LPX_FAR_WORDS + tid % LPX_FAR_WORDS] = 0;
}

__global__ __launch_bounds__(SEG_THREADS) void seed_kernel(const uint64_t *__restrict__ zsorted, SegParams prm,
                                                            SegState *__restrict__ st, long long *__restrict__ facc,
                                                            const FrameState *__restrict__ frame, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<2>(fs);
    __shared__ __attribute__((aligned(16))) float zbuf[SEED_LDS];
    __shared__ float s_sum;
    __shared__ uint32_t s_cut;
    zsorted = lpx_slot(zsorted, fs);
    st = lpx_slot(st, fs);
    facc = lpx_slot(facc, fs);
    seg_bind(prm, lpx_slot(frame, fs));
    const uint32_t s = lpx_blk.x;
    const uint32_t ns = prm.n_per;
    const uint64_t *zs = zsorted + (size_t)s * ns;
    const uint32_t tid = threadIdx.x;

    seg_far_reset(facc, s, tid);

    // first index with z > z_floor (sorted ascending): upper bound
    if (tid == 0)
    {
        uint32_t lo = 0, hi = ns;
        while (lo < hi)
        {
            const uint32_t mid = lo + (hi - lo) / 2;
            const float z = lpx_key_float((uint32_t)zs[mid]);
            if (z > prm.z_floor)
                hi = mid;
            else
                lo = mid + 1;
        }
        s_cut = (lo < ns) ? lo : 0u;  // :171-182 no point above the floor -> nothing dropped
        s_sum = 0.0f;
    }
    __syncthreads();
    const uint32_t cut = s_cut;
    const uint32_t nrem = ns - cut;
    const uint32_t n_rep = min(nrem, prm.n_lpr);
    // sequential float sum in ascending z (:193-197): staged through LDS, added by one lane
    for (uint32_t base = 0; base < n_rep; base += SEED_LDS)
    {
        const uint32_t cnt = min((uint32_t)SEED_LDS, n_rep - base);
        for (uint32_t i = tid; i < cnt; i += SEG_THREADS)
            zbuf[i] = lpx_key_float((uint32_t)zs[cut + base + i]);
        __syncthreads();
        if (tid == 0)
        {
            // strictly sequential adds (bit-exact with the reference's loop); the LDS reads are hoisted
            // sixteen at a time so that only the 4-cycle add chain is serial
            float sum = s_sum;
            uint32_t i = 0;
            for (; i + 16 <= cnt; i += 16)
            {
                const float4 a = *(const float4 *)&zbuf[i], b = *(const float4 *)&zbuf[i + 4];
                const float4 c = *(const float4 *)&zbuf[i + 8], d = *(const float4 *)&zbuf[i + 12];
                sum += a.x;
                sum += a.y;
                sum += a.z;
                sum += a.w;
                sum += b.x;
                sum += b.y;
                sum += b.z;
                sum += b.w;
                sum += c.x;
                sum += c.y;
                sum += c.z;
                sum += c.w;
                sum += d.x;
                sum += d.y;
                sum += d.z;
                sum += d.w;
            }
            for (; i < cnt; ++i)
                sum += zbuf[i];
            s_sum = sum;
        }
        __syncthreads();
    }
    if (tid == 0)
    {
        const float z_mean = s_sum / (float)n_rep;
        const float z_max = z_mean + prm.seed_thr;
        // first index in the remainder with z > z_max; none -> no seeds (Q4)
        uint32_t lo = 0, hi = nrem;
        while (lo < hi)
        {
            const uint32_t mid = lo + (hi - lo) / 2;
            const float z = lpx_key_float((uint32_t)zs[cut + mid]);
            if (z > z_max)
                hi = mid;
            else
                lo = mid + 1;
        }
        const uint32_t n_seed = (lo < nrem) ? lo : 0u;
        SegState o;
        o.lo_excl = (cut > 0) ? prm.z_floor : -INFINITY;
        o.hi_incl = z_max;
        o.has_seeds = n_seed > 0;
        o.failed = (ns < 3) ? 2u : 0u;  // :224-229 nothing is labelled
        o.plane[0] = o.plane[1] = o.plane[2] = o.plane[3] = 0.0f;
        o.fitted = 0;
        o.thr = 0.0f;
        o.pad[0] = o.pad[1] = 0;
        st[s] = o;
        st[LPX_MAX_PARTITIONS + s].pad[1] = SEG_STATE_STALE;  // (set 1 still holds the last call's record)
    }
}

// ------------------------------------------------------------------------------------------------
// seeds without sorting the segment: the statistics of extract_initial_seeds (:151-217) only need the
// n_lpr LOWEST z above the floor in ascending order (the float sum at :193-197 is sequential), plus two
// counts.  One workgroup per segment keeps the segment's keys in registers, finds the key of rank n_rep
// by a 4-round radix select on LDS histograms (9 + 9 + 9 + 5 bits, one private histogram per wavefront:
// ground z values share their leading bits and would serialise a shared one), collects the keys below it, sorts
// those <= 8192 keys in LDS (bitonic) and sums them with one lane.  Same results as seed_kernel over the
// sorted segment; replaces five radix-sort passes over all N points.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;  // a frame alone on the device: sixteen wavefronts, 24 keys per thread
// In launch chains the same kernel runs with SEL_NARROW threads and 96 keys per thread: under load a 1024-thread workgroup
// waits until ONE compute unit has sixteen free wave slots (seed_select_kernel: 0.20 ms alone, 3.0 ms with sixteen
// chains in flight -- fifteen times, where the 256-thread kernels of a chain wait four to ten), a 256-thread workgroup
// fits anywhere.
constexpr int SEL_NARROW = 256;
constexpr uint32_t SEL_MAX_POINTS = 24576;                 // points per segment: 24 (96) keys per thread
constexpr uint32_t SEL_MAX_LPR = 8192;                     // keys sorted in LDS
constexpr int SEL_BINS = 512;  // per wavefront; the private histograms alias the sort buffer

template <int THREADS>
__device__ __forceinline__ uint32_t sel_block_sum(uint32_t v, uint32_t *red, uint32_t tid)
{
    v = lpx_wave_sum_u32(v);
    if ((tid % WAVE) == 0)
        red[tid / WAVE] = v;
    __syncthreads();
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < THREADS / WAVE; ++i)
        s += red[i];
    __syncthreads();
    return s;
}

// compare-exchanges between registers a and a + D of one thread (a holds the lower index of the pair), all T / 2 of them
template <int T, int D>
__device__ __forceinline__ void sel_ce_regs(uint32_t (&e)[T], uint32_t w, uint32_t lane, uint32_t k2)
{
#pragma unroll
    for (int a = 0; a < T; ++a)
        if ((a & D) == 0)
        {
            const bool up = (((w * (T * WAVE) + a * WAVE + lane) & k2) == 0);
            const uint32_t mn = min(e[a], e[a + D]), mx = max(e[a], e[a + D]);
            e[a] = up ? mn : mx;
            e[a + D] = up ? mx : mn;
        }
}

// the <= SEL_MAX_LPR keys in s_buf (padded with 0xffffffff to n_sort, a power of two), ascending; all THREADS threads
template <int THREADS>
__device__ __forceinline__ void sel_bitonic_sort(uint32_t *s_buf, uint32_t n_sort, uint32_t tid)
{
    // Bitonic sort, ascending, of the T keys per thread i = 64 T w + 64 t + lane (T = 8 with sixteen wavefronts, 32 with
    // four).  A compare-exchange at distance j2 pairs: the same lane of another register (64 <= j2 < 64 T), another lane
    // of the same register (j2 < 64, one cross-lane read) or another wavefront's block (j2 >= 64 T, through LDS with
    // workgroup barriers).
    constexpr int T = (int)SEL_MAX_LPR / THREADS;
    constexpr uint32_t WBLK = (uint32_t)T * WAVE;  // keys of one wavefront
    static_assert(T == 8 || T == 32, "register stages are written for 8 or 32 keys per thread");
    const uint32_t w = tid / WAVE, lane = tid % WAVE;
    uint32_t e[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
        e[t] = s_buf[w * WBLK + t * WAVE + lane];
    for (uint32_t k2 = 2; k2 <= n_sort; k2 <<= 1)
        for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1)
        {
            if (j2 >= WBLK)
            {
                __syncthreads();
#pragma unroll
                for (int t = 0; t < T; ++t)
                    s_buf[w * WBLK + t * WAVE + lane] = e[t];
                __syncthreads();
#pragma unroll
                for (int t = 0; t < T; ++t)
                {
                    const uint32_t i = w * WBLK + t * WAVE + lane;
                    const uint32_t pv = s_buf[(i ^ j2) & (SEL_MAX_LPR - 1)];
                    const bool keep_min = ((i & j2) == 0) == ((i & k2) == 0);
                    e[t] = keep_min ? min(e[t], pv) : max(e[t], pv);
                }
            }
            else if (j2 >= (uint32_t)WAVE)
            {
                if (j2 == 64)
                    sel_ce_regs<T, 1>(e, w, lane, k2);
                else if (j2 == 128)
                    sel_ce_regs<T, 2>(e, w, lane, k2);
                else if (j2 == 256)
                    sel_ce_regs<T, 4>(e, w, lane, k2);
                else if (T > 8)
                {
                    if (j2 == 512)
                        sel_ce_regs<T, (T > 8 ? 8 : 1)>(e, w, lane, k2);
                    else
                        sel_ce_regs<T, (T > 8 ? 16 : 1)>(e, w, lane, k2);
                }
            }
            else
            {
#pragma unroll
                for (int t = 0; t < T; ++t)
                {
                    const uint32_t i = w * WBLK + t * WAVE + lane;
                    const uint32_t pv = (uint32_t)__shfl_xor((int)e[t], (int)j2, WAVE);
                    const bool keep_min = ((i & j2) == 0) == ((i & k2) == 0);
                    e[t] = keep_min ? min(e[t], pv) : max(e[t], pv);
                }
            }
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t)
        s_buf[w * WBLK + t * WAVE + lane] = e[t];
    __syncthreads();
}

// mean of the n_rep lowest z (float bits in s_buf, ascending) + threshold; thread 0 only
__device__ __forceinline__ float sel_sequential_zmax(const uint32_t *s_buf, uint32_t n_rep, float seed_thr)
{
    // strictly sequential adds in ascending z (bit-exact with the reference's loop, :193-197); the LDS reads
    // are hoisted sixteen at a time so that only the add chain is serial
    const float *zf = (const float *)s_buf;
    float sum = 0.0f;
    uint32_t i = 0;
    for (; i + 16 <= n_rep; i += 16)
    {
        const float4 a = *(const float4 *)&zf[i], b = *(const float4 *)&zf[i + 4];
        const float4 c4 = *(const float4 *)&zf[i + 8], d = *(const float4 *)&zf[i + 12];
        sum += a.x;
        sum += a.y;
        sum += a.z;
        sum += a.w;
        sum += b.x;
        sum += b.y;
        sum += b.z;
        sum += b.w;
        sum += c4.x;
        sum += c4.y;
        sum += c4.z;
        sum += c4.w;
        sum += d.x;
        sum += d.y;
        sum += d.z;
        sum += d.w;
    }
    for (; i < n_rep; ++i)
        sum += zf[i];
    return sum / (float)n_rep + seed_thr;
}

// (Plane pass 0 -- the moments of the seeds -- was folded into this kernel once: it holds every z of the segment in
// registers, x and y of the seeds are two more loads per point.  One launch less, 0.013 ms more per 64-frame chain alone,
// and 2 % LESS throughput with sixteen chains in flight: this is a 1024-thread workgroup, the kind that waits longest
// for a compute unit under load, and everything added to it is added to that wait.  Taken out again.)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void seed_select_kernel(const float *__restrict__ ZS, SegParams prm,
                                                                   SegState *__restrict__ st,
                                                                   long long *__restrict__ facc,
                                                                   const FrameState *__restrict__ frame, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<2>(fs);
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[SEL_MAX_LPR];
    constexpr int SEL_PTS = (int)SEL_MAX_POINTS / THREADS;  // keys per thread, in registers
    constexpr int NW = THREADS / WAVE;
    static_assert(SEL_BINS * NW <= (int)SEL_MAX_LPR, "private histograms alias the sort buffer");
    static_assert(SEL_BINS % THREADS == 0 || THREADS % SEL_BINS == 0, "bins per thread");
    uint32_t *s_hist = s_buf;  // [wavefront][bin] during the select, before the buffer is filled
    __shared__ uint32_t s_red[NW];
    __shared__ uint32_t s_pick[2];  // bin, count before the bin
    __shared__ uint32_t s_fill;
    __shared__ float s_zmax;
    ZS = lpx_slot(ZS, fs);
    st = lpx_slot(st, fs);
    facc = lpx_slot(facc, fs);
    seg_bind(prm, lpx_slot(frame, fs));
    const uint32_t s = lpx_blk.x, tid = threadIdx.x;
    const uint32_t ns = prm.n_per;
    const uint32_t base = s * ns;
    seg_far_reset(facc, s, tid);

    uint32_t key[SEL_PTS];
    // (slot u of a thread holds a point of the segment iff tid + u THREADS < ns: recomputed where needed -- 96 slots per
    // thread do not fit a 32-bit mask)
#define SEL_VALID(u) (tid + (uint32_t)(u) * THREADS < ns)
#pragma unroll
    for (int u = 0; u < SEL_PTS; ++u)
        key[u] = SEL_VALID(u) ? lpx_float_key(ZS[base + tid + u * THREADS]) : 0xffffffffu;
    // points at or below the floor are dropped, unless that leaves nothing (:171-182)
    const uint32_t floor_key = lpx_float_key(prm.z_floor);
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < SEL_PTS; ++u)
        c += SEL_VALID(u) && key[u] <= floor_key;
    const uint32_t c_floor = sel_block_sum<THREADS>(c, s_red, tid);
    const uint32_t cut = (c_floor < ns) ? c_floor : 0u;
    const uint32_t nrem = ns - cut;
    const uint32_t n_rep = min(nrem, prm.n_lpr);
    // slot u is in the remainder:
#define SEL_REM(u) (SEL_VALID(u) && (cut == 0 || key[u] > floor_key))

    // key K of rank n_rep (1-based) among the remainder, `need` = how many copies of K belong to the n_rep lowest
    uint32_t prefix = 0, pmask = 0, need = n_rep;
    if (n_rep)
    {
        const int shifts[4] = {23, 14, 5, 0}, bits[4] = {9, 9, 9, 5};
        const uint32_t wv = tid / WAVE;
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            for (uint32_t i = tid; i < SEL_BINS * NW; i += THREADS)
                s_hist[i] = 0;
            __syncthreads();
            const uint32_t dmask = (1u << bits[r]) - 1u;
#pragma unroll
            for (int u = 0; u < SEL_PTS; ++u)
                if (SEL_REM(u) && (key[u] & pmask) == prefix)
                    atomicAdd(&s_hist[wv * SEL_BINS + ((key[u] >> shifts[r]) & dmask)], 1u);
            __syncthreads();
            // bin totals over the wavefronts, then the first bin whose inclusive prefix count reaches `need`: thread t owns
            // the BPT consecutive bins BPT t .. (one with 1024 threads -- the upper half idles --, two with 256)
            constexpr int BPT = SEL_BINS > THREADS ? SEL_BINS / THREADS : 1;
            uint32_t hb[BPT], h = 0;
#pragma unroll
            for (int q = 0; q < BPT; ++q)
            {
                hb[q] = 0;
                if (tid * BPT + q < (uint32_t)SEL_BINS)
                    for (int ww = 0; ww < NW; ++ww)
                        hb[q] += s_hist[ww * SEL_BINS + tid * BPT + q];
                h += hb[q];
            }
            const uint32_t incl = lpx_wave_incl_scan_u32(h);
            if ((tid % WAVE) == WAVE - 1)
                s_red[tid / WAVE] = incl;
            __syncthreads();
            uint32_t wbase = 0;
            for (uint32_t i = 0; i < tid / WAVE; ++i)
                wbase += s_red[i];
            uint32_t before = wbase + incl - h;  // keys in bins below this thread's first
#pragma unroll
            for (int q = 0; q < BPT; ++q)
            {
                if (tid * BPT + q < (uint32_t)SEL_BINS && before < need && need <= before + hb[q])
                {
                    s_pick[0] = tid * BPT + q;
                    s_pick[1] = before;
                }
                before += hb[q];
            }
            __syncthreads();
            prefix |= s_pick[0] << shifts[r];
            pmask |= dmask << shifts[r];
            need -= s_pick[1];
            __syncthreads();
        }
    }
    const uint32_t K = prefix;
    // the n_rep lowest keys: everything below K, then `need` copies of K; padded to a power of two for the sort
    uint32_t n_sort = 1;
    while (n_sort < n_rep)
        n_sort <<= 1;
    if (tid == 0)
        s_fill = 0;
    for (uint32_t i = tid; i < SEL_MAX_LPR; i += THREADS)
        s_buf[i] = 0xffffffffu;
    __syncthreads();
    if (n_rep)
    {
        // one LDS atomic per wavefront and slot, not per key
        const unsigned long long lt = lpx_lanemask_lt();
#pragma unroll
        for (int u = 0; u < SEL_PTS; ++u)
        {
            const bool take = SEL_REM(u) && key[u] < K;
            const unsigned long long tm = __ballot(take);
            if (tm)
            {
                uint32_t pos = 0;
                if ((tid % WAVE) == 0)
                    pos = atomicAdd(&s_fill, (uint32_t)__popcll(tm));
                pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
                if (take)
                    s_buf[pos + __popcll(tm & lt)] = key[u];
            }
        }
        __syncthreads();
        const uint32_t below = s_fill;  // == n_rep - need
        for (uint32_t i = tid; i < need; i += THREADS)
            s_buf[below + i] = K;
        __syncthreads();
        sel_bitonic_sort<THREADS>(s_buf, n_sort, tid);
        // keys -> float bits in place, so that the summing lane reads floats
        for (uint32_t i = tid; i < n_sort; i += THREADS)
            s_buf[i] = __float_as_uint(lpx_key_float(s_buf[i]));
    }
    __syncthreads();
    if (tid == 0)
        s_zmax = sel_sequential_zmax(s_buf, n_rep, prm.seed_thr);
    __syncthreads();
    const float z_max = s_zmax;
    // points of the remainder up to z_max; none above z_max -> no seeds (Q4)
    c = 0;
#pragma unroll
    for (int u = 0; u < SEL_PTS; ++u)
        c += SEL_REM(u) && !(lpx_key_float(key[u]) > z_max);
    const uint32_t cnt_le = sel_block_sum<THREADS>(c, s_red, tid);
    if (tid == 0)
    {
        const uint32_t n_seed = (cnt_le < nrem) ? cnt_le : 0u;
        SegState o;
        o.lo_excl = (cut > 0) ? prm.z_floor : -INFINITY;
        o.hi_incl = z_max;
        o.has_seeds = n_seed > 0;
        o.failed = (ns < 3) ? 2u : 0u;  // :224-229 nothing is labelled
        o.plane[0] = o.plane[1] = o.plane[2] = o.plane[3] = 0.0f;
        o.fitted = 0;
        o.thr = 0.0f;
        o.pad[0] = o.pad[1] = 0;
        st[s] = o;
        st[LPX_MAX_PARTITIONS + s].pad[1] = SEG_STATE_STALE;  // (set 1 still holds the last call's record)
    }
}
#undef SEL_REM
#undef SEL_VALID

// ------------------------------------------------------------------------------------------------
// The same selection for segments that do not fit one workgroup's registers (BASELINE's 1M- and 5M-point clouds hold
// 83k and 208k points per segment).  They used to take the full (segment, z) sort -- a gather launch that also built
// 64-bit keys and five radix passes over all N of them, 1.5 GB of the 10.7 GB a 5M-point frame moves -- for statistics
// that need the n_lpr lowest z of a segment.  Here the radix select runs over the keys where they lie: four histogram
// passes over ZS (9 + 9 + 9 + 5 bits; tiles of 8192 keys, one workgroup each, private LDS histograms added to the
// segment's 512 global bins), every workgroup of pass r first reading the bin that pass r - 1 settled on; then one
// pass that collects the keys below the key of rank n_rep, and one workgroup per segment that sorts and sums them
// exactly like seed_select_kernel.  The rank is taken among ALL keys of the segment: the keys at or below the floor
// are its `cut` lowest, so the n_rep lowest of the remainder are ranks cut + 1 .. cut + n_rep (:171-182).  Whether any
// remainder point lies above z_max -- the only use of the count at :205-217 -- follows from the segment's largest key.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t SELW_TILE = 8192;
constexpr int SELW_THREADS = 256;
struct SelWide  // per segment, in the 64-bit key scratch (unused on this path)
{
    uint32_t c_floor;  // keys at or below the floor
    uint32_t max_key;  // the segment's largest key
    uint32_t fill;     // keys collected
    uint32_t pad;
    uint32_t hist[4][SEL_BINS];
    uint32_t keys[SEL_MAX_LPR];
};
struct SelPick
{
    uint32_t cut, nrem, n_rep, prefix, pmask, need;
};

// What the histograms of rounds [0, upto) say (all NT threads of the workgroup; s_scr: NT / 64 + 2 words)
template <int NT>
__device__ __forceinline__ SelPick selw_pick(const SelWide *sw, uint32_t ns, uint32_t n_lpr, int upto, uint32_t *s_scr,
                                             uint32_t tid)
{
    SelPick p;
    const uint32_t c_floor = sw->c_floor;
    p.cut = (c_floor < ns) ? c_floor : 0u;  // :171-182 no point above the floor -> nothing dropped
    p.nrem = ns - p.cut;
    p.n_rep = min(p.nrem, n_lpr);
    p.prefix = 0;
    p.pmask = 0;
    uint32_t need = p.cut + p.n_rep;  // 1-based rank among all keys of the segment
    const int shifts[4] = {23, 14, 5, 0}, bits[4] = {9, 9, 9, 5};
    for (int r = 0; r < upto; ++r)
    {
        uint32_t h0 = 0, h1 = 0;  // thread t < 256 owns bins 2 t and 2 t + 1
        if (tid < (uint32_t)SEL_BINS / 2u)
        {
            h0 = sw->hist[r][2u * tid];
            h1 = sw->hist[r][2u * tid + 1u];
        }
        const uint32_t hs = h0 + h1;
        const uint32_t incl = lpx_wave_incl_scan_u32(hs);
        if ((tid % WAVE) == WAVE - 1)
            s_scr[tid / WAVE] = incl;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t i = 0; i < tid / WAVE; ++i)
            wbase += s_scr[i];
        const uint32_t before = wbase + incl - hs;  // keys in bins below 2 t
        if (tid < (uint32_t)SEL_BINS / 2u)
        {
            if (before < need && need <= before + h0)
            {
                s_scr[NT / WAVE] = 2u * tid;
                s_scr[NT / WAVE + 1] = before;
            }
            else if (before + h0 < need && need <= before + hs)
            {
                s_scr[NT / WAVE] = 2u * tid + 1u;
                s_scr[NT / WAVE + 1] = before + h0;
            }
        }
        __syncthreads();
        p.prefix |= s_scr[NT / WAVE] << shifts[r];
        p.pmask |= ((1u << bits[r]) - 1u) << shifts[r];
        need -= s_scr[NT / WAVE + 1];
        __syncthreads();
    }
    p.need = need;
    return p;
}

__global__ void selw_clear_kernel(SelWide *__restrict__ sw, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<2>(fs);
    sw = lpx_slot(sw, fs) + lpx_blk.x;
    uint32_t *w = (uint32_t *)sw;
    for (uint32_t i = threadIdx.x; i < 4u + 4u * SEL_BINS; i += blockDim.x)
        w[i] = 0;
}

template <int R>  // R = 0 .. 3: histogram of round R; R = 4: collect the keys below the key the four rounds settled on
__global__ __launch_bounds__(SELW_THREADS) void selw_pass_kernel(const float *__restrict__ ZS, SegParams prm,
                                                                  SelWide *__restrict__ sw,
                                                                  const FrameState *__restrict__ frame, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<2>(fs);
    __shared__ uint32_t s_hist[SELW_THREADS / WAVE][SEL_BINS];
    __shared__ uint32_t s_scr[SELW_THREADS / WAVE + 2];
    ZS = lpx_slot(ZS, fs);
    seg_bind(prm, lpx_slot(frame, fs));
    const uint32_t s = lpx_blk.y, tid = threadIdx.x, ns = prm.n_per;
    const uint32_t t0 = lpx_blk.x * SELW_TILE;
    if (t0 >= ns)
        return;
    SelWide *w = lpx_slot(sw, fs) + s;
    const uint32_t base = s * ns, floor_key = lpx_float_key(prm.z_floor);
    const SelPick pk = selw_pick<SELW_THREADS>(w, ns, prm.n_lpr, R < 4 ? R : 4, s_scr, tid);
    const int shifts[4] = {23, 14, 5, 0}, bits[4] = {9, 9, 9, 5};
    if (R < 4)
    {
        for (uint32_t i = tid; i < (SELW_THREADS / WAVE) * SEL_BINS; i += SELW_THREADS)
            (&s_hist[0][0])[i] = 0;
        __syncthreads();
    }
    const uint32_t wv = tid / WAVE;
    const unsigned long long lt = lpx_lanemask_lt();
    uint32_t cfl = 0, mx = 0;
    for (uint32_t i0 = 0; i0 < SELW_TILE; i0 += 4 * SELW_THREADS)
    {
        uint32_t k[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)  // four keys per thread and trip: the loads go out together
        {
            const uint32_t q = t0 + i0 + u * SELW_THREADS + tid;
            in[u] = q < ns;
            k[u] = in[u] ? lpx_float_key(ZS[base + q]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            if (R == 0 && in[u])
            {
                cfl += k[u] <= floor_key;
                mx = max(mx, k[u]);
            }
            if (R < 4)
            {
                if (in[u] && (k[u] & pk.pmask) == pk.prefix)
                    atomicAdd(&s_hist[wv][(k[u] >> shifts[R & 3]) & ((1u << bits[R & 3]) - 1u)], 1u);
            }
            else
            {
                // the n_rep lowest of the remainder without the copies of K itself: below K, above the floor if any
                // key was dropped there; one atomic per wavefront and slot
                const bool take = in[u] && k[u] < pk.prefix && (pk.cut == 0u || k[u] > floor_key);
                const unsigned long long tm = __ballot(take);
                if (tm)
                {
                    uint32_t pos = 0;
                    if ((tid % WAVE) == 0)
                        pos = atomicAdd(&w->fill, (uint32_t)__popcll(tm));
                    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
                    if (take)
                        w->keys[pos + __popcll(tm & lt)] = k[u];
                }
            }
        }
    }
    if (R < 4)
    {
        __syncthreads();
        for (uint32_t b = tid; b < (uint32_t)SEL_BINS; b += SELW_THREADS)
        {
            uint32_t h = 0;
#pragma unroll
            for (int ww = 0; ww < SELW_THREADS / WAVE; ++ww)
                h += s_hist[ww][b];
            if (h)
                atomicAdd(&w->hist[R & 3][b], h);
        }
    }
    if (R == 0)
    {
        cfl = lpx_wave_sum_u32(cfl);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
        if ((tid % WAVE) == 0)
        {
            if (cfl)
                atomicAdd(&w->c_floor, cfl);
            atomicMax(&w->max_key, mx);
        }
    }
}

__global__ __launch_bounds__(SEL_THREADS) void selw_final_kernel(SegParams prm, const SelWide *__restrict__ sw,
                                                                  SegState *__restrict__ st, long long *__restrict__ facc,
                                                                  const FrameState *__restrict__ frame, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<2>(fs);
    __shared__ __attribute__((aligned(16))) uint32_t s_buf[SEL_MAX_LPR];
    __shared__ uint32_t s_scr[SEL_THREADS / WAVE + 2];
    __shared__ float s_zmax, s_zmin;
    st = lpx_slot(st, fs);
    facc = lpx_slot(facc, fs);
    seg_bind(prm, lpx_slot(frame, fs));
    const uint32_t s = lpx_blk.x, tid = threadIdx.x, ns = prm.n_per;
    const SelWide *w = lpx_slot(sw, fs) + s;
    seg_far_reset(facc, s, tid);
    if (ns == 0)
    {
        if (tid == 0)
        {
            const float none = (float)ns;  // (a frame of the batch with fewer points than partitions)
            SegState o;
            o.lo_excl = -INFINITY;
            o.hi_incl = none / none + prm.seed_thr;  // as seed_select_kernel leaves it: the mean of nothing
            o.has_seeds = 0;
            o.failed = 2u;
            o.plane[0] = o.plane[1] = o.plane[2] = o.plane[3] = 0.0f;
            o.fitted = 0;
            o.thr = 0.0f;
            o.pad[0] = o.pad[1] = 0;
            st[s] = o;
            st[LPX_MAX_PARTITIONS + s].pad[1] = SEG_STATE_STALE;  // (set 1 still holds the last call's record)
        }
        return;
    }
    const SelPick pk = selw_pick<SEL_THREADS>(w, ns, prm.n_lpr, 4, s_scr, tid);
    const uint32_t K = pk.prefix, n_rep = pk.n_rep, below = n_rep - pk.need;  // == w->fill
    uint32_t n_sort = 1;
    while (n_sort < n_rep)
        n_sort <<= 1;
    for (uint32_t i = tid; i < SEL_MAX_LPR; i += SEL_THREADS)
        s_buf[i] = i < below ? w->keys[i] : (i < n_rep ? K : 0xffffffffu);
    __syncthreads();
    sel_bitonic_sort<SEL_THREADS>(s_buf, n_sort, tid);
    // keys -> float bits in place, so that the summing lane reads floats
    for (uint32_t i = tid; i < n_sort; i += SEL_THREADS)
        s_buf[i] = __float_as_uint(lpx_key_float(s_buf[i]));
    __syncthreads();
    if (tid == 0)
    {
        s_zmin = __uint_as_float(s_buf[0]);
        s_zmax = sel_sequential_zmax(s_buf, n_rep, prm.seed_thr);
    }
    __syncthreads();
    if (tid == 0)
    {
        // seeds = the points of the remainder up to z_max; none above z_max -> no seeds (Q4).  Some remainder point is
        // up to z_max iff the lowest one is; some lies above iff the segment's highest point does (when keys were cut
        // at the floor, the highest point is above it and belongs to the remainder).
        const float z_max = s_zmax;
        const bool some_le = !(s_zmin > z_max), some_gt = lpx_key_float(w->max_key) > z_max;
        SegState o;
        o.lo_excl = (pk.cut > 0) ? prm.z_floor : -INFINITY;
        o.hi_incl = z_max;
        o.has_seeds = some_le && some_gt;
        o.failed = (ns < 3) ? 2u : 0u;  // :224-229 nothing is labelled
        o.plane[0] = o.plane[1] = o.plane[2] = o.plane[3] = 0.0f;
        o.fitted = 0;
        o.thr = 0.0f;
        o.pad[0] = o.pad[1] = 0;
        st[s] = o;
        st[LPX_MAX_PARTITIONS + s].pad[1] = SEG_STATE_STALE;  // (set 1 still holds the last call's record)
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 Jacobi SVD (Eigen 3.4 JacobiSVD<Matrix3f> algorithm as used at src/segmentation.cpp:87-94)
// ------------------------------------------------------------------------------------------------
struct JRot
{
    float c, s;
};

__device__ JRot make_jacobi(float x, float y, float z)
{
    JRot j;
    const float deno = 2.0f * fabsf(y);
    if (deno < FLT_MIN)
    {
        j.c = 1.0f;
        j.s = 0.0f;
        return j;
    }
    const float tau = (x - z) / deno;
    const float w = sqrtf(tau * tau + 1.0f);
    float t;
    if (tau > 0.0f)
        t = 1.0f / (tau + w);
    else
        t = 1.0f / (tau - w);
    const float sign_t = t > 0.0f ? 1.0f : -1.0f;
    const float n = 1.0f / sqrtf(t * t + 1.0f);
    j.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    j.c = n;
    return j;
}

__device__ void rot_left(float *w, int p, int q, JRot j)
{
    if (j.c == 1.0f && j.s == 0.0f)
        return;
    for (int i = 0; i < 3; ++i)
    {
        const float xi = w[p * 3 + i], yi = w[q * 3 + i];
        w[p * 3 + i] = j.c * xi + j.s * yi;
        w[q * 3 + i] = -j.s * xi + j.c * yi;
    }
}

__device__ void rot_right(float *w, int p, int q, JRot j)
{
    const float c = j.c, s = -j.s;
    if (c == 1.0f && s == 0.0f)
        return;
    for (int i = 0; i < 3; ++i)
    {
        const float xi = w[i * 3 + p], yi = w[i * 3 + q];
        w[i * 3 + p] = c * xi + s * yi;
        w[i * 3 + q] = -s * xi + c * yi;
    }
}

// returns false if the matrix is not finite
__device__ bool jacobi_svd3(const float *a, float *v)
{
    float w[9];
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i)
    {
        const float m = fabsf(a[i]);
        if (!(m <= scale))
            scale = m;
    }
    if (!isfinite(scale))
        return false;
    if (scale == 0.0f)
        scale = 1.0f;
    for (int i = 0; i < 9; ++i)
        w[i] = a[i] / scale;
    for (int i = 0; i < 9; ++i)
        v[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    const float precision = 2.0f * FLT_EPSILON;
    const float consider_as_zero = FLT_MIN;
    float max_diag = fmaxf(fabsf(w[0]), fmaxf(fabsf(w[4]), fabsf(w[8])));
    bool finished = false;
    while (!finished)
    {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q)
            {
                const float threshold = fmaxf(consider_as_zero, precision * max_diag);
                if (fabsf(w[p * 3 + q]) > threshold || fabsf(w[q * 3 + p]) > threshold)
                {
                    finished = false;
                    float m00 = w[p * 3 + p], m01 = w[p * 3 + q], m10 = w[q * 3 + p], m11 = w[q * 3 + q];
                    JRot rot1;
                    const float t = m00 + m11;
                    const float d = m10 - m01;
                    if (fabsf(d) < FLT_MIN)
                    {
                        rot1.s = 0.0f;
                        rot1.c = 1.0f;
                    }
                    else
                    {
                        const float u = t / d;
                        const float tmp = sqrtf(1.0f + u * u);
                        rot1.s = 1.0f / tmp;
                        rot1.c = u / tmp;
                    }
                    if (!(rot1.c == 1.0f && rot1.s == 0.0f))
                    {
                        const float a0 = m00, a1 = m01, b0 = m10, b1 = m11;
                        m00 = rot1.c * a0 + rot1.s * b0;
                        m01 = rot1.c * a1 + rot1.s * b1;
                        m10 = -rot1.s * a0 + rot1.c * b0;
                        m11 = -rot1.s * a1 + rot1.c * b1;
                    }
                    const JRot j_right = make_jacobi(m00, m01, m11);
                    JRot j_left;
                    {
                        const float c2 = j_right.c, s2 = -j_right.s;
                        j_left.c = rot1.c * c2 - rot1.s * s2;
                        j_left.s = rot1.c * s2 + rot1.s * c2;
                    }
                    rot_left(w, p, q, j_left);
                    rot_right(w, p, q, j_right);
                    rot_right(v, p, q, j_right);
                    max_diag = fmaxf(max_diag, fmaxf(fabsf(w[p * 3 + p]), fabsf(w[q * 3 + q])));
                }
            }
    }
    float sv[3];
    for (int i = 0; i < 3; ++i)
        sv[i] = fabsf(w[i * 3 + i]) * scale;
    for (int i = 0; i < 3; ++i)
    {
        int pos = i;
        float best = sv[i];
        for (int k = i + 1; k < 3; ++k)
            if (sv[k] > best)
            {
                best = sv[k];
                pos = k;
            }
        if (best == 0.0f)
            break;
        if (pos != i)
        {
            const float ts = sv[i];
            sv[i] = sv[pos];
            sv[pos] = ts;
            for (int r = 0; r < 3; ++r)
            {
                const float tv = v[r * 3 + i];
                v[r * 3 + i] = v[r * 3 + pos];
                v[r * 3 + pos] = tv;
            }
        }
    }
    return true;
}

typedef __int128 i128;

__device__ double i128_to_double(i128 v)
{
    const bool neg = v < 0;
    const unsigned __int128 mag = neg ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
    const uint64_t hi = (uint64_t)(mag >> 64), lo = (uint64_t)mag;
    const double d = (double)hi * 18446744073709551616.0 + (double)lo;
    return neg ? -d : d;
}

// A member point with a coordinate beyond +-2048 m (rare: a spurious far return, or a cloud in a map / UTM
// frame) does not fit the int32 lanes.  Its exact moments go straight to a per-segment global accumulator:
// q = round(clamp(v, +-2^24 m) * 2^16) = h * 2^20 + l with 0 <= l < 2^20, and every product as the three
// partial sums hh, hl + lh, ll (each below 2^41, so 64-bit atomics never carry).
__device__ __forceinline__ long long far_fix(float v)
{
    v = fminf(fmaxf(v, -FIX_CLAMP), FIX_CLAMP);
    return __float2ll_rn(v * FIX_SCALE);
}

__device__ __noinline__ void far_accumulate(long long *fa, float x, float y, float z)
{
    const long long q[3] = {far_fix(x), far_fix(y), far_fix(z)};
    long long h[3], l[3];
    for (int i = 0; i < 3; ++i)
    {
        h[i] = q[i] >> 20;
        l[i] = q[i] & 0xfffffLL;
    }
    unsigned long long *f = (unsigned long long *)fa;
    atomicAdd(f + 0, 1ull);
    atomicAdd(f + 1, (unsigned long long)q[0]);
    atomicAdd(f + 2, (unsigned long long)q[1]);
    atomicAdd(f + 3, (unsigned long long)q[2]);
    int w = 4;
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b)
        {
            atomicAdd(f + w + 0, (unsigned long long)(h[a] * h[b]));
            atomicAdd(f + w + 1, (unsigned long long)(h[a] * l[b] + l[a] * h[b]));
            atomicAdd(f + w + 2, (unsigned long long)(l[a] * l[b]));
            w += 3;
        }
}

__device__ __forceinline__ bool is_near(float x, float y, float z)
{
    return fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) < FIX_LIMIT;
}

// moments words: 0 n, 1 sx, 2 sy, 3 sz, 4.. (hi, lo) of xx, xy, xz, yy, yz, zz; far (may be null): the
// LPX_FAR_WORDS of far_accumulate
__device__ bool plane_from_moments(const long long *m, const long long *far, float *plane)
{
    uint64_t cnt = (uint64_t)m[0];
    i128 sx = m[1], sy = m[2], sz = m[3];
    i128 q[6];
    for (int i = 0; i < 6; ++i)
        q[i] = ((i128)m[4 + 2 * i] << 32) + (i128)m[5 + 2 * i];
    if (far)
    {
        cnt += (uint64_t)far[0];
        sx += far[1];
        sy += far[2];
        sz += far[3];
        for (int i = 0; i < 6; ++i)
            q[i] += ((i128)far[4 + 3 * i] << 40) + ((i128)far[5 + 3 * i] << 20) + (i128)far[6 + 3 * i];
    }
    if (cnt < 3)
        return false;
    const double n = (double)cnt;
    const double den = n * (double)(cnt - 1);
    const double inv16 = 1.0 / 65536.0, inv32 = inv16 * inv16;
    const float cx = (float)((i128_to_double(sx) / n) * inv16);
    const float cy = (float)((i128_to_double(sy) / n) * inv16);
    const float cz = (float)((i128_to_double(sz) / n) * inv16);
    const i128 N = (i128)cnt;
    const float cxx = (float)((i128_to_double(N * q[0] - sx * sx) / den) * inv32);
    const float cxy = (float)((i128_to_double(N * q[1] - sx * sy) / den) * inv32);
    const float cxz = (float)((i128_to_double(N * q[2] - sx * sz) / den) * inv32);
    const float cyy = (float)((i128_to_double(N * q[3] - sy * sy) / den) * inv32);
    const float cyz = (float)((i128_to_double(N * q[4] - sy * sz) / den) * inv32);
    const float czz = (float)((i128_to_double(N * q[5] - sz * sz) / den) * inv32);
    const float cov[9] = {cxx, cxy, cxz, cxy, cyy, cyz, cxz, cyz, czz};
    float v[9];
    if (!jacobi_svd3(cov, v))
        return false;
    const float a = v[2], b = v[5], c = v[8];
    plane[0] = a;
    plane[1] = b;
    plane[2] = c;
    plane[3] = (a * cx + b * cy) + c * cz;
    return true;
}

// ------------------------------------------------------------------------------------------------
// K2+K3 fused pass.  Pass t tests every point of the segment against predicate t
//   t == 0 : seed predicate (z window)                         (:243, :199-216)
//   t >= 1 : signed distance to plane t-1 < thr * |normal|     (:287-307, Q1)
// and, unless FINAL, accumulates the moments of the members for plane t (:261-273).
//
// No cross-block traffic inside a launch: every block leaves the exact integer moments of ITS members in a 128-byte
// row of its own (seg_part, set t & 1) and pass t + 1 starts by reducing the rows of its segment and solving the 3x3
// problem -- every block for itself, identically (integer sums are order-independent, the solve is deterministic),
// while the sixteen points per thread it requested first are on their way from HBM.  (Rounds 1-3 had sixteen
// same-address 64-bit atomics per block, a ticket, and the solve on ONE lane at the very end of the kernel, after
// which the next launch could start: the non-final pass took 44 us against 18 us for the final one on the same loads.)
// Block 0 of a segment publishes the state for the host-visible planes (set t & 1 of seg_state; a pass reads set
// (t - 1) & 1, so nobody reads what it writes).  FINAL writes the per-point flag and per-block ground/obstacle counts
// for the compaction.  Points are loaded four at a time (16-byte loads): a block covers chunk consecutive points
// starting at a multiple of four, clipped to its segment (seg_block_range).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void seg_block_range(const SegParams &prm, uint32_t s, uint32_t b, uint32_t &base,
                                                uint32_t &lo, uint32_t &hi)
{
    const uint32_t seg_lo = s * prm.n_per, seg_hi = seg_lo + prm.n_per;
    base = (seg_lo & ~3u) + b * prm.chunk;  // first point of the block's 16-byte-aligned window
    lo = max(base, seg_lo);
    hi = min(base + prm.chunk, seg_hi);
    if (hi < lo)
        hi = lo;
}

// One WAVEFRONT per block.  Every block of a pass pays for the head -- a dependent chain of ~10k cycles on a single
// instruction stream (the 3x3 Jacobi with its correctly rounded divisions and square roots) -- and with 256-thread
// blocks three of four wavefronts sat at a barrier meanwhile, holding registers and wave slots (measured: 47 us per
// pass against 44 us for the atomics-and-ticket form it replaced).  A one-wavefront block has nobody waiting for it:
// the ~1900 blocks of a 64-frame launch are all resident at once (two per SIMD), their heads overlap one another and
// the loads of the other blocks, and the block reduction is a DPP reduction without LDS or barriers.
constexpr int PASS_THREADS = WAVE;
constexpr int PASS_QUADS = 4;                                           // (general form) quads per lane and trip
constexpr uint32_t PASS_TRIP = 4u * PASS_THREADS * PASS_QUADS;          // points of a trip: 1024
// The lean loop keeps a RING of PASS_RING quads (four consecutive points: three 16-byte loads) per lane: quad g of the
// block is lane's points base + 4 (lane + 64 g) ..., a wavefront instruction reads 1 KB; as soon as a quad has been
// processed its registers take the load of the quad PASS_RING further on, so 7/8 of the 96 staging registers are in
// flight at any time (two alternating sets of four quads: half of them) -- 21 KB per wavefront.
#ifndef LPX_PASS_RING
#define LPX_PASS_RING 4
#endif
constexpr int PASS_RING = LPX_PASS_RING;
constexpr uint32_t PASS_QUAD_POINTS = 4u * PASS_THREADS;                // points of one quad row of a wavefront: 256
static_assert(SEG_CHUNK % (2u * PASS_TRIP) == 0 && (2u * PASS_TRIP) % (PASS_RING * PASS_QUAD_POINTS) == 0,
              "a block is a whole number of rings");

typedef float pass_v4f __attribute__((ext_vector_type(4)));

// The plane of pass t - 1 from the 16 moment words (lane l holds word l & 15, every lane the reduced total): the nine
// quantities that need a 128-bit product, a conversion and a double-precision division -- three centroid coordinates,
// six covariances -- are computed by nine lanes AT ONCE (one instruction stream either way: nine times fewer
// instructions than nine evaluations in a row), the same operations in the same order as plane_from_moments, so the
// results are bit-identical to it and to the oracle.  Then the 3x3 solve, identically on every lane.
__device__ __forceinline__ bool plane_from_moment_lanes(long long v, uint32_t lane, float *plane)
{
    const uint64_t cnt = (uint64_t)__shfl(v, 0, WAVE);
    if (cnt < 3)
        return false;
    // lane 0..2: sums sx, sy, sz (word 1 + lane); lane 3 + k: second moment k = xx, xy, xz, yy, yz, zz (words 4 + 2 k, 5 + 2 k)
    const int k = (int)lane - 3;
    const int ka = k < 3 ? 0 : (k < 5 ? 1 : 2);               // first factor of moment k: x, x, x, y, y, z
    const int kb = k < 3 ? k : (k < 5 ? k - 2 : 2);           // second factor:            x, y, z, y, z, z
    const bool first = lane < 3, used = lane < 9;
    const long long lo = __shfl(v, first ? 1 + (int)lane : (used ? 5 + 2 * k : 0), WAVE);
    const long long hi = __shfl(v, (used && !first) ? 4 + 2 * k : 0, WAVE);
    const long long ya = __shfl(v, (used && !first) ? 1 + ka : 0, WAVE);
    const long long zb = __shfl(v, (used && !first) ? 1 + kb : 0, WAVE);
    const i128 X = first ? (i128)lo : (((i128)hi << 32) + (i128)lo);
    const i128 N = (i128)cnt;
    const i128 num = first ? X : (N * X - (i128)ya * (i128)zb);
    const double n = (double)cnt;
    const double den = n * (double)(cnt - 1);
    const double inv16 = 1.0 / 65536.0, inv32 = inv16 * inv16;
    const float r = (float)((i128_to_double(num) / (first ? n : den)) * (first ? inv16 : inv32));
#define LPX_RL(l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), (l)))
    const float cx = LPX_RL(0), cy = LPX_RL(1), cz = LPX_RL(2);
    const float cxx = LPX_RL(3), cxy = LPX_RL(4), cxz = LPX_RL(5), cyy = LPX_RL(6), cyz = LPX_RL(7), czz = LPX_RL(8);
#undef LPX_RL
    const float cov[9] = {cxx, cxy, cxz, cxy, cyy, cyz, cxz, cyz, czz};
    float vv[9];
    if (!jacobi_svd3(cov, vv))
        return false;
    const float a = vv[2], b = vv[5], c = vv[8];
    plane[0] = a;
    plane[1] = b;
    plane[2] = c;
    plane[3] = (a * cx + b * cy) + c * cz;
    return true;
}

// signed distance of a point to a plane, scaled by |normal| (Q1): ONE definition for every kernel that classifies a point
// (plane passes lean and general, labels_direct_kernel) -- no FMA (-ffp-contract=off), this order of operations
__device__ __forceinline__ float plane_dist(float x, float y, float z, float pa, float pb, float pc, float pd)
{
    return ((x * pa + y * pb) + z * pc) - pd;
}

// ---- the general form of the loop, for frames that hold a point beyond +-2048 m (rare: a spurious far return, a cloud in
// a map frame): member points that do not fit the int32 lanes go to the segment's far accumulator one by one.  Kept out
// of line, loading for itself, so that its registers and its call do not shape the allocation of the lean loop.
struct PassGeneric
{
    long long m[10];        // n, sx, sy, sz, xx, xy, xz, yy, yz, zz
    uint32_t cnt_g, cnt_o;
};
template <bool FINAL>
__device__ __noinline__ void pass_block_generic(const float *XS, const float *YS, const float *ZS, uint32_t base,
                                                uint32_t lo, uint32_t hi, uint32_t trips, uint32_t lane,
                                                const SegState *sstp, uint32_t t, uint32_t I, long long *fa,
                                                uint8_t *flags, PassGeneric *out)
{
    const SegState sst = *sstp;
    const bool any_far = true;
    const bool skip = sst.failed == 2;
    const bool dead = sst.failed != 0;
    const float pa = sst.plane[0], pb = sst.plane[1], pc = sst.plane[2], pd = sst.plane[3];
    const float thr = sst.thr;
    const bool use_seed = (t == 0);
    const bool seeds_ok = sst.has_seeds != 0;
    struct
    {
        uint32_t I;
    } prm = {I};
    long long a_n = 0, a_x = 0, a_y = 0, a_z = 0, a_xx = 0, a_xy = 0, a_xz = 0, a_yy = 0, a_yz = 0, a_zz = 0;
    uint32_t cnt_g = 0, cnt_o = 0;
    auto process = [&](const pass_v4f *x4, const pass_v4f *y4, const pass_v4f *z4, uint32_t trip) {
#pragma unroll
        for (int u = 0; u < PASS_QUADS; ++u)
        {
            const uint32_t p0 = base + 4u * (lane + PASS_THREADS * (u + PASS_QUADS * trip));
            if (p0 >= hi)
                continue;
            uint32_t fword = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                const uint32_t p = p0 + e;
                const bool in = p >= lo && p < hi;
                const float x = x4[u][e], y = y4[u][e], z = z4[u][e];
                bool member;
                if (use_seed)
                    member = seeds_ok && (z > sst.lo_excl) && (z <= sst.hi_incl);
                else
                {
                    const float dist = plane_dist(x, y, z, pa, pb, pc, pd);
                    member = dist < thr;
                }
                member = member && !dead && in;
                if (FINAL)
                {
                    // number_of_iterations == 0: seeds are ground, the rest stays UNKNOWN (:243-247)
                    const uint32_t f = (!in || skip) ? 0u : (member ? 1u : ((prm.I == 0 && !dead) ? 0u : 2u));
                    fword |= f << (8 * e);
                    cnt_g += (f == 1u);
                    cnt_o += (f == 2u);
                }
                else if (member && any_far && !is_near(x, y, z))
                    far_accumulate(fa, x, y, z);
                else if (member)
                {
                    const int qx = __float2int_rn(x * FIX_SCALE);
                    const int qy = __float2int_rn(y * FIX_SCALE);
                    const int qz = __float2int_rn(z * FIX_SCALE);
                    a_n += 1;
                    a_x += qx;
                    a_y += qy;
                    a_z += qz;
                    a_xx += (long long)qx * qx;
                    a_xy += (long long)qx * qy;
                    a_xz += (long long)qx * qz;
                    a_yy += (long long)qy * qy;
                    a_yz += (long long)qy * qz;
                    a_zz += (long long)qz * qz;
                }
            }
            if (FINAL)
            {
                if (p0 >= lo && p0 + 4u <= hi)
                    *(uint32_t *)(flags + p0) = fword;  // p0 is a multiple of four
                else
                    for (int e = 0; e < 4; ++e)
                        if (p0 + e >= lo && p0 + e < hi)
                            flags[p0 + e] = (uint8_t)(fword >> (8 * e));
            }
        }
    };
    for (uint32_t trip = 0; trip < trips; ++trip)
    {
        pass_v4f xa[PASS_QUADS], ya[PASS_QUADS], za[PASS_QUADS];
#pragma unroll
        for (int u = 0; u < PASS_QUADS; ++u)
        {
            const uint32_t p_ = base + 4u * (lane + PASS_THREADS * (u + PASS_QUADS * trip));
            if (p_ < hi)
            {
                xa[u] = *(const pass_v4f *)(XS + p_);
                ya[u] = *(const pass_v4f *)(YS + p_);
                za[u] = *(const pass_v4f *)(ZS + p_);
            }
        }
        process(xa, ya, za, trip);
    }
    const long long r[10] = {a_n, a_x, a_y, a_z, a_xx, a_xy, a_xz, a_yy, a_yz, a_zz};
    for (int i = 0; i < 10; ++i)
        out->m[i] = r[i];
    out->cnt_g = cnt_g;
    out->cnt_o = cnt_o;
}

// ---- the streaming loop of a pass block, lean form ----
// A pass is issue-bound as much as memory-bound: the first form of this loop spent 63 vector and 20 scalar instructions
// per point (per-point scalar branches on conditions that are uniform for the whole block -- the kind of pass, "the
// frame has a far point", "the segment is dead" --, 64-bit adds for sums that fit 32 bits for eight points, bounds
// tests on every point of blocks that lie wholly inside their segment), the vector ALUs of a compute unit were 86 %
// busy and pass 0 (two compares per point), the plane passes (six multiply-adds) and the final pass (no moments at
// all) took the same 29-32 us per 96 MB.  Here everything uniform is decided once per block (template parameters
// behind scalar branches around the whole loop), members are selected instead of branched on (a non-member adds
// zeros), the three coordinate sums and the count run in 32-bit lanes flushed every eight points (|q| < 2^27), and only
// the first and the last block of a segment test positions against the segment's bounds: 29 vector instructions per
// point in a plane pass, 10 in the final one.  Integer sums: the result does not depend on any of this.
struct PassUniform
{
    float pa, pb, pc, pd, thr;     // plane t - 1 and its threshold
    float lo_excl, hi_incl;        // pass 0: the seed window
    uint32_t base, lo, span;       // first position of the block's window; the segment's points are lo <= p < lo + span
    uint32_t fm, fn;               // FINAL: flag of a member / of a non-member (0 when the segment is not labelled)
};
struct PassAcc
{
    long long x = 0, y = 0, z = 0, xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0;
    uint32_t n = 0, in = 0;        // members; FINAL with EDGE: points of the segment seen
    int sx = 0, sy = 0, sz = 0;    // 32-bit partial sums of at most eight points
    __device__ __forceinline__ void flush()
    {
        x += sx;
        y += sy;
        z += sz;
        sx = sy = sz = 0;
    }
};

// (Packed float32 arithmetic -- v_pk_mul_f32 / v_pk_add_f32 on pairs of points, bit-identical -- was built and measured:
// 0.338 -> 0.376 ms per 32-frame 1M-point chain and -1.3 % under load; the packed forms are no faster per point here and
// cost the moves that pair the operands.  Scalar it stays.)
template <bool FINAL, bool SEED, bool EDGE>
__device__ __forceinline__ void pass_quad_lean(const pass_v4f x4, const pass_v4f y4, const pass_v4f z4, uint32_t p0,
                                               const PassUniform &U, PassAcc &A, uint8_t *flags)
{
    uint32_t fword = 0, inq = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
    {
        const float x = x4[e], y = y4[e], z = z4[e];
        bool member;
        if (SEED)
            member = (z > U.lo_excl) && (z <= U.hi_incl);
        else
        {
            const float dist = plane_dist(x, y, z, U.pa, U.pb, U.pc, U.pd);
            member = dist < U.thr;
        }
        bool in = true;
        if (EDGE)
        {
            in = (p0 + (uint32_t)e - U.lo) < U.span;  // (unsigned: positions before the segment wrap around)
            member = member && in;
        }
        A.n += member ? 1u : 0u;
        if (FINAL)
        {
            const uint32_t f = member ? U.fm : U.fn;
            fword |= (EDGE && !in ? 0u : f) << (8 * e);
            if (EDGE)
                inq |= (in ? 1u : 0u) << e;
        }
        else
        {
            const int qx = member ? __float2int_rn(x * FIX_SCALE) : 0;
            const int qy = member ? __float2int_rn(y * FIX_SCALE) : 0;
            const int qz = member ? __float2int_rn(z * FIX_SCALE) : 0;
            A.sx += qx;
            A.sy += qy;
            A.sz += qz;
            A.xx += (long long)qx * qx;
            A.xy += (long long)qx * qy;
            A.xz += (long long)qx * qz;
            A.yy += (long long)qy * qy;
            A.yz += (long long)qy * qz;
            A.zz += (long long)qz * qz;
        }
    }
    if (FINAL)
    {
        if (!EDGE || inq == 0xfu)
            *(uint32_t *)(flags + p0) = fword;  // p0 is a multiple of four
        else
            for (int e = 0; e < 4; ++e)
                if ((inq >> e) & 1u)
                    flags[p0 + e] = (uint8_t)(fword >> (8 * e));
        if (EDGE)
            A.in += __popc(inq);
    }
}

// S_{t+1} from S_t and the rows of pass t (one wavefront, every lane the same result): the rows of the segment summed
// (integer sums: order-independent), the 3x3 problem solved -- nine lanes for the nine double-precision quotients, the
// bit-exact Jacobi once.  Out of line: called at the tail of a pass by the block that arrives last, or at the head of
// the next pass by every block (PassParams::head_solve); its ~100 registers of 128-bit arithmetic stay out of the
// allocation of the streaming loop.  Rows are read past the caches (agent scope): in the tail form other blocks of the
// same launch have just written them.
__device__ __noinline__ void pass_solve(const long long *rows, uint32_t bps, const SegState *prev, const long long *fa,
                                        uint32_t any_far, float odt, uint32_t lane, SegState *out)
{
    const uint32_t row = lane >> 4, word = lane & 15u;
    long long tot = 0;
    for (uint32_t r0 = 0; r0 < bps; r0 += 32)
    {
        long long q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            const uint32_t r = r0 + row + 4u * k;
            q[k] = r < bps ? __hip_atomic_load(rows + (size_t)r * LPX_ACC_WORDS + word, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)
                           : 0ll;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            tot += q[k];
    }
    tot += __shfl_xor(tot, 16, WAVE);
    tot += __shfl_xor(tot, 32, WAVE);  // every lane: the segment's total of word lane & 15
    SegState o = *prev;  // sticky flags carry over
    o.pad[0] = 0;        // a fresh ticket counter
    if (o.failed == 0)
    {
        float plane[4];
        bool ok;
        if (any_far)
        {
            // rare (a coordinate beyond +-2048 m): the far-point limbs join the sums; the plain evaluation
            long long m[LPX_ACC_WORDS], fm[LPX_FAR_WORDS];
#pragma unroll
            for (int i = 0; i < LPX_ACC_WORDS; ++i)
                m[i] = __shfl(tot, i, WAVE);
            for (int i = 0; i < LPX_FAR_WORDS; ++i)  // (accumulated by memory-side atomics of every block)
                fm[i] = __hip_atomic_load(fa + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = plane_from_moments(m, fm, plane);
        }
        else
            ok = plane_from_moment_lanes(tot, lane, plane);
        // fewer than 3 ground points or a failed solve: everything is an obstacle (:251-259, :275-283)
        if (!ok)
            o.failed = 1;
        else
        {
            o.plane[0] = plane[0];
            o.plane[1] = plane[1];
            o.plane[2] = plane[2];
            o.plane[3] = plane[3];
            o.thr = odt * sqrtf((plane[0] * plane[0] + plane[1] * plane[1]) + plane[2] * plane[2]);
            o.fitted = 1;
        }
    }
    *out = o;
}

// S_t of a segment as another block of the SAME launch published it (chained passes): twelve words read past the caches
// by twelve lanes, handed round by readlane
__device__ __forceinline__ SegState seg_state_load_agent(const SegState *p, uint32_t lane)
{
    constexpr int NW = (int)(sizeof(SegState) / sizeof(uint32_t));
    static_assert(NW == 12, "SegState is twelve words");
    const uint32_t w = __hip_atomic_load((const uint32_t *)p + (lane < (uint32_t)NW ? lane : 0u), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    uint32_t v[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i)
        v[i] = (uint32_t)__builtin_amdgcn_readlane((int)w, i);
    SegState o;
    __builtin_memcpy(&o, v, sizeof o);
    return o;
}

// CHAINED: the block belongs to a launch that holds ALL passes of the frames of a call (plane_chain_kernel): it waits
// for the state it tests against -- published by a block of the same launch -- instead of being started after it.
template <bool FINAL, bool CHAINED>
__device__ __forceinline__ void plane_pass_block(const float *__restrict__ XS, const float *__restrict__ YS,
                                                 const float *__restrict__ ZS, SegParams prm, uint32_t t, SegState *st,
                                                 long long *part, long long *facc, uint8_t *__restrict__ flags,
                                                 uint32_t *__restrict__ blk_counts, const FrameState *__restrict__ frame,
                                                 uint32_t s, uint32_t b)
{
    seg_bind(prm, frame);
    const bool any_far = frame->has_far != 0;
    const uint32_t lane = threadIdx.x;
    const uint32_t nb = prm.P * prm.bps;
    uint32_t base, lo, hi;
    seg_block_range(prm, s, b, base, lo, hi);
    const uint32_t trips = prm.chunk / PASS_TRIP;  // even (seg_geometry)
    const uint32_t G = prm.chunk / PASS_QUAD_POINTS;  // quads per lane: a multiple of PASS_RING (seg_geometry)

    // The ring of the lean loop.  Buffer loads: the three descriptors live in scalar registers, a load's address is the
    // lane's constant byte offset plus a scalar offset -- no vector address arithmetic at all -- and reads beyond the
    // frame's points return zeros (the last block of the last segment is clipped by the descriptor, not by a branch).
    pass_v4f X[PASS_RING], Y[PASS_RING], Z[PASS_RING];
    const uint32_t bytes = 4u * ((prm.n + 3u) & ~3u);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)XS, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)YS, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void *)ZS, 0, bytes, 0x00020000);
    const uint32_t voff = lane * 16u;
    typedef unsigned int pass_v4u __attribute__((ext_vector_type(4)));
#define LPX_RING_LOAD(i, g)                                                                                            \
    {                                                                                                                  \
        /* quad g of the block; past the block's last quad: an offset beyond the descriptor's range (zeros, no access) */ \
        const uint32_t so_ = (uint32_t)(g) < G ? 4u * (base + PASS_QUAD_POINTS * (uint32_t)(g)) : bytes;               \
        X[i] = __builtin_bit_cast(pass_v4f, (pass_v4u)__builtin_amdgcn_raw_buffer_load_b128(rx, voff, so_, 0));        \
        Y[i] = __builtin_bit_cast(pass_v4f, (pass_v4u)__builtin_amdgcn_raw_buffer_load_b128(ry, voff, so_, 0));        \
        Z[i] = __builtin_bit_cast(pass_v4f, (pass_v4u)__builtin_amdgcn_raw_buffer_load_b128(rz, voff, so_, 0));        \
    }

    // ---- the state this pass tests against: S_t, set t & 1 of seg_state -- the seed window (t == 0, from the seed
    // selection) or plane t - 1, which the LAST block of pass t - 1 to finish its segment solved and published (tail below)
    // Tail form (launches of more blocks than the device holds at once: early segments solve while later ones stream):
    // read it.  Head form (launches that are resident all at once -- a frame alone, a chain of 120k-point frames: there
    // every segment finishes together and a solve at the tail is ~10 us that nothing overlaps): every block computes S_t
    // itself from S_{t-1} and the rows of pass t - 1, identically, and block 0 publishes it for the compaction's planes.
    SegState sst;
    if (prm.head_solve && t > 0)
    {
        const SegState prev = st[(size_t)((t + 1u) & 1u) * LPX_MAX_PARTITIONS + s];
        pass_solve(part + (size_t)((t + 1u) & 1u) * prm.part_stride + (size_t)s * prm.bps * LPX_ACC_WORDS, prm.bps, &prev,
                   facc + ((size_t)((t - 1u) % 3u) * LPX_MAX_PARTITIONS + s) * LPX_FAR_WORDS, any_far ? 1u : 0u, prm.odt,
                   lane, &sst);
        if (b == 0 && lane == 0)
            st[(size_t)(t & 1u) * LPX_MAX_PARTITIONS + s] = sst;
        // the far set pass t + 1 accumulates into was read by pass t - 1 and is free
        if (any_far && b == 0 && lane < LPX_FAR_WORDS)
            facc[((size_t)((t + 1u) % 3u) * LPX_MAX_PARTITIONS + s) * LPX_FAR_WORDS + lane] = 0;
    }
    else if (CHAINED && t > 0)
    {
        // S_t comes from the block of pass t - 1 that finished this segment last -- a block of THIS launch with a lower
        // workgroup number (pass-major grid), i.e. one that was started before this one: the wait cannot deadlock however
        // few blocks are resident (see plane_chain_kernel).  pad[1] is the pass the record belongs to (release / acquire).
        SegState *const cur = st + (size_t)(t & 1u) * LPX_MAX_PARTITIONS + s;
        // (relaxed agent-scope loads: past the caches; the record itself is read the same way right after)
        if (__hip_atomic_load(&cur->pad[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t)
        {
            const uint64_t t0 = wall_clock64();  // 100 MHz
            while (__hip_atomic_load(&cur->pad[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != t)
            {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > 400000000ull)
                {
                    // four seconds: the dispatch-order premise does not hold on this machine.  Report instead of hanging
                    // (the frame fails with LPX_ERR_INTERNAL; LPX_PASS_CHAIN=0 of the development build is the way out)
                    if (lane == 0)
                        atomicCAS((uint32_t *)&frame->status, 0u, (uint32_t)(-LPX_ERR_INTERNAL));
                    break;
                }
            }
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        sst = seg_state_load_agent(cur, lane);
    }
    else
        sst = st[(size_t)(t & 1u) * LPX_MAX_PARTITIONS + s];
    const bool skip = sst.failed == 2;       // < 3 points: nothing labelled
    const bool dead = sst.failed != 0;       // all obstacle
    const float pa = sst.plane[0], pb = sst.plane[1], pc = sst.plane[2], pd = sst.plane[3];
    const float thr = sst.thr;
    const bool use_seed = (t == 0);
    const bool seeds_ok = sst.has_seeds != 0;
    long long *fa = facc + ((size_t)(t % 3u) * LPX_MAX_PARTITIONS + s) * LPX_FAR_WORDS;

    long long a_n = 0, a_x = 0, a_y = 0, a_z = 0, a_xx = 0, a_xy = 0, a_xz = 0, a_yy = 0, a_yz = 0, a_zz = 0;
    uint32_t cnt_g = 0, cnt_o = 0;

    if (any_far)
    {
        // rare (a coordinate beyond +-2048 m somewhere in the frame): the general form, out of line
        PassGeneric g;
        pass_block_generic<FINAL>(XS, YS, ZS, base, lo, hi, trips, lane, &sst, t, prm.I, fa, flags, &g);
        a_n = g.m[0], a_x = g.m[1], a_y = g.m[2], a_z = g.m[3];
        a_xx = g.m[4], a_xy = g.m[5], a_xz = g.m[6], a_yy = g.m[7], a_yz = g.m[8], a_zz = g.m[9];
        cnt_g = g.cnt_g, cnt_o = g.cnt_o;
    }
    else
    {
        // the lean form (pass_trip_lean): what is uniform for the block is decided here, once
        PassUniform U;
        const bool alive = !dead && (!use_seed || seeds_ok);
        U.pa = pa, U.pb = pb, U.pc = pc, U.pd = pd;
        U.thr = alive ? thr : -INFINITY;            // nobody is a member of a dead segment
        U.lo_excl = alive ? sst.lo_excl : INFINITY;
        U.hi_incl = sst.hi_incl;
        U.base = base, U.lo = lo, U.span = hi - lo;
        U.fm = skip ? 0u : 1u;
        U.fn = skip ? 0u : ((prm.I == 0 && !dead) ? 0u : 2u);  // number_of_iterations == 0: the rest stays UNKNOWN (:243-247)
        const bool edge = lo != base || hi != base + prm.chunk;  // the block is clipped by its segment's bounds
        PassAcc A;
// One instance of the loop: the ring is filled, then every quad is processed and its registers take the load of the
// quad PASS_RING further on (sched_barrier: in that order -- left to itself the compiler copies the ring aside, requests
// all eight reloads and processes the copies: 256 registers).
#define LPX_PASS_RUN(SEEDV, EDGEV)                                                                                    \
    {                                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < PASS_RING; ++i) LPX_RING_LOAD(i, i)                                     \
        for (uint32_t g0 = 0; g0 < G; g0 += PASS_RING)                                                                \
        {                                                                                                             \
            _Pragma("unroll") for (int i = 0; i < PASS_RING; ++i)                                                     \
            {                                                                                                         \
                pass_quad_lean<FINAL, SEEDV, EDGEV>(X[i], Y[i], Z[i], base + 4u * lane + PASS_QUAD_POINTS * (g0 + i), \
                                                    U, A, flags);                                                     \
                LPX_RING_LOAD(i, g0 + i + PASS_RING)                                                                  \
                if (!FINAL && (i & 1))                                                                                \
                    A.flush();                                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                    \
            }                                                                                                         \
        }                                                                                                             \
    }
        if (use_seed)
        {
            if (edge)
            {
                LPX_PASS_RUN(true, true)
            }
            else
            {
                LPX_PASS_RUN(true, false)
            }
        }
        else if (edge)
        {
            LPX_PASS_RUN(false, true)
        }
        else
        {
            LPX_PASS_RUN(false, false)
        }
#undef LPX_PASS_RUN
        if (FINAL)
        {
            const uint32_t seen = edge ? A.in : G * 4u;  // points of the segment this lane flagged
            cnt_g = U.fm == 1u ? A.n : 0u;
            cnt_o = U.fn == 2u ? seen - A.n : 0u;
        }
        else
        {
            A.flush();
            a_n = (long long)A.n;
            a_x = A.x, a_y = A.y, a_z = A.z;
            a_xx = A.xx, a_xy = A.xy, a_xz = A.xz, a_yy = A.yy, a_yz = A.yz, a_zz = A.zz;
        }
    }
#undef LPX_RING_LOAD

    if (FINAL)
    {
        cnt_g = lpx_wave_sum_u32(cnt_g);
        cnt_o = lpx_wave_sum_u32(cnt_o);
        if (lane == 0)
        {
            blk_counts[s * prm.bps + b] = cnt_g;
            blk_counts[nb + s * prm.bps + b] = cnt_o;
            if (s == 0 && b == 0)
                blk_counts[2 * nb] = 0;  // sentinel: the exclusive scan leaves the grand total here
        }
        return;
    }

    // ---- tail: this block's moments, and -- for the block that arrives last -- the segment's next plane ----
    // 16 words: n, sx, sy, sz, then (hi, lo) limbs of the six second moments, reduced over the wavefront and left in this
    // block's 128-byte row of set t & 1 by sixteen lanes with ONE write-through store instruction.  Then the block takes
    // a ticket of its segment (an agent-scope atomic behind the drained stores); the block that draws the last one reads
    // all rows of the segment past every cache, sums them (integer sums: order-independent), solves the 3x3 problem --
    // nine lanes for the nine double-precision quotients, the bit-exact Jacobi once -- and publishes S_{t+1} (set
    // (t + 1) & 1) for the next launch.  So the solve runs ONCE per segment and pass, while the other segments still
    // stream, instead of at the head of every block of the next pass (rounds 3-4: ~10 us of dependent instructions that
    // each of a launch's ~1900 blocks had to wait for before it could test its first point -- and ~100 registers of
    // 128-bit arithmetic live beside the staging registers of the loop).  Publication follows the write-through form of
    // the hand-off: every store of the payload sc1 and drained before the ticket, every load of it sc1.
    long long v[LPX_ACC_WORDS];
    v[0] = a_n;
    v[1] = a_x;
    v[2] = a_y;
    v[3] = a_z;
    const long long sm[6] = {a_xx, a_xy, a_xz, a_yy, a_yz, a_zz};
#pragma unroll
    for (int i = 0; i < 6; ++i)
    {
        v[4 + 2 * i] = sm[i] >> 32;
        v[5 + 2 * i] = sm[i] & 0xffffffffLL;
    }
    long long mine = 0;
#pragma unroll
    for (int i = 0; i < LPX_ACC_WORDS; ++i)
    {
        const long long tot = lpx_wave_sum_i64(v[i]);  // valid in lane 0
        const long long bc = __shfl(tot, 0, WAVE);
        mine = lane == (uint32_t)i ? bc : mine;
    }
    long long *rows = part + (size_t)(t & 1u) * prm.part_stride + (size_t)s * prm.bps * LPX_ACC_WORDS;
    if (lane < LPX_ACC_WORDS)
        __hip_atomic_store(rows + (size_t)b * LPX_ACC_WORDS + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prm.head_solve)
        return;  // (the next launch reads the rows: every block of it solves at its head)
    // The hand-off.  The payload (this block's row, its far-point atomics) goes out as agent-scope atomic stores --
    // write-through to the device's coherence point -- and is DRAINED (s_waitcnt) before the ticket is taken; the block
    // that draws the last ticket reads every row with agent-scope atomic loads, past the caches.  That is the hand-off
    // form of the CDNA guide.  Compiler ordering is pinned by the two signal fences (formal compiler barriers: ADVICE
    // round 5 -- the waitcnt builtin alone is not one on paper).  The memory-model form -- a release fence and an
    // ACQ_REL ticket at agent scope -- was built and measured in round 6: the legalizer turns the release into a
    // write-back of the XCD's whole L2 per block, 92 -> 403 us per pass of a 32-frame 1M-point chain.  Not used.
    SegState *const cur = st + (size_t)(t & 1u) * LPX_MAX_PARTITIONS + s;
    uint32_t ticket = 0;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the row has left for memory
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (lane == 0)
        ticket = __hip_atomic_fetch_add(&cur->pad[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0, WAVE);
    if (ticket != prm.bps - 1u)
        return;
    // the last block of the segment: every row is in memory
    SegState o;
    pass_solve(rows, prm.bps, &sst, fa, any_far ? 1u : 0u, prm.odt, lane, &o);
    o.pad[1] = t + 1u;  // the pass this record is the state of
    // the far set pass t + 1 accumulates into was read at the end of pass t - 2 and is free
    if (any_far && lane < LPX_FAR_WORDS)
        __hip_atomic_store(facc + ((size_t)((t + 1u) % 3u) * LPX_MAX_PARTITIONS + s) * LPX_FAR_WORDS + lane, 0ll,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // what pass t + 1 tests against and the compaction's planes: the record (write-through words), then -- behind a release
    // fence -- the pass number a chained block of pass t + 1 waits for
    SegState *const nxt = st + (size_t)((t + 1u) & 1u) * LPX_MAX_PARTITIONS + s;
    {
        constexpr int NW = (int)(sizeof(SegState) / sizeof(uint32_t));
        uint32_t v[NW];
        __builtin_memcpy(v, &o, sizeof o);
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            mine = lane == (uint32_t)i ? v[i] : mine;
        if (lane < (uint32_t)NW - 1u)  // every word but the last, pad[1]
            __hip_atomic_store((uint32_t *)nxt + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the same hand-off form as the ticket above: write-through payload, drained, then the word that is waited for)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (lane == 0)
            __hip_atomic_store(&nxt->pad[1], t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool FINAL>
__global__ __launch_bounds__(PASS_THREADS, LPX_PASS_MINWAVES) void plane_pass_kernel(const float *__restrict__ XS,
                                                                   const float *__restrict__ YS,
                                                                   const float *__restrict__ ZS, SegParams prm,
                                                                   uint32_t t, SegState *st, long long *part,
                                                                   long long *facc, uint8_t *__restrict__ flags,
                                                                   uint32_t *__restrict__ blk_counts,
                                                                   const FrameState *__restrict__ frame, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<2>(fs);
    XS = lpx_slot(XS, fs);
    YS = lpx_slot(YS, fs);
    ZS = lpx_slot(ZS, fs);
    st = lpx_slot(st, fs);
    part = lpx_slot(part, fs);
    facc = lpx_slot(facc, fs);
    flags = lpx_slot(flags, fs);
    blk_counts = lpx_slot(blk_counts, fs);
    frame = lpx_slot(frame, fs);
    plane_pass_block<FINAL, false>(XS, YS, ZS, prm, t, st, part, facc, flags, blk_counts, frame, lpx_blk.y, lpx_blk.x);
}

// ALL passes of the frames of a call in ONE launch.  Grid (blocks of a segment, segments, (I + 1) x frames): workgroup
// numbers run pass-major -- every block of pass t of every frame before any block of pass t + 1 -- and a block of pass
// t + 1 waits for its segment's S_{t+1}, which the block of pass t that finishes the segment last solves and publishes
// (tail form).  Why that cannot deadlock without a cooperative launch: the hardware starts the workgroups of a launch in
// the order of their numbers (round-robin over the XCDs, in order on each), and a block only ever waits for blocks with
// LOWER numbers; the lowest-numbered unfinished block of the launch therefore waits for nobody, has been started (its
// XCD started it before anything above it) and finishes -- by induction everything does, whatever else occupies the
// device.  (A wait that lasts four seconds reports LPX_ERR_INTERNAL instead of hanging.)  What it buys: between the
// launches of a chain the device drained and refilled -- the last segments' solves (~10 us each of dependent double
// precision) and the ramp of ~3000 one-wavefront blocks, four times per 1M-point chain; here the next pass's blocks
// stream while the stragglers of the last one solve, and the x-sorted SoA of a frame is re-read while the Infinity
// Cache still holds it.  The XCD-affine renumbering permutes workgroups inside groups of eight z values: frames of ONE
// pass when the frame count is a multiple of eight (the host leaves it on only then).
__global__ __launch_bounds__(PASS_THREADS, LPX_PASS_MINWAVES) void plane_chain_kernel(const float *__restrict__ XS,
                                                                    const float *__restrict__ YS,
                                                                    const float *__restrict__ ZS, SegParams prm,
                                                                    uint32_t frames, SegState *st, long long *part,
                                                                    long long *facc, uint8_t *__restrict__ flags,
                                                                    uint32_t *__restrict__ blk_counts,
                                                                    const FrameState *__restrict__ frame, size_t fs)
{
    LpxBlock lpx_blk = lpx_block<2>(fs);
    const uint32_t t = lpx_blk.z / frames;
    lpx_blk.z -= t * frames;
    XS = lpx_slot(XS, fs);
    YS = lpx_slot(YS, fs);
    ZS = lpx_slot(ZS, fs);
    st = lpx_slot(st, fs);
    part = lpx_slot(part, fs);
    facc = lpx_slot(facc, fs);
    flags = lpx_slot(flags, fs);
    blk_counts = lpx_slot(blk_counts, fs);
    frame = lpx_slot(frame, fs);
    if (t == prm.I)
        plane_pass_block<true, true>(XS, YS, ZS, prm, t, st, part, facc, flags, blk_counts, frame, lpx_blk.y, lpx_blk.x);
    else
        plane_pass_block<false, true>(XS, YS, ZS, prm, t, st, part, facc, flags, blk_counts, frame, lpx_blk.y, lpx_blk.x);
}

// ------------------------------------------------------------------------------------------------
// compaction: flags -> labels (original order), ground / obstacle index lists in output-cloud
// order (:331-343, Q7) and the obstacle SoA handed to clustering.
// ------------------------------------------------------------------------------------------------
template <bool SCATTER_LABELS>  // small frames: the labels by original index from here (they merge in the frame's L2)
__global__ __launch_bounds__(SEG_THREADS) void compact_kernel(const uint8_t *__restrict__ flags,
                                                               const uint32_t *__restrict__ sidx,
                                                               const float *__restrict__ XS,
                                                               const float *__restrict__ YS,
                                                               const float *__restrict__ ZS, SegParams prm,
                                                               const uint32_t *__restrict__ blk_offs,
                                                               uint32_t *__restrict__ labels,
                                                               uint32_t *__restrict__ gidx, uint32_t *__restrict__ oidx,
                                                               float *__restrict__ OX, float *__restrict__ OY,
                                                               float *__restrict__ OZ, float4 *__restrict__ nodes,
                                                               const SegState *__restrict__ st,
                                                               float *__restrict__ planes, FrameState *frame, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<0>(fv.fs);
    __shared__ uint32_t wg[SEG_WAVES], wo[SEG_WAVES];
    __shared__ uint32_t s_base[3][SEG_WAVES];
    flags = lpx_slot(flags, fv.fs);
    sidx = lpx_slot(sidx, fv.fs);
    XS = lpx_slot(XS, fv.fs);
    YS = lpx_slot(YS, fv.fs);
    ZS = lpx_slot(ZS, fv.fs);
    blk_offs = lpx_slot(blk_offs, fv.fs);
    OX = lpx_slot(OX, fv.fs);
    OY = lpx_slot(OY, fv.fs);
    OZ = lpx_slot(OZ, fv.fs);
    nodes = lpx_slot(nodes, fv.fs);
    st = lpx_slot(st, fv.fs);
    frame = lpx_slot(frame, fv.fs);
    labels = lpx_user(labels, fv.upitch);
    gidx = lpx_user(gidx, fv.upitch);
    oidx = lpx_user(oidx, fv.upitch);
    planes = lpx_user(planes, 4u * prm.P);
    seg_bind(prm, frame);
    const uint32_t s = lpx_blk.y, b = lpx_blk.x;
    const uint32_t tid = threadIdx.x, lane = tid % WAVE, w = tid / WAVE;
    const uint32_t nb = prm.P * prm.bps;
    uint32_t base_unused, lo, hi;  // the block's points: the same ranges as the plane passes that counted them
    seg_block_range(prm, s, b, base_unused, lo, hi);
    // The block's offsets into the two output lists from the RAW per-block counts the final plane pass left
    // ([ground counts | obstacle counts], nb each): every block sums what lies before it by itself -- a few hundred words
    // at most for a 120k-point frame, L2-resident -- instead of a scan kernel between the two launches (under load every
    // launch of a chain, however small, waits ~0.4 ms for its turn).
    const uint32_t me = s * prm.bps + b;
    uint32_t sg_all = 0, sg_before = 0, so_before = 0, so_all = 0;
    for (uint32_t i = tid; i < nb; i += SEG_THREADS)
    {
        const uint32_t g = blk_offs[i], o = blk_offs[nb + i];
        sg_all += g;
        so_all += o;
        sg_before += i < me ? g : 0u;
        so_before += i < me ? o : 0u;
    }
    sg_all = lpx_wave_sum_u32(sg_all);
    sg_before = lpx_wave_sum_u32(sg_before);
    so_before = lpx_wave_sum_u32(so_before);
    so_all = lpx_wave_sum_u32(so_all);
    if (lane == 0)
    {
        s_base[0][w] = sg_all;
        s_base[1][w] = sg_before;
        s_base[2][w] = so_before;
        wg[w] = so_all;  // (wg / wo are filled with the wave counts further down, after the barrier below)
    }
    __syncthreads();
    uint32_t total_g = 0, gbase = 0, obase = 0, total_o = 0;
#pragma unroll
    for (int i = 0; i < SEG_WAVES; ++i)
    {
        total_g += s_base[0][i];
        gbase += s_base[1][i];
        obase += s_base[2][i];
        total_o += wg[i];
    }
    __syncthreads();
    const unsigned long long lt = lpx_lanemask_lt();

    // each wave owns a contiguous quarter of the chunk, CR rounds of 64 points at a time: the flags, sorted indices and
    // coordinates of all CR rounds are requested before the first is used (a round used to be a chain of three
    // dependent loads, sixteen rounds per wavefront: the wavefront spent its life waiting -- and held its slot)
    constexpr int CR = 16;
    const uint32_t span = hi > lo ? hi - lo : 0;
    const uint32_t per_w = (span + SEG_WAVES - 1) / SEG_WAVES;
    const uint32_t wlo = min(lo + w * per_w, hi), whi = min(wlo + per_w, hi);
    uint32_t cg = 0, co = 0;
    for (uint32_t p0 = wlo; p0 < whi; p0 += WAVE * CR)
    {
        uint32_t f[CR];
#pragma unroll
        for (int r = 0; r < CR; ++r)
        {
            const uint32_t p = p0 + r * WAVE + lane;
            f[r] = (p < whi) ? flags[p] : 0u;
        }
#pragma unroll
        for (int r = 0; r < CR; ++r)
        {
            cg += __popcll(__ballot(f[r] == 1u));
            co += __popcll(__ballot(f[r] == 2u));
        }
    }
    if (lane == 0)
    {
        wg[w] = cg;
        wo[w] = co;
    }
    __syncthreads();
    uint32_t gpos = gbase, opos = obase;
    for (uint32_t i = 0; i < w; ++i)
    {
        gpos += wg[i];
        opos += wo[i];
    }
    const bool want_hash = gridDim.z == 1;  // single-frame calls: the host may be handed this cloud back (lpx_cluster)
    uint64_t hsum = 0, hsum2 = 0;
    for (uint32_t p0 = wlo; p0 < whi; p0 += WAVE * CR)
    {
        uint32_t f[CR], si[CR];
        float ox[CR], oy[CR], oz[CR];
#pragma unroll
        for (int r = 0; r < CR; ++r)
        {
            const uint32_t p = p0 + r * WAVE + lane;
            const bool in = p < whi;
            f[r] = in ? flags[p] : 0u;  // (L2 hits: the count pass has just read them)
            si[r] = in ? sidx[p] : 0u;
            ox[r] = in ? XS[p] : 0.0f;
            oy[r] = in ? YS[p] : 0.0f;
            oz[r] = in ? ZS[p] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < CR; ++r)
        {
            const uint32_t p = p0 + r * WAVE + lane;
            const bool in = p < whi;
            const unsigned long long mg = __ballot(f[r] == 1u), mo = __ballot(f[r] == 2u);
            if (in)
            {
                const uint32_t i = si[r];
                if (SCATTER_LABELS)
                    labels[i] = f[r];  // (large frames: labels_direct_kernel, no scatter)
                if (f[r] == 1u)
                    gidx[gpos + __popcll(mg & lt)] = i;
                else if (f[r] == 2u)
                {
                    const uint32_t d = opos + __popcll(mo & lt);
                    oidx[d] = i;
                    OX[d] = ox[r];
                    OY[d] = oy[r];
                    OZ[d] = oz[r];
                    nodes[d] = make_float4(ox[r], oy[r], oz[r], __uint_as_float(d));  // kd-tree input, KDTree::rebuild :185-189
                    if (want_hash)
                    {
                        hsum += lpx_obstacle_mix(d, __float_as_uint(ox[r]), __float_as_uint(oy[r]), __float_as_uint(oz[r]));
                        hsum2 += lpx_obstacle_mix2(d, __float_as_uint(ox[r]), __float_as_uint(oy[r]), __float_as_uint(oz[r]));
                    }
                }
            }
            gpos += __popcll(mg);
            opos += __popcll(mo);
        }
    }
    if (want_hash)
    {
        hsum = (uint64_t)lpx_wave_sum_i64((long long)hsum);  // wrapping sum, valid in lane 0
        if (lane == 0 && hsum)
            atomicAdd((unsigned long long *)&frame->obs_hash, (unsigned long long)hsum);
        hsum2 = (uint64_t)lpx_wave_sum_i64((long long)hsum2);
        if (lane == 0 && hsum2)
            atomicAdd((unsigned long long *)&frame->obs_hash2, (unsigned long long)hsum2);
    }
    if (s == 0 && b == 0)
    {
        // the N mod P highest-x points belong to no segment (Q2); written UNKNOWN (Q3)
        if (SCATTER_LABELS)
            for (uint32_t p = prm.P * prm.n_per + tid; p < prm.n; p += SEG_THREADS)
                labels[sidx[p]] = LPX_LABEL_UNKNOWN;
        if (tid == 0)
        {
            frame->n_ground = total_g;
            frame->n_obstacle = frame->status ? 0u : total_o;  // a frame in error is not clustered
        }
        if (planes)
            for (uint32_t i = tid; i < prm.P * 4; i += SEG_THREADS)
                planes[i] = st[i / 4].plane[i % 4];
    }
}

// The labels in ORIGINAL order, WITHOUT a scatter (round 6).  The compaction walks the cloud in x-sorted order and knows a
// point's label there; its original index is random with respect to that order, so writing the caller's label from
// there was one memory request per point whatever its width (4-byte stores: 3.5 x the algorithmic bytes moved, 15 % of
// the HBM roofline on 1M-point frames; one byte per point into an L2-resident scratch plus a widening pass: 19 % -- the
// lines merge, the requests stay: 11.6 M L2 requests per 8 M points, 8 M of them this scatter).  A label is a function
// of the point alone once the planes are final: its segment follows from its x key and index against the P segment
// boundaries of the sorted order -- (key, index) pairs, compared the way the stable sort orders them -- and its side of
// the segment's plane from final_label(), the very expression the final plane pass evaluates.  So this kernel streams the
// caller's records once more in input order (16 of every 32 bytes, coalesced) and stores the labels coalesced.
// Identical results by construction (same floats, same expressions, -ffp-contract=off); the N mod P points beyond the
// last segment are UNKNOWN (Q2, Q3) as before.
__device__ __forceinline__ uint32_t final_label(float lo_excl, float hi_incl, float pa, float pb, float pc, float pd, float thr,
                                                uint32_t code, uint32_t I, float x, float y, float z)
{
    // code: bit 0 nothing is labelled (failed == 2), bit 1 dead (failed != 0), bit 2 the segment has seeds
    const bool skip = code & 1u, dead = code & 2u, seeds_ok = code & 4u;
    bool member;
    if (I == 0)
        member = seeds_ok && (z > lo_excl) && (z <= hi_incl);  // the only pass is pass 0: the seed window (:243)
    else
    {
        const float dist = plane_dist(x, y, z, pa, pb, pc, pd);  // what pass_quad_lean / pass_block_generic evaluate
        member = dist < thr;
    }
    member = member && !dead;
    // number_of_iterations == 0: seeds are ground, the rest stays UNKNOWN (:243-247)
    return skip ? LPX_LABEL_UNKNOWN : (member ? LPX_LABEL_GROUND : ((I == 0 && !dead) ? LPX_LABEL_UNKNOWN : LPX_LABEL_OBSTACLE));
}

constexpr int LBL_THREADS = 256, LBL_PER = 4;
constexpr uint32_t LABELS_DIRECT_FROM = 262144u;  // points of the largest frame of a call
__global__ __launch_bounds__(LBL_THREADS) void labels_direct_kernel(const float4 *__restrict__ rec, LpxRecLayout lay,
                                                                    const float *__restrict__ XS,
                                                                    const uint32_t *__restrict__ sidx, SegParams prm,
                                                                    const SegState *__restrict__ st,
                                                                    uint32_t *__restrict__ labels,
                                                                    const FrameState *__restrict__ frame, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<0>(fv.fs);
    __shared__ unsigned long long s_bound[LPX_MAX_PARTITIONS];  // (key << 32 | index) of the first point of segments 1 .. P
    __shared__ float s_pl[LPX_MAX_PARTITIONS][7];
    __shared__ uint32_t s_code[LPX_MAX_PARTITIONS];
    if (lay.stride == 0)
        rec = lpx_slot(rec, fv.fs);  // the arena's copy of the cloud
    else
        rec = (const float4 *)((const char *)rec + (size_t)lpx_blk.z * lay.pitch * lay.stride);  // the records of the call
    XS = lpx_slot(XS, fv.fs);
    sidx = lpx_slot(sidx, fv.fs);
    st = lpx_slot(st, fv.fs);
    frame = lpx_slot(frame, fv.fs);
    labels = lpx_user(labels, fv.upitch);
    seg_bind(prm, frame);
    const uint32_t n = prm.n, P = prm.P, n_per = prm.n_per;
    if (lpx_blk.x * (uint32_t)(LBL_THREADS * LBL_PER) >= n)
        return;
    for (uint32_t sgm = threadIdx.x; sgm < P; sgm += LBL_THREADS)
    {
        // boundary sgm + 1: the first sorted position that no longer belongs to segment sgm
        const uint32_t pos = (sgm + 1u) * n_per;
        s_bound[sgm] = (n_per && pos < n) ? (((unsigned long long)lpx_float_key(XS[pos]) << 32) | sidx[pos]) : ~0ull;
        const SegState ss = st[sgm];
        s_pl[sgm][0] = ss.lo_excl, s_pl[sgm][1] = ss.hi_incl;
        s_pl[sgm][2] = ss.plane[0], s_pl[sgm][3] = ss.plane[1], s_pl[sgm][4] = ss.plane[2], s_pl[sgm][5] = ss.plane[3];
        s_pl[sgm][6] = ss.thr;
        s_code[sgm] = (ss.failed == 2 ? 1u : 0u) | (ss.failed != 0 ? 2u : 0u) | (ss.has_seeds ? 4u : 0u);
    }
    __syncthreads();
    const uint32_t i0 = lpx_blk.x * (uint32_t)(LBL_THREADS * LBL_PER) + threadIdx.x;
    float4 q[LBL_PER];
#pragma unroll
    for (int u = 0; u < LBL_PER; ++u)
    {
        const uint32_t i = i0 + u * LBL_THREADS;
        q[u] = lpx_rec_xyz(rec, i < n ? i : 0u, lay);
    }
#pragma unroll
    for (int u = 0; u < LBL_PER; ++u)
    {
        const uint32_t i = i0 + u * LBL_THREADS;
        if (i >= n)
            continue;
        const unsigned long long me = ((unsigned long long)lpx_float_key(q[u].x) << 32) | i;
        // the segment: how many boundaries lie at or before this point in (key, index) order
        uint32_t lo = 0, hi = P;  // first sgm in [lo, hi) with s_bound[sgm] > me
        while (lo < hi)
        {
            const uint32_t mid = (lo + hi) / 2;
            if (s_bound[mid] <= me)
                lo = mid + 1;
            else
                hi = mid;
        }
        uint32_t lab = LPX_LABEL_UNKNOWN;  // lo == P: one of the N mod P highest-x points (or n < P: no segment at all)
        if (lo < P && n_per)
            lab = final_label(s_pl[lo][0], s_pl[lo][1], s_pl[lo][2], s_pl[lo][3], s_pl[lo][4], s_pl[lo][5], s_pl[lo][6],
                              s_code[lo], prm.I, q[u].x, q[u].y, q[u].z);
        labels[i] = lab;
    }
}

// ------------------------------------------------------------------------------------------------
// egress: the two clouds Processor::process builds right after segment() (reference src/processor.cpp:152-163:
// ground -> PointXYZRGBL(x, y, z, 220, 220, 220, label 0), obstacle -> PointXYZRGBL(x, y, z, 0, 255, 0, label 1))
// and then memcpy's into the data[] of the two published PointCloud2 messages (src/conversions.cpp:164-193).
// Record = 32 bytes as pcl::PointXYZRGBL lays them out (PCL 1.12 point_types: float x, y, z, 1.0f | b, g, r, a = 255
// | uint32 label | 8 bytes of padding, written as zero).  One thread per record, two 16-byte stores.
// ------------------------------------------------------------------------------------------------
__global__ void colour_kernel(const float4 *__restrict__ P4, LpxRecLayout lay,
                              const uint32_t *__restrict__ gidx, const uint32_t *__restrict__ oidx,
                              const FrameState *__restrict__ frame, float4 *__restrict__ grec,
                              float4 *__restrict__ orec, FV fv)
{
    const LpxBlock lpx_blk = lpx_block<0>(fv.fs);
    if (lay.stride == 0)
        P4 = lpx_slot(P4, fv.fs);  // the arena's copy of the cloud
    else
        P4 = (const float4 *)((const char *)P4 + (size_t)lpx_blk.z * lay.pitch * lay.stride);  // the records of the call
    frame = lpx_slot(frame, fv.fs);
    gidx = lpx_user(gidx, fv.upitch);
    oidx = lpx_user(oidx, fv.upitch);
    grec = lpx_user(grec, 2u * fv.upitch);
    orec = lpx_user(orec, 2u * fv.upitch);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    const uint32_t ng = frame->n_ground, no = frame->n_obstacle;
    if (i >= ng + no)
        return;
    const bool g = i < ng;
    const uint32_t j = g ? i : i - ng;
    const uint32_t k = g ? gidx[j] : oidx[j];
    float4 *dst = (g ? grec : orec) + 2 * (size_t)j;
    const float4 q = lpx_rec_xyz(P4, k, lay);
    dst[0] = make_float4(q.x, q.y, q.z, 1.0f);
    const uint32_t rgba = g ? 0xffdcdcdcu : 0xff00ff00u;  // a r g b from the top byte down: bytes b, g, r, a in memory
    dst[1] = make_float4(__uint_as_float(rgba), __uint_as_float(g ? 0u : 1u), 0.0f, 0.0f);
}

__global__ void dbg_all_seed_kernel(SegState *st, long long *facc)
{
    seg_far_reset(facc, 0, threadIdx.x);
    if (threadIdx.x == 0)
    {
        SegState o;
        o.lo_excl = -INFINITY;
        o.hi_incl = INFINITY;
        o.has_seeds = 1;
        o.failed = 0;
        o.plane[0] = o.plane[1] = o.plane[2] = o.plane[3] = 0.0f;
        o.fitted = 0;
        o.thr = 0.0f;
        o.pad[0] = o.pad[1] = 0;
        st[0] = o;
    }
}

__global__ void dbg_plane_out_kernel(const SegState *st, float *out)
{
    st += LPX_MAX_PARTITIONS;  // the state the head of pass 1 published (set 1)
    if (threadIdx.x < 4)
        out[threadIdx.x] = st[0].plane[threadIdx.x];
    if (threadIdx.x == 4)
        out[4] = st[0].failed ? 1.0f : 0.0f;
}

__global__ void fill_u32_kernel(uint32_t *p, uint32_t v, uint32_t n)
{
    const LpxBlock lpx_blk = lpx_block<0>(0);
    const uint32_t i = lpx_blk.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = v;
}
}  // namespace

static uint32_t bits_for(uint32_t v)  // number of bits needed to represent values 0..v
{
    uint32_t b = 0;
    while (v)
    {
        ++b;
        v >>= 1;
    }
    return b ? b : 1;
}

// n: point bound of the largest frame; hist: where the first pass of lpx_sort_pairs expects its tile histograms, or null
static void launch_ingest(lpx_ctx *ctx, uint32_t n, const void *d_pts, size_t stride, float *X, float *Y, float *Z,
                          float4 *P4, uint32_t *key, uint32_t *val, FrameState *frame, float4 *nodes, uint32_t *hist)
{
    const XyzOff off = {ctx->in_off[0], ctx->in_off[1], ctx->in_off[2]};
    const bool aligned = (((uintptr_t)d_pts | stride | off.x | off.y | off.z) & 3u) == 0;
    const dim3 grid((n + LPX_SORT_TILE - 1) / LPX_SORT_TILE, 1, ctx->cur_b);
    if (aligned)
        hipLaunchKernelGGL(ingest_kernel<true>, grid, dim3(INGEST_THREADS), 0, ctx->stream, (const char *)d_pts, stride,
                           off, X, Y, Z, P4, key, val, frame, nodes, hist, lpx_fv(ctx));
    else
        hipLaunchKernelGGL(ingest_kernel<false>, grid, dim3(INGEST_THREADS), 0, ctx->stream, (const char *)d_pts, stride,
                           off, X, Y, Z, P4, key, val, frame, nodes, hist, lpx_fv(ctx));
}

int lpx_run_colour(lpx_ctx *ctx, uint32_t n_max, const uint32_t *d_gidx, const uint32_t *d_oidx, void *d_grec,
                   void *d_orec)
{
    if (n_max == 0)
        return LPX_OK;
    // the coordinates: the arena's copy of the cloud, or -- when the segmentation made none (rec_direct) -- the records
    // of that call where they lie
    const uint32_t table_off[3] = {0, 4, 8};
    const LpxRecLayout lay = ctx->rec_direct ? lpx_rec_layout(ctx->rec_ptr, ctx->rec_stride, ctx->rec_off, ctx->rec_pitch)
                                             : lpx_rec_layout(ctx->pts4.p, 0, table_off, 0);
    hipLaunchKernelGGL(colour_kernel, dim3((n_max + 255) / 256, 1, ctx->cur_b), dim3(256), 0, ctx->stream,
                       (const float4 *)(ctx->rec_direct ? ctx->rec_ptr : ctx->pts4.p), lay, d_gidx, d_oidx,
                       (const FrameState *)ctx->frame.p, (float4 *)d_grec, (float4 *)d_orec, lpx_fv(ctx));
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_frame_init(lpx_ctx *ctx, const uint32_t *n_points, bool as_obstacles)
{
    NArr na;
    for (uint32_t b = 0; b < LPX_MAX_BATCH; ++b)
        na.v[b] = b < ctx->cur_b ? n_points[b] : 0u;
    hipLaunchKernelGGL(frame_init_kernel, dim3(1, 1, ctx->cur_b), dim3(WAVE), 0, ctx->stream, (FrameState *)ctx->frame.p,
                       na, as_obstacles ? 1u : 0u, ctx->fs_tag);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_write_counts(lpx_ctx *ctx, uint32_t *d_counts)
{
    hipLaunchKernelGGL(counts_kernel, dim3(1, 1, ctx->cur_b), dim3(WAVE), 0, ctx->stream,
                       (const FrameState *)ctx->frame.p, d_counts, ctx->fs_tag);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

// plane of ALL n points (stride-12 device input) through the moment + Jacobi path; out[0..3] plane, out[4] failed
// Launch geometry of the plane passes and the compaction from prm.n_per (the largest segment of the call): blocks of
// SEG_CHUNK points, larger ones once a segment would need more than 256 of them (every block of pass t + 1 reads the
// rows of ALL blocks of its segment) up to 256 points per lane (the bound of the int64 moment lanes); one block more
// where a segment's 16-byte-aligned window starts before the segment.
static void seg_geometry(SegParams &prm, uint32_t cap_n)
{
    uint32_t chunk = SEG_CHUNK;
    const uint32_t quantum = 2u * PASS_TRIP;  // an even number of trips per block
    // long segments (BASELINE's 1M- and 5M-point clouds: 83k / 208k points each): 8192-point blocks -- the plane passes
    // run at the same rate, the compaction (same geometry: fewer per-block offset sums, longer runs per lane) 16 % faster
    if (prm.n_per >= 65536u && chunk < 2u * SEG_CHUNK)
        chunk = 2u * SEG_CHUNK;
    if (prm.n_per / 256u > chunk)
    {
        chunk = ((prm.n_per / 256u + quantum - 1) / quantum) * quantum;
        if (chunk > 256u * PASS_THREADS)
            chunk = 256u * PASS_THREADS;  // 256 points per lane: the bound of the int64 moment lanes
    }
    prm.chunk = chunk;
    prm.bps = prm.n_per ? (prm.n_per + 3u + chunk - 1) / chunk : 1;
    prm.part_stride = (uint32_t)(LPX_SEG_MAX_BLOCKS(cap_n) * LPX_ACC_WORDS);
    prm.head_solve = 1;  // (lpx_pass_form decides per launch)
}

// Where the 3x3 solve of a plane pass runs (plane_pass_kernel): at the head of every block when the blocks of a launch
// are resident all at once (three one-wavefront blocks per SIMD: 3072), at the tail of the last block of a segment
// otherwise.  LPX_PASS_SOLVE=head|tail (development build) forces one form: the tests run both against the oracle.
static uint32_t lpx_pass_form(uint32_t blocks_of_a_launch)
{
    static const char *e = LPX_KNOB("LPX_PASS_SOLVE");
    if (e)
        return strcmp(e, "tail") == 0 ? 0u : 1u;
    return blocks_of_a_launch <= 3072u ? 1u : 0u;
}

int lpx_dbg_plane_run(lpx_ctx *ctx, const void *d_pts, uint32_t n, float *d_out)
{
    FrameState *frame = (FrameState *)ctx->frame.p;
    SegParams prm;
    prm.n = n;
    prm.n_per = n;
    prm.P = 1;
    prm.I = 1;
    seg_geometry(prm, ctx->cap_n);
    prm.head_solve = lpx_pass_form(prm.bps);
    prm.z_floor = 0.0f;
    prm.seed_thr = 0.0f;
    prm.odt = 0.0f;
    prm.n_lpr = 0;
    float *XS = (float *)ctx->XS.p, *YS = (float *)ctx->YS.p, *ZS = (float *)ctx->ZS.p;
    SegState *sst = (SegState *)ctx->seg_state.p;
    long long *part = (long long *)ctx->seg_part.p;
    const FV fv = lpx_fv(ctx);
    int rc = lpx_frame_init(ctx, &n, false);
    if (rc)
        return rc;
    if (n)
        launch_ingest(ctx, n, d_pts, 12, XS, YS, ZS, nullptr, nullptr, nullptr, frame, nullptr, nullptr);
    // pass 0 leaves the moments of every point, the head of pass 1 (run as the final pass) solves and publishes the plane
    hipLaunchKernelGGL(dbg_all_seed_kernel, dim3(1), dim3(128), 0, ctx->stream, sst, (long long *)ctx->seg_far.p);
    hipLaunchKernelGGL((plane_pass_kernel<false>), dim3(prm.bps, 1), dim3(PASS_THREADS), 0, ctx->stream, XS, YS, ZS, prm,
                       0u, sst, part, (long long *)ctx->seg_far.p, (uint8_t *)ctx->flags.p, (uint32_t *)ctx->blk_counts.p,
                       (const FrameState *)frame, fv.fs);
    hipLaunchKernelGGL((plane_pass_kernel<true>), dim3(prm.bps, 1), dim3(PASS_THREADS), 0, ctx->stream, XS, YS, ZS, prm,
                       1u, sst, part, (long long *)ctx->seg_far.p, (uint8_t *)ctx->flags.p, (uint32_t *)ctx->blk_counts.p,
                       (const FrameState *)frame, fv.fs);
    hipLaunchKernelGGL(dbg_plane_out_kernel, dim3(1), dim3(64), 0, ctx->stream, sst, d_out);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

int lpx_ingest_obstacles(lpx_ctx *ctx, const void *d_pts, size_t stride, uint32_t m)
{
    FrameState *frame = (FrameState *)ctx->frame.p;
    int rc = lpx_frame_init(ctx, &m, true);
    if (rc)
        return rc;
    if (m)
    {
        launch_ingest(ctx, m, d_pts, stride, (float *)ctx->OX.p, (float *)ctx->OY.p, (float *)ctx->OZ.p, nullptr, nullptr,
                      nullptr, frame, (float4 *)ctx->nodes.p, nullptr);
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

// n_points: the point count of every frame of the call (ctx->cur_b entries)
int lpx_run_segment(lpx_ctx *ctx, const void *d_pts, size_t stride, const uint32_t *n_points, const lpx_seg_cfg *cfg,
                    uint32_t *d_labels, uint32_t *d_gidx, uint32_t *d_oidx, float *d_planes)
{
    FrameState *frame = (FrameState *)ctx->frame.p;
    const uint32_t P = cfg->number_of_planar_partitions;
    const uint32_t I = cfg->number_of_iterations;
    if (P == 0 || P > LPX_MAX_PARTITIONS || I > LPX_MAX_ITERATIONS)
        return lpx_fail(ctx, LPX_ERR_ARG, "partitions must be 1..%u and iterations 0..%u", LPX_MAX_PARTITIONS,
                        LPX_MAX_ITERATIONS);
    hipStream_t st = ctx->stream;
    const uint32_t B = ctx->cur_b;
    const FV fv = lpx_fv(ctx);
    uint32_t n = 0;  // the largest frame sizes every launch; kernels use their own frame's count
    for (uint32_t b = 0; b < B; ++b)
        n = n_points[b] > n ? n_points[b] : n;
    int rc = lpx_frame_init(ctx, n_points, false);
    if (rc)
        return rc;
    if (B == 1 && d_planes && n / P == 0)  // no segment gets a point: nothing else writes the planes
        LPX_HIP(ctx, hipMemsetAsync(d_planes, 0, sizeof(float) * 4 * P, st));
    if (n == 0)
    {
        if (B > 1 && d_planes)
            LPX_HIP(ctx, hipMemsetAsync(d_planes, 0, sizeof(float) * 4 * P * B, st));
        return LPX_OK;
    }

    SegParams prm;
    prm.n = n;
    prm.n_per = n / P;
    prm.P = P;
    prm.I = I;
    seg_geometry(prm, ctx->cap_n);
    prm.z_floor = -1.5f * cfg->sensor_height_m;
    prm.seed_thr = cfg->initial_seed_threshold;
    prm.odt = cfg->orthogonal_distance_threshold;
    prm.n_lpr = cfg->number_of_lower_point_representatives;

    float4 *P4 = (float4 *)ctx->pts4.p;  // the cloud in original order, 16-byte records
    float *XS = (float *)ctx->XS.p, *YS = (float *)ctx->YS.p, *ZS = (float *)ctx->ZS.p;
    const dim3 blk(256), grd((n + 255) / 256, 1, B);

    uint32_t *first_hist = lpx_sort_first_hist(ctx, n);  // the ingest also counts the lowest key byte per sort tile
    // segments that fit one workgroup's registers get their seed statistics by selection; larger ones (or more
    // representatives than the LDS sort holds) by the full (segment, z) sort
    const bool select_seeds = prm.n_per <= SEL_MAX_POINTS && prm.n_lpr <= SEL_MAX_LPR;
    // (the selection path needs nothing from gather_kernel but the x-sorted SoA: the last pass of the sort writes it)
    // ... and longer segments by the same selection spread over many workgroups (selw_*), unless the development build
    // says LPX_SEEDS=sort
    static const char *seeds_env = LPX_KNOB("LPX_SEEDS");
    const bool select_wide = !select_seeds && prm.n_lpr <= SEL_MAX_LPR && sizeof(SelWide) * (size_t)P <= ctx->key64_a.bytes &&
                             !(seeds_env && strcmp(seeds_env, "sort") == 0);
#ifdef LPX_NO_FUSED_GATHER
    const bool fused_gather = false;
#else
    const bool fused_gather = (select_seeds || select_wide) && !(B == 1 && prm.n_per == 0);
#endif
    // With the gather inside the last sort pass nobody needs a copy of the cloud in input order: that pass (and
    // lpx_coloured_clouds*) read x, y, z of a point from the records of the call where they lie -- one request per
    // point either way -- and the ingest writes 4 bytes per point (the x key) instead of 20.
    ctx->rec_ptr = d_pts;
    ctx->rec_stride = stride;
    ctx->rec_off[0] = ctx->in_off[0], ctx->rec_off[1] = ctx->in_off[1], ctx->rec_off[2] = ctx->in_off[2];
    ctx->rec_pitch = ctx->upitch;
    // (lpx_set_record_copy: the caller recycles its input buffer before it asks for the coloured clouds -- the ingest then
    // writes the 16-byte copy as it did before round 5, and the last sort pass and colour_kernel read THAT)
    const bool direct = fused_gather && !ctx->keep_copy;
    ctx->rec_direct = direct;
    {
        StageTimer tm(ctx, ST_INGEST);
        launch_ingest(ctx, n, d_pts, stride, nullptr, nullptr, nullptr, direct ? (float4 *)nullptr : P4,
                      (uint32_t *)ctx->key_a.p, (uint32_t *)nullptr, frame, nullptr, first_hist);  // (values: iota_vals)
    }
    LpxSortGather sg;
    sg.records = direct ? d_pts : (const void *)P4;
    sg.x = XS, sg.y = YS, sg.z = ZS;
    if (direct)
    {
        sg.stride = stride;
        sg.off[0] = ctx->in_off[0], sg.off[1] = ctx->in_off[1], sg.off[2] = ctx->in_off[2];
        sg.pitch = ctx->upitch;
    }
    uint32_t *skeys = nullptr, *sidx = nullptr;
    {
        StageTimer tm(ctx, ST_XSORT);
        rc = lpx_sort_pairs(ctx, (uint32_t *)ctx->key_a.p, (uint32_t *)ctx->key_b.p, (uint32_t *)ctx->val_a.p,
                            (uint32_t *)ctx->val_b.p, n, &frame->n_in, 32, &skeys, &sidx, first_hist != nullptr,
                            fused_gather ? &sg : nullptr, true);
        if (rc)
            return rc;
    }
    if (B == 1 && prm.n_per == 0)
    {
        // fewer points than partitions: no segment holds a point, every label is UNKNOWN
        hipLaunchKernelGGL(fill_u32_kernel, dim3((n + 255) / 256), blk, 0, st, d_labels, LPX_LABEL_UNKNOWN, n);
        LPX_HIP(ctx, hipGetLastError());
        return LPX_OK;
    }
    if (!fused_gather)
    {
        StageTimer tm(ctx, ST_GATHER);
        hipLaunchKernelGGL(gather_kernel, dim3((n + 256 * GATHER_ITEMS - 1) / (256 * GATHER_ITEMS), 1, B), blk, 0, st, sidx,
                           (const float4 *)P4, XS, YS, ZS,
                           (select_seeds || select_wide) ? (uint64_t *)nullptr : (uint64_t *)ctx->key64_a.p, prm,
                           (const FrameState *)frame, fv.fs);
    }
    SegState *sst = (SegState *)ctx->seg_state.p;
    long long *part = (long long *)ctx->seg_part.p;
    long long *facc = (long long *)ctx->seg_far.p;
    if (select_seeds)
    {
        StageTimer tm(ctx, ST_SEEDS);
#ifndef LPX_SEED_WIDE_IN_CHAINS  // (A/B switch of tools/build_variant.sh)
        if (B > 1)  // launch chains: the narrow form (a 1024-thread workgroup waits longest for a compute unit under load)
            hipLaunchKernelGGL(seed_select_kernel<SEL_NARROW>, dim3(P, 1, B), dim3(SEL_NARROW), 0, st, ZS, prm, sst, facc,
                               (const FrameState *)frame, fv.fs);
        else
#endif
            hipLaunchKernelGGL(seed_select_kernel<SEL_THREADS>, dim3(P, 1, B), dim3(SEL_THREADS), 0, st, ZS, prm, sst, facc,
                               (const FrameState *)frame, fv.fs);
    }
    else if (select_wide)
    {
        StageTimer tm(ctx, ST_SEEDS);
        SelWide *sw = (SelWide *)ctx->key64_a.p;
        const dim3 gw((prm.n_per + SELW_TILE - 1) / SELW_TILE, P, B), bw(SELW_THREADS);
        hipLaunchKernelGGL(selw_clear_kernel, dim3(P, 1, B), dim3(256), 0, st, sw, fv.fs);
        hipLaunchKernelGGL(selw_pass_kernel<0>, gw, bw, 0, st, ZS, prm, sw, (const FrameState *)frame, fv.fs);
        hipLaunchKernelGGL(selw_pass_kernel<1>, gw, bw, 0, st, ZS, prm, sw, (const FrameState *)frame, fv.fs);
        hipLaunchKernelGGL(selw_pass_kernel<2>, gw, bw, 0, st, ZS, prm, sw, (const FrameState *)frame, fv.fs);
        hipLaunchKernelGGL(selw_pass_kernel<3>, gw, bw, 0, st, ZS, prm, sw, (const FrameState *)frame, fv.fs);
        hipLaunchKernelGGL(selw_pass_kernel<4>, gw, bw, 0, st, ZS, prm, sw, (const FrameState *)frame, fv.fs);
        hipLaunchKernelGGL(selw_final_kernel, dim3(P, 1, B), dim3(SEL_THREADS), 0, st, prm, (const SelWide *)sw, sst, facc,
                           (const FrameState *)frame, fv.fs);
    }
    else
    {
        uint64_t *zsorted = nullptr;
        {
            StageTimer tm(ctx, ST_ZSORT);
            rc = lpx_sort_keys64(ctx, (uint64_t *)ctx->key64_a.p, (uint64_t *)ctx->key64_b.p, n, &frame->n_in,
                                 32 + bits_for(P), &zsorted);
            if (rc)
                return rc;
        }
        StageTimer tm(ctx, ST_SEEDS);
        hipLaunchKernelGGL(seed_kernel, dim3(P, 1, B), dim3(SEG_THREADS), 0, st, zsorted, prm, sst, facc,
                           (const FrameState *)frame, fv.fs);
    }
    const uint32_t nb = P * prm.bps;
    if (sizeof(uint32_t) * (2 * (size_t)nb + 2) > ctx->blk_counts.bytes ||
        sizeof(long long) * LPX_ACC_WORDS * (size_t)nb > ctx->seg_part.bytes / 2)
        return lpx_fail(ctx, LPX_ERR_INTERNAL, "block tables of %u blocks do not fit the workspace", nb);
    uint32_t *blk_counts = (uint32_t *)ctx->blk_counts.p;
    {
        StageTimer tm(ctx, ST_PLANE);
        // One launch per pass, 256-thread workgroups that schedule anywhere.  (Round 1-2 gave a single frame ONE launch
        // for all passes -- a 1024-thread workgroup per segment with its points in registers: under load that workgroup
        // waited for a whole CU to drain (3.2 ms against 0.5), and alone it has since been overtaken too: 0.145 ms
        // against 0.095 ms for the six launches of a 123k-point frame.  Removed.)
        static const char *chain_env = LPX_KNOB("LPX_PASS_CHAIN");  // development build: 1 / 0 forces the form
        // one launch for all passes where the passes are separate launches of MORE blocks than the device holds (the tail
        // form's domain: chains of 1M-point frames); launches that are resident at once keep the head form, whose
        // blocks wait for nothing
        const bool chained = I >= 1 && (size_t)(I + 1) * B <= 65535u &&
                             (chain_env ? chain_env[0] == '1' : lpx_pass_form(prm.bps * P * B) == 0u);
        if (chained)
        {
            prm.head_solve = 0;
            // (the XCD-affine renumbering only where a group of eight z values holds frames of ONE pass)
            const size_t fs_chain = (B % 8u == 0) ? fv.fs : (fv.fs & ~(size_t)(1u << 2));
            hipLaunchKernelGGL(plane_chain_kernel, dim3(prm.bps, P, (I + 1) * B), dim3(PASS_THREADS), 0, st, XS, YS, ZS, prm,
                               B, sst, part, facc, (uint8_t *)ctx->flags.p, blk_counts, (const FrameState *)frame, fs_chain);
        }
        else
        {
            const dim3 g2(prm.bps, P, B);
            prm.head_solve = lpx_pass_form(prm.bps * P * B);
            for (uint32_t t = 0; t < I; ++t)
                hipLaunchKernelGGL((plane_pass_kernel<false>), g2, dim3(PASS_THREADS), 0, st, XS, YS, ZS, prm, t, sst, part,
                                   facc, (uint8_t *)ctx->flags.p, blk_counts, (const FrameState *)frame, fv.fs);
            hipLaunchKernelGGL((plane_pass_kernel<true>), g2, dim3(PASS_THREADS), 0, st, XS, YS, ZS, prm, I, sst, part,
                               facc, (uint8_t *)ctx->flags.p, blk_counts, (const FrameState *)frame, fv.fs);
        }
    }
    {
        StageTimer tm(ctx, ST_COMPACT);
        // The labels by original index.  Frames above LABELS_DIRECT_FROM points: recomputed from the records in input order
        // (labels_direct_kernel: no scatter; 32 x 1M chain: compaction 583 -> 198 + 234 us).  Smaller frames: scattered from
        // the compaction as 4-byte stores -- a 123k-point frame's 0.5 MB of labels merge in its L2, and the second launch and
        // the second pass over the records would only cost (64-frame stream chain: 0.090 against 0.118 ms).
        static const char *ld_env = LPX_KNOB("LPX_LABELS_DIRECT");  // development build: 1 / 0 forces the form (tests)
        const bool labels_direct = ld_env ? ld_env[0] == '1' : n > LABELS_DIRECT_FROM;
#define LPX_COMPACT_ARGS                                                                                               \
    dim3(prm.bps, P, B), dim3(SEG_THREADS), 0, st, (const uint8_t *)ctx->flags.p, sidx, XS, YS, ZS, prm, blk_counts,      \
        d_labels, d_gidx, d_oidx, (float *)ctx->OX.p, (float *)ctx->OY.p, (float *)ctx->OZ.p, (float4 *)ctx->nodes.p,     \
        sst + (size_t)(I & 1u) * LPX_MAX_PARTITIONS, /* the state the head of the final pass published */               \
        d_planes, frame, fv
        if (labels_direct)
            hipLaunchKernelGGL(compact_kernel<false>, LPX_COMPACT_ARGS);
        else
            hipLaunchKernelGGL(compact_kernel<true>, LPX_COMPACT_ARGS);
#undef LPX_COMPACT_ARGS
        static const uint32_t table_off[3] = {0, 4, 8};
        const LpxRecLayout lay = direct ? lpx_rec_layout(d_pts, stride, ctx->in_off, ctx->upitch)
                                        : lpx_rec_layout(P4, 0, table_off, 0);
        if (labels_direct)
            hipLaunchKernelGGL(labels_direct_kernel, dim3((n + LBL_THREADS * LBL_PER - 1) / (LBL_THREADS * LBL_PER), 1, B),
                           dim3(LBL_THREADS), 0, st, (const float4 *)(direct ? d_pts : (const void *)P4), lay,
                           (const float *)XS, (const uint32_t *)sidx, prm,
                           (const SegState *)(sst + (size_t)(I & 1u) * LPX_MAX_PARTITIONS), d_labels,
                           (const FrameState *)frame, fv);
    }
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}
