// lpx_primitives.hip -- stable LSD radix sort and exclusive scan for gfx950 (wave64).
//
// Replaces the two std::sort(std::execution::par, ...) index sorts of the reference
// (src/segmentation.cpp:119-122, :165-168) and orders component members for the FEC replay.
// Sorting iota by key with a STABLE sort gives the canonical (key, index) tie order.
//
// Layout: a tile is 4 wavefronts x ITEMS rounds x 64 lanes; element order inside a tile is
// (wave, round, lane), i.e. ascending address, so ranks computed per (wave, digit) are stable.
// Per pass: histogram (digit-major [256][blocks]) -> single-block exclusive scan -> scatter.
#include "lpx_internal.h"

namespace
{
constexpr int SORT_THREADS = 256;
constexpr int SORT_WAVES = SORT_THREADS / WAVE;
constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
static_assert(SORT_TILE == (int)LPX_SORT_TILE, "the arena sizes the histogram table with LPX_SORT_TILE");
constexpr int RADIX = 256;
// Flag in the `block_major` / `large` word of the two kernels below: every key is BELOW the frame's element count n (the
// component sort: the keys are positions of the obstacle cloud).  The host sizes the passes for its BOUND of n -- 17 bits
// for a 123k-point frame -- while the frame holds ~50k obstacle points: in a pass whose digit starts at or above
// log2(n) every key has digit 0 and the stable pass is the identity, so the scatter only copies its tile (coalesced, no
// ranking, no table) and the histogram pass does nothing: the third pass of every KITTI frame.
constexpr int SORT_KEYS_BELOW_N = 2;
// Flag in the `large` word of radix_scatter_kernel: the small (block-major) table holds exclusive prefixes over the tiles
// and, in row nblocks, the digit totals (hist_cols_kernel).  Launch chains: every scatter workgroup summing the rows of
// all tiles itself is 61 x 61 KiB of L2 reads per pass and frame -- seven times the keys -- and a workgroup that waits for
// two trips of 32 rows holds its wave slots for them; one 256-thread workgroup per frame sums them once.  A single frame
// keeps the fused form: one launch less per pass is worth more there.
constexpr int SORT_PREFIXED = 4;

// Histogram of one digit per tile.  Small tables (frame-sized inputs, <= FUSED_SCAN_MAX_BLOCKS tiles) are stored
// block-major [nblocks][256] and never scanned: every scatter block sums the few rows it needs itself.  Large
// tables are stored digit-major [256][nblocks] for hist_rows_kernel.
template <typename KeyT>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const KeyT *__restrict__ keys, uint32_t n_max,
                                                                   const uint32_t *__restrict__ d_n, uint32_t shift,
                                                                   uint32_t *__restrict__ hist, uint32_t nblocks,
                                                                   int block_major, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<1>(fs);
    __shared__ uint32_t h[RADIX];
    keys = lpx_slot(keys, fs);
    d_n = lpx_slot(d_n, fs);
    hist = lpx_slot(hist, fs);
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    if ((block_major & SORT_KEYS_BELOW_N) && shift < 32u && n <= (1u << shift))
        return;  // the pass is the identity (radix_scatter_kernel): nobody reads this table
    const uint32_t tid = threadIdx.x;
    if ((block_major & SORT_PREFIXED) && lpx_blk.x * SORT_TILE >= n)
        return;  // (hist_cols_kernel reads the rows of the frame's own tiles only)
    block_major &= 1;
    h[tid] = 0;
    __syncthreads();
    const uint32_t base = lpx_blk.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r)
    {
        const uint32_t e = base + r * SORT_THREADS + tid;
        if (e < n)
            atomicAdd(&h[(uint32_t)(keys[e] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    if (block_major)
        hist[lpx_blk.x * RADIX + tid] = h[tid];
    else
        hist[tid * nblocks + lpx_blk.x] = h[tid];
}

// GATHER (last pass of the x sort): the sorted VALUES are indices into a table of 16-byte records {x, y, z, .}; the
// pass fetches the record of every value it places and writes the x-sorted SoA itself -- what gather_kernel did in a
// launch of its own, reading the sorted indices back.  The sorted keys are not written (nobody reads them).
template <typename KeyT, bool HAS_VALS, bool GATHER = false>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const KeyT *__restrict__ keys_in,
                                                                      KeyT *__restrict__ keys_out,
                                                                      const uint32_t *__restrict__ vals_in,
                                                                      uint32_t *__restrict__ vals_out, uint32_t n_max,
                                                                      const uint32_t *__restrict__ d_n, uint32_t shift,
                                                                      const uint32_t *__restrict__ offs,
                                                                      uint32_t nblocks, int large, size_t fs,
                                                                      const float4 *__restrict__ rec = nullptr,
                                                                      float *__restrict__ GX = nullptr,
                                                                      float *__restrict__ GY = nullptr,
                                                                      float *__restrict__ GZ = nullptr,
                                                                      LpxRecLayout lay = LpxRecLayout())
{
    const LpxBlock lpx_blk = lpx_block<1>(fs);
    __shared__ uint32_t wcnt[SORT_WAVES][RADIX];
    __shared__ uint32_t dsum[SORT_WAVES], dsum2[SORT_WAVES];
    __shared__ uint32_t gofs[RADIX];
    __shared__ KeyT stage_k[SORT_TILE];
    __shared__ uint32_t stage_v[HAS_VALS ? SORT_TILE : 1];
    keys_in = lpx_slot(keys_in, fs);
    keys_out = lpx_slot(keys_out, fs);
    vals_in = lpx_slot(vals_in, fs);
    vals_out = lpx_slot(vals_out, fs);
    d_n = lpx_slot(d_n, fs);
    offs = lpx_slot(offs, fs);
    if (GATHER)
    {
        if (lay.stride == 0)
            rec = lpx_slot(rec, fs);  // the arena's table of this frame slot
        else
            rec = (const float4 *)((const char *)rec + (size_t)lpx_blk.z * lay.pitch * lay.stride);  // the caller's frame
        GX = lpx_slot(GX, fs);
        GY = lpx_slot(GY, fs);
        GZ = lpx_slot(GZ, fs);
    }
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    const uint32_t tid = threadIdx.x;
    const uint32_t w = tid / WAVE, lane = tid % WAVE;
    if (!GATHER && (large & SORT_KEYS_BELOW_N) && shift < 32u && n <= (1u << shift))
    {
        // identity pass (SORT_KEYS_BELOW_N above): the tile is copied as it is
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r)
        {
            const uint32_t e = lpx_blk.x * SORT_TILE + r * SORT_THREADS + tid;
            if (e < n)
            {
                keys_out[e] = keys_in[e];
                if (HAS_VALS)
                    vals_out[e] = vals_in ? vals_in[e] : e;
            }
        }
        return;
    }
    const bool prefixed = (large & SORT_PREFIXED) != 0;
    if (prefixed && lpx_blk.x * SORT_TILE >= n)
        return;  // an empty tile (the host launches for its bound of n)
    large &= 1;
    for (int i = tid; i < SORT_WAVES * RADIX; i += SORT_THREADS)
        (&wcnt[0][0])[i] = 0;
    __syncthreads();

    const uint32_t chunk = lpx_blk.x * SORT_TILE + w * (SORT_ITEMS * WAVE);
    const unsigned long long lt = lpx_lanemask_lt();
    KeyT k[SORT_ITEMS];
    uint32_t v[SORT_ITEMS];
    uint32_t loc[SORT_ITEMS];
    // every global load of the block goes out before anything waits: the eight keys and values of the thread here, the
    // rows of the histogram table right below (they do not depend on the keys), the ranking after both
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r)
    {
        const uint32_t e = chunk + r * WAVE + lane;
        const bool valid = e < n;
        k[r] = valid ? keys_in[e] : (KeyT)0;
        if (HAS_VALS)
            v[r] = valid ? (vals_in ? vals_in[e] : e) : 0u;  // (no value array: the values are the positions 0, 1, 2, ...)
    }
    // Global offset of (digit tid, this block) = counts of the lower digits in all blocks + counts of this digit in
    // the lower blocks.  Small tables hold raw counts, block-major: thread tid sums its column (coalesced rows, at
    // most FUSED_SCAN_MAX_BLOCKS of them).  Large tables (hist_rows_kernel) hold per-digit exclusive prefixes over
    // the blocks and, behind the table, the 256 digit totals.
    uint32_t below = 0, t = 0;
    if (large)
    {
        below = offs[tid * nblocks + lpx_blk.x];
        t = offs[RADIX * nblocks + tid];
    }
    else if (prefixed)
    {
        below = offs[lpx_blk.x * RADIX + tid];
        t = offs[nblocks * RADIX + tid];
    }
    else
    {
        // (32 rows per trip: the 61 rows of a 123k-point frame are two round trips instead of eight -- this loop was
        // most of a scatter block's lifetime, and a workgroup that waits holds its wave slots: with twenty chains in
        // flight the device runs out of wave slots, not of bandwidth)
        for (uint32_t b0 = 0; b0 < nblocks; b0 += 32)
        {
            uint32_t c[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                c[u] = (b0 + u < nblocks) ? offs[(b0 + u) * RADIX + tid] : 0u;
#pragma unroll
            for (int u = 0; u < 32; ++u)
            {
                t += c[u];
                below += (b0 + u < lpx_blk.x) ? c[u] : 0u;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r)
    {
        const uint32_t e = chunk + r * WAVE + lane;
        const bool valid = e < n;
        const uint32_t d = (uint32_t)(k[r] >> shift) & (RADIX - 1);
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b)
        {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        if (!valid)
            m = 0;
        const uint32_t rank = __popcll(m & lt);
        uint32_t old = 0;
        if (valid)
            old = wcnt[w][d];
        // every lane of the group has read before the leader writes: one wave, in-order LDS
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0)
            wcnt[w][d] = old + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
        loc[r] = old + rank;
    }
    __syncthreads();
    // Thread tid answers for digit tid: dbase = keys of lower digits in the whole array (exclusive scan of the digit
    // totals over the workgroup), tl = keys of lower digits in THIS tile (the same for the tile's own counts).
    uint32_t ctile = 0;
#pragma unroll
    for (int ww = 0; ww < SORT_WAVES; ++ww)
        ctile += wcnt[ww][tid];
    uint32_t dbase = 0, tl = 0;
    {
        const uint32_t incl = lpx_wave_incl_scan_u32(t), incl2 = lpx_wave_incl_scan_u32(ctile);
        if (lane == WAVE - 1)
        {
            dsum[w] = incl;
            dsum2[w] = incl2;
        }
        __syncthreads();
        for (uint32_t i = 0; i < w; ++i)
        {
            dbase += dsum[i];
            tl += dsum2[i];
        }
        dbase += incl - t;
        tl += incl2 - ctile;
    }
    // The keys are first put in order INSIDE the tile (LDS), then written out position by position: consecutive lanes
    // hold consecutive positions of the tile, and the ~8 keys a digit owns in a tile of 2048 go to consecutive
    // addresses -- one 32-byte request instead of eight.  (Written straight from the registers, lane by lane to ~55
    // different places per store instruction, the kernel was bound by the request rate of its scattered 4-byte stores:
    // 12 % of a launch chain's resident wavefront time for seven passes.)
    {
        uint32_t run = tl;  // digit `tid`: where wavefront ww's keys of the digit start inside the tile
#pragma unroll
        for (int ww = 0; ww < SORT_WAVES; ++ww)
        {
            const uint32_t c = wcnt[ww][tid];
            wcnt[ww][tid] = run;
            run += c;
        }
        gofs[tid] = dbase + below - tl;  // position in the tile -> position in the output, for keys of this digit
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r)
    {
        const uint32_t e = chunk + r * WAVE + lane;
        if (e < n)
        {
            const uint32_t d = (uint32_t)(k[r] >> shift) & (RADIX - 1);
            const uint32_t lp = wcnt[w][d] + loc[r];
            stage_k[lp] = k[r];
            if (HAS_VALS)
                stage_v[lp] = v[r];
        }
    }
    __syncthreads();
    const uint32_t tile_base = lpx_blk.x * SORT_TILE;
    const uint32_t tile_n = tile_base < n ? min(n - tile_base, (uint32_t)SORT_TILE) : 0u;
    if (GATHER)
    {
        uint32_t vv[SORT_ITEMS], dd[SORT_ITEMS];
        float4 q[SORT_ITEMS];
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r)  // the eight record fetches of a thread go out together
        {
            const uint32_t j = r * SORT_THREADS + tid;
            const bool in = j < tile_n;
            vv[r] = in ? stage_v[HAS_VALS ? j : 0] : 0u;
            dd[r] = in ? gofs[(uint32_t)(stage_k[j] >> shift) & (RADIX - 1)] + j : 0u;
            q[r] = lpx_rec_xyz(rec, vv[r], lay);
        }
#pragma unroll
        for (int r = 0; r < SORT_ITEMS; ++r)
            if (r * SORT_THREADS + tid < tile_n)
            {
                vals_out[dd[r]] = vv[r];
                GX[dd[r]] = q[r].x;
                GY[dd[r]] = q[r].y;
                GZ[dd[r]] = q[r].z;
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r)
    {
        const uint32_t j = r * SORT_THREADS + tid;
        if (j < tile_n)
        {
            const KeyT kk = stage_k[j];
            const uint32_t dst = gofs[(uint32_t)(kk >> shift) & (RADIX - 1)] + j;
            keys_out[dst] = kk;
            if (HAS_VALS)
                vals_out[dst] = stage_v[j];
        }
    }
}

// single-block exclusive scan.  One iteration covers SCAN_GROUPS x SCAN_THREADS x 4 = 16 Ki elements
// (the radix histograms of a 120k-point frame) with coalesced 4-element accesses and ONE barrier.
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_GROUPS = 4;
constexpr int SCAN_WAVES = SCAN_THREADS / WAVE;

// in == out is allowed (no __restrict__): every thread reads its inputs before it writes them.
// THREADS: 1024 for a frame alone on the device; 256 inside launch chains -- under load a 1024-thread workgroup waits
// until ONE compute unit has sixteen free wave slots (scan_kernel: 5 us alone, 0.86 ms with twenty chains in flight,
// twice what the other small kernels of a chain wait), a 256-thread workgroup fits anywhere.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void scan_kernel(const uint32_t *in, uint32_t *out, uint32_t n_max,
                                                        const uint32_t *d_n, uint64_t *d_total, size_t fs)
{
    constexpr int SCAN_THREADS = THREADS, SCAN_WAVES = THREADS / WAVE;
    const LpxBlock lpx_blk = lpx_block<1>(fs);
    __shared__ uint32_t wsum[2][SCAN_GROUPS][SCAN_WAVES];
    in = lpx_slot(in, fs);
    out = lpx_slot(out, fs);
    d_n = lpx_slot(d_n, fs);
    d_total = lpx_slot(d_total, fs);
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    const uint32_t tid = threadIdx.x, lane = tid % WAVE, w = tid / WAVE;
    unsigned long long carry = 0;
    uint32_t it = 0;
    for (uint32_t base = 0; base < n; base += SCAN_GROUPS * SCAN_THREADS * 4, ++it)
    {
        uint32_t a[SCAN_GROUPS][4], tsum[SCAN_GROUPS], incl[SCAN_GROUPS];
#pragma unroll
        for (int g = 0; g < SCAN_GROUPS; ++g)
        {
            const uint32_t e = base + g * (SCAN_THREADS * 4) + tid * 4;
            tsum[g] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                a[g][i] = (e + i < n) ? in[e + i] : 0u;
                tsum[g] += a[g][i];
            }
        }
#pragma unroll
        for (int g = 0; g < SCAN_GROUPS; ++g)
        {
            incl[g] = lpx_wave_incl_scan_u32(tsum[g]);
            if (lane == WAVE - 1)
                wsum[it & 1][g][w] = incl[g];
        }
        __syncthreads();
        uint32_t gbase = 0;
#pragma unroll
        for (int g = 0; g < SCAN_GROUPS; ++g)
        {
            uint32_t wbase = 0, tot = 0;
#pragma unroll
            for (int i = 0; i < SCAN_WAVES; ++i)
            {
                const uint32_t s = wsum[it & 1][g][i];
                if (i < (int)w)
                    wbase += s;
                tot += s;
            }
            const uint32_t e = base + g * (SCAN_THREADS * 4) + tid * 4;
            uint32_t run = (uint32_t)carry + gbase + wbase + (incl[g] - tsum[g]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                if (e + i < n)
                    out[e + i] = run;
                run += a[g][i];
            }
            gbase += tot;
        }
        carry += gbase;
    }
    if (tid == 0 && d_total)
        *d_total = carry;
}

// Small radix tables of a launch chain (SORT_PREFIXED): one workgroup per frame, thread d walks column d of the block-major
// counts -- coalesced rows, sixteen in flight -- and leaves the exclusive prefix over the tiles in their place and the
// digit's total in row nblocks.
__global__ __launch_bounds__(RADIX) void hist_cols_kernel(uint32_t *hist, uint32_t nblocks, uint32_t n_max,
                                                           const uint32_t *__restrict__ d_n, uint32_t shift, int flags,
                                                           size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<1>(fs);
    hist = lpx_slot(hist, fs);
    d_n = lpx_slot(d_n, fs);
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    if ((flags & SORT_KEYS_BELOW_N) && shift < 32u && n <= (1u << shift))
        return;  // identity pass: nobody reads the table
    const uint32_t tiles = (n + SORT_TILE - 1) / SORT_TILE;
    const uint32_t nb = tiles < nblocks ? tiles : nblocks;
    const uint32_t tid = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < nb; b0 += 16)
    {
        uint32_t c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            c[u] = (b0 + u < nb) ? hist[(b0 + u) * RADIX + tid] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (b0 + u < nb)
            {
                hist[(b0 + u) * RADIX + tid] = run;
                run += c[u];
            }
    }
    hist[nblocks * RADIX + tid] = run;
}

// Large radix tables (more than FUSED_SCAN_MAX_BLOCKS tiles): one workgroup per digit row turns its nblocks
// counts into exclusive prefixes in place and leaves the row total behind the table; the scatter kernel adds the
// prefix over the 256 totals.  (A single workgroup scanning the whole 256 x nblocks table took 0.32 ms per pass
// on a 5M-point frame.)
__global__ __launch_bounds__(SORT_THREADS) void hist_rows_kernel(uint32_t *hist, uint32_t nblocks, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<1>(fs);
    __shared__ uint32_t wsum[2][SORT_WAVES];
    hist = lpx_slot(hist, fs);
    uint32_t *row = hist + (size_t)lpx_blk.x * nblocks;
    const uint32_t tid = threadIdx.x, lane = tid % WAVE, w = tid / WAVE;
    uint32_t carry = 0, it = 0;
    for (uint32_t base = 0; base < nblocks; base += SORT_THREADS * 4, ++it)
    {
        const uint32_t e = base + tid * 4;
        uint32_t a[4], tsum = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            a[i] = (e + i < nblocks) ? row[e + i] : 0u;
            tsum += a[i];
        }
        const uint32_t incl = lpx_wave_incl_scan_u32(tsum);
        if (lane == WAVE - 1)
            wsum[it & 1][w] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i)
        {
            const uint32_t sv = wsum[it & 1][i];
            if (i < (int)w)
                wbase += sv;
            tot += sv;
        }
        uint32_t run = carry + wbase + (incl - tsum);
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            if (e + i < nblocks)
                row[e + i] = run;
            run += a[i];
        }
        carry += tot;
    }
    if (tid == 0)
        hist[(size_t)RADIX * nblocks + lpx_blk.x] = carry;
}

// Exclusive scan of a long array in three launches: tile sums, single-block scan of the sums, tiles again with
// their bases.  Tile = 1024 threads x 4 elements.
constexpr int XS_TILE = SCAN_THREADS * 4;

__device__ __forceinline__ uint32_t xs_block_excl(uint32_t v, uint32_t *wsum, uint32_t tid, uint32_t &total)
{
    const uint32_t lane = tid % WAVE, w = tid / WAVE;
    const uint32_t incl = lpx_wave_incl_scan_u32(v);
    if (lane == WAVE - 1)
        wsum[w] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_WAVES; ++i)
    {
        const uint32_t sv = wsum[i];
        if (i < (int)w)
            wbase += sv;
        tot += sv;
    }
    total = tot;
    return wbase + incl - v;
}

template <bool APPLY>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const uint32_t *in, uint32_t *out, uint32_t n_max,
                                                                   const uint32_t *d_n, uint32_t *tile_sums, size_t fs)
{
    const LpxBlock lpx_blk = lpx_block<1>(fs);
    __shared__ uint32_t wsum[SCAN_WAVES];
    in = lpx_slot(in, fs);
    out = lpx_slot(out, fs);
    d_n = lpx_slot(d_n, fs);
    tile_sums = lpx_slot(tile_sums, fs);
    const uint32_t n = d_n ? min(*d_n, n_max) : n_max;
    const uint32_t tid = threadIdx.x;
    const uint32_t e = lpx_blk.x * XS_TILE + tid * 4;
    if (lpx_blk.x * (uint32_t)XS_TILE >= n)
    {
        if (!APPLY && tid == 0)
            tile_sums[lpx_blk.x] = 0;
        return;
    }
    uint32_t a[4], tsum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        a[i] = (e + i < n) ? in[e + i] : 0u;
        tsum += a[i];
    }
    uint32_t total;
    const uint32_t excl = xs_block_excl(tsum, wsum, tid, total);
    if (!APPLY)
    {
        if (tid == 0)
            tile_sums[lpx_blk.x] = total;
        return;
    }
    uint32_t run = tile_sums[lpx_blk.x] + excl;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        if (e + i < n)
            out[e + i] = run;
        run += a[i];
    }
}

}  // namespace

static inline uint32_t sort_blocks(uint32_t n)
{
    return n == 0 ? 1u : (n + SORT_TILE - 1) / SORT_TILE;
}

constexpr uint32_t FUSED_SCAN_MAX_BLOCKS = 128;

// hist buffer: bytes [0,8) scan-total scratch, [64,...) the table.  Sized with the frame arena (lpx_ensure_capacity).
static int ensure_hist(lpx_ctx *ctx, uint32_t nblocks)
{
    const size_t need = 64 + (size_t)RADIX * (nblocks + 1) * sizeof(uint32_t);  // + the row of digit totals
    if (ctx->hist.bytes >= need && ctx->hist.p)
        return LPX_OK;
    return lpx_fail(ctx, LPX_ERR_INTERNAL, "histogram table of %u blocks does not fit the workspace", nblocks);
}

int lpx_exclusive_scan(lpx_ctx *ctx, const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *d_n,
                       uint64_t *d_total)
{
    // long arrays: tile sums (scratch: the radix table, free outside a sort) -> their scan -> tiles with bases.
    // Launch chains of many frames take ONE launch up to half a million elements per frame instead (one 256-thread
    // workgroup per frame walks its array in steps of 4096): three launches are three waits for a turn on a loaded
    // device (~0.4 ms each), the walk of a 53k-element array is ~30 us.
    const uint32_t tiles = (n + XS_TILE - 1) / XS_TILE;
    const bool chain = ctx->cur_b > 1;
    if (n > (chain ? 128u : 16u) * XS_TILE && ctx->hist.p && 64 + sizeof(uint32_t) * (size_t)tiles <= ctx->hist.bytes &&
        (const void *)in != (const void *)((char *)ctx->hist.p + 64))
    {
        uint32_t *sums = (uint32_t *)((char *)ctx->hist.p + 64);
        const dim3 grid(tiles, 1, ctx->cur_b);
        hipLaunchKernelGGL(scan_tiles_kernel<false>, grid, dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, d_n, sums,
                           ctx->fs_tag);
        // (the tile sums of a frame are a few hundred words: in a chain a 256-thread workgroup, which a loaded device
        // places at once, where a 1024-thread one waited 4.5 ms per 1M-point chain for a whole compute unit)
        if (chain)
            hipLaunchKernelGGL(scan_kernel<256>, dim3(1, 1, ctx->cur_b), dim3(256), 0, ctx->stream, (const uint32_t *)sums,
                               sums, tiles, (const uint32_t *)nullptr, d_total, ctx->fs_tag);
        else
            hipLaunchKernelGGL(scan_kernel<SCAN_THREADS>, dim3(1, 1, ctx->cur_b), dim3(SCAN_THREADS), 0, ctx->stream,
                               (const uint32_t *)sums, sums, tiles, (const uint32_t *)nullptr, d_total, ctx->fs_tag);
        hipLaunchKernelGGL(scan_tiles_kernel<true>, grid, dim3(SCAN_THREADS), 0, ctx->stream, in, out, n, d_n, sums,
                           ctx->fs_tag);
        LPX_HIP(ctx, hipGetLastError());
        return LPX_OK;
    }
    if (chain)
        hipLaunchKernelGGL(scan_kernel<256>, dim3(1, 1, ctx->cur_b), dim3(256), 0, ctx->stream, in, out, n, d_n, d_total,
                           ctx->fs_tag);
    else
        hipLaunchKernelGGL(scan_kernel<SCAN_THREADS>, dim3(1, 1, ctx->cur_b), dim3(SCAN_THREADS), 0, ctx->stream, in, out, n,
                           d_n, d_total, ctx->fs_tag);
    LPX_HIP(ctx, hipGetLastError());
    return LPX_OK;
}

uint32_t *lpx_sort_first_hist(lpx_ctx *ctx, uint32_t n)
{
    const uint32_t nblocks = sort_blocks(n);
    if (nblocks > FUSED_SCAN_MAX_BLOCKS || ensure_hist(ctx, nblocks) != LPX_OK)
        return nullptr;
    return (uint32_t *)((char *)ctx->hist.p + 64);
}

int lpx_sort_pairs(lpx_ctx *ctx, uint32_t *keys_a, uint32_t *keys_b, uint32_t *vals_a, uint32_t *vals_b, uint32_t n,
                   const uint32_t *d_n, uint32_t bits, uint32_t **keys_out, uint32_t **vals_out, bool first_hist_ready,
                   const LpxSortGather *gather, bool iota_vals, bool keys_below_n)
{
    const uint32_t nblocks = sort_blocks(n);
    int rc = ensure_hist(ctx, nblocks);
    if (rc)
        return rc;
    uint32_t *hist = (uint32_t *)((char *)ctx->hist.p + 64);
    const int large = nblocks > FUSED_SCAN_MAX_BLOCKS;
    // (the identity shortcut needs the per-frame count on the device and the small-table form: hist_rows_kernel would
    // scan a table nobody filled)
    // ... and no gathering last pass: that scatter form has no identity branch, it would rank against a table the
    // histogram pass skipped.  No caller combines the two; the shortcut is simply off then.
    const int below = (keys_below_n && d_n && !large && !gather) ? SORT_KEYS_BELOW_N : 0;
    const int prefixed = (!large && ctx->cur_b > 1) ? SORT_PREFIXED : 0;
    uint32_t *ka = keys_a, *kb = keys_b, *va = vals_a, *vb = vals_b;
    const uint32_t B = ctx->cur_b;
    const size_t fs = ctx->fs_tag;
    for (uint32_t shift = 0; shift < bits; shift += 8)
    {
        if (!(shift == 0 && first_hist_ready && !large))
            hipLaunchKernelGGL((radix_hist_kernel<uint32_t>), dim3(nblocks, 1, B), dim3(SORT_THREADS), 0, ctx->stream, ka,
                               n, d_n, shift, hist, nblocks, (large ? 0 : 1) | below | prefixed, fs);
        if (large)
            hipLaunchKernelGGL(hist_rows_kernel, dim3(RADIX, 1, B), dim3(SORT_THREADS), 0, ctx->stream, hist, nblocks, fs);
        else if (prefixed)
            hipLaunchKernelGGL(hist_cols_kernel, dim3(1, 1, B), dim3(RADIX), 0, ctx->stream, hist, nblocks, n, d_n, shift,
                               below, fs);
        // iota_vals: the values going in are the positions themselves -- the first pass makes them up instead of
        // reading an array somebody had to write first (4 bytes per element written and read back, for nothing)
        const uint32_t *vin = (shift == 0 && iota_vals) ? (const uint32_t *)nullptr : va;
        if (gather && shift + 8 >= bits)  // the last pass also fetches the records the sorted values name
            hipLaunchKernelGGL((radix_scatter_kernel<uint32_t, true, true>), dim3(nblocks, 1, B), dim3(SORT_THREADS), 0,
                               ctx->stream, ka, kb, vin, vb, n, d_n, shift, hist, nblocks, large | prefixed, fs,
                               (const float4 *)gather->records, gather->x, gather->y, gather->z,
                               lpx_rec_layout(gather->records, gather->stride, gather->off, gather->pitch));
        else
            hipLaunchKernelGGL((radix_scatter_kernel<uint32_t, true>), dim3(nblocks, 1, B), dim3(SORT_THREADS), 0,
                               ctx->stream, ka, kb, vin, vb, n, d_n, shift, hist, nblocks, large | below | prefixed, fs);
        uint32_t *t = ka;
        ka = kb;
        kb = t;
        t = va;
        va = vb;
        vb = t;
    }
    LPX_HIP(ctx, hipGetLastError());
    *keys_out = ka;
    *vals_out = va;
    return LPX_OK;
}

int lpx_sort_keys64(lpx_ctx *ctx, uint64_t *keys_a, uint64_t *keys_b, uint32_t n, const uint32_t *d_n, uint32_t bits,
                    uint64_t **keys_out)
{
    const uint32_t nblocks = sort_blocks(n);
    int rc = ensure_hist(ctx, nblocks);
    if (rc)
        return rc;
    uint32_t *hist = (uint32_t *)((char *)ctx->hist.p + 64);
    const int large = nblocks > FUSED_SCAN_MAX_BLOCKS;
    uint64_t *ka = keys_a, *kb = keys_b;
    const uint32_t B = ctx->cur_b;
    const size_t fs = ctx->fs_tag;
    for (uint32_t shift = 0; shift < bits; shift += 8)
    {
        hipLaunchKernelGGL((radix_hist_kernel<uint64_t>), dim3(nblocks, 1, B), dim3(SORT_THREADS), 0, ctx->stream, ka, n,
                           d_n, shift, hist, nblocks, !large, fs);
        if (large)
            hipLaunchKernelGGL(hist_rows_kernel, dim3(RADIX, 1, B), dim3(SORT_THREADS), 0, ctx->stream, hist, nblocks, fs);
        hipLaunchKernelGGL((radix_scatter_kernel<uint64_t, false>), dim3(nblocks, 1, B), dim3(SORT_THREADS), 0,
                           ctx->stream, ka, kb, (const uint32_t *)nullptr, (uint32_t *)nullptr, n, d_n, shift, hist,
                           nblocks, large, fs);
        uint64_t *t = ka;
        ka = kb;
        kb = t;
    }
    LPX_HIP(ctx, hipGetLastError());
    *keys_out = ka;
    return LPX_OK;
}
