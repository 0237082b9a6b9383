// lpx_grid.hip -- exact connected components of the d-graph from a grid of clique cells (search mode, frames below
// 400k obstacle points): what makes "one sequencer per component" possible without any radius list.
// (The partition itself still follows src/clustering.cpp:69-124 in lpx_cluster.hip; a BFS never leaves its component.)
#include "lpx_kd_shared.h"

#include <string.h>
#include <stdlib.h>

namespace
{
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

}  // namespace

#ifdef LPX_DEV_KNOBS
// development build only: the components from a sweep over the cloud's own order (measured, not adopted) live outside
// the product sources
#include "../../experiments/sweep_components.inc"
#endif

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


