// tools/probe/read_bw.hip -- the ceiling of a READ-ONLY stream on MI355X in the access shape of plane_pass_kernel
// (three SoA float arrays, one wavefront per block of CHUNK consecutive points, 16-byte buffer loads, a ring of R quads
// per lane) and in a few other shapes, with a trivial amount of arithmetic per point: what `roofline.plane_passes.frac`
// can be at most in this shape.  Build: hipcc -O3 --offload-arch=gfx950 read_bw.hip -o read_bw; run: ./read_bw [Mpoints]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float V4 __attribute__((ext_vector_type(4)));
typedef unsigned int U4 __attribute__((ext_vector_type(4)));

// plane-pass shape: WAVES wavefronts per block, each its own CHUNK points; ring of R quads per lane
template <int R, int WAVES, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void ring_read(const float *X, const float *Y, const float *Z, uint32_t n,
                                                              uint32_t chunk, float *out)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t base = (blockIdx.x * WAVES + w) * chunk;
    if (base >= n)
        return;
    const uint32_t bytes = 4u * n;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)Y, 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void *)Z, 0, bytes, 0x00020000);
    const uint32_t voff = lane * 16u, G = chunk / 256u;
    V4 x[R], y[R], z[R];
#define LD(i, g)                                                                                                       \
    {                                                                                                                  \
        const uint32_t so = (uint32_t)(g) < G ? 4u * (base + 256u * (uint32_t)(g)) : bytes;                            \
        x[i] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(rx, voff, so, 0));                     \
        y[i] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(ry, voff, so, 0));                     \
        z[i] = __builtin_bit_cast(V4, (U4)__builtin_amdgcn_raw_buffer_load_b128(rz, voff, so, 0));                     \
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
        LD(i, i)
    float acc = 0.0f;
    for (uint32_t g0 = 0; g0 < G; g0 += R)
    {
#pragma unroll
        for (int i = 0; i < R; ++i)
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc += (x[i][e] * 0.5f + y[i][e] * 0.25f) + z[i][e];
            LD(i, g0 + i + R)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef LD
    if (acc == 123.456f)
        out[blockIdx.x] = acc;
}

// the classic: 256 threads, U independent 16-byte loads per thread and array, grid-stride
template <int U>
__global__ __launch_bounds__(256) void stride_read(const V4 *X, const V4 *Y, const V4 *Z, size_t n16, float *out)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += U * stride)
    {
        V4 a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n16)
            {
                a[u] = X[i + u * stride];
                b[u] = Y[i + u * stride];
                c[u] = Z[i + u * stride];
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * stride < n16)
                acc += (a[u].x + b[u].y) + c[u].z;
    }
    if (acc == 123.456f)
        out[blockIdx.x] = acc;
}

template <class F>
static double run(F launch, double bytes, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return bytes * reps / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv)
{
    const uint32_t n = (argc > 1 ? (uint32_t)atoi(argv[1]) : 64u) << 20;  // points per array
    float *X, *Y, *Z, *out;
    hipMalloc(&X, 4ull * n + 64);
    hipMalloc(&Y, 4ull * n + 64);
    hipMalloc(&Z, 4ull * n + 64);
    hipMalloc(&out, 4ull << 20);
    hipMemset(X, 0, 4ull * n);
    hipMemset(Y, 0, 4ull * n);
    hipMemset(Z, 0, 4ull * n);
    const double bytes = 12.0 * n;
    printf("read-only stream of three float arrays, %u M points each (%.0f MB)\n", n >> 20, bytes / 1e6);
#define RING(R, WAVES, MINW, CHUNK)                                                                                    \
    printf("  ring %d quads, %d wave(s) per block, min %d waves/SIMD, %5d points per wave: %.2f TB/s\n", R, WAVES, MINW, \
           CHUNK, run([&] { hipLaunchKernelGGL((ring_read<R, WAVES, MINW>), dim3((n / CHUNK + WAVES - 1) / WAVES),     \
                                               dim3(64 * WAVES), 0, 0, X, Y, Z, n, (uint32_t)CHUNK, out); }, bytes, 10));
    RING(4, 1, 3, 4096)
    RING(4, 1, 4, 4096)
    RING(4, 1, 8, 4096)
    RING(8, 1, 2, 4096)
    RING(8, 1, 3, 4096)
    RING(4, 1, 3, 8192)
    RING(4, 1, 3, 16384)
    RING(8, 1, 2, 16384)
    RING(4, 4, 3, 4096)
    RING(4, 4, 4, 8192)
    RING(2, 1, 8, 4096)
    for (int g : {1024, 2048, 4096, 8192})
        printf("  grid-stride float4 x 3 arrays, 8 loads per thread and array, %d blocks: %.2f TB/s\n", g,
               run([&] { hipLaunchKernelGGL((stride_read<8>), dim3(g), dim3(256), 0, 0, (const V4 *)X, (const V4 *)Y,
                                            (const V4 *)Z, (size_t)n / 4, out); }, bytes, 10));
    return 0;
}
