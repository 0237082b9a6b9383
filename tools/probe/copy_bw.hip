// tools/probe/copy_bw.hip -- which shape of a plain device copy reaches the achievable HBM bandwidth on MI355X
// (MI355X_MICROARCH.md quotes 6.29 TB/s for a V4 copy).  Build: hipcc -O3 --offload-arch=gfx950 copy_bw.hip -o copy_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float V4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_stride(const V4 *__restrict__ src, V4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride)
    {
        V4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (NT)
                __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else
                dst[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride)
        dst[i] = src[i];
}

// every workgroup owns one contiguous tile of U * 256 V4 (no grid stride)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_tile(const V4 *__restrict__ src, V4 *__restrict__ dst, size_t n16)
{
    const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    V4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n16)
            v[u] = NT ? __builtin_nontemporal_load(src + base + u * 256) : src[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n16)
        {
            if (NT)
                __builtin_nontemporal_store(v[u], dst + base + u * 256);
            else
                dst[base + u * 256] = v[u];
        }
}

template <class F>
static double run(F launch, size_t bytes, int reps)
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
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? atol(argv[1]) : 1024) << 20;
    void *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 2, bytes);
    const size_t n16 = bytes / 16;
    const V4 *s = (const V4 *)a;
    V4 *d = (V4 *)b;
    for (int wg : {2048, 4096, 8192, 16384})
    {
        printf("stride U4 plain  wg %5d: %.0f GB/s\n", wg, run([&] { copy_stride<4, false><<<wg, 256>>>(s, d, n16); }, bytes, 10));
        printf("stride U4 nt     wg %5d: %.0f GB/s\n", wg, run([&] { copy_stride<4, true><<<wg, 256>>>(s, d, n16); }, bytes, 10));
        printf("stride U8 nt     wg %5d: %.0f GB/s\n", wg, run([&] { copy_stride<8, true><<<wg, 256>>>(s, d, n16); }, bytes, 10));
    }
    printf("tile U1 plain: %.0f GB/s\n", run([&] { copy_tile<1, false><<<(n16 + 255) / 256, 256>>>(s, d, n16); }, bytes, 10));
    printf("tile U1 nt   : %.0f GB/s\n", run([&] { copy_tile<1, true><<<(n16 + 255) / 256, 256>>>(s, d, n16); }, bytes, 10));
    printf("tile U4 plain: %.0f GB/s\n", run([&] { copy_tile<4, false><<<(n16 + 1023) / 1024, 256>>>(s, d, n16); }, bytes, 10));
    printf("tile U4 nt   : %.0f GB/s\n", run([&] { copy_tile<4, true><<<(n16 + 1023) / 1024, 256>>>(s, d, n16); }, bytes, 10));
    printf("tile U8 nt   : %.0f GB/s\n", run([&] { copy_tile<8, true><<<(n16 + 2047) / 2048, 256>>>(s, d, n16); }, bytes, 10));
    printf("hipMemcpyDtoD: %.0f GB/s\n", run([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes, 10));
    return 0;
}
