// Micro-benchmark (not product code): device copy rate of a cost-volume-sized buffer (1 GiB) as a function of the bytes
// per lane and access -- the yardstick the aggregation passes (K4: 8 bytes per lane, 512 bytes per wave) are priced against.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/copy_rate.hip -o tools/ubench/copy_rate && tools/ubench/copy_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) dst[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// a wave streams its own contiguous chunk (like a K4 wave marching along a line): 64 lanes x sizeof(T) per step
template <typename T, int DEPTH>
__global__ __launch_bounds__(256) void k_copy_lines(const T* __restrict__ src, T* __restrict__ dst, size_t n, size_t per_wave)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const size_t b = wave * per_wave, e = b + per_wave < n ? b + per_wave : n;
    for (size_t i = b + lane; i < e; i += 64 * DEPTH) {
        T v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) if (i + u * 64 < e) v[u] = src[i + u * 64];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) if (i + u * 64 < e) dst[i + u * 64] = v[u];
    }
}

template <typename F>
static double timeit(F launch, size_t bytes, int reps = 5)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return 2.0 * bytes / (best * 1e-3) / 1e12;
}

int main()
{
    const size_t bytes = (size_t)1920 * 1080 * 128 * 4; // V of the 1080p, D = 128 cost volume
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    printf("buffer %.3f GB (read + written once per copy)\n", bytes / 1e9);
    printf("hipMemcpyAsync D2D                         %.2f TB/s\n", timeit([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, bytes));
    for (int grid : {2048, 8192, 32768}) {
        printf("grid-stride  4 B/lane x4  grid %6d       %.2f TB/s\n", grid, timeit([&] { hipLaunchKernelGGL((k_copy<float, 4>), dim3(grid), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4); }, bytes));
        printf("grid-stride  8 B/lane x4  grid %6d       %.2f TB/s\n", grid, timeit([&] { hipLaunchKernelGGL((k_copy<float2, 4>), dim3(grid), dim3(256), 0, 0, (const float2*)a, (float2*)b, bytes / 8); }, bytes));
        printf("grid-stride 16 B/lane x4  grid %6d       %.2f TB/s\n", grid, timeit([&] { hipLaunchKernelGGL((k_copy<float4, 4>), dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16); }, bytes));
    }
    // per-wave contiguous streams: 4096 waves (4 per SIMD) and 8192 waves, 8 loads in flight per lane
    for (int waves : {4096, 8192, 16384}) {
        const int grid = waves / 4;
        printf("wave streams  4 B/lane x8  %5d waves       %.2f TB/s\n", waves, timeit([&] { hipLaunchKernelGGL((k_copy_lines<float, 8>), dim3(grid), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4, (bytes / 4 + waves - 1) / waves); }, bytes));
        printf("wave streams  8 B/lane x8  %5d waves       %.2f TB/s\n", waves, timeit([&] { hipLaunchKernelGGL((k_copy_lines<float2, 8>), dim3(grid), dim3(256), 0, 0, (const float2*)a, (float2*)b, bytes / 8, (bytes / 8 + waves - 1) / waves); }, bytes));
        printf("wave streams 16 B/lane x8  %5d waves       %.2f TB/s\n", waves, timeit([&] { hipLaunchKernelGGL((k_copy_lines<float4, 8>), dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16, (bytes / 16 + waves - 1) / waves); }, bytes));
    }
    return 0;
}
