// Micro-benchmark (not product code): does a working set that fits the 256 MiB Infinity Cache (die-level L3) stream faster
// than HBM?  (Round-4 review item 6: SURVEY 7.2 / K4 pointed at plane residency -- a slab-major aggregation would run all 8
// passes over 8-16 disparities x all pixels while the slab stays on die.)
//   8 ping-pong passes A -> B -> A ... (read slab + write slab each) over slabs of 8 ... 1062 MB,
//   shape 1: one float4 per thread (the best copy shape of copy_ceiling.hip)
//   shape 2: marching waves (one wave per "line" of 512-byte steps, 8 loads in flight, then 8 stores: K4's access pattern)
//   shape 3: in place (A -> A, element-wise): the footprint is ONE slab
// Prints TB/s of read + written bytes, average over passes 2..8 (pass 1 warms the cache).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l3_resident.hip -o tools/ubench/l3_resident && tools/ubench/l3_resident
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_flat(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 v = src[i]; v.x += 1.0f; dst[i] = v; }
}
__global__ __launch_bounds__(256) void k_inplace(float4* __restrict__ a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { float4 v = a[i]; v.x += 1.0f; a[i] = v; }
}
// wave = one line of `steps` pieces of 512 bytes, `stride` bytes apart (a column of a [rows][lines][128 floats] volume)
__global__ __launch_bounds__(64) void k_march(const float2* __restrict__ src, float2* __restrict__ dst, int lines, int steps, size_t stride2)
{
    const int b = blockIdx.x, per = (lines + 7) / 8, line = (b & 7) * per + (b >> 3); // XCD-aware: XCD x gets a contiguous band
    if ((b >> 3) >= per || line >= lines) return;
    const float2* s = src + (size_t)line * 64 + threadIdx.x;
    float2* d = dst + (size_t)line * 64 + threadIdx.x;
    for (int m = 0; m < steps; m += 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = m + u < steps ? s[(size_t)(m + u) * stride2] : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; u++) if (m + u < steps) { v[u].x += 1.0f; d[(size_t)(m + u) * stride2] = v[u]; }
    }
}

int main()
{
    const size_t full = (size_t)1920 * 1080 * 128 * 4;
    float *a, *b;
    CK(hipMalloc(&a, full));
    CK(hipMalloc(&b, full));
    CK(hipMemset(a, 0, full));
    CK(hipMemset(b, 0, full));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("slab MB | ping-pong flat float4 | ping-pong marching waves | in place flat float4   (TB/s of read + written bytes, passes 2..8)\n");
    const int rows_list[] = {8, 16, 32, 64, 96, 128, 192, 256, 540, 1080};
    for (int rows : rows_list) { // slab = rows x 1920 pixels x 128 floats
        const size_t bytes = (size_t)rows * 1920 * 512, n4 = bytes / 16;
        double rate[3] = {0, 0, 0};
        for (int shape = 0; shape < 3; shape++) {
            for (int rep = 0; rep < 3; rep++) { // best of 3
                float ms = 0.f;
                for (int pass = 0; pass < 8; pass++) {
                    float* s = (pass & 1) ? b : a;
                    float* d = (pass & 1) ? a : b;
                    if (pass == 1) CK(hipEventRecord(e0, 0));
                    if (shape == 0) hipLaunchKernelGGL(k_flat, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)s, (float4*)d, n4);
                    else if (shape == 1) hipLaunchKernelGGL(k_march, dim3((unsigned)(((1920 + 7) / 8) * 8)), dim3(64), 0, 0, (const float2*)s, (float2*)d, 1920, rows, (size_t)1920 * 64);
                    else hipLaunchKernelGGL(k_inplace, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (float4*)a, n4);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double r = 7.0 * 2.0 * bytes / (ms * 1e-3) / 1e12;
                if (r > rate[shape]) rate[shape] = r;
            }
        }
        printf("%7.1f | %21.2f | %24.2f | %21.2f\n", bytes / 1e6, rate[0], rate[1], rate[2]);
    }
    return 0;
}
