// Micro-benchmark (not product code): the HBM yardstick of the box, settled on ONE box in one run (round-3 review item 4).
//   (a) the guide's shape: float4 grid-stride copy (MI355X_MICROARCH.md: 6.29 TB/s), several grids / unrolls, plain and
//       non-temporal (streaming) loads and stores;
//   (b) read-only and write-only streams of the same buffer (what a copy could reach if the two directions did not interact);
//   (c) hipMemcpyAsync D2D (what bench.py used to report as device_copy_GBps);
//   (d) the marching shapes of K4 / K5: one wave per image line, 8 bytes per lane, 8 loads in flight, XCD-aware mapping --
//       column march (step = one image row) and row march (step = one pixel), and the same with 2 / 4 ADJACENT lines per
//       workgroup so that consecutive waves of a workgroup touch consecutive 512-byte pieces (same DRAM page);
// on the 1080p / D = 128 volume (1.06 GB).  Prints TB/s of read + written bytes (read-only / write-only: of the one direction).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/copy_ceiling.hip -o tools/ubench/copy_ceiling && tools/ubench/copy_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float vf4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ src4, float4* __restrict__ dst4, size_t n)
{
    const vf4* __restrict__ src = reinterpret_cast<const vf4*>(src4);
    vf4* __restrict__ dst = reinterpret_cast<vf4*>(dst4);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        vf4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
// one float4 per thread, no loop (the simplest "float4 copy")
__global__ __launch_bounds__(256) void k_copy4_flat(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read4(const float4* __restrict__ src, float* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc; // (never true: keeps the loads)
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_write4(float4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// Marching shapes.  Volume [H][W][128] f32; a wave owns the 512 bytes of a pixel (8 bytes per lane) and marches along a line;
// a workgroup = LPW waves on LPW ADJACENT lines (for a column march: adjacent columns = consecutive 512-byte pieces of an
// image row; for a row march: adjacent rows, 983 KB apart).  XCD-aware: block b runs on XCD b % 8 and takes the lines of band
// b % 8 (k_aggregate*.h's mapping).  DEPTH loads in flight, issued as a batch, then stored.
template <int DEPTH, bool VERT, int LPW>
__global__ __launch_bounds__(64 * LPW) void k_march(const float2* __restrict__ src, float2* __restrict__ dst, int W, int H, int per_xcd)
{
    extern __shared__ float lds_dummy[];
    const int b = (int)blockIdx.x;
    const int grp = (b & 7) * per_xcd + (b >> 3);
    const int nlines = VERT ? W : H;
    const int line = grp * LPW + (int)(threadIdx.x >> 6);
    if ((b >> 3) >= per_xcd || line >= nlines) return;
    const int N = VERT ? H : W;
    const size_t estep = (VERT ? (size_t)W : 1) * 64; // float2 elements per step
    const size_t pix0 = VERT ? (size_t)line : (size_t)line * W;
    const float2* sp = src + pix0 * 64 + (threadIdx.x & 63);
    float2* dp = dst + pix0 * 64 + (threadIdx.x & 63);
    for (int j = 0; j < N; j += DEPTH) {
        float2 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) v[u] = sp[(size_t)(j + u < N ? j + u : N - 1) * estep];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) if (j + u < N) dp[(size_t)(j + u) * estep] = v[u];
    }
}

static hipEvent_t e0, e1;
template <typename F>
static double best_ms(F launch, int reps = 6)
{
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
    return best;
}
template <int DEPTH, bool VERT, int LPW>
static void run_march(const char* a, char* b, int waves_per_cu)
{
    const int W = 1920, H = 1080;
    const int nlines = VERT ? W : H, groups = (nlines + LPW - 1) / LPW, per_xcd = (groups + 7) / 8;
    const size_t lds = ((size_t)(160 * 1024 / waves_per_cu) * LPW) & ~(size_t)511;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_march<DEPTH, VERT, LPW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double ms = best_ms([&] { hipLaunchKernelGGL((k_march<DEPTH, VERT, LPW>), dim3(per_xcd * 8), dim3(64 * LPW), lds > 160 * 1024 ? 160 * 1024 : lds, 0, (const float2*)a, (float2*)b, W, H, per_xcd); });
    printf("march %s  8 B/lane depth %2d  %d line(s)/workgroup  %2d waves/CU   %.3f ms  %.2f TB/s\n", VERT ? "columns" : "rows   ", DEPTH, LPW,
           waves_per_cu, ms, 2.0 * W * H * 512 / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)1920 * 1080 * 128 * 4, n4 = bytes / 16;
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs, buffer %.3f GB\n", prop.name, prop.multiProcessorCount, bytes / 1e9);
    auto tb = [&](double ms, double dirs) { return dirs * bytes / (ms * 1e-3) / 1e12; };
    double best_copy = 0;
    {
        const double ms = best_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpyAsync D2D                                   %.3f ms  %.2f TB/s\n", ms, tb(ms, 2));
    }
    {
        const double ms = best_ms([&] { hipLaunchKernelGGL(k_copy4_flat, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
        printf("float4 copy, one element per thread (%zu blocks)   %.3f ms  %.2f TB/s\n", (n4 + 255) / 256, ms, tb(ms, 2));
        if (tb(ms, 2) > best_copy) best_copy = tb(ms, 2);
    }
    for (int grid : {1024, 2048, 4096, 8192, 16384, 32768, 65536}) {
#define ROW(U, NT)                                                                                                                   \
    {                                                                                                                                \
        const double ms = best_ms([&] { hipLaunchKernelGGL((k_copy4<U, NT>), dim3(grid), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); }); \
        printf("float4 grid-stride copy  grid %6d unroll %d %s   %.3f ms  %.2f TB/s\n", grid, U, NT ? "non-temporal" : "plain       ", ms, tb(ms, 2)); \
        if (tb(ms, 2) > best_copy) best_copy = tb(ms, 2);                                                                            \
    }
        ROW(1, false) ROW(2, false) ROW(4, false) ROW(4, true)
#undef ROW
    }
    for (int grid : {2048, 8192, 32768}) {
        const double mr = best_ms([&] { hipLaunchKernelGGL((k_read4<4>), dim3(grid), dim3(256), 0, 0, (const float4*)a, (float*)b, n4); });
        const double mw = best_ms([&] { hipLaunchKernelGGL((k_write4<4>), dim3(grid), dim3(256), 0, 0, (float4*)b, n4); });
        printf("read-only stream  grid %6d   %.3f ms  %.2f TB/s   |   write-only stream   %.3f ms  %.2f TB/s\n", grid, mr, tb(mr, 1), mw, tb(mw, 1));
    }
    printf("best copy kernel of this run: %.2f TB/s (guide: 6.29; spec peak 8.0)\n", best_copy);
    // the marching shapes of K4 / K5 (what a plain copy reaches with their access pattern)
    run_march<8, true, 1>(a, b, 8);
    run_march<8, true, 1>(a, b, 16);
    run_march<8, true, 2>(a, b, 8);
    run_march<8, true, 4>(a, b, 8);
    run_march<8, true, 4>(a, b, 16);
    run_march<16, true, 4>(a, b, 8);
    run_march<8, false, 1>(a, b, 8);
    run_march<8, false, 2>(a, b, 8);
    run_march<8, false, 4>(a, b, 8);
    run_march<16, false, 1>(a, b, 8);
    return 0;
}
