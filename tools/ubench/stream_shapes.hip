// Micro-benchmark (not product code): what a plain COPY reaches when it is issued with the access shapes of the
// aggregation passes (K4) on the 1080p / D = 128 volume [1080][1920][128] f32 -- the ceiling each pass shape is priced
// against.  One wave per workgroup, XCD-aware block -> line mapping as in k_aggregate*.h, occupancy limited through the
// dynamic-LDS size (waves per CU = 160 KiB / lds).
//   H pass: a wave marches along a row (step = one pixel = 512 B), VPL floats per lane (VPL = 1: 256 of the 512 bytes)
//   V pass: a wave marches down a column (step = one row = 983 040 B)
//   DEPTH loads in flight per wave (issued as a batch, then stored)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int DEPTH, bool VERT>
__global__ __launch_bounds__(64) void k_march(const T* __restrict__ src, T* __restrict__ dst, int W, int H, int chunks, int nseg, int seg_len, int per_xcd)
{
    extern __shared__ float lds_dummy[];
    const int b = (int)blockIdx.x;
    const int gw = (b & 7) * per_xcd + (b >> 3);
    const int nlines = (VERT ? W : H) * chunks;
    if ((b >> 3) >= per_xcd || gw >= nlines * nseg) return;
    const int seg = gw / nlines, line = gw - seg * nlines;
    const int fixed = line / chunks, chunk = line - fixed * chunks;
    const int N = VERT ? H : W;
    const int s0 = seg * seg_len, s1 = s0 + seg_len < N ? s0 + seg_len : N;
    const size_t pix_step = VERT ? (size_t)W : 1;
    const size_t pix0 = VERT ? (size_t)fixed : (size_t)fixed * W;
    const size_t estep = pix_step * chunks * 64; // elements of T per step
    const T* sp = src + pix0 * chunks * 64 + chunk * 64 + threadIdx.x;
    T* dp = dst + pix0 * chunks * 64 + chunk * 64 + threadIdx.x;

    for (int j = s0; j < s1; j += DEPTH) {
        T v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) v[u] = sp[(size_t)(j + u < s1 ? j + u : s1 - 1) * estep];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) if (j + u < s1) dp[(size_t)(j + u) * estep] = v[u];
    }
}

static hipEvent_t e0, e1;
template <typename T, int DEPTH, bool VERT>
static void run(const char* a, char* b, int nseg, int waves_per_cu)
{
    const int W = 1920, H = 1080, Dp = 128;
    const int vpl = sizeof(T) / 4;
    const int chunks = Dp / (64 * vpl);
    const int N = VERT ? H : W;
    const int seg_len = (N + nseg - 1) / nseg;
    const long long waves = (long long)(VERT ? W : H) * chunks * nseg;
    const int per_xcd = (int)((waves + 7) / 8);
    const size_t lds = (size_t)(160 * 1024 / waves_per_cu) & ~(size_t)511;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_march<T, DEPTH, VERT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    double best = 1e30;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_march<T, DEPTH, VERT>), dim3(per_xcd * 8), dim3(64), lds, 0, (const T*)a, (T*)b, W, H, chunks, nseg, seg_len, per_xcd);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = 2.0 * W * H * Dp * 4;
    printf("%s VPL=%d depth %2d nseg %2d (%6lld waves, %2d per CU)   %.3f ms  %.2f TB/s\n", VERT ? "V" : "H", vpl, DEPTH, nseg, waves, waves_per_cu, best, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)1920 * 1080 * 128 * 4;
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // today's shapes: one float per lane, 16 waves per CU
    run<float, 8, false>(a, b, 2, 16);
    run<float, 8, false>(a, b, 5, 16);
    run<float, 8, true>(a, b, 1, 16);
    run<float, 8, true>(a, b, 2, 16);
    run<float, 16, true>(a, b, 1, 16);
    // two floats per lane, 8 waves per CU (register ring of 138 VGPRs)
    run<float2, 8, false>(a, b, 1, 8);
    run<float2, 8, false>(a, b, 2, 8);
    run<float2, 8, false>(a, b, 3, 8);
    run<float2, 8, false>(a, b, 5, 8);
    run<float2, 16, false>(a, b, 2, 8);
    run<float2, 16, false>(a, b, 5, 8);
    run<float2, 8, true>(a, b, 1, 8);
    run<float2, 12, true>(a, b, 1, 8);
    run<float2, 16, true>(a, b, 1, 8);
    run<float2, 16, true>(a, b, 2, 8);
    run<float2, 24, true>(a, b, 1, 8);
    // two floats per lane if 12 / 16 waves per CU were possible
    run<float2, 8, false>(a, b, 3, 12);
    run<float2, 8, true>(a, b, 1, 12);
    run<float2, 8, false>(a, b, 4, 16);
    run<float2, 8, true>(a, b, 2, 16);
    // four floats per lane (one wave = two pixels' worth is not possible; shown for the 16 B/lane shape): skipped
    return 0;
}
