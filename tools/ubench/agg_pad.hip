// Micro-benchmark (not product code): which part of the marching-ring aggregation pass limits it?
// Variants: 0 = pure streaming copy with the same prefetch structure, 1 = + LDS ring write,
//           2 = + 1 LDS read (arms = 0 equivalent), 3 = full emit (arms from records)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int PF, int MODE, bool VERT>
__global__ __launch_bounds__(64) void k(const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ rec,
                                        int W, int H, int Dp, int L, int per_xcd, int pad)
{
    extern __shared__ float ring_all[];
    const int R = 2 * L + 1;
    const int lane = threadIdx.x;
    float* ring = ring_all + lane;
    const int chunks = Dp >> 6;
    const int N = VERT ? H : W;
    const int nlines = (VERT ? W : H) * chunks;
    const int b = blockIdx.x;
    const int gw = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || gw >= nlines) return;
    const int fixed = gw / chunks, chunk = gw - fixed * chunks;
    const long long pix_step = VERT ? W : 1;
    const long long pix0 = VERT ? fixed : (long long)fixed * W;
    const long long rowpitch = (long long)W * Dp + pad;
    const long long fstep = VERT ? ((long long)W * Dp + pad) : (long long)Dp;
    const float* sp = src + (VERT ? pix0 * Dp : (long long)fixed * rowpitch) + chunk * 64 + lane;
    float* dp = dst + (VERT ? pix0 * Dp : (long long)fixed * rowpitch) + chunk * 64 + lane;
    const uint32_t* rp = rec + (long long)fixed * N;
    float pf[PF]; uint32_t pr[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) { pf[u] = sp[(long long)u * fstep]; pr[u] = rp[u]; }
    int slot = 0;
    int j = 0;
    for (; j + PF <= N; j += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const float v = pf[u]; const uint32_t r = pr[u];
            const int e = min(j + u + PF, N - 1);
            pf[u] = sp[(long long)e * fstep]; pr[u] = rp[e];
            float acc = v;
            if (MODE >= 1) { ring[slot * 64] = v; }
            if (MODE == 2) { acc = ring[slot * 64]; }
            if (MODE == 3) {
                const int a_lo = r & 255u, a_hi = (r >> 8) & 255u;
                int idx = slot - a_lo - a_hi; if (idx < 0) idx += R;   // (entries behind us; same cost as the real thing)
                int n = a_lo + a_hi + 1; acc = 0.f;
                while (n > 0) { float t[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { int s2 = idx + k; if (s2 >= R) s2 -= R; t[k] = ring[s2 * 64]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) if (k < n) acc += t[k];
                    idx += 8; if (idx >= R) idx -= R; n -= 8; }
                acc = acc / (float)(r >> 16);
            }
            if (MODE >= 1) { slot = slot + 1 == R ? 0 : slot + 1; }
            dp[(long long)(j + u) * fstep] = acc;
        }
    }
}

static int g_pad = 0;
template <int PF, int MODE, bool VERT>
float run(const float* a, float* b, const uint32_t* rec, int W, int H, int Dp, int L, int reps)
{
    const long long nlines = (long long)(VERT ? W : H) * (Dp / 64);
    const int per_xcd = (int)((nlines + 7) / 8);
    const size_t lds = (size_t)(2 * L + 1) * 256;
    CK(hipFuncSetAttribute((const void*)&k<PF, MODE, VERT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<PF, MODE, VERT>), dim3(per_xcd * 8), dim3(64), lds, 0, a, b, rec, W, H, Dp, L, per_xcd, g_pad);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<PF, MODE, VERT>), dim3(per_xcd * 8), dim3(64), lds, 0, a, b, rec, W, H, Dp, L, per_xcd, g_pad);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

__global__ void copyk(const float4* a, float4* b, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i]; }

int main(int argc, char** argv)
{
    const int W = 1920, H = 1080, Dp = 128, L = argc > 1 ? atoi(argv[1]) : 34;
    const int arm = argc > 2 ? atoi(argv[2]) : 0;
    g_pad = getenv("PAD") ? atoi(getenv("PAD")) : 0;
    const size_t n = (size_t)(W * Dp + g_pad) * H + 4096;
    float *a, *b; uint32_t* rec;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&rec, (size_t)W * H * 4));
    std::vector<uint32_t> hr((size_t)W * H, (uint32_t)arm | ((uint32_t)arm << 8) | (1u << 16));
    CK(hipMemcpy(rec, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(a, 0, n * 4));
    const double gb = 2.0 * n * 4 / 1e9;
    { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); copyk<<<4096, 256>>>((float4*)a, (float4*)b, n / 4); hipDeviceSynchronize();
      hipEventRecord(e0); for (int i = 0; i < 5; i++) copyk<<<4096, 256>>>((float4*)a, (float4*)b, n / 4); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); printf("float4 copy: %.3f ms  %.0f GB/s\n", ms / 5, gb / (ms / 5) * 1e3); }
#define RUN(PF, MODE, VERT) { float ms = run<PF, MODE, VERT>(a, b, rec, W, H, Dp, L, 5); printf("PF=%2d mode=%d %s L=%d arm=%d: %.3f ms  %.0f GB/s\n", PF, MODE, VERT ? "V" : "H", L, arm, ms, gb / ms * 1e3); }
    RUN(8, 0, false) RUN(8, 2, false) RUN(8, 0, true) RUN(8, 2, true) RUN(16, 2, true)
    return 0;
}
