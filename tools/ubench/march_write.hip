// march_write.hip -- what does a WRITE-ONLY (or read-only) marching stream reach, and how does it depend on the step cadence?
// The fused-cost first pass of the aggregation only writes the volume and still takes 0.34 (noise) / 0.41 ms (structured): is that
// the store stream of its access shape, or the instruction stream of a step?  Volume [1080][1920][128] f32; a wave owns the 512
// bytes of a pixel (8 B per lane) and marches along a row piece (rows cut into `segs` pieces: 2 pieces = the 2047-wave launch of
// the register-ring row passes) or down a column.  Per step: `spin` dependent v_add (a stand-in for the work of a step), then the
// store (mode 0), the load (mode 1: 8 in flight) or both (mode 2).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/march_write.hip -o tools/ubench/march_write && tools/ubench/march_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool VERT, int MODE>
__global__ __launch_bounds__(64) void k_march(const float2* __restrict__ src, float2* __restrict__ dst, int W, int H, int segs, int spin)
{
    const int nlines = VERT ? W : H, N = VERT ? H : W;
    const int wave = (int)blockIdx.x;
    if (wave >= nlines * segs) return;
    // XCD-aware like the kernels: block b runs on XCD b % 8 and takes a line of band b % 8
    const int per = (nlines * segs + 7) / 8, idx = (wave & 7) * per + (wave >> 3);
    if (idx >= nlines * segs) return;
    const int line = idx / segs, seg = idx - line * segs;
    const int len = (N + segs - 1) / segs, j0 = seg * len, j1 = j0 + len < N ? j0 + len : N;
    const size_t estep = (VERT ? (size_t)W : 1) * 64;
    const size_t pix0 = VERT ? (size_t)line : (size_t)line * W;
    const float2* sp = src + pix0 * 64 + threadIdx.x;
    float2* dp = dst + pix0 * 64 + threadIdx.x;
    float acc = (float)threadIdx.x;
    float2 v[8];
    for (int j = j0; j < j1; j += 8) {
        if (MODE != 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = sp[(size_t)(j + u < j1 ? j + u : j1 - 1) * estep];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            for (int s = 0; s < spin; s++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(1.0f));
            float2 o = MODE != 0 ? v[u] : make_float2(acc, 2.f);
            if (MODE == 1) { acc += o.x + o.y; continue; }
            if (j + u < j1) dp[(size_t)(j + u) * estep] = o;
        }
    }
    if (MODE == 1 && acc == 12345.678f) dst[0] = make_float2(acc, acc);
}

// mode 3: the copy as the register-ring kernels issue it -- ONE load and ONE store per step, 8 loads in flight, the slot of step s
// re-loaded for step s + 8 right after it was taken over (asm loads, hand-counted wait: at most 14 younger operations outstanding)
template <bool VERT>
__global__ __launch_bounds__(64) void k_march_interleaved(const float2* __restrict__ src, float2* __restrict__ dst, int W, int H, int segs, int spin)
{
    const int nlines = VERT ? W : H, N = VERT ? H : W;
    const int wave = (int)blockIdx.x;
    const int per = (nlines * segs + 7) / 8, idx = (wave & 7) * per + (wave >> 3);
    if (wave >= (nlines * segs + 7) / 8 * 8 || idx >= nlines * segs) return;
    const int line = idx / segs, seg = idx - line * segs;
    const int len = (N + segs - 1) / segs, j0 = seg * len, j1 = j0 + len < N ? j0 + len : N;
    const size_t estep = (VERT ? (size_t)W : 1) * 64;
    const size_t pix0 = VERT ? (size_t)line : (size_t)line * W;
    const float2* sp = src + pix0 * 64 + threadIdx.x + (size_t)j0 * estep;
    float2* dp = dst + pix0 * 64 + threadIdx.x + (size_t)j0 * estep;
    float acc = 0.f;
    float2 pf[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pf[u]) : "v"(sp) : "memory"); sp += estep; }
    int j = j0;
    for (; j + 16 <= j1; j += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float2 v;
            asm volatile("s_waitcnt vmcnt(14)\n\tv_mov_b64 %0, %1" : "=&v"(v) : "v"(pf[u]) : "memory");
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pf[u]) : "v"(sp) : "memory");
            sp += estep;
            for (int s = 0; s < spin; s++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(1.0f));
            v.x += acc;
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dp), "v"(v) : "memory");
            dp += estep;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(pf[5]), "v"(pf[6]), "v"(pf[7]) : "memory");
    // (the last < 16 elements of a piece are left out: 1.5 % of the volume)
}

static hipEvent_t e0, e1;
template <typename F>
static double best_ms(F launch)
{
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 5; r++) {
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
template <bool VERT, int MODE>
static void run(const char* a, char* b, int segs, int spin)
{
    const int W = 1920, H = 1080, waves = (VERT ? W : H) * segs;
    const double ms = best_ms([&] { hipLaunchKernelGGL((k_march<VERT, MODE>), dim3((waves + 7) / 8 * 8), dim3(64), 0, 0, (const float2*)a, (float2*)b, W, H, segs, spin); });
    const double bytes = (MODE == 2 ? 2.0 : 1.0) * W * H * 512;
    printf("%-10s %-7s %d piece(s) per line (%4d waves)  %3d dependent adds per step   %.3f ms  %.2f TB/s\n", MODE == 0 ? "write-only" : (MODE == 1 ? "read-only" : "copy"),
           VERT ? "columns" : "rows", segs, waves, spin, ms, bytes / (ms * 1e-3) / 1e12);
}

template <bool VERT>
static void run_il(const char* a, char* b, int segs, int spin)
{
    const int W = 1920, H = 1080, waves = (VERT ? W : H) * segs;
    const double ms = best_ms([&] { hipLaunchKernelGGL((k_march_interleaved<VERT>), dim3((waves + 7) / 8 * 8), dim3(64), 0, 0, (const float2*)a, (float2*)b, W, H, segs, spin); });
    printf("%-10s %-7s %d piece(s) per line (%4d waves)  %3d dependent adds per step   %.3f ms  %.2f TB/s   (1 load + 1 store per step, 8 in flight)\n", "copy", VERT ? "columns" : "rows", segs, waves, spin, ms,
           2.0 * W * H * 512 / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)1920 * 1080 * 128 * 4;
    char *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int spin : {0, 16, 32, 64, 96}) {
        run<false, 0>(a, b, 1, spin);
        run<false, 0>(a, b, 2, spin);
        run<false, 0>(a, b, 4, spin);
        run<true, 0>(a, b, 1, spin);
        run<true, 0>(a, b, 2, spin);
    }
    for (int spin : {0, 32, 64}) {
        run<false, 1>(a, b, 1, spin);
        run<false, 1>(a, b, 2, spin);
        run<true, 1>(a, b, 1, spin);
        run<false, 2>(a, b, 2, spin);
        run<true, 2>(a, b, 1, spin);
    }
    for (int spin : {0, 8, 16, 24, 32}) {
        run<false, 2>(a, b, 2, spin);
        run_il<false>(a, b, 2, spin);
        run<true, 2>(a, b, 1, spin);
        run_il<true>(a, b, 1, spin);
    }
    return 0;
}
