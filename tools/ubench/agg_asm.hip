// Micro-benchmark (not product code): lean marching-ring step with asm-issued prefetch loads and
// hand-counted s_waitcnt vmcnt (the compiler's loop-carried vmcnt model drains the queue otherwise).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float run_sum(float acc, const float* q, int cnt)
{
    while (cnt >= 8) {
        const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320], t6 = q[384], t7 = q[448];
        acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
        q += 512; cnt -= 8;
    }
    if (cnt > 0) {
        const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320], t6 = q[384];
        switch (cnt) {
        case 7: acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; break;
        case 6: acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; break;
        case 5: acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; break;
        case 4: acc += t0; acc += t1; acc += t2; acc += t3; break;
        case 3: acc += t0; acc += t1; acc += t2; break;
        case 2: acc += t0; acc += t1; break;
        default: acc += t0; break;
        }
    }
    return acc;
}

template <int PF, bool VERT, bool DIVIDE>
__global__ __launch_bounds__(64) void k(const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ rec,
                                        int W, int H, int Dp, int L, int per_xcd)
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
    const long long fstep = pix_step * Dp;
    const float* sp = src + pix0 * Dp + chunk * 64 + lane;
    float* dp = dst + pix0 * Dp + chunk * 64 + lane;
    const uint32_t* rp = rec + (long long)fixed * N;
    float pf[PF]; uint32_t pr[PF];
    // prologue: PF asm loads (2 VMEM ops each)
#pragma unroll
    for (int u = 0; u < PF; u++) {
        asm volatile("global_load_dword %0, %1, off" : "=v"(pf[u]) : "v"(sp) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "=v"(pr[u]) : "v"(rp) : "memory");
        sp += fstep; rp += 1;
    }
    int slot_w = 0, slot_m = 0;
    for (int j = 0; j + 2 * PF <= N; j += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            // consume slot u: every VMEM op older than the (PF-1) younger steps x 3 ops must have landed
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(pf[u]), "+v"(pr[u]) : "n"(3 * (PF - 1)) : "memory");
            const float v = pf[u];
            const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)pr[u]);
            asm volatile("global_load_dword %0, %1, off" : "=v"(pf[u]) : "v"(sp) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(pr[u]) : "v"(rp) : "memory");
            sp += fstep; rp += 1;
            ring[slot_w * 64] = v;
            slot_w = slot_w + 1 == R ? 0 : slot_w + 1;
            const int a_lo = r & 255u, a_hi = (r >> 8) & 255u;
            const int n = a_lo + a_hi + 1;
            int idx = slot_m - (a_lo + a_hi); // microbench: window = entries behind the newest
            if (idx < 0) idx += R;
            const int n1 = min(n, R - idx);
            float acc = run_sum(0.0f, ring + idx * 64, n1);
            if (n > n1) acc = run_sum(acc, ring, n - n1);
            if (DIVIDE) { const uint32_t c = r >> 16; if (c != 1u) acc = acc / (float)c; }
            slot_m = slot_m + 1 == R ? 0 : slot_m + 1;
            *dp = acc; dp += fstep;   // the one compiler-issued VMEM op per step
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PF, bool VERT, bool DIVIDE>
float run(const float* a, float* b, const uint32_t* rec, int W, int H, int Dp, int L, int reps)
{
    const long long nlines = (long long)(VERT ? W : H) * (Dp / 64);
    const int per_xcd = (int)((nlines + 7) / 8);
    const size_t lds = (size_t)(2 * L + 1) * 256 * (getenv("LDSX") ? atoi(getenv("LDSX")) : 1);
    CK(hipFuncSetAttribute((const void*)&k<PF, VERT, DIVIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<PF, VERT, DIVIDE>), dim3(per_xcd * 8), dim3(64), lds, 0, a, b, rec, W, H, Dp, L, per_xcd);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<PF, VERT, DIVIDE>), dim3(per_xcd * 8), dim3(64), lds, 0, a, b, rec, W, H, Dp, L, per_xcd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int W = 1920, H = 1080, Dp = 128, L = 34;
    const size_t n = (size_t)W * H * Dp;
    float *a, *b; uint32_t* rec;
    CK(hipMalloc(&a, n * 4 + (1 << 20))); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&rec, (size_t)W * H * 4 + (1 << 20)));
    CK(hipMemset(a, 0, n * 4));
    const double gb = 2.0 * n * 4 / 1e9;
    for (int arm : {0, 3, 6, 12}) {
        const uint32_t cntv = arm == 0 ? 1u : 37u;
        std::vector<uint32_t> hr((size_t)W * H, (uint32_t)arm | ((uint32_t)arm << 8) | (cntv << 16));
        CK(hipMemcpy(rec, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
#define RUN(PF, VERT, DIV) { float ms = run<PF, VERT, DIV>(a, b, rec, W, H, Dp, L, 5); printf("arm=%2d PF=%2d %s div=%d: %.3f ms  %.0f GB/s\n", arm, PF, VERT ? "V" : "H", DIV, ms, gb / ms * 1e3); }
        RUN(8, false, false) RUN(8, false, true) RUN(16, false, true) RUN(8, true, false) RUN(8, true, true) RUN(16, true, true) RUN(20, true, true)
    }
    return 0;
}
