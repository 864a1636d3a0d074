// Micro-benchmark (not product code): lean marching-ring step, VALU-side addressing, mirrored ring.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// LDS layout per wave: ring of R entries (256 B each) + 7 mirror entries + 1 dummy entry
template <int PF, bool VERT, bool DIVIDE, int SUMMODE>
__global__ __launch_bounds__(64) void k(const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ rec,
                                        int W, int H, int Dp, int L, int per_xcd)
{
    extern __shared__ float ring_all[];
    const int R = 2 * L + 1;
    const int lane = threadIdx.x;
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
    int vzero; asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const float* sp = src + pix0 * Dp + chunk * 64 + lane;     // next element to prefetch
    float* dp = dst + pix0 * Dp + chunk * 64 + lane;           // next output
    const uint32_t* rp = rec + (long long)fixed * N + vzero;
    // LDS byte offsets (VGPR) relative to the dynamic LDS base
    const unsigned RB = (unsigned)R * 256u;
    unsigned pw = lane * 4u;  // write position (entry slot_w)
    unsigned pm = lane * 4u;  // position of entry m (no halo in this microbench: window = entries behind)
    char* lds = (char*)ring_all;
    float pf[PF]; uint32_t pr[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) { pf[u] = sp[(long long)u * fstep]; pr[u] = rp[u]; }
    sp += (long long)PF * fstep; rp += PF;
    for (int j = 0; j + 2 * PF <= N; j += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const float v = pf[u]; const uint32_t rv = pr[u];
            pf[u] = *sp; sp += fstep; pr[u] = rp[u];
            // push
            *(float*)(lds + pw) = v;
            { const unsigned mir = pw < 7u * 256u ? pw + RB : (RB + 7u * 256u + lane * 4u); *(float*)(lds + mir) = v; }
            pw += 256u; pw = pw >= RB ? pw - RB : pw;
            // emit
            const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)rv);
            const int a_lo = r & 255u, a_hi = (r >> 8) & 255u;
            int cnt = a_lo + a_hi + 1;
            unsigned p = pm + RB - (unsigned)(a_lo + a_hi) * 256u;   // entries behind (microbench): start = m - (lo+hi)
            p = p >= RB ? p - RB : p;
            float acc = 0.f;
            if (SUMMODE == 0) {
                while (cnt >= 8) {
                    const float* q = (const float*)(lds + p);
                    const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320], t6 = q[384], t7 = q[448];
                    acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
                    p += 2048u; p = p >= RB ? p - RB : p; cnt -= 8;
                }
                if (cnt > 0) {
                    const float* q = (const float*)(lds + p);
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
            } else { // VALU-masked partial chunk
                int c2 = cnt;
                while (c2 > 0) {
                    const float* q = (const float*)(lds + p);
                    float t[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) t[k] = q[k * 64];
                    if (c2 >= 8) {
#pragma unroll
                        for (int k = 0; k < 8; k++) acc += t[k];
                    } else {
                        const int cv = c2 + vzero;
#pragma unroll
                        for (int k = 0; k < 8; k++) acc += (k < cv ? t[k] : 0.0f);
                    }
                    p += 2048u; p = p >= RB ? p - RB : p; c2 -= 8;
                }
            }
            if (DIVIDE) { const uint32_t c = r >> 16; if (c != 1u) acc = acc / (float)c; }
            pm += 256u; pm = pm >= RB ? pm - RB : pm;
            *dp = acc; dp += fstep;
        }
        rp += PF;
    }
}

template <int PF, bool VERT, bool DIVIDE, int SUMMODE>
float run(const float* a, float* b, const uint32_t* rec, int W, int H, int Dp, int L, int reps)
{
    const long long nlines = (long long)(VERT ? W : H) * (Dp / 64);
    const int per_xcd = (int)((nlines + 7) / 8);
    const size_t lds = (size_t)(2 * L + 1 + 8) * 256;
    CK(hipFuncSetAttribute((const void*)&k<PF, VERT, DIVIDE, SUMMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<PF, VERT, DIVIDE, SUMMODE>), dim3(per_xcd * 8), dim3(64), lds, 0, a, b, rec, W, H, Dp, L, per_xcd);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<PF, VERT, DIVIDE, SUMMODE>), dim3(per_xcd * 8), dim3(64), lds, 0, a, b, rec, W, H, Dp, L, per_xcd);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int W = 1920, H = 1080, Dp = 128, L = 34;
    const size_t n = (size_t)W * H * Dp;
    float *a, *b; uint32_t* rec;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&rec, (size_t)W * H * 4));
    CK(hipMemset(a, 0, n * 4));
    const double gb = 2.0 * n * 4 / 1e9;
    for (int arm : {0, 3, 6, 12}) {
        const uint32_t cntv = arm == 0 ? 1u : 37u;
        std::vector<uint32_t> hr((size_t)W * H, (uint32_t)arm | ((uint32_t)arm << 8) | (cntv << 16));
        CK(hipMemcpy(rec, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
#define RUN(PF, VERT, DIV, SM) { float ms = run<PF, VERT, DIV, SM>(a, b, rec, W, H, Dp, L, 5); printf("arm=%2d PF=%2d %s div=%d sum=%d: %.3f ms  %.0f GB/s\n", arm, PF, VERT ? "V" : "H", DIV, SM, ms, gb / ms * 1e3); }
        RUN(8, false, false, 0) RUN(8, false, true, 0) RUN(8, false, false, 1) RUN(8, true, false, 0) RUN(8, true, true, 0) RUN(12, true, true, 0)
    }
    return 0;
}
