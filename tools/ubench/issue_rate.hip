// Micro-benchmark (not product code): cycles per instruction of ONE wave for the instruction patterns the latency-bound
// kernels of this pipeline consist of (K4 register-ring step, K5 scanline step, K11 median level), and how the rate
// scales with 1 / 2 / 4 / 8 waves per SIMD.  s_memtime brackets 64 repetitions of a 32-instruction pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_rate.hip -o tools/ubench/issue_rate && tools/ubench/issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

// each PATTERN is 32 "units"; UNIT_INSTR = instructions per unit
template <int P>
__global__ __launch_bounds__(1024) void k(float* out, long long* cycles, int reps)
{
    float a = threadIdx.x * 0.001f, b = 1.0f, c = 2.0f, d = 3.0f, e = 4.0f, f = 5.0f, g = 6.0f, h = 7.0f;
    int s0 = reps, s1 = 3, s2 = 5;
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int r = 0; r < reps; r++) {
        if constexpr (P == 0) { // dependent v_add_f32 chain
            asm volatile(REP32("v_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(b));
        } else if constexpr (P == 1) { // 8 independent v_add_f32 chains, round robin
            asm volatile(REP4("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                              "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(1.0f));
        } else if constexpr (P == 2) { // dependent s_add_u32 chain
            asm volatile(REP32("s_add_u32 %0, %0, %1\n\t") : "+s"(s1) : "s"(s2) : "scc");
        } else if constexpr (P == 3) { // alternating independent SALU / VALU (two chains)
            asm volatile(REP8("s_add_u32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\ts_add_u32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\t")
                         : "+s"(s1), "+v"(a) : "s"(s2), "v"(b) : "scc");
        } else if constexpr (P == 4) { // v_readlane -> s_add -> v_add with that SGPR -> ... (cross-pipe dependency chain; 4 instr per unit x 8)
            asm volatile(REP8("v_readlane_b32 %1, %0, 3\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %0\n\t")
                         : "+v"(s0), "+s"(s1) : : "scc");
        } else if constexpr (P == 5) { // wave minimum as in k_scanline: 6 dependent DPP mins with s_nop 1 + readlane (14 instr per unit x 2)
            asm volatile(REP4("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                              "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                              "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                              "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a));
        } else if constexpr (P == 6) { // the register-ring push: s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off (3 instr per unit x 10 + 2)
            asm volatile(REP8("s_set_gpr_idx_on %1, gpr_idx(DST)\n\tv_mov_b32 v100, %0\n\ts_set_gpr_idx_off\n\tv_add_f32 %0, %0, %0\n\t")
                         : "+v"(a) : "s"(s1) : "m0", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108");
        } else if constexpr (P == 7) { // dependent v_min3 / v_med3 chain (median kernel)
            asm volatile(REP8("v_min3_f32 %0, %0, %1, %2\n\tv_med3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %1, %2\n\tv_med3_f32 %0, %0, %2, %1\n\t")
                         : "+v"(a) : "v"(b), "v"(c));
        } else if constexpr (P == 8) { // v_cndmask with SGPR-pair mask after v_cmp (select chain)
            asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_gt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t")
                         : "+v"(a) : "v"(b), "v"(c) : "vcc");
        } else if constexpr (P == 9) { // dependent VALU chain with one independent SALU between every two VALU (K4 / K5 mix)
            asm volatile(REP8("v_add_f32 %0, %0, %2\n\ts_add_u32 %1, %1, 1\n\tv_add_f32 %0, %0, %2\n\ts_lshl_b32 %1, %1, 1\n\t")
                         : "+v"(a), "+s"(s1) : "v"(b) : "scc");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + (float)(s0 + s1 + s2);
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int P>
static void run(const char* name, int instr_per_rep)
{
    float* out;
    long long* cyc;
    CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    CK(hipMalloc(&cyc, 256 * 8 * 4 * sizeof(long long)));
    const int reps = 64;
    printf("%-70s", name);
    for (int wps : {1, 2, 4, 8}) { // waves per SIMD: blocks of (wps*4) waves... one block per CU
        const int threads = 64 * 4 * (wps > 4 ? 4 : wps), blocks_per_cu = wps > 4 ? 2 : 1;
        const int blocks = 256 * blocks_per_cu;
        CK(hipMemset(cyc, 0, 256 * 8 * 4 * sizeof(long long)));
        hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        std::vector<long long> hc(blocks * (threads / 64));
        CK(hipMemcpy(hc.data(), cyc, hc.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0;
        for (long long v : hc) avg += (double)v;
        avg /= (double)hc.size();
        printf("  %dw/SIMD: %6.2f cyc/instr/wave", wps, avg / (double)(reps * instr_per_rep));
    }
    printf("\n");
    CK(hipFree(out));
    CK(hipFree(cyc));
}

int main()
{
    run<0>("dependent v_add_f32 chain", 32);
    run<1>("8 independent v_add_f32 chains", 32);
    run<2>("dependent s_add_u32 chain", 32);
    run<3>("alternating SALU / VALU, two independent chains", 32);
    run<4>("v_readlane -> s_add -> v_add(sgpr) -> v_add chain", 32);
    run<5>("DPP min chain with s_nop 1 (wave-min of k_scanline; nops counted)", 32);
    run<6>("gpr-idx push (idx_on, v_mov, idx_off) + dependent v_add", 32);
    run<7>("dependent v_min3 / v_med3 / v_max3 chain", 32);
    run<8>("v_cmp + v_cndmask chain through vcc", 32);
    run<9>("dependent VALU chain with an independent SALU after every VALU", 32);
    return 0;
}
