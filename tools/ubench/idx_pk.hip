// Micro-test (not product code): does the VGPR index mode (s_set_gpr_idx_on) apply to the 64-bit operands of
// v_pk_add_f32 (VOP3P) and v_mov_b64 on gfx950?  The two-disparities-per-lane register ring of the aggregation pass
// (k_aggregate_rr2.h) keeps ring slot s in the even-aligned VGPR pair v[V0+2s : V0+2s+1] and wants ONE packed add per
// ring entry (M0 = 2 * slot).  Prints PASS / FAIL per primitive:
//   push32   two v_mov_b32 under gpr_idx(DST)                (known to work: same as the one-float ring)
//   push64   one v_mov_b64 under gpr_idx(DST)
//   pkadd    v_pk_add_f32 acc, v[pair], acc under gpr_idx(SRC0), entered through a computed jump (8 bytes per add)
//   add32x2  two v_add_f32 per entry under gpr_idx(SRC0)      (fallback form)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
#define RING_CLOBBERS "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159"

// MODE: 0 = push32 + add32x2, 1 = push64 + add32x2, 2 = push32 + pkadd, 3 = push64 + pkadd
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(96))) void k(const f2* __restrict__ src, f2* __restrict__ dst, int n, int R, int span)
{
    const int lane = threadIdx.x;
    const f2* sp = src + (size_t)blockIdx.x * n * 64 + lane;
    f2* dp = dst + (size_t)blockIdx.x * n * 64 + lane;
    int slot = 0;
    for (int i = 0; i < n; i++) {
        const f2 v = sp[(size_t)i * 64];
        const int s2 = __builtin_amdgcn_readfirstlane(2 * slot);
        if constexpr (MODE & 1)
            asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b64 v[96:97], %1\n\ts_set_gpr_idx_off" ::"s"(s2), "v"(v) : "m0", RING_CLOBBERS);
        else
            asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v96, %1\n\tv_mov_b32 v97, %2\n\ts_set_gpr_idx_off" ::"s"(s2), "v"(v.x), "v"(v.y) : "m0", RING_CLOBBERS);
        // sum of the last c = min(span, i + 1) entries in the order oldest .. newest (no wrap handling: two runs)
        const int c = span < i + 1 ? span : i + 1;
        int idx = slot + 1 - c; // first slot of the span (may be negative: wrapped)
        f2 acc = {0.0f, 0.0f};
        for (int part = 0; part < 2; part++) {
            int b, cnt;
            if (idx < 0) { b = part == 0 ? idx + R : 0; cnt = part == 0 ? -idx : c + idx; }
            else { b = idx; cnt = part == 0 ? c : 0; }
            if (cnt <= 0) continue;
            // block of 8 entries entered late (cnt <= 8 per run here: loop)
            while (cnt > 0) {
                const int cc = cnt < 8 ? cnt : 8;
                const int m = __builtin_amdgcn_readfirstlane(2 * (b + cc));
                if constexpr (MODE & 2) {
                    const int off = __builtin_amdgcn_readfirstlane(12 + 8 * (8 - cc));
                    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                                 "s_getpc_b64 vcc\n\t"
                                 "s_add_u32 vcc_lo, vcc_lo, %2\n\t"
                                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                                 "s_setpc_b64 vcc\n\t"
                                 "v_pk_add_f32 %0, v[80:81], %0\n\t"
                                 "v_pk_add_f32 %0, v[82:83], %0\n\t"
                                 "v_pk_add_f32 %0, v[84:85], %0\n\t"
                                 "v_pk_add_f32 %0, v[86:87], %0\n\t"
                                 "v_pk_add_f32 %0, v[88:89], %0\n\t"
                                 "v_pk_add_f32 %0, v[90:91], %0\n\t"
                                 "v_pk_add_f32 %0, v[92:93], %0\n\t"
                                 "v_pk_add_f32 %0, v[94:95], %0\n\t"
                                 "s_set_gpr_idx_off"
                                 : "+v"(acc) : "s"(m), "s"(off) : "m0", "scc", "vcc", RING_CLOBBERS);
                } else {
                    const int off = __builtin_amdgcn_readfirstlane(12 + 8 * (8 - cc));
                    float ax = acc.x, ay = acc.y;
                    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\t"
                                 "s_getpc_b64 vcc\n\t"
                                 "s_add_u32 vcc_lo, vcc_lo, %3\n\t"
                                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                                 "s_setpc_b64 vcc\n\t"
                                 "v_add_f32_e32 %0, v80, %0\n\tv_add_f32_e32 %1, v81, %1\n\t"
                                 "v_add_f32_e32 %0, v82, %0\n\tv_add_f32_e32 %1, v83, %1\n\t"
                                 "v_add_f32_e32 %0, v84, %0\n\tv_add_f32_e32 %1, v85, %1\n\t"
                                 "v_add_f32_e32 %0, v86, %0\n\tv_add_f32_e32 %1, v87, %1\n\t"
                                 "v_add_f32_e32 %0, v88, %0\n\tv_add_f32_e32 %1, v89, %1\n\t"
                                 "v_add_f32_e32 %0, v90, %0\n\tv_add_f32_e32 %1, v91, %1\n\t"
                                 "v_add_f32_e32 %0, v92, %0\n\tv_add_f32_e32 %1, v93, %1\n\t"
                                 "v_add_f32_e32 %0, v94, %0\n\tv_add_f32_e32 %1, v95, %1\n\t"
                                 "s_set_gpr_idx_off"
                                 : "+v"(ax), "+v"(ay) : "s"(m), "s"(off) : "m0", "scc", "vcc", RING_CLOBBERS);
                    acc.x = ax; acc.y = ay;
                }
                b += cc;
                cnt -= cc;
            }
        }
        dp[(size_t)i * 64] = acc;
        slot = slot + 1 == R ? 0 : slot + 1;
    }
}

template <int MODE>
static void run(const f2* dsrc, f2* ddst, int n, int R, int span, int blocks, const std::vector<float>& want, const char* name)
{
    CK(hipMemset(ddst, 0, (size_t)blocks * n * 64 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, dsrc, ddst, n, R, span);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<float> got((size_t)n * 128);
    CK(hipMemcpy(got.data(), ddst, got.size() * 4, hipMemcpyDeviceToHost)); // block 0 only
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); i++) bad += memcmp(&got[i], &want[i], 4) != 0;
    printf("%-28s %s  %.3f ms  mismatches(block 0) %zu of %zu\n", name, bad ? "FAIL" : "PASS", best, bad, got.size());
}

int main()
{
    const int n = 512, R = 32, span = 13, blocks = 2048;
    std::vector<float> src((size_t)blocks * n * 128);
    srand(11);
    for (auto& x : src) x = (float)(rand() % 1000) * 0.37f + 0.001f * (rand() % 7);
    std::vector<float> want((size_t)n * 128);
    for (int i = 0; i < n; i++) {
        const int c = span < i + 1 ? span : i + 1;
        for (int l = 0; l < 128; l++) {
            float acc = 0.0f;
            for (int k = i + 1 - c; k <= i; k++) acc += src[(size_t)k * 128 + l];
            want[(size_t)i * 128 + l] = acc;
        }
    }
    f2 *dsrc, *ddst;
    CK(hipMalloc(&dsrc, src.size() * 4)); CK(hipMalloc(&ddst, src.size() * 4));
    CK(hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    run<0>(dsrc, ddst, n, R, span, blocks, want, "push32 + add32x2");
    run<1>(dsrc, ddst, n, R, span, blocks, want, "push64 + add32x2");
    run<2>(dsrc, ddst, n, R, span, blocks, want, "push32 + pkadd");
    run<3>(dsrc, ddst, n, R, span, blocks, want, "push64 + pkadd");
    return 0;
}
