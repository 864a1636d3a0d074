// Micro-benchmark (not product code): a register-resident ring addressed with the VGPR index mode
// (s_set_gpr_idx_on, wave-uniform indices) versus the LDS ring, for the ordered span sums of the aggregation pass.
// Checks the results against a host computation and prints ns per span entry.
//   MODE 0: LDS ring (reference structure)   MODE 1: gpr-idx loop   MODE 2: gpr-idx, 16 adds unrolled with a computed
//   jump, no nop between "s_add m0" and the next indexed add   MODE 3: same with s_nop 0   MODE 4: same with s_nop 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define RING_CLOBBERS "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"

// 16 indexed adds, entered late (skip = 16 - cnt); NOPS = "" | "s_nop 0\n\t" | "s_nop 1\n\t" (each block = add + s_add [+ nop])
#define ADD1(NOP) "v_add_f32_e32 %0, v56, %0\n\ts_add_u32 m0, m0, 1\n\t" NOP
#define ADD16(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP) ADD1(NOP)

template <int MODE>
__device__ __forceinline__ float span_sum(float acc, int idx, int cnt)
{
    if constexpr (MODE == 1) {
        asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n"
                     "1:\n\tv_add_f32_e32 %0, v56, %0\n\t"
                     "s_add_u32 m0, m0, 1\n\t"
                     "s_sub_u32 %2, %2, 1\n\t"
                     "s_cmp_lg_u32 %2, 0\n\t"
                     "s_cbranch_scc1 1b\n\t"
                     "s_set_gpr_idx_off"
                     : "+v"(acc), "+s"(idx), "+s"(cnt) :: "m0", "scc", "vcc", RING_CLOBBERS);
    } else {
        while (cnt > 0) {
            const int c = cnt < 16 ? cnt : 16;
            constexpr int BLK = MODE == 2 ? 8 : 12; // bytes per add block
            const int off = 12 + BLK * (16 - c);
#define SPAN_ASM(NOP)                                                                                  \
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"                                               \
                 "s_getpc_b64 vcc\n\t"                                                                  \
                 "s_add_u32 vcc_lo, vcc_lo, %2\n\t"                                                     \
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"                                                     \
                 "s_setpc_b64 vcc\n\t" ADD16(NOP) "s_set_gpr_idx_off"                                   \
                 : "+v"(acc) : "s"(idx), "s"(off) : "m0", "scc", "vcc", RING_CLOBBERS)
            if constexpr (MODE == 2) SPAN_ASM("");
            else if constexpr (MODE == 3) SPAN_ASM("s_nop 0\n\t");
            else SPAN_ASM("s_nop 1\n\t");
            idx += c;
            cnt -= c;
        }
    }
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(56))) void k(const float* __restrict__ src, float* __restrict__ dst,
                                                                              const uint32_t* __restrict__ rec, int n, int R)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const float* sp = src + (size_t)blockIdx.x * n * 64 + lane;
    float* dp = dst + (size_t)blockIdx.x * n * 64 + lane;
    int slot = 0;
    for (int i = 0; i < n; i++) {
        const float v = sp[(size_t)i * 64];
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec[i]);
        if constexpr (MODE == 0) lds[slot * 64 + lane] = v;
        else asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v56, %1\n\ts_set_gpr_idx_off" ::"s"(slot), "v"(v) : "m0", RING_CLOBBERS);
        const int lo = (int)(r & 255u), cnt = (int)((r >> 8) & 255u); // span = the cnt entries that end lo entries behind the newest... see host
        int idx = slot - lo;
        if (idx < 0) idx += R;
        const int c1 = cnt < R - idx ? cnt : R - idx;
        float acc = 0.0f;
        if constexpr (MODE == 0) {
            for (int k = 0; k < c1; k++) acc += lds[(idx + k) * 64 + lane];
            for (int k = 0; k < cnt - c1; k++) acc += lds[k * 64 + lane];
        } else {
            acc = span_sum<MODE>(acc, idx, c1);
            if (cnt > c1) acc = span_sum<MODE>(acc, 0, cnt - c1);
        }
        dp[(size_t)i * 64] = acc;
        slot = slot + 1 == R ? 0 : slot + 1;
    }
}

template <int MODE>
static void run(const float* dsrc, float* ddst, const uint32_t* drec, int n, int R, int blocks, const std::vector<float>& want, const char* name)
{
    CK(hipMemset(ddst, 0, (size_t)blocks * n * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), MODE == 0 ? R * 256 : 0, 0, dsrc, ddst, drec, n, R);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    std::vector<float> got((size_t)n * 64);
    CK(hipMemcpy(got.data(), ddst, got.size() * 4, hipMemcpyDeviceToHost)); // block 0 only
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); i++) bad += memcmp(&got[i], &want[i], 4) != 0;
    printf("%-34s %.3f ms  mismatches(block 0) %zu of %zu\n", name, best, bad, got.size());
}

int main()
{
    const int n = 1024, R = 69, blocks = 256 * 16;
    std::vector<float> src((size_t)blocks * n * 64);
    std::vector<uint32_t> rec(n);
    srand(7);
    for (auto& x : src) x = (float)(rand() % 1000) * 0.37f + 0.001f * (rand() % 7);
    double tot = 0;
    for (int i = 0; i < n; i++) {
        int cnt = 1 + rand() % (i + 1 < R ? i + 1 : R); // only entries that exist
        if (rand() % 4) cnt = cnt < 14 ? cnt : 1 + rand() % 14;
        int lo = cnt - 1 + rand() % ((i + 1 < R ? i + 1 : R) - cnt + 1); // span = entries i-lo .. i-lo+cnt-1 (<= i)
        rec[i] = (uint32_t)lo | ((uint32_t)cnt << 8);
        tot += cnt;
    }
    std::vector<float> want((size_t)n * 64);
    for (int i = 0; i < n; i++) {
        const int lo = rec[i] & 255, cnt = (rec[i] >> 8) & 255;
        for (int l = 0; l < 64; l++) {
            float acc = 0.0f;
            for (int k = 0; k < cnt; k++) acc += src[(size_t)(i - lo + k) * 64 + l];
            want[(size_t)i * 64 + l] = acc;
        }
    }
    float *dsrc, *ddst; uint32_t* drec;
    CK(hipMalloc(&dsrc, src.size() * 4)); CK(hipMalloc(&ddst, src.size() * 4)); CK(hipMalloc(&drec, rec.size() * 4));
    CK(hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    printf("mean span %.1f entries, %d lines x %d steps\n", tot / n, blocks, n);
    run<0>(dsrc, ddst, drec, n, R, blocks, want, "LDS ring (compiler loop)");
    run<1>(dsrc, ddst, drec, n, R, blocks, want, "gpr-idx loop");
    run<2>(dsrc, ddst, drec, n, R, blocks, want, "gpr-idx unrolled, no nop");
    run<3>(dsrc, ddst, drec, n, R, blocks, want, "gpr-idx unrolled, s_nop 0");
    run<4>(dsrc, ddst, drec, n, R, blocks, want, "gpr-idx unrolled, s_nop 1");
    return 0;
}
