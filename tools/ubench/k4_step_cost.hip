// Micro-benchmark (not product code): what do the building blocks of one step of the full-ring aggregation pass
// (k_aggregate_rr2.h: RR2_STEP) cost on gfx950, one wave alone on its SIMD and two waves sharing it (the kernel's occupancy)?
// A step of k_agg_rr2 takes ~1090 cycles per wave (0.46 ms for 1013 steps at 2.4 GHz) for ~70 instructions; SQ counters: 0.44-0.47
// of the wave cycles waiting to issue, more scalar than vector instructions.  Cutting the scalar decode chain out of the step
// (round 4, batched decode) changed nothing.  This prints cycles per iteration (s_memtime, one CU) for:
//   push        s_set_gpr_idx_on / v_mov_b64 v[96:97] / s_set_gpr_idx_off
//   adds14      14 dependent v_pk_add_f32, plain registers
//   adds14_idx  the same under the VGPR index mode (idx_on, 14 adds, idx_off)
//   run14       idx_on + s_getpc / s_add / s_addc / s_setpc + 35-add block entered at 21 + idx_off   (rr2_run with 14 entries)
//   run14x2     two such runs on two accumulators back to back (two independent chains)
//   adds14x2    28 v_pk_add_f32 on two accumulators INTERLEAVED (what instruction-level parallelism would buy)
//   decode      v_readlane + the 12 dependent scalar instructions of the record decode
//   divide      v_cvt + v_pk_mul + 2 v_pk_fma (Markstein)
//   step        push + decode + run14 + divide + store address add: the whole step without memory operations
//   step_mem    step + the prefetch load / store pair (global_load_dwordx2 + global_store_dwordx2) with vmcnt(14)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/k4_step_cost.hip -o tools/ubench/k4_step_cost && tools/ubench/k4_step_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
#define CLOB "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239"

#define ADD35(A)                                                                                                       \
    "v_pk_add_f32 " A ", v[26:27], " A "\n\tv_pk_add_f32 " A ", v[28:29], " A "\n\tv_pk_add_f32 " A ", v[30:31], " A "\n\t" \
    "v_pk_add_f32 " A ", v[32:33], " A "\n\tv_pk_add_f32 " A ", v[34:35], " A "\n\tv_pk_add_f32 " A ", v[36:37], " A "\n\t" \
    "v_pk_add_f32 " A ", v[38:39], " A "\n\tv_pk_add_f32 " A ", v[40:41], " A "\n\tv_pk_add_f32 " A ", v[42:43], " A "\n\t" \
    "v_pk_add_f32 " A ", v[44:45], " A "\n\tv_pk_add_f32 " A ", v[46:47], " A "\n\tv_pk_add_f32 " A ", v[48:49], " A "\n\t" \
    "v_pk_add_f32 " A ", v[50:51], " A "\n\tv_pk_add_f32 " A ", v[52:53], " A "\n\tv_pk_add_f32 " A ", v[54:55], " A "\n\t" \
    "v_pk_add_f32 " A ", v[56:57], " A "\n\tv_pk_add_f32 " A ", v[58:59], " A "\n\tv_pk_add_f32 " A ", v[60:61], " A "\n\t" \
    "v_pk_add_f32 " A ", v[62:63], " A "\n\tv_pk_add_f32 " A ", v[64:65], " A "\n\tv_pk_add_f32 " A ", v[66:67], " A "\n\t" \
    "v_pk_add_f32 " A ", v[68:69], " A "\n\tv_pk_add_f32 " A ", v[70:71], " A "\n\tv_pk_add_f32 " A ", v[72:73], " A "\n\t" \
    "v_pk_add_f32 " A ", v[74:75], " A "\n\tv_pk_add_f32 " A ", v[76:77], " A "\n\tv_pk_add_f32 " A ", v[78:79], " A "\n\t" \
    "v_pk_add_f32 " A ", v[80:81], " A "\n\tv_pk_add_f32 " A ", v[82:83], " A "\n\tv_pk_add_f32 " A ", v[84:85], " A "\n\t" \
    "v_pk_add_f32 " A ", v[86:87], " A "\n\tv_pk_add_f32 " A ", v[88:89], " A "\n\tv_pk_add_f32 " A ", v[90:91], " A "\n\t" \
    "v_pk_add_f32 " A ", v[92:93], " A "\n\tv_pk_add_f32 " A ", v[94:95], " A "\n\t"

__device__ __forceinline__ void push(int s2, f2 v)
{
    s2 = __builtin_amdgcn_readfirstlane(s2);
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b64 v[96:97], %1\n\ts_set_gpr_idx_off" ::"s"(s2), "v"(v) : "m0", CLOB);
}
__device__ __forceinline__ void run(f2& acc, int m2, int off)
{
    m2 = __builtin_amdgcn_readfirstlane(m2);
    off = __builtin_amdgcn_readfirstlane(off);
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "s_getpc_b64 vcc\n\ts_add_u32 vcc_lo, vcc_lo, %2\n\ts_addc_u32 vcc_hi, vcc_hi, 0\n\ts_setpc_b64 vcc\n\t" ADD35("%0")
                 "s_set_gpr_idx_off"
                 : "+v"(acc) : "s"(m2), "s"(off) : "m0", "scc", "vcc", CLOB);
}
#define ADD14(A, B)                                                                                                    \
    "v_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\t" \
    "v_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\t" \
    "v_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\t" \
    "v_pk_add_f32 " A ", " B ", " A "\n\tv_pk_add_f32 " A ", " B ", " A "\n\t"

// MODE selects the pattern; every wave of the workgroup runs it n times and wave 0 reports the cycles of ITS loop
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(96))) void k(long long* __restrict__ cycles, float* __restrict__ buf, const uint32_t* __restrict__ recs, int n, int R)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    asm volatile("" ::: CLOB);
    f2 acc = {1.0f, 2.0f}, acc2 = {3.0f, 4.0f}, v = {(float)lane, 1.0f};
    const uint32_t myrec = recs[lane];
    int w1 = wave;
    float* sp = buf + ((size_t)wave * 64 + lane) * 2;
    f2 pf = *reinterpret_cast<f2*>(sp);
    long long t0 = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        if constexpr (MODE == 0) { push(2 * w1, v); w1 = w1 + 1 == R ? 0 : w1 + 1; }
        if constexpr (MODE == 1) asm volatile(ADD14("%0", "%1") : "+v"(acc) : "v"(v));
        if constexpr (MODE == 2) asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t" ADD14("%0", "v[96:97]") "s_set_gpr_idx_off" : "+v"(acc) : "s"(__builtin_amdgcn_readfirstlane(2 * w1)) : "m0", CLOB);
        if constexpr (MODE == 3) run(acc, 2 * (w1 + 14), 12 + 8 * 21);
        if constexpr (MODE == 4) { run(acc, 2 * (w1 + 14), 12 + 8 * 21); run(acc2, 2 * (w1 + 20), 12 + 8 * 21); }
        if constexpr (MODE == 5) {
            asm volatile("v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\t"
                         "v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\t"
                         "v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\t"
                         "v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\t"
                         "v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\t"
                         "v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\t"
                         "v_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1\n\tv_pk_add_f32 %0, %2, %0\n\tv_pk_add_f32 %1, %2, %1"
                         : "+v"(acc), "+v"(acc2) : "v"(v));
        }
        if constexpr (MODE == 6 || MODE >= 8) { // record decode (RR2_EMIT): readlane, then the dependent scalar chain
            const uint32_t r_ = (uint32_t)__builtin_amdgcn_readlane((int)myrec, i & 63);
            const int alo_ = (int)(r_ & 255u), an_ = (int)((r_ >> 8) & 255u);
            uint32_t i1_ = (uint32_t)(w1 - alo_);
            i1_ = i1_ < i1_ + (uint32_t)R ? i1_ : i1_ + (uint32_t)R;
            int n1_ = an_ < R - (int)i1_ ? an_ : R - (int)i1_;
            n1_ = n1_ > 35 ? 35 : (n1_ < 1 ? 1 : n1_);
            const int m2 = 2 * ((int)i1_ + n1_), off = 12 + 8 * 35 - 8 * n1_;
            if constexpr (MODE == 6) { asm volatile("" ::"s"(__builtin_amdgcn_readfirstlane(m2)), "s"(__builtin_amdgcn_readfirstlane(off))); }
            else {
                if constexpr (MODE == 9) {
                    f2 t;
                    asm volatile("s_waitcnt vmcnt(14)\n\tv_mov_b64 %0, %1" : "=&v"(t) : "v"(pf) : "memory");
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pf) : "v"(sp) : "memory");
                    v = t;
                }
                push(2 * w1, v);
                w1 = w1 + 1 == R ? 0 : w1 + 1;
                f2 a = {0.0f, 0.0f};
                run(a, m2 < 2 * 72 ? m2 : 2 * 72, off);
                const float cf = (float)(r_ >> 16), y = __int_as_float(0x3d000000);
                f2 q0 = a * y, cfv = {cf, cf};
                f2 r = __builtin_elementwise_fma(-cfv, q0, a);
                acc = __builtin_elementwise_fma(r, f2{y, y}, q0);
                if constexpr (MODE == 9) *reinterpret_cast<f2*>(sp) = acc;
                else asm volatile("" ::"v"(acc));
            }
        }
        if constexpr (MODE == 7) {
            const float cf = (float)(i & 255), y = __int_as_float(0x3d000000);
            f2 q0 = acc * y, cfv = {cf, cf};
            f2 r = __builtin_elementwise_fma(-cfv, q0, acc);
            acc = __builtin_elementwise_fma(r, f2{y, y}, q0);
        }
    }
    long long t1 = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(pf) : "memory");
    if (lane == 0) cycles[wave] = t1 - t0;
    if (acc.x + acc2.x + acc.y + acc2.y == 12345.678f) buf[lane] = acc.x + (float)w1;
}

template <int MODE>
static void run_mode(const char* name, long long* dcyc, float* dbuf, const uint32_t* drec, int n)
{
    printf("%-12s", name);
    for (int waves : {4, 8}) { // one / two waves per SIMD of the one CU the workgroup runs on
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, dcyc, dbuf, drec, n, 69);
        CK(hipDeviceSynchronize());
        long long c[8];
        CK(hipMemcpy(c, dcyc, sizeof(c), hipMemcpyDeviceToHost));
        double s = 0;
        for (int w = 0; w < waves; w++) s += (double)c[w];
        // s_memtime ticks (the shader clock counter on this chip: the first run, which took a tick for 10 ns, printed ten times
        // these figures as "ns" -- profiles/r4_ubench_k4_step_cost.txt; the whole step of the real kernel takes ~450 ns)
        printf("   %d waves/SIMD: %8.1f ticks/iter/wave", waves / 4, s / waves / n);
    }
    printf("\n");
}

int main()
{
    long long* dcyc;
    float* dbuf;
    uint32_t* drec;
    CK(hipMalloc(&dcyc, 64 * sizeof(long long)));
    CK(hipMalloc(&dbuf, 1 << 20));
    CK(hipMemset(dbuf, 0, 1 << 20));
    uint32_t rec[64];
    for (int i = 0; i < 64; i++) { const int arm = 3 + (i * 7) % 12; rec[i] = (uint32_t)(35 - arm + 34 + 1) | ((uint32_t)(2 * arm + 1) << 8) | (49u << 16); }
    CK(hipMalloc(&drec, sizeof(rec)));
    CK(hipMemcpy(drec, rec, sizeof(rec), hipMemcpyHostToDevice));
    const int n = 20000;
    printf("one workgroup on one CU; s_memtime ticks per iteration and wave\n");
    run_mode<0>("push", dcyc, dbuf, drec, n);
    run_mode<1>("adds14", dcyc, dbuf, drec, n);
    run_mode<2>("adds14_idx", dcyc, dbuf, drec, n);
    run_mode<3>("run14", dcyc, dbuf, drec, n);
    run_mode<4>("run14x2", dcyc, dbuf, drec, n);
    run_mode<5>("adds14x2", dcyc, dbuf, drec, n);
    run_mode<6>("decode", dcyc, dbuf, drec, n);
    run_mode<7>("divide", dcyc, dbuf, drec, n);
    run_mode<8>("step", dcyc, dbuf, drec, n);
    run_mode<9>("step_mem", dcyc, dbuf, drec, n);
    return 0;
}
