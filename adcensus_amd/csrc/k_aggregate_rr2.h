// k_aggregate_rr2.h -- full-ring aggregation passes, third generation: TWO disparities per lane, the ring in registers as
// even-aligned VGPR PAIRS, one packed add (v_pk_add_f32) per ring entry.  Included by k_aggregate.hip after
// k_aggregate_rr.h (whose primitives / emulation scaffolding it reuses).
//
// Same semantics as agg_rr_body (cross_aggregator.cpp:327-394: every output = sequential f32 sum from 0.0f in the order
// t = -arm .. +arm over the pixel's own arm span; the second pass of an iteration divided by the support count) and the
// same marching-ring control flow (bulk 8-byte records, biased arm + one slot counter, one block of 35 indexed adds
// entered through a computed jump, Markstein division).  What changes:
//   * a wave owns 128 consecutive disparities (512 contiguous bytes) of every pixel of its line: lane l holds
//     d = 2l, 2l+1.  Ring slot s = v[RR2_V0 + 2s : RR2_V0 + 2s + 1]; the VGPR index mode applies to the 64-bit operands
//     of v_mov_b64 / v_pk_add_f32 (M0 = 2 * slot; checked on gfx950 by tools/ubench/idx_pk.hip), so every wave-uniform
//     instruction of a step (record decode, slot arithmetic, jump set-up, waits) and every add now covers twice the
//     bytes.  The two sums of a lane are independent IEEE chains: bit-identical to two one-float lanes.
//   * 144 ring registers + 96 for the compiler = 240 VGPRs -> 2 waves per SIMD (the one-float ring: 128 -> 4): the same
//     bytes in flight per SIMD, half the instructions per byte.  SQ counters of the one-float body (profiles/
//     r3_sq_all_noise.md): 53 % of the wave cycles waiting for an issue slot, 14.5 % in scalar instructions.
//   * COSTIN (first pass of the pipeline): the matching cost (cost_computor.cpp:82-121) is computed in registers on the
//     same body.  Lane l needs the right-image pixels of columns x - d0 and x - d0 - 1 (d0 = its even disparity): two
//     lane windows A (even d) and B (odd d); when the wave advances one pixel, A' = B shifted up one lane with the new
//     column entering at lane 0 (DPP wave_shr:1) and B' = A -- three DPP moves per step for 128 disparities.
// Needs Dp % 128 == 0 and 2L+1 <= RR2_SLOTS; otherwise the one-float body runs.
#pragma once

#define RR2_BLK 35
#define RR2_PF 8
#define RR2_V0 96     // ring slot s = v[96 + 2s : 97 + 2s]; the compiler keeps to v0..v95 (amdgpu_num_vgpr)
#define RR2_SLOTS 72  // v96 .. v239

// fused-cost inputs (k_cost_records): right records padded with out-of-image markers on both sides
struct AggCostIn {
    const uint4* rrec; // right records, row pitch rpitch, first real column at index padl
    const uint4* lrec; // left records [H][W]
    const float* lut_ad;
    const float* lut_census;
    int rpitch, padl, dmin, D;
};

#ifndef RR_EMUL
typedef float rr2_f2 __attribute__((ext_vector_type(2)));
RR_FN rr2_f2 rr2_make(float a, float b) { rr2_f2 r; r.x = a; r.y = b; return r; }
#define RR2_LDS_TABLES extern __shared__ __attribute__((aligned(16))) float rr2_lds[]
#define RR2_VLOAD(DST, PTR) asm volatile("global_load_dwordx2 %0, %1, off" ADC_VOL_NT_STR : "=v"(DST) : "v"(PTR) : "memory")
#define RR2_WAIT_TAKE(DST, SRC, N) asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b64 %0, %1" : "=&v"(DST) : "v"(SRC), "n"(N) : "memory")
#define RR2_DRAIN8(D, S)                                                                                             \
    asm volatile("s_waitcnt vmcnt(0)\n\t"                                                                            \
                 "v_mov_b64 %0, %8\n\tv_mov_b64 %1, %9\n\tv_mov_b64 %2, %10\n\tv_mov_b64 %3, %11\n\t"               \
                 "v_mov_b64 %4, %12\n\tv_mov_b64 %5, %13\n\tv_mov_b64 %6, %14\n\tv_mov_b64 %7, %15"                  \
                 : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(D[4]), "=&v"(D[5]), "=&v"(D[6]), "=&v"(D[7]) \
                 : "v"(S[0]), "v"(S[1]), "v"(S[2]), "v"(S[3]), "v"(S[4]), "v"(S[5]), "v"(S[6]), "v"(S[7])            \
                 : "memory")
// bulk block of cost records: lane LN loads the 2 x 3 dwords of entry IDX_R (right, index into the padded row) / IDX_L (left)
#define RR2_CREC_LOAD(NR0, NR1, NR2, NL0, NL1, NL2, IDX_R, IDX_L)                                                    \
    do {                                                                                                             \
        const int LN = lane;                                                                                         \
        const uint4* pr_ = rbase + (IDX_R);                                                                          \
        const uint4* pl_ = lrow + (IDX_L);                                                                           \
        asm volatile("global_load_dword %0, %1, off" : "=v"(NR0) : "v"(pr_) : "memory");                             \
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(NR1) : "v"(pr_) : "memory");                    \
        asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(NR2) : "v"(pr_) : "memory");                    \
        asm volatile("global_load_dword %0, %1, off" : "=v"(NL0) : "v"(pl_) : "memory");                             \
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(NL1) : "v"(pl_) : "memory");                    \
        asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(NL2) : "v"(pl_) : "memory");                    \
    } while (0)
#define RR2_CREC_TAKE(WAIT, C0, C1, C2, C3, C4, C5, N0, N1, N2, N3, N4, N5)                                          \
    asm volatile(WAIT "v_mov_b32 %0, %6\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\t"              \
                      "v_mov_b32 %4, %10\n\tv_mov_b32 %5, %11"                                                       \
                 : "=&v"(C0), "=&v"(C1), "=&v"(C2), "=&v"(C3), "=&v"(C4), "=&v"(C5)                                  \
                 : "v"(N0), "v"(N1), "v"(N2), "v"(N3), "v"(N4), "v"(N5) : "memory")
#define RR2_KEEPALIVE6(A, B, C, D, E, F) asm volatile("s_waitcnt vmcnt(0)" ::"v"(A), "v"(B), "v"(C), "v"(D), "v"(E), "v"(F) : "memory")
#define RR2_REC_TAKE1(WAIT, C1, N1) asm volatile(WAIT "v_mov_b32 %0, %1" : "=&v"(C1) : "v"(N1) : "memory")
#define RR2_KEEPALIVE2(A, B) asm volatile("s_waitcnt vmcnt(0)" ::"v"(A), "v"(B) : "memory")
#define RR2_KEEPALIVE1(A) asm volatile("s_waitcnt vmcnt(0)" ::"v"(A) : "memory")
// the two lane windows advance one pixel: A' = B shifted up one lane, the new column (wave-uniform) enters at lane 0; B' = A
#define RR2_WIN_STEP(NB, NC0, NC1)                                                                                   \
    do {                                                                                                             \
        const uint32_t a0_ = wA0, a1_ = wA1, a2_ = wA2;                                                              \
        wA0 = (uint32_t)__builtin_amdgcn_update_dpp((int)(NB), (int)wB0, 0x138, 0xf, 0xf, false);                    \
        wA1 = (uint32_t)__builtin_amdgcn_update_dpp((int)(NC0), (int)wB1, 0x138, 0xf, 0xf, false);                   \
        wA2 = (uint32_t)__builtin_amdgcn_update_dpp((int)(NC1), (int)wB2, 0x138, 0xf, 0xf, false);                   \
        wB0 = a0_; wB1 = a1_; wB2 = a2_;                                                                             \
    } while (0)
#define RR2_SAD(A, B) __builtin_amdgcn_sad_u8((A), (B), 0u)
#define RR2_POPC(X) ((uint32_t)__popc(X))

#define RR2_CLOBBERS "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239"
// ring[slot] = v
RR_FN void rr2_push(int slot, rr2_f2 v)
{
    const int s2 = RR_UNIFORM(2 * slot);
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b64 v[96:97], %1\n\ts_set_gpr_idx_off" ::"s"(s2), "v"(v)
                 : "m0", RR2_CLOBBERS);
}
// One run: acc += ring[m - c], ..., ring[m - 1] in this order, c = (12 + 8*RR2_BLK - off) / 8 <= RR2_BLK entries.  The adds
// name the pairs v[V0-70 : V0-69] .. v[V0-2 : V0-1]; the hardware adds M0 = 2*m to the register number of src0 and the
// computed jump enters the block at position 35 - c, so the pairs actually read are ring slots m - c .. m - 1.  One
// packed add = 8 bytes (VOP3P); 12 = the three 4-byte scalar instructions between the value s_getpc returns and the
// first add.
RR_FN void rr2_run(rr2_f2& acc, int m2, int off)
{
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "s_getpc_b64 vcc\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %2\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n\t"
                 "v_pk_add_f32 %0, v[26:27], %0\n\t"
                 "v_pk_add_f32 %0, v[28:29], %0\n\t"
                 "v_pk_add_f32 %0, v[30:31], %0\n\t"
                 "v_pk_add_f32 %0, v[32:33], %0\n\t"
                 "v_pk_add_f32 %0, v[34:35], %0\n\t"
                 "v_pk_add_f32 %0, v[36:37], %0\n\t"
                 "v_pk_add_f32 %0, v[38:39], %0\n\t"
                 "v_pk_add_f32 %0, v[40:41], %0\n\t"
                 "v_pk_add_f32 %0, v[42:43], %0\n\t"
                 "v_pk_add_f32 %0, v[44:45], %0\n\t"
                 "v_pk_add_f32 %0, v[46:47], %0\n\t"
                 "v_pk_add_f32 %0, v[48:49], %0\n\t"
                 "v_pk_add_f32 %0, v[50:51], %0\n\t"
                 "v_pk_add_f32 %0, v[52:53], %0\n\t"
                 "v_pk_add_f32 %0, v[54:55], %0\n\t"
                 "v_pk_add_f32 %0, v[56:57], %0\n\t"
                 "v_pk_add_f32 %0, v[58:59], %0\n\t"
                 "v_pk_add_f32 %0, v[60:61], %0\n\t"
                 "v_pk_add_f32 %0, v[62:63], %0\n\t"
                 "v_pk_add_f32 %0, v[64:65], %0\n\t"
                 "v_pk_add_f32 %0, v[66:67], %0\n\t"
                 "v_pk_add_f32 %0, v[68:69], %0\n\t"
                 "v_pk_add_f32 %0, v[70:71], %0\n\t"
                 "v_pk_add_f32 %0, v[72:73], %0\n\t"
                 "v_pk_add_f32 %0, v[74:75], %0\n\t"
                 "v_pk_add_f32 %0, v[76:77], %0\n\t"
                 "v_pk_add_f32 %0, v[78:79], %0\n\t"
                 "v_pk_add_f32 %0, v[80:81], %0\n\t"
                 "v_pk_add_f32 %0, v[82:83], %0\n\t"
                 "v_pk_add_f32 %0, v[84:85], %0\n\t"
                 "v_pk_add_f32 %0, v[86:87], %0\n\t"
                 "v_pk_add_f32 %0, v[88:89], %0\n\t"
                 "v_pk_add_f32 %0, v[90:91], %0\n\t"
                 "v_pk_add_f32 %0, v[92:93], %0\n\t"
                 "v_pk_add_f32 %0, v[94:95], %0\n\t"
                 "s_set_gpr_idx_off"
                 : "+v"(acc)
                 : "s"(m2), "s"(off)
                 : "m0", "scc", "vcc", RR2_CLOBBERS);
}
#else
// --------------------------------------------------------------------------------------------- emulation primitives
struct rr2_f2 { float x, y; };
RR_FN rr2_f2 rr2_make(float a, float b) { rr2_f2 r; r.x = a; r.y = b; return r; }
struct Rr2EmulCost { float lut[768 + 64]; };
static Rr2EmulCost rr2_emul_cost;
#define RR2_LDS_TABLES float* const rr2_lds = rr2_emul_cost.lut
#define RR2_VLOAD(DST, PTR) ((DST) = *reinterpret_cast<const rr2_f2*>(PTR))
#define RR2_WAIT_TAKE(DST, SRC, N) ((DST) = (SRC))
#define RR2_DRAIN8(D, S) do { for (int i_ = 0; i_ < 8; i_++) D[i_] = S[i_]; } while (0)
#define RR2_CREC_LOAD(NR0, NR1, NR2, NL0, NL1, NL2, IDX_R, IDX_L)                                                    \
    do {                                                                                                             \
        for (int LN = 0; LN < 64; LN++) {                                                                            \
            const uint4* pr_ = rbase + (IDX_R);                                                                      \
            const uint4* pl_ = lrow + (IDX_L);                                                                       \
            (NR0).v[LN] = pr_->x; (NR1).v[LN] = pr_->y; (NR2).v[LN] = pr_->z;                                        \
            (NL0).v[LN] = pl_->x; (NL1).v[LN] = pl_->y; (NL2).v[LN] = pl_->z;                                        \
        }                                                                                                            \
    } while (0)
#define RR2_CREC_TAKE(WAIT, C0, C1, C2, C3, C4, C5, N0, N1, N2, N3, N4, N5) do { C0 = N0; C1 = N1; C2 = N2; C3 = N3; C4 = N4; C5 = N5; } while (0)
#define RR2_KEEPALIVE6(A, B, C, D, E, F) ((void)0)
#define RR2_REC_TAKE1(WAIT, C1, N1) do { C1 = N1; } while (0)
#define RR2_KEEPALIVE2(A, B) ((void)0)
#define RR2_KEEPALIVE1(A) ((void)0)
// One lane at a time: the window of this lane after the step that brings entry `win_j` in is, by construction, the record
// of column win_j - d0 (A) / win_j - d0 - 1 (B) -- read directly; the shifting form itself (A' = shr(B), B' = A) is
// checked as a whole-wave model by tests/test_emul.py::test_rr2_cost_windows_shift_identity.
#define RR2_WIN_STEP(NB, NC0, NC1)                                                                                   \
    do {                                                                                                             \
        win_j++;                                                                                                     \
        const uint4 ra_ = rrow[win_j - 2 * lane], rb_ = rrow[win_j - 2 * lane - 1];                                  \
        assert(lane != 0 || (ra_.x == (NB) && ra_.y == (NC0) && ra_.z == (NC1)));                                   \
        wA0 = ra_.x; wA1 = ra_.y; wA2 = ra_.z; wB0 = rb_.x; wB1 = rb_.y; wB2 = rb_.z;                                \
    } while (0)
RR_FN uint32_t rr2_sad_u8(uint32_t a, uint32_t b)
{
    uint32_t s = 0;
    for (int k = 0; k < 4; k++) {
        const int x = (int)((a >> (8 * k)) & 255u), y = (int)((b >> (8 * k)) & 255u);
        s += (uint32_t)(x > y ? x - y : y - x);
    }
    return s;
}
#define RR2_SAD(A, B) rr2_sad_u8((A), (B))
#define RR2_POPC(X) ((uint32_t)__builtin_popcount(X))
RR_FN void rr2_push(int slot, rr2_f2 v)
{
    assert(slot >= 0 && slot < RR2_SLOTS);
    rr_emul.vgpr[RR2_V0 + 2 * slot] = v.x;
    rr_emul.vgpr[RR2_V0 + 2 * slot + 1] = v.y;
}
RR_FN void rr2_run(rr2_f2& acc, int m2, int off)
{
    assert((off - 12) % 8 == 0 && m2 % 2 == 0);
    const int first = (off - 12) / 8; // position of the first add executed
    assert(first >= 0 && first < RR2_BLK);
    for (int p = first; p < RR2_BLK; p++) {
        const int reg = (RR2_V0 - 2 * RR2_BLK + 2 * p) + m2; // named register + M0
        assert(reg >= RR2_V0 && reg + 1 < RR2_V0 + 2 * RR2_SLOTS && reg % 2 == 0);
        acc.x += rr_emul.vgpr[reg];
        acc.y += rr_emul.vgpr[reg + 1];
    }
}
#endif

// acc += ring[idx], ring[idx+1], ... (cnt >= 1 entries, no wrap), in this order
RR_FN rr2_f2 rr2_sum(rr2_f2 acc, int idx, int cnt)
{
    idx = RR_UNIFORM(idx);
    cnt = RR_UNIFORM(cnt);
    while (__builtin_expect(cnt > RR2_BLK, 0)) { // rare: 4 % of the spans of a natural image (kept out of the fall-through path)
        rr2_run(acc, 2 * (idx + RR2_BLK), 12);
        idx += RR2_BLK;
        cnt -= RR2_BLK;
    }
    rr2_run(acc, 2 * (idx + cnt), 12 + 8 * RR2_BLK - 8 * cnt);
    return acc;
}

// One PIECE: outputs [m0, m1) of line (fixed, chunk).
template <bool VERT, bool DIVIDE, bool COSTIN>
RR_FN void agg_rr2_piece(const float* __restrict__ src, float* __restrict__ dst,
                         const uint2* __restrict__ rec, // {lob | span<<8 | count<<16, RN(1/count)}, line-major
                         int W, int H, int Dp, int L, int fixed, int chunk, int m0, int m1, float* rr2_lds, const AggCostIn& ci)
{
    const int R = 2 * L + 1;
    const int lane = RR_LANE;
    const int N = VERT ? H : W;
    // [m0, m1) = outputs of this piece; [lo, hi) = entries it reads
    const int lo = adc_imax(0, m0 - L);
    const int hi = adc_imin(N, m1 + L);

    const long long pix_step = VERT ? (long long)W : 1LL;
    const long long pix0 = VERT ? (long long)fixed : (long long)fixed * W;
    const long long fstep = pix_step * Dp; // floats per step
#ifdef RR2_FAKE_MEM // (timing experiment only: 1 = every steady-state load AND store of a wave goes to ONE address, 2 = the loads, 3 = the
                    // stores -- the results are garbage; profiles/r4_k4_fake_mem.txt)
#define RR2_MSTEP_LD (RR2_FAKE_MEM == 3 ? fstep : 0)
#define RR2_MSTEP_ST (RR2_FAKE_MEM == 2 ? fstep : 0)
#else
#define RR2_MSTEP_LD fstep
#define RR2_MSTEP_ST fstep
#endif
    const float* sp = src + pix0 * Dp + chunk * 128 + 2 * lane;
    float* dpn = dst + pix0 * Dp + chunk * 128 + 2 * lane + (long long)m0 * fstep; // outputs leave in increasing order from m0
    const uint2* rl = rec + (long long)fixed * N;                                  // records of this line
#define RR2_LD(J) (*reinterpret_cast<const rr2_f2*>(sp + (long long)(J) * fstep))

    // ---- fused cost state: two lane windows {bgrx, census lo, census hi} of the right image (A: even d, B: odd d)
    uint32_t wA0 = 0, wA1 = 0, wA2 = 0, wB0 = 0, wB1 = 0, wB2 = 0;
    const uint4* rbase = nullptr; // padded right-record row of this line
    const uint4* rrow = nullptr;  // rrow[x] = right record of column x - d_first (lane 0's even disparity)
    const uint4* lrow = nullptr;
    int roff = 0;                 // rrow == rbase + roff
    bool pad0 = false, pad1 = false;
    float* const lutA = rr2_lds;  // A[766] then C[64]
    float* const lutC = rr2_lds + 768;
#ifdef RR_EMUL
    int win_j = 0;
#endif
    if constexpr (COSTIN) {
        const int d_first = chunk * 128 + ci.dmin;
        pad0 = chunk * 128 + 2 * lane >= ci.D;
        pad1 = chunk * 128 + 2 * lane + 1 >= ci.D;
        rbase = ci.rrec + (size_t)fixed * ci.rpitch;
        roff = ci.padl - d_first;
        rrow = rbase + roff;
        lrow = ci.lrec + (size_t)fixed * W;
        // windows of the entry BEFORE the first one (columns lo-1-d0 and lo-2-d0); the first step shifts them into place
        int ga = roff + lo - 1 - 2 * lane, gb = ga - 1;
        ga = ga < 0 ? 0 : (ga >= ci.rpitch ? ci.rpitch - 1 : ga); // index 0 is a marker column (padl >= 1)
        gb = gb < 0 ? 0 : (gb >= ci.rpitch ? ci.rpitch - 1 : gb);
        const uint4 qa = rbase[ga], qb = rbase[gb];
        wA0 = qa.x; wA1 = qa.y; wA2 = qa.z;
        wB0 = qb.x; wB1 = qb.y; wB2 = qb.z;
#ifdef RR_EMUL
        win_j = lo - 1;
#endif
    }
// matching cost of the next entry: (RB, RC0, RC1) = right pixel of the new column entering at lane 0, (LB, LC0, LC1) =
// left pixel of the entry (all wave-uniform).  == ((1 - ea) + 1) - ec of cost_computor.cpp:117 through the host tables;
// right pixel outside the image -> 1.0f (:101-104); padding disparities (d >= D) -> 0.0f
#define RR2_COST(RB, RC0, RC1, LB, LC0, LC1, OUT)                                                                    \
    do {                                                                                                             \
        RR2_WIN_STEP(RB, RC0, RC1);                                                                                  \
        const uint32_t adA_ = RR2_SAD(wA0, (uint32_t)(LB)), adB_ = RR2_SAD(wB0, (uint32_t)(LB));                     \
        const uint32_t hmA_ = RR2_POPC(wA1 ^ (uint32_t)(LC0)) + RR2_POPC(wA2 ^ (uint32_t)(LC1));                     \
        const uint32_t hmB_ = RR2_POPC(wB1 ^ (uint32_t)(LC0)) + RR2_POPC(wB2 ^ (uint32_t)(LC1));                     \
        float cA_ = lutA[adA_ < 766u ? adA_ : 765u] - lutC[hmA_ & 63u];                                              \
        float cB_ = lutA[adB_ < 766u ? adB_ : 765u] - lutC[hmB_ & 63u];                                              \
        cA_ = wA0 == 0xFFFFFFFFu ? 1.0f : cA_;                                                                       \
        cB_ = wB0 == 0xFFFFFFFFu ? 1.0f : cB_;                                                                       \
        OUT = rr2_make(pad0 ? 0.0f : cA_, pad1 ? 0.0f : cB_);                                                        \
    } while (0)
// cost of entry J from plain (uniform) loads: phase A and the tail
#define RR2_COST_AT(J, OUT)                                                                                          \
    do {                                                                                                             \
        const uint4 rn_ = rrow[(J)], ln_ = lrow[(J)];                                                                \
        RR2_COST(rn_.x, rn_.y, rn_.z, ln_.x, ln_.y, ln_.z, OUT);                                                     \
    } while (0)

    // ---- arm records.  Emit index te = tbase + pos: output m0 + te uses record m0 + te.  Lane l of a block holds the
    // record of emit index tbase + l; the next block is in flight while the current one is used.
    rr_lanes32 c1x, c1y, n1x, n1y;
    int pos = 0, tbase = 0;
#define RR2_REC_ISSUE(TB)                                                                                            \
    do {                                                                                                             \
        if constexpr (DIVIDE) RR_REC_LOAD2(n1x, n1y, adc_imin(m0 + (TB) + LN, N - 1));                               \
        else RR_REC_LOAD1(n1x, adc_imin(m0 + (TB) + LN, N - 1));                                                     \
    } while (0)
// take over the block in flight (WAIT = "" inside the steady state: it was issued >= 64 steps, i.e. >= 64 younger
// vector-memory operations ago, and the counter tracks at most 63)
#define RR2_REC_TAKE(WAIT)                                                                                           \
    do {                                                                                                             \
        if constexpr (DIVIDE) RR_REC_TAKE2(WAIT, c1x, c1y, n1x, n1y);                                                \
        else RR2_REC_TAKE1(WAIT, c1x, n1x);                                                                          \
    } while (0)
#define RR2_REC_ADVANCE(WAIT)                                                                                        \
    if (pos == 64) {                                                                                                 \
        RR2_REC_TAKE(WAIT);                                                                                          \
        tbase += 64;                                                                                                 \
        pos = 0;                                                                                                     \
        RR2_REC_ISSUE(tbase + 64);                                                                                   \
    }
    RR2_REC_ISSUE(0);
    RR2_REC_TAKE(RR_WAIT_ALL_STR);
    RR2_REC_ISSUE(64);

    // Ring slots.  Entry e sits in slot (e - lo) mod R; w1 = slot of the next entry.  In the steady state output m is
    // summed right after entry m + L was pushed, i.e. with w1 = slot(m) + L + 1: the records carry the BIASED arm
    // lob = arm_lo + L + 1, so the first slot of the span is simply (w1 - lob) mod R.  Where outputs leave without a push
    // (image end) the slot is computed from the indices (RR2_SLOT).
    int w1 = 0;
#define RR2_SLOT(M) (((M) + L + 1 - lo) % R)
#define RR2_PUSH(VAL) do { rr2_push(w1, (VAL)); w1 = w1 + 1 == R ? 0 : w1 + 1; } while (0)
// output m = m0 + te (record at lane POS of the current block); W1 = slot of entry m + L + 1; exactly one store
#define RR2_EMIT(POS, W1)                                                                                            \
    do {                                                                                                             \
        const uint32_t r_ = RR_READLANE(c1x, (POS));                                                                 \
        const int alo_ = (int)(r_ & 255u), an_ = (int)((r_ >> 8) & 255u);                                            \
        uint32_t i1_ = (uint32_t)((W1) - alo_);                                                                      \
        i1_ = i1_ < i1_ + (uint32_t)R ? i1_ : i1_ + (uint32_t)R; /* min_u32: wraps a negative index */                \
        const int n1_ = adc_imin(an_, R - (int)i1_);                                                                 \
        rr2_f2 acc_ = rr2_sum(rr2_make(0.0f, 0.0f), (int)i1_, n1_); /* t = -arm .. +arm */                            \
        if (__builtin_expect(an_ > n1_, 0)) acc_ = rr2_sum(acc_, 0, an_ - n1_); /* wrapped part */                    \
        if constexpr (DIVIDE) {                                                                                      \
            const float y_ = RR_BITS_TO_F32(RR_READLANE(c1y, (POS)));                                                \
            const float cf_ = (float)(r_ >> 16);                                                                     \
            acc_ = rr2_make(rr_divide(acc_.x, cf_, y_), rr_divide(acc_.y, cf_, y_)); /* cross_aggregator.cpp:389 */  \
        }                                                                                                            \
        ADC_VOL_STORE(reinterpret_cast<rr2_f2*>(dpn), acc_);                                                         \
        dpn += RR2_MSTEP_ST;                                                                                         \
    } while (0)

    // ---- phase A: entries lo .. jB-1 precede the first output's look-ahead (no output yet)
    const int jB = adc_imin(hi, m0 + L);
    if constexpr (COSTIN) {
        for (int j = lo; j < jB; j++) {
            rr2_f2 v;
            RR2_COST_AT(j, v);
            RR2_PUSH(v);
        }
    } else {
        for (int j = lo; j < jB; j += RR2_PF) {
            rr2_f2 tv[RR2_PF];
#pragma unroll
            for (int u = 0; u < RR2_PF; u++) tv[u] = RR2_LD(adc_imin(j + u, jB - 1));
#pragma unroll
            for (int u = 0; u < RR2_PF; u++)
                if (j + u < jB) RR2_PUSH(tv[u]);
        }
    }
    // ---- phase B: entry j arrives, output m = j - L leaves
    RR_WAITALL(); // the manual vmcnt bookkeeping starts from an empty queue
    int j = jB;
    if (j + 2 * RR2_PF <= hi) {
        if constexpr (COSTIN) {
            // No data load: the entry is computed from cost records that arrive in BULK like the arm records (lane l of a
            // block holds the 2 x 3 dwords of entry bx0 + l, six v_readlane per step).  The only vector-memory operation
            // of a step is its store, so a block issued 64 steps ago has landed when it is taken over ("" wait).
            rr_lanes32 cR0, cR1, cR2, cL0, cL1, cL2, nR0, nR1, nR2, nL0, nL1, nL2;
            int bx0 = j, bpos = 0;
#define RR2_CREC_ISSUE(X0)                                                                                           \
    RR2_CREC_LOAD(nR0, nR1, nR2, nL0, nL1, nL2, adc_imax(0, adc_imin((X0) + LN + roff, ci.rpitch - 1)), adc_imin((X0) + LN, W - 1))
#define RR2_CREC_NEXT(WAIT)                                                                                          \
    if (bpos == 64) {                                                                                                \
        RR2_CREC_TAKE(WAIT, cR0, cR1, cR2, cL0, cL1, cL2, nR0, nR1, nR2, nL0, nL1, nL2);                             \
        bx0 += 64;                                                                                                   \
        bpos = 0;                                                                                                    \
        RR2_CREC_ISSUE(bx0 + 64);                                                                                    \
    }
            RR2_CREC_ISSUE(bx0);
            RR2_CREC_TAKE(RR_WAIT_ALL_STR, cR0, cR1, cR2, cL0, cL1, cL2, nR0, nR1, nR2, nL0, nL1, nL2);
            RR2_CREC_ISSUE(bx0 + 64);
#define RR2_STEPC(U)                                                                                                 \
    do {                                                                                                             \
        const int li_ = bpos + (U);                                                                                  \
        rr2_f2 v_;                                                                                                   \
        RR2_COST(RR_READLANE(cR0, li_), RR_READLANE(cR1, li_), RR_READLANE(cR2, li_), RR_READLANE(cL0, li_),         \
                 RR_READLANE(cL1, li_), RR_READLANE(cL2, li_), v_);                                                  \
        RR2_PUSH(v_);                                                                                                \
        RR2_EMIT(pos + (U), w1);                                                                                     \
    } while (0)
            for (; j + RR2_PF <= hi; j += RR2_PF) {
                RR2_REC_ADVANCE("");
                RR2_CREC_NEXT("");
                RR2_STEPC(0); RR2_STEPC(1); RR2_STEPC(2); RR2_STEPC(3); RR2_STEPC(4); RR2_STEPC(5); RR2_STEPC(6); RR2_STEPC(7);
                pos += RR2_PF;
                bpos += RR2_PF;
            }
            // the blocks in flight are never used: wait for them before their registers die
            RR2_KEEPALIVE6(nR0, nR1, nR2, nL0, nL1, nL2);
#undef RR2_CREC_ISSUE
#undef RR2_CREC_NEXT
#undef RR2_STEPC
        } else {
            rr2_f2 pf[RR2_PF];
            const float* spn = sp + (long long)j * fstep;
#pragma unroll
            for (int u = 0; u < RR2_PF; u++) {
                RR2_VLOAD(pf[u], spn);
                spn += RR2_MSTEP_LD;
            }
// one step; WAITN = vector-memory operations younger than slot U's load that may stay in flight
#define RR2_STEP(U, WAITN)                                                                                           \
    do {                                                                                                             \
        rr2_f2 v_;                                                                                                   \
        RR2_WAIT_TAKE(v_, pf[U], WAITN);                                                                             \
        RR2_VLOAD(pf[U], spn);                                                                                       \
        spn += RR2_MSTEP_LD;                                                                                         \
        RR2_PUSH(v_);                                                                                                \
        RR2_EMIT(pos + (U), w1); /* exactly one compiler-issued vector-memory operation (a store) */                 \
    } while (0)
            // first block: younger operations = prologue loads of slots U+1.. (1 each) + 2 per finished step
            static_assert(RR2_PF == 8, "the peeled first block is written for RR2_PF == 8");
            RR2_STEP(0, 7); RR2_STEP(1, 8); RR2_STEP(2, 9); RR2_STEP(3, 10); RR2_STEP(4, 11); RR2_STEP(5, 12); RR2_STEP(6, 13); RR2_STEP(7, 14);
            j += RR2_PF;
            pos += RR2_PF;
            // steady state: younger operations = the reissue step's own store + 2 per younger step = 1 + 2*(RR2_PF-1) = 15;
            // wait for <= 14 (one stricter; the bulk record loads only make it more conservative)
            for (; j + 2 * RR2_PF <= hi; j += RR2_PF) {
                RR2_REC_ADVANCE("");
                RR2_STEP(0, 14); RR2_STEP(1, 14); RR2_STEP(2, 14); RR2_STEP(3, 14); RR2_STEP(4, 14); RR2_STEP(5, 14); RR2_STEP(6, 14); RR2_STEP(7, 14);
                pos += RR2_PF;
            }
#undef RR2_STEP
            // drain: the RR2_PF entries still in flight are entries j .. j+RR2_PF-1 (all < hi)
            RR2_REC_ADVANCE(RR_WAIT_ALL_STR);
            rr2_f2 df[RR2_PF];
            RR2_DRAIN8(df, pf);
#pragma unroll
            for (int u = 0; u < RR2_PF; u++) {
                RR2_PUSH(df[u]);
                RR2_EMIT(pos + u, w1);
            }
            j += RR2_PF;
            pos += RR2_PF;
        }
    }
    // ---- tail of phase B (< 2*RR2_PF entries left) and phase C (outputs whose look-ahead ends beyond the last entry, at
    // the image end): one generic loop; the remaining entries are loaded up front
    {
        rr2_f2 tv[2 * RR2_PF];
        if constexpr (!COSTIN) {
#pragma unroll
            for (int u = 0; u < 2 * RR2_PF; u++) tv[u] = RR2_LD(adc_imin(j + u, hi - 1));
        }
        const int nm = m1 - m0; // outputs in total; tbase + pos of them are done
        int u = 0;
#pragma unroll 1
        while (tbase + pos < nm) {
            const int m = m0 + tbase + pos;
            if (j < hi && j <= m + L) { // output m still waits for entry j
                rr2_f2 v;
                if constexpr (COSTIN) {
                    RR2_COST_AT(j, v);
                } else {
                    v = tv[0];
#pragma unroll
                    for (int q = 1; q < 2 * RR2_PF; q++) v = u == q ? tv[q] : v;
                }
                RR2_PUSH(v);
                u++;
                j++;
                if (j < hi && j <= m + L) continue;
            }
            RR2_REC_ADVANCE(RR_WAIT_ALL_STR);
            RR2_EMIT(pos, RR2_SLOT(m));
            pos++;
        }
    }
    // the record block in flight is never used: wait for it before its registers die (a late-landing load would
    // otherwise overwrite whatever the compiler put there)
    if constexpr (DIVIDE) RR2_KEEPALIVE2(n1x, n1y);
    else RR2_KEEPALIVE1(n1x);
#undef RR2_LD
#undef RR2_COST
#undef RR2_COST_AT
#undef RR2_REC_ISSUE
#undef RR2_REC_TAKE
#undef RR2_REC_ADVANCE
#undef RR2_SLOT
#undef RR2_PUSH
#undef RR2_EMIT
}

// A wave = one CHUNK of the flattened output index space (line-major: line * N + m), chunk_len outputs: one piece when the
// chunk lies inside a line, two when it straddles a line end, several whole lines when chunk_len > N.  With
// chunk_len = ceil(lines * N / wave slots) every slot of the chip gets the same number of steps: a 1080p row pass
// (1080 lines on 2048 slots) otherwise runs 5 segments per row in 2.6 -> 3 rounds with a 2L halo per segment
// (17 % more entries read, profiles/r3_hseg_fetch.txt); whole-line segments are the special case N % chunk_len == 0.
template <bool VERT, bool DIVIDE, bool COSTIN>
RR_FN void agg_rr2_body(const float* __restrict__ src, float* __restrict__ dst, const uint2* __restrict__ rec, int W, int H, int Dp,
                        int L, int chunk_len, int nwaves, int per_xcd, const int* __restrict__ armmax, int small_variant, int small_L,
                        const AggCostIn& ci)
{
    static_assert(!COSTIN || (!VERT && !DIVIDE), "the fused cost is for the first (row, non-dividing) pass");
    if (agg_gate_skip(armmax, small_variant, small_L, VERT)) return;
    const int chunks = Dp / 128;
    const int N = VERT ? H : W;
    const long long total = (long long)(VERT ? W : H) * chunks * N;
    const int b = RR_BLOCK; // XCD-aware mapping: block b runs on XCD b % 8, each XCD gets a contiguous band of chunks
    const int gw = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || gw >= nwaves) return;
    RR2_LDS_TABLES;
#ifndef RR_EMUL
    if constexpr (COSTIN) {
        for (int i = RR_LANE; i < 766; i += 64) rr2_lds[i] = ci.lut_ad[i];
        rr2_lds[768 + RR_LANE] = ci.lut_census[RR_LANE];
    }
#endif
    long long c = (long long)gw * chunk_len;
    const long long c1 = c + chunk_len < total ? c + chunk_len : total;
#pragma unroll 1
    while (c < c1) {
        const int line = (int)(c / N);
        const int m0 = (int)(c - (long long)line * N);
        const int m1 = (int)((long long)m0 + (c1 - c) < (long long)N ? (long long)m0 + (c1 - c) : (long long)N);
        const int fixed = line / chunks;
        agg_rr2_piece<VERT, DIVIDE, COSTIN>(src, dst, rec, W, H, Dp, L, fixed, line - fixed * chunks, m0, m1, rr2_lds, ci);
        c += m1 - m0;
    }
}
