// k_aggregate_rr.h -- the full-ring aggregation passes with the ring(s) in REGISTERS (included by k_aggregate.hip).
//
// Same semantics as agg_march_body (cross_aggregator.cpp:327-394: every output = sequential f32 sum from 0.0f in the
// order t = -arm .. +arm over the pixel's own arm span, the second pass of an iteration divided by the support
// count), same marching-ring mapping (a wave owns 64 disparities of one image line and marches along it), but built
// around what the round-1 SQ counters showed: the full-ring pass is bound by the CU's SCALAR unit (wave-uniform
// control), not by HBM or LDS.  Per step this body spends ~28 scalar instructions where the first version spent 47:
//   * records are 8 bytes {lob | span<<8 | count<<16, RN(1/count)} and arrive in BULK: every 64 steps lane l loads
//     the record of output base+l (one 8-byte load per 64 steps instead of one scalar-feeding load per step), each
//     step picks its record with two v_readlane -- no per-step record load, no pointer arithmetic, no readfirstlane;
//   * lob = arm_lo + L + 1 is the BIASED arm: in the steady state output m is summed right after entry m + L was
//     pushed, so the first ring slot of its span is (write slot - lob) mod R -- ONE slot counter per ring, and the
//     wrap-around is a single unsigned min: idx = min_u32(idx, idx + R);
//   * the ordered sum is ONE block of 35 indexed adds entered late through a computed jump (96 % of the spans of a
//     natural image are <= 35 entries; longer ones loop): no block loop, no min/shift bookkeeping per 16 entries;
//   * the correctly rounded division by the (wave-uniform) support count is Markstein's sequence on the precomputed
//     correctly rounded reciprocal y = RN(1/c):  q0 = RN(x*y), r = fma(-c, q0, x) [exact], q = fma(r, y, q0) == RN(x/c)
//     -- 4 vector instructions instead of 12, no branch for c == 1 (the sequence returns x itself).  Verified
//     EXHAUSTIVELY against IEEE division for every count 1..65535 and every binary32 significand
//     (tools/markstein_check.c; results scale by powers of two, sums of costs are far from overflow/underflow).
// PAIR (dividing pass of iteration k + first pass of iteration k+1, same direction): the first pass's outputs go into
// a SECOND register ring (v128..v199) instead of HBM; as soon as output m exists, output m-L of the second pass is
// summed from it.  200 VGPRs = 2 waves per SIMD, but a launch moves one volume in and one out for two passes of work:
// natural images take 5 launches (H | V+V | H+H | V+V | H) instead of 8, like short-arm images on the LDS rings.
//
// The control flow below (segments, halos, slots, record positions, tails) is compiled a second time for the CPU with
// RR_EMUL defined (tests/emul/emul_rr.cpp): the primitives that touch registers / memory asynchronously are macros
// with a device form (inline asm) and an emulation form (a modelled VGPR file, with the same M0-relative addressing).
#pragma once

#ifndef ADC_VOL_NT_STR // (adc_internal.h defines the same; this header is also compiled for the CPU, without it)
// Streaming hint for the volume accesses (every element is read once and written once per pass; 1 GB >> every cache):
// the loads / stores of the marching kernels (K4, K5, K6) are marked non-temporal: same-box A/B at 1080p, K4 launch 0.394 -> 0.376 ms
// (0.462 -> 0.445 in the slower clock state of the same box), scanline stage -1 %, right-view WTA -4 % (profiles/r5_ab_nontemporal.txt);
// -DADC_VOL_NT=0 switches it off (tools/build_variant.sh).
#if (!defined(ADC_VOL_NT) || ADC_VOL_NT) && !defined(RR_EMUL) // (the CPU build of this header has no such builtin)
#define ADC_VOL_NT_STR " nt"
#define ADC_VOL_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define ADC_VOL_NT_STR ""
#define ADC_VOL_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
#endif

#define RR_BLK 35
#define RR_PF 8

#ifndef RR_EMUL
// ------------------------------------------------------------------------------------------------ device primitives
#define RR_FN __device__ __forceinline__
typedef uint32_t rr_lanes32; // one VGPR seen across the 64 lanes
#define RR_LANE ((int)threadIdx.x)
#define RR_BLOCK ((int)blockIdx.x)
#define RR_UNIFORM(X) __builtin_amdgcn_readfirstlane(X) /* wave-uniform by construction; a no-op when already scalar */
#define RR_READLANE(X, POS) ((uint32_t)__builtin_amdgcn_readlane((int)(X), (POS)))
#define RR_BITS_TO_F32(U) __uint_as_float(U)
// asynchronous 4-byte load into a register that the compiler does not track (waited for by RR_WAIT_TAKE / RR_DRAIN8)
#define RR_VLOAD(DST, PTR) asm volatile("global_load_dword %0, %1, off" : "=v"(DST) : "v"(PTR) : "memory")
// wait until at most N younger vector-memory operations are outstanding, then read the landed register -- ONE statement,
// so nothing can be hoisted above the wait
#define RR_WAIT_TAKE(DST, SRC, N) asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=&v"(DST) : "v"(SRC), "n"(N) : "memory")
#define RR_WAITALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define RR_DRAIN8(D, S)                                                                                              \
    asm volatile("s_waitcnt vmcnt(0)\n\t"                                                                            \
                 "v_mov_b32 %0, %8\n\tv_mov_b32 %1, %9\n\tv_mov_b32 %2, %10\n\tv_mov_b32 %3, %11\n\t"               \
                 "v_mov_b32 %4, %12\n\tv_mov_b32 %5, %13\n\tv_mov_b32 %6, %14\n\tv_mov_b32 %7, %15"                  \
                 : "=&v"(D[0]), "=&v"(D[1]), "=&v"(D[2]), "=&v"(D[3]), "=&v"(D[4]), "=&v"(D[5]), "=&v"(D[6]), "=&v"(D[7]) \
                 : "v"(S[0]), "v"(S[1]), "v"(S[2]), "v"(S[3]), "v"(S[4]), "v"(S[5]), "v"(S[6]), "v"(S[7])            \
                 : "memory")
// record block: lane LN loads record IDX (an expression of LN) of the line rl
#define RR_REC_LOAD2(NX, NY, IDX)                                                                                    \
    do {                                                                                                             \
        const int LN = lane;                                                                                         \
        const uint2* p_ = rl + (IDX);                                                                                \
        asm volatile("global_load_dword %0, %1, off" : "=v"(NX) : "v"(p_) : "memory");                               \
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(NY) : "v"(p_) : "memory");                      \
    } while (0)
#define RR_REC_LOAD1(NX, IDX)                                                                                        \
    do {                                                                                                             \
        const int LN = lane;                                                                                         \
        const uint2* p_ = rl + (IDX);                                                                                \
        asm volatile("global_load_dword %0, %1, off" : "=v"(NX) : "v"(p_) : "memory");                               \
    } while (0)
#define RR_REC_TAKE3(WAIT, C1, C2, C3, N1, N2, N3)                                                                   \
    asm volatile(WAIT "v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"                                     \
                 : "=&v"(C1), "=&v"(C2), "=&v"(C3) : "v"(N1), "v"(N2), "v"(N3) : "memory")
#define RR_REC_TAKE2(WAIT, C1, C2, N1, N2)                                                                           \
    asm volatile(WAIT "v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(C1), "=&v"(C2) : "v"(N1), "v"(N2) : "memory")
#define RR_WAIT_ALL_STR "s_waitcnt vmcnt(0)\n\t"
#define RR_KEEPALIVE3(A, B, C) asm volatile("s_waitcnt vmcnt(0)" ::"v"(A), "v"(B), "v"(C) : "memory")

#define RR_CLOBBERS1 "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define RR_CLOBBERS2 "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199"
// ring1[slot] = v   (ring 1 = v56..v127, ring 2 = v128..v199; slot is wave-uniform)
RR_FN void rr_push1(int slot, float v)
{
    slot = RR_UNIFORM(slot);
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v56, %1\n\ts_set_gpr_idx_off" ::"s"(slot), "v"(v)
                 : "m0", RR_CLOBBERS1);
}
RR_FN void rr_push2(int slot, float v)
{
    slot = RR_UNIFORM(slot);
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v128, %1\n\ts_set_gpr_idx_off" ::"s"(slot), "v"(v)
                 : "m0", RR_CLOBBERS2);
}
// One run: acc += ring[m - c], ..., ring[m - 1] in this order, c = (12 + 4*RR_BLK - off) / 4 <= RR_BLK entries.  The adds
// name the registers V0-35 .. V0-1; the hardware adds M0 = m to the register number, and the computed jump enters the
// block at position 35 - c, so the registers actually read are v[V0 + m - c] .. v[V0 + m - 1] -- all inside the ring
// (the named registers only appear as encodings).  One add = 4 bytes; 12 = the three 4-byte scalar instructions between
// the value s_getpc returns and the first add.
RR_FN void rr_run1(float& acc, int m, int off)
{
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "s_getpc_b64 vcc\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %2\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n\t"
                 "v_add_f32_e32 %0, v21, %0\n\t"
                 "v_add_f32_e32 %0, v22, %0\n\t"
                 "v_add_f32_e32 %0, v23, %0\n\t"
                 "v_add_f32_e32 %0, v24, %0\n\t"
                 "v_add_f32_e32 %0, v25, %0\n\t"
                 "v_add_f32_e32 %0, v26, %0\n\t"
                 "v_add_f32_e32 %0, v27, %0\n\t"
                 "v_add_f32_e32 %0, v28, %0\n\t"
                 "v_add_f32_e32 %0, v29, %0\n\t"
                 "v_add_f32_e32 %0, v30, %0\n\t"
                 "v_add_f32_e32 %0, v31, %0\n\t"
                 "v_add_f32_e32 %0, v32, %0\n\t"
                 "v_add_f32_e32 %0, v33, %0\n\t"
                 "v_add_f32_e32 %0, v34, %0\n\t"
                 "v_add_f32_e32 %0, v35, %0\n\t"
                 "v_add_f32_e32 %0, v36, %0\n\t"
                 "v_add_f32_e32 %0, v37, %0\n\t"
                 "v_add_f32_e32 %0, v38, %0\n\t"
                 "v_add_f32_e32 %0, v39, %0\n\t"
                 "v_add_f32_e32 %0, v40, %0\n\t"
                 "v_add_f32_e32 %0, v41, %0\n\t"
                 "v_add_f32_e32 %0, v42, %0\n\t"
                 "v_add_f32_e32 %0, v43, %0\n\t"
                 "v_add_f32_e32 %0, v44, %0\n\t"
                 "v_add_f32_e32 %0, v45, %0\n\t"
                 "v_add_f32_e32 %0, v46, %0\n\t"
                 "v_add_f32_e32 %0, v47, %0\n\t"
                 "v_add_f32_e32 %0, v48, %0\n\t"
                 "v_add_f32_e32 %0, v49, %0\n\t"
                 "v_add_f32_e32 %0, v50, %0\n\t"
                 "v_add_f32_e32 %0, v51, %0\n\t"
                 "v_add_f32_e32 %0, v52, %0\n\t"
                 "v_add_f32_e32 %0, v53, %0\n\t"
                 "v_add_f32_e32 %0, v54, %0\n\t"
                 "v_add_f32_e32 %0, v55, %0\n\t"
                 "s_set_gpr_idx_off"
                 : "+v"(acc)
                 : "s"(m), "s"(off)
                 : "m0", "scc", "vcc", RR_CLOBBERS1);
}
RR_FN void rr_run2(float& acc, int m, int off)
{
    asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                 "s_getpc_b64 vcc\n\t"
                 "s_add_u32 vcc_lo, vcc_lo, %2\n\t"
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                 "s_setpc_b64 vcc\n\t"
                 "v_add_f32_e32 %0, v93, %0\n\t"
                 "v_add_f32_e32 %0, v94, %0\n\t"
                 "v_add_f32_e32 %0, v95, %0\n\t"
                 "v_add_f32_e32 %0, v96, %0\n\t"
                 "v_add_f32_e32 %0, v97, %0\n\t"
                 "v_add_f32_e32 %0, v98, %0\n\t"
                 "v_add_f32_e32 %0, v99, %0\n\t"
                 "v_add_f32_e32 %0, v100, %0\n\t"
                 "v_add_f32_e32 %0, v101, %0\n\t"
                 "v_add_f32_e32 %0, v102, %0\n\t"
                 "v_add_f32_e32 %0, v103, %0\n\t"
                 "v_add_f32_e32 %0, v104, %0\n\t"
                 "v_add_f32_e32 %0, v105, %0\n\t"
                 "v_add_f32_e32 %0, v106, %0\n\t"
                 "v_add_f32_e32 %0, v107, %0\n\t"
                 "v_add_f32_e32 %0, v108, %0\n\t"
                 "v_add_f32_e32 %0, v109, %0\n\t"
                 "v_add_f32_e32 %0, v110, %0\n\t"
                 "v_add_f32_e32 %0, v111, %0\n\t"
                 "v_add_f32_e32 %0, v112, %0\n\t"
                 "v_add_f32_e32 %0, v113, %0\n\t"
                 "v_add_f32_e32 %0, v114, %0\n\t"
                 "v_add_f32_e32 %0, v115, %0\n\t"
                 "v_add_f32_e32 %0, v116, %0\n\t"
                 "v_add_f32_e32 %0, v117, %0\n\t"
                 "v_add_f32_e32 %0, v118, %0\n\t"
                 "v_add_f32_e32 %0, v119, %0\n\t"
                 "v_add_f32_e32 %0, v120, %0\n\t"
                 "v_add_f32_e32 %0, v121, %0\n\t"
                 "v_add_f32_e32 %0, v122, %0\n\t"
                 "v_add_f32_e32 %0, v123, %0\n\t"
                 "v_add_f32_e32 %0, v124, %0\n\t"
                 "v_add_f32_e32 %0, v125, %0\n\t"
                 "v_add_f32_e32 %0, v126, %0\n\t"
                 "v_add_f32_e32 %0, v127, %0\n\t"
                 "s_set_gpr_idx_off"
                 : "+v"(acc)
                 : "s"(m), "s"(off)
                 : "m0", "scc", "vcc", RR_CLOBBERS2);
}
#else
// --------------------------------------------------------------------------------------------- emulation primitives
// One lane of one wave at a time: rr_emul.lane / rr_emul.block select it, rr_emul.vgpr models that lane's register file
// (NaN-poisoned outside the rings by the driver), registers seen across lanes (the record blocks) are 64-entry arrays.
#include <assert.h>
#include <math.h>
#include <string.h>
#define RR_FN static inline
struct rr_lanes32 { uint32_t v[64]; };
struct RrEmulState { int lane, block; float vgpr[256]; long reads_lo, reads_hi; };
static RrEmulState rr_emul;
#define RR_LANE (rr_emul.lane)
#define RR_BLOCK (rr_emul.block)
#define RR_UNIFORM(X) (X)
#define RR_READLANE(X, POS) (assert((POS) >= 0 && (POS) < 64), (X).v[(POS)])
RR_FN float rr_bits_to_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define RR_BITS_TO_F32(U) rr_bits_to_f32(U)
#define RR_VLOAD(DST, PTR) ((DST) = *(PTR))
#define RR_WAIT_TAKE(DST, SRC, N) ((DST) = (SRC))
#define RR_WAITALL() ((void)0)
#define RR_DRAIN8(D, S) do { for (int i_ = 0; i_ < 8; i_++) D[i_] = S[i_]; } while (0)
#define RR_REC_LOAD2(NX, NY, IDX)                                                                                    \
    do {                                                                                                             \
        for (int LN = 0; LN < 64; LN++) {                                                                            \
            const uint2* p_ = rl + (IDX);                                                                            \
            (NX).v[LN] = p_->x;                                                                                      \
            (NY).v[LN] = p_->y;                                                                                      \
        }                                                                                                            \
    } while (0)
#define RR_REC_LOAD1(NX, IDX)                                                                                        \
    do {                                                                                                             \
        for (int LN = 0; LN < 64; LN++) (NX).v[LN] = (rl + (IDX))->x;                                                \
    } while (0)
#define RR_REC_TAKE3(WAIT, C1, C2, C3, N1, N2, N3) do { C1 = N1; C2 = N2; C3 = N3; } while (0)
#define RR_REC_TAKE2(WAIT, C1, C2, N1, N2) do { C1 = N1; C2 = N2; } while (0)
#define RR_WAIT_ALL_STR ""
#define RR_KEEPALIVE3(A, B, C) ((void)0)
RR_FN void rr_push1(int slot, float v) { assert(slot >= 0 && 56 + slot < 128); rr_emul.vgpr[56 + slot] = v; }
RR_FN void rr_push2(int slot, float v) { assert(slot >= 0 && 128 + slot < 200); rr_emul.vgpr[128 + slot] = v; }
RR_FN void rr_run_emul(float& acc, int m, int off, int v0)
{
    assert((off - 12) % 4 == 0);
    const int first = (off - 12) / 4; // position of the first add executed
    assert(first >= 0 && first < RR_BLK);
    for (int p = first; p < RR_BLK; p++) {
        const int reg = (v0 - RR_BLK + p) + m; // named register + M0
        assert(reg >= v0 && reg < v0 + 72);
        acc += rr_emul.vgpr[reg];
    }
}
RR_FN void rr_run1(float& acc, int m, int off) { rr_run_emul(acc, m, off, 56); }
RR_FN void rr_run2(float& acc, int m, int off) { rr_run_emul(acc, m, off, 128); }
#endif

// acc += ring[idx], ring[idx+1], ... (cnt >= 1 entries, no wrap), in this order
RR_FN float rr_sum1(float acc, int idx, int cnt)
{
    idx = RR_UNIFORM(idx);
    cnt = RR_UNIFORM(cnt);
    while (cnt > RR_BLK) { // rare: 4 % of the spans of a natural image
        rr_run1(acc, idx + RR_BLK, 12);
        idx += RR_BLK;
        cnt -= RR_BLK;
    }
    rr_run1(acc, idx + cnt, 12 + 4 * RR_BLK - 4 * cnt);
    return acc;
}
RR_FN float rr_sum2(float acc, int idx, int cnt)
{
    idx = RR_UNIFORM(idx);
    cnt = RR_UNIFORM(cnt);
    while (cnt > RR_BLK) {
        rr_run2(acc, idx + RR_BLK, 12);
        idx += RR_BLK;
        cnt -= RR_BLK;
    }
    rr_run2(acc, idx + cnt, 12 + 4 * RR_BLK - 4 * cnt);
    return acc;
}

// x / c, correctly rounded, from y = RN(1/c) (Markstein; see the header comment).  c and y are wave-uniform.
RR_FN float rr_divide(float x, float cf, float y)
{
    const float q0 = x * y;
    const float r = __builtin_fmaf(-cf, q0, x);
    return __builtin_fmaf(r, y, q0);
}

// Which launches of a pass do any work -- the launch gate shared by every marching body (agg_march_body, agg_rr_body,
// agg_rr2_body).  armmax[0] / [1] = longest horizontal / vertical arm of the image (k_build_arms), written before the
// aggregation starts and not changed by it.
//   small_variant < 0   the host launched exactly the variant that applies: run.
//   0 / 1               debug surface, two launches per pass: the full-ring (0) / small-ring (1) variant runs when the arms
//                       of THIS direction do not / do fit small_L.
//   2                   ring depth assumed from the previous Match of the handle: a longer arm skips the pass and raises
//                       armmax[3]; adc_wait redoes the aggregation and what follows with the full ring.
//   3 / 4               TWO PLANS enqueued back to back (streams that alternate between short-arm and long-arm images): every
//                       launch of the assumed-depth plan (3) runs when BOTH directions fit their assumed depths
//                       (small_L = depth_h | depth_v << 16; 0x7fff = that direction runs the full ring anyway), every launch
//                       of the full-ring plan (4) when they do not -- exactly one of the two plans does the work, no host
//                       round trip, no redo.
RR_FN bool agg_gate_skip(const int* armmax, int small_variant, int small_L, bool vert)
{
    if (small_variant < 0) return false;
    if (small_variant >= 3) {
        const bool fits = armmax[0] <= (small_L & 0xffff) && armmax[1] <= ((small_L >> 16) & 0xffff);
        return (small_variant == 3) != fits;
    }
    const bool fits_small = armmax[vert ? 1 : 0] <= small_L;
    if (small_variant == 2) {
#ifndef RR_EMUL
        if (!fits_small && RR_BLOCK == 0 && RR_LANE == 0) const_cast<int*>(armmax)[3] = 1;
#endif
        return !fits_small;
    }
    return (small_variant != 0) != fits_small;
}

template <bool VERT, bool DIVIDE, bool PAIR>
RR_FN void agg_rr_body(const float* __restrict__ src, float* __restrict__ dst,
                       const uint2* __restrict__ rec, // {lob | span<<8 | count<<16, RN(1/count)}, line-major
                       int W, int H, int Dp, int L, int seg_len, int nseg, int per_xcd,
                       const int* __restrict__ armmax, int small_variant, int small_L, float* __restrict__ sink)
{
    static_assert(!PAIR || DIVIDE, "a pair = dividing pass + the following non-dividing pass");
    if (agg_gate_skip(armmax, small_variant, small_L, VERT)) return;
    const int R = 2 * L + 1;
    const int lane = RR_LANE;
    const int chunks = Dp / 64;
    const int N = VERT ? H : W;
    const int nlines = (VERT ? W : H) * chunks;
    const int b = RR_BLOCK; // XCD-aware mapping: block b runs on XCD b % 8, each XCD gets a contiguous band of lines
    const int gw = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || gw >= nlines * nseg) return;
    const int seg = gw / nlines;
    const int line = gw - seg * nlines;
    const int fixed = line / chunks;
    const int chunk = line - fixed * chunks;

    // [s0, s1) = outputs this wave delivers; [m0, m1) = outputs of the (first) pass it computes; [lo, hi) = entries it reads
    const int s0 = seg * seg_len;
    const int s1 = adc_imin(N, s0 + seg_len);
    if (s0 >= s1) return;
    const int m0 = PAIR ? adc_imax(0, s0 - L) : s0;
    const int m1 = PAIR ? adc_imin(N, s1 + L) : s1;
    const int lo = adc_imax(0, m0 - L);
    const int hi = adc_imin(N, m1 + L);

    const long long pix_step = VERT ? (long long)W : 1LL;
    const long long pix0 = VERT ? (long long)fixed : (long long)fixed * W;
    const long long fstep = pix_step * Dp;
    const float* sp = src + pix0 * Dp + chunk * 64 + lane;
    float* dpn = dst + pix0 * Dp + chunk * 64 + lane + (long long)s0 * fstep; // outputs leave in increasing order from s0
    const uint2* rl = rec + (long long)fixed * N;                              // records of this line
    float* const sinkp = sink + ((b & 1023) * 64 + lane);                      // store target of halo steps (PAIR)

    // ---- record streams.  Emit index te = tbase + pos: first-pass output m0 + te uses record m0 + te (stream 1) and,
    // PAIR, the second-pass output m0 - L + te that becomes ready at the same step uses record m0 - L + te (stream 2):
    // ONE position counter.  Lane l of a block holds the record of emit index tbase + l; the next block is in flight
    // while the current one is used.  The second-pass output exists from te = t2 on (s >= s0); t2rel = t2 - tbase is
    // what the lane position is compared with.
    rr_lanes32 c1x, c1y, n1x, n1y, c2x, n2x;
    int pos = 0, tbase = 0;
    const int t2 = s0 + L - m0;
    int t2rel = t2;
#define RR_REC_ISSUE(TB)                                                                                             \
    do {                                                                                                             \
        RR_REC_LOAD2(n1x, n1y, adc_imin(m0 + (TB) + LN, N - 1));                                                     \
        if constexpr (PAIR) RR_REC_LOAD1(n2x, adc_imax(0, adc_imin(m0 - L + (TB) + LN, N - 1)));                     \
    } while (0)
// take over the block in flight (WAIT = "" inside the steady state: it was issued >= 64 steps, i.e. >= 128 younger
// vector-memory operations ago, and the counter tracks at most 63)
#define RR_REC_TAKE(WAIT)                                                                                            \
    do {                                                                                                             \
        if constexpr (PAIR) RR_REC_TAKE3(WAIT, c1x, c1y, c2x, n1x, n1y, n2x);                                        \
        else RR_REC_TAKE2(WAIT, c1x, c1y, n1x, n1y);                                                                 \
    } while (0)
#define RR_REC_ADVANCE(WAIT)                                                                                         \
    if (pos == 64) {                                                                                                 \
        RR_REC_TAKE(WAIT);                                                                                           \
        tbase += 64;                                                                                                 \
        t2rel -= 64;                                                                                                 \
        pos = 0;                                                                                                     \
        RR_REC_ISSUE(tbase + 64);                                                                                    \
    }
    RR_REC_ISSUE(0);
    RR_REC_TAKE(RR_WAIT_ALL_STR);
    RR_REC_ISSUE(64);

    // Ring slots.  Entry e sits in ring-1 slot (e - lo) mod R; w1 = slot of the next entry.  In the steady state output m
    // is summed right after entry m + L was pushed, i.e. with w1 = slot(m) + L + 1: the records carry the BIASED arm
    // lob = arm_lo + L + 1, so the first slot of the span is simply (w1 - lob) mod R -- no second counter.  Same for
    // ring 2 (first-pass output m in slot (m - m0) mod R, w2 = slot of the next one): second-pass output s is summed
    // right after first-pass output s + L was pushed.  Where outputs leave without a push (image end) the slot is
    // computed from the indices (RR_SLOT1 / RR_SLOT2).
    int w1 = 0;
    int w2 = 0;
#define RR_SLOT1(M) (((M) + L + 1 - lo) % R)
#define RR_SLOT2(S) (((S) + L + 1 - m0) % R)
#define RR_WRAP_INC(S) do { (S) = (S) + 1 == R ? 0 : (S) + 1; } while (0)
#define RR_PUSH(VAL) do { rr_push1(w1, (VAL)); RR_WRAP_INC(w1); } while (0)

// second-pass output s = m0 - L + te; W2 = ring-2 slot of first-pass output s + L + 1
#define RR_EMIT2(POS, W2)                                                                                            \
    do {                                                                                                             \
        const uint32_t q_ = RR_READLANE(c2x, (POS));                                                                 \
        const int blo_ = (int)(q_ & 255u), bn_ = (int)((q_ >> 8) & 255u);                                            \
        uint32_t i2_ = (uint32_t)((W2) - blo_);                                                                      \
        i2_ = i2_ < i2_ + (uint32_t)R ? i2_ : i2_ + (uint32_t)R; /* min_u32: wraps a negative index */                \
        const int k1_ = adc_imin(bn_, R - (int)i2_);                                                                 \
        float a2_ = rr_sum2(0.0f, (int)i2_, k1_);                                                                    \
        if (bn_ > k1_) a2_ = rr_sum2(a2_, 0, bn_ - k1_);                                                             \
        *dpn = a2_;                                                                                                  \
        dpn += fstep;                                                                                                \
    } while (0)

// first-pass output m = m0 + te (record at lane POS of the current block); W1 = ring-1 slot of entry m + L + 1;
// IN_LOOP: exactly one store per call
#define RR_EMIT(POS, IN_LOOP, W1)                                                                                    \
    do {                                                                                                             \
        const uint32_t r_ = RR_READLANE(c1x, (POS));                                                                 \
        const int alo_ = (int)(r_ & 255u), an_ = (int)((r_ >> 8) & 255u);                                            \
        uint32_t i1_ = (uint32_t)((W1) - alo_);                                                                      \
        i1_ = i1_ < i1_ + (uint32_t)R ? i1_ : i1_ + (uint32_t)R;                                                     \
        const int n1_ = adc_imin(an_, R - (int)i1_);                                                                 \
        float acc_ = rr_sum1(0.0f, (int)i1_, n1_); /* t = -arm .. +arm */                                             \
        if (an_ > n1_) acc_ = rr_sum1(acc_, 0, an_ - n1_); /* wrapped part */                                         \
        if constexpr (DIVIDE) {                                                                                      \
            const float y_ = RR_BITS_TO_F32(RR_READLANE(c1y, (POS)));                                                \
            acc_ = rr_divide(acc_, (float)(r_ >> 16), y_); /* cross_aggregator.cpp:389 */                            \
        }                                                                                                            \
        if constexpr (PAIR) {                                                                                        \
            rr_push2(w2, acc_);                                                                                      \
            RR_WRAP_INC(w2);                                                                                         \
            if ((POS) >= t2rel) RR_EMIT2(POS, w2); /* s = m0 - L + te >= s0 (and < s1 because m < m1 <= s1 + L) */     \
            else if (IN_LOOP) *sinkp = acc_; /* keeps the vector-memory operation count of a step constant */         \
        } else {                                                                                                     \
            *dpn = acc_;                                                                                             \
            dpn += fstep;                                                                                            \
        }                                                                                                            \
    } while (0)

    // ---- phase A: entries lo .. jB-1 precede the first output's look-ahead (no output yet)
    const int jB = adc_imin(hi, m0 + L);
    for (int j = lo; j < jB; j += RR_PF) {
        float tv[RR_PF];
#pragma unroll
        for (int u = 0; u < RR_PF; u++) tv[u] = sp[(long long)adc_imin(j + u, jB - 1) * fstep];
#pragma unroll
        for (int u = 0; u < RR_PF; u++)
            if (j + u < jB) RR_PUSH(tv[u]);
    }
    // ---- phase B: entry j arrives, first-pass output m = j - L leaves (PAIR: and second-pass output m - L)
    RR_WAITALL(); // the manual vmcnt bookkeeping starts from an empty queue
    int j = jB;
    if (j + 2 * RR_PF <= hi) {
        float pf[RR_PF];
        const float* spn = sp + (long long)j * fstep;
#pragma unroll
        for (int u = 0; u < RR_PF; u++) {
            RR_VLOAD(pf[u], spn);
            spn += fstep;
        }
// one step; WAITN = vector-memory operations younger than slot U's load that may stay in flight
#define RR_STEP(U, WAITN)                                                                                            \
    do {                                                                                                             \
        float v_;                                                                                                    \
        RR_WAIT_TAKE(v_, pf[U], WAITN);                                                                              \
        RR_VLOAD(pf[U], spn);                                                                                        \
        spn += fstep;                                                                                                \
        RR_PUSH(v_);                                                                                                 \
        RR_EMIT(pos + (U), true, w1); /* exactly one compiler-issued vector-memory operation (a store) */            \
    } while (0)
        // first block: younger operations = prologue loads of slots U+1.. (1 each) + 2 per finished step
        static_assert(RR_PF == 8, "the peeled first block is written for RR_PF == 8");
        RR_STEP(0, 7); RR_STEP(1, 8); RR_STEP(2, 9); RR_STEP(3, 10); RR_STEP(4, 11); RR_STEP(5, 12); RR_STEP(6, 13); RR_STEP(7, 14);
        j += RR_PF;
        pos += RR_PF;
        // steady state: younger operations = the reissue step's own store + 2 per younger step = 1 + 2*(RR_PF-1) = 15;
        // wait for <= 14 (one stricter; the bulk record loads only make it more conservative)
        for (; j + 2 * RR_PF <= hi; j += RR_PF) {
            RR_REC_ADVANCE("");
            RR_STEP(0, 14); RR_STEP(1, 14); RR_STEP(2, 14); RR_STEP(3, 14); RR_STEP(4, 14); RR_STEP(5, 14); RR_STEP(6, 14); RR_STEP(7, 14);
            pos += RR_PF;
        }
#undef RR_STEP
        // drain: the RR_PF entries still in flight are entries j .. j+RR_PF-1 (all < hi)
        RR_REC_ADVANCE(RR_WAIT_ALL_STR);
        float df[RR_PF];
        RR_DRAIN8(df, pf);
#pragma unroll
        for (int u = 0; u < RR_PF; u++) {
            RR_PUSH(df[u]);
            RR_EMIT(pos + u, false, w1);
        }
        j += RR_PF;
        pos += RR_PF;
    }
    // ---- tail of phase B (< 2*RR_PF entries left) and phase C (outputs whose look-ahead ends beyond the last entry, at
    // the image end): one generic loop; the remaining entries are loaded up front
    {
        float tv[2 * RR_PF];
#pragma unroll
        for (int u = 0; u < 2 * RR_PF; u++) tv[u] = sp[(long long)adc_imin(j + u, hi - 1) * fstep];
        const int nm = m1 - m0; // first-pass outputs in total; tbase + pos of them are done
        int u = 0;
#pragma unroll 1
        while (tbase + pos < nm) {
            const int m = m0 + tbase + pos;
            if (j < hi && j <= m + L) { // output m still waits for entry j
                float v = tv[0];
#pragma unroll
                for (int q = 1; q < 2 * RR_PF; q++) v = u == q ? tv[q] : v;
                RR_PUSH(v);
                u++;
                j++;
                if (j < hi && j <= m + L) continue;
            }
            RR_REC_ADVANCE(RR_WAIT_ALL_STR);
            RR_EMIT(pos, false, RR_SLOT1(m));
            pos++;
        }
        // ---- PAIR: second-pass outputs whose look-ahead ends beyond the last first-pass output (image end)
        if constexpr (PAIR) {
            const int te_end = t2 + (s1 - s0);
#pragma unroll 1
            while (tbase + pos < te_end) {
                RR_REC_ADVANCE(RR_WAIT_ALL_STR);
                if (pos >= t2rel) RR_EMIT2(pos, RR_SLOT2(m0 - L + tbase + pos));
                pos++;
            }
        }
    }
    // the record block in flight is never used: wait for it before its registers die (a late-landing load would
    // otherwise overwrite whatever the compiler put there)
    RR_KEEPALIVE3(n1x, n1y, n2x);
#undef RR_REC_ISSUE
#undef RR_REC_TAKE
#undef RR_REC_ADVANCE
#undef RR_WRAP_INC
#undef RR_SLOT1
#undef RR_SLOT2
#undef RR_PUSH
#undef RR_EMIT
#undef RR_EMIT2
}
