// k_aggregate.hip -- K4 cross-based cost aggregation (the roofline kernel of the pipeline).
//
// Replaces CrossAggregator::{Aggregate, AggregateInArms} (cross_aggregator.cpp:89-118,327-394):
// 4 iterations (H-first, V-first, H-first, V-first); per iteration two 1-D passes over the whole
// volume, each output = sequential f32 sum (from 0.0f, in the order t = -arm ... +arm, which is
// observable in the WTA result -- SURVEY.md A.4) of the inputs inside the pixel's own arm; the second
// pass of an iteration divides by the support count.
//
// MI355X mapping ("marching ring"):
//   * layout [y][x][Dp], lanes = disparities: a wave owns 64 consecutive floats (256 B) of every
//     pixel of ONE image line (a row for the H pass, a column for the V pass) and marches along it.
//     Arms and support counts are wave-uniform (scalar loads, no divergence).
//   * every input element is read from HBM exactly once per pass (+ the 34-pixel halo at segment
//     ends): the last 2L+1 line entries live in an LDS ring (256 B x (2L+1) per wave, each lane
//     only ever reads what it wrote itself, so no barriers), the next AGG_PF entries are in flight
//     in registers (software prefetch) -- that is what keeps ~10 MB of loads outstanding chip-wide.
//   * algorithmic bytes per pass: read V + write V (+4 B/pixel arms, +2 B/pixel counts), V = 4*W*H*Dp.
#include "adc_internal.h"
#include "adc_device_fn.h"
#include <mutex>
#include "k_aggregate_rr.h"
#include "k_aggregate_rr2.h"

// ---------------------------------------------------------------------------------- marching ring
// One 64-lane workgroup (= one wave) per line segment.
//   * LDS per wave = (2L+1) x 256 B (17.25 KiB at L = 34) => 9 waves per CU, 2304 chip-wide: all 2160
//     H-lines of a 1080p / D=128 volume run in ONE round;
//   * the next AGG_PF entries and their packed {arm_lo, arm_hi, count} records are in flight in
//     registers.  These prefetch loads are issued from inline asm and waited for with a hand-counted
//     s_waitcnt vmcnt: hipcc's loop-carried vmcnt model otherwise drains the queue to a depth of ~3 steps
//     (tools/ubench/agg_lean.hip vs agg_asm.hip).  VMEM ops per steady-state step, in program order:
//     [data load][record load] ... [output store]; vmcnt retires in order on gfx9-family hardware.
// Measured (tools/ubench): the same load/store structure without the ordered sum streams at the
// float4-copy rate (4.8-5.4 TB/s on MI355X); the full step is bound by its dependent chain
// (LDS write -> LDS reads -> ordered adds -> store) at 9 waves per CU.
#define AGG_PF 8
#ifndef ADC_K4_TAILLOOP
#define ADC_K4_TAILLOOP 0 // A/B switch: 1 = element-wise tail loop in the compact span sum
#endif
#ifndef ADC_K4_DIAG
#define ADC_K4_DIAG 0 // timing experiments only (wrong results): 1 = no span sums, 2 = no division, 3 = spans capped at 4
#endif

// ordered partial sum over cnt consecutive ring entries starting at q (lane-private column, stride 64 floats)
__device__ __forceinline__ float agg_run(float acc, const float* q, int cnt)
{
    while (cnt >= 8) {
        const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320], t6 = q[384], t7 = q[448];
        acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
        q += 512;
        cnt -= 8;
    }
    switch (cnt) { // exact-length tail (reads only what it adds)
    case 7: { const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320], t6 = q[384];
              acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; } break;
    case 6: { const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256], t5 = q[320];
              acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; } break;
    case 5: { const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192], t4 = q[256];
              acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; } break;
    case 4: { const float t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192]; acc += t0; acc += t1; acc += t2; acc += t3; } break;
    case 3: { const float t0 = q[0], t1 = q[64], t2 = q[128]; acc += t0; acc += t1; acc += t2; } break;
    case 2: { const float t0 = q[0], t1 = q[64]; acc += t0; acc += t1; } break;
    case 1: acc += q[0]; break;
    default: break;
    }
    return acc;
}

// Compact form for the small-ring kernels (spans of at most 2*8+1 entries): same ordered sum, a fraction of the code --
// the fully unrolled form above is inlined four times per step of the pair kernel and made its steady-state loop
// ~38 KB of instructions.
// (V = float, or a pair of floats when a lane owns two disparities: the two sums are independent chains)
typedef float agg_f2 __attribute__((ext_vector_type(2)));
template <int VPL> struct AggT { typedef float V; };
template <> struct AggT<2> { typedef agg_f2 V; };
template <class V>
__device__ __forceinline__ V agg_run_compact(V acc, const V* q, int cnt)
{
#pragma nounroll
    for (; cnt >= 4; cnt -= 4, q += 256) {
        const V t0 = q[0], t1 = q[64], t2 = q[128], t3 = q[192];
        acc += t0; acc += t1; acc += t2; acc += t3;
    }
#if ADC_K4_TAILLOOP
#pragma nounroll
    for (; cnt > 0; cnt--, q += 64) acc += q[0];
#else
    // the last 1..3 entries with all their reads in flight at once (an element-wise loop pays one LDS round trip and
    // four scalar instructions per entry; on short-arm images almost every span ends here)
    if (cnt == 3) {
        const V t0 = q[0], t1 = q[64], t2 = q[128];
        acc += t0; acc += t1; acc += t2;
    } else if (cnt == 2) {
        const V t0 = q[0], t1 = q[64];
        acc += t0; acc += t1;
    } else if (cnt == 1) {
        acc += q[0];
    }
#endif
    return acc;
}
template <bool SMALL_, class V>
__device__ __forceinline__ V agg_sum(V acc, const V* q, int cnt)
{
    if constexpr (SMALL_) return agg_run_compact<V>(acc, q, cnt);
    else {
        static_assert(SMALL_ || sizeof(V) == sizeof(float), "the full ring is used with one disparity per lane");
        return agg_run(acc, q, cnt);
    }
}

// SMALL: the small-ring launch (own kernel name in profiles; compact summation code).
// COSTIN (first pass of the production pipeline, rows, non-dividing): there is no input volume -- each entry of the
// line, i.e. the AD-Census matching cost of pixel (x, y) for this wave's 64 disparities (cost_computor.cpp:82-121), is
// computed in registers.  Lane l owns disparity d = d_first + l and needs the right-image pixel of column x - d: as the
// wave marches in x that column moves one lane per step, so the window {bgrx, census} is a DPP wave_shr:1 shift with
// the one new column (x - d_first) entering at lane 0; the new column and the left pixel of x arrive as uniform
// 16-byte records (k_cost_records, rows padded with out-of-image markers), prefetched like the arm records.
// AD = v_sad_u8 of the packed colours, Hamming = two v_bcnt on the xor of the census words, cost = A[ad] - C[hm] from
// the host-built tables (in LDS behind the ring) -- bit-identical to k_cost.  Saves writing V and reading it back.
// (struct AggCostIn: k_aggregate_rr2.h)
typedef unsigned long long agg_u64;

// PAIR: TWO consecutive passes of the same direction in one launch (the dividing second pass of an iteration and the
// non-dividing first pass of the next one: V1+V2, H2+H3, V3+V4).  The first pass's outputs are not stored but pushed
// into a second ring (and their arm records into a small record ring); as soon as output m of the first pass
// exists, output m-L of the second pass can be summed from the second ring -- same ordered sums, same division, so
// the result is bit-identical, but the intermediate volume never travels to HBM and back (8 passes -> 5 launches).
// Needs two rings per wave, so it is used with the small ring only (armmax <= small_L, decided on the host).
// VPL = disparities per lane (1, or 2 with the small ring when Dp is a multiple of 128): with two, a wave owns 512
// contiguous bytes of every pixel of its line, all wave-uniform work of a step (records, ring slots, branches, waits) is
// shared by twice the data and the two sums of a lane are independent chains.
// REGRING (full ring, plain pass): the ring lives in REGISTERS -- VGPRs v56..v127, outside the range the compiler may
// allocate for this kernel (amdgpu_num_vgpr) -- and is addressed with the VGPR index mode (s_set_gpr_idx_on: the
// ring slots are wave-uniform, M0 holds the index).  A span entry costs one indexed v_add_f32 and one s_add on M0
// instead of an LDS round trip, there is no LDS allocation at all, and 128 VGPRs allow 16 waves per CU where the
// 17 KiB LDS ring allowed 9.  Same entries, same order, same adds: bit-identical (tools/ubench/regring.hip).
#define AGG_RING_V0 56
#define AGG_RING_REGS 72
#define AGG_RING_CLOBBERS                                                                                             \
    "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",  \
        "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88",    \
        "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103",       \
        "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117",    \
        "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
// ring[slot] = v
__device__ __forceinline__ void agg_reg_push(int slot, float v)
{
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v56, %1\n\ts_set_gpr_idx_off" ::"s"(slot), "v"(v)
                 : "m0", AGG_RING_CLOBBERS);
}
// acc += ring[idx], ring[idx+1], ... (cnt entries, no wrap), in this order: blocks of 16 indexed adds entered late
// through a computed jump.  The index register M0 is set ONCE per block: the adds name the registers v40..v55 and the
// hardware adds M0 = idx + c to the register number, so block position p reads v[40 + p + idx + c]; entering at position
// 16 - c makes that v[56 + idx] ... v[56 + idx + c - 1].  (v40..v55 only appear as encodings: every register actually
// read lies in the ring.)  One add = 4 bytes; 12 = the three scalar instructions between the value s_getpc returns and
// the first add.  A first version advanced M0 with an s_add per entry: 13 more scalar instructions per step on a
// kernel whose scalar unit is saturated (SQ counters).
__device__ __forceinline__ float agg_reg_sum(float acc, int idx, int cnt)
{
    static_assert(AGG_RING_V0 == 56, "the add block below names v40..v55 = AGG_RING_V0 - 16 + position");
    while (cnt > 0) {
        const int c = cnt < 16 ? cnt : 16;
        const int off = 12 + 4 * (16 - c);
        const int m = idx + c;
        asm volatile("s_set_gpr_idx_on %1, gpr_idx(SRC0)\n\t"
                     "s_getpc_b64 vcc\n\t"
                     "s_add_u32 vcc_lo, vcc_lo, %2\n\t"
                     "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"
                     "s_setpc_b64 vcc\n\t"
                     "v_add_f32_e32 %0, v40, %0\n\tv_add_f32_e32 %0, v41, %0\n\tv_add_f32_e32 %0, v42, %0\n\tv_add_f32_e32 %0, v43, %0\n\t"
                     "v_add_f32_e32 %0, v44, %0\n\tv_add_f32_e32 %0, v45, %0\n\tv_add_f32_e32 %0, v46, %0\n\tv_add_f32_e32 %0, v47, %0\n\t"
                     "v_add_f32_e32 %0, v48, %0\n\tv_add_f32_e32 %0, v49, %0\n\tv_add_f32_e32 %0, v50, %0\n\tv_add_f32_e32 %0, v51, %0\n\t"
                     "v_add_f32_e32 %0, v52, %0\n\tv_add_f32_e32 %0, v53, %0\n\tv_add_f32_e32 %0, v54, %0\n\tv_add_f32_e32 %0, v55, %0\n\t"
                     "s_set_gpr_idx_off"
                     : "+v"(acc)
                     : "s"(m), "s"(off)
                     : "m0", "scc", "vcc", AGG_RING_CLOBBERS);
        idx += c;
        cnt -= c;
    }
    return acc;
}

template <bool VERT, bool DIVIDE, bool SMALL, bool COSTIN, bool PAIR, int VPL, bool REGRING>
__device__ __forceinline__ void agg_march_body(const float* __restrict__ src, float* __restrict__ dst,
                                               const uint32_t* __restrict__ rec, // {lo, hi, count16} per pixel, line-major
                                               int W, int H, int Dp, int L, int seg_len, int nseg, int per_xcd,
                                               const int* __restrict__ armmax, int small_variant, int small_L,
                                               const AggCostIn& ci)
{
    static_assert(!REGRING || (!SMALL && !PAIR && VPL == 1), "register ring here: the fused-cost first pass (plain passes and pairs: k_aggregate_rr.h)");
    static_assert(!COSTIN || (!VERT && !DIVIDE), "the fused cost is for the first (row, non-dividing) pass");
    static_assert(!PAIR || (DIVIDE && !COSTIN), "a fused pair = dividing pass + the following non-dividing pass");
    static_assert(VPL == 1 || (VPL == 2 && SMALL && !COSTIN), "two disparities per lane: small ring, no fused cost");
    typedef typename AggT<VPL>::V V;
    const V vzero = (V)(0.0f);
    // small_variant >= 0 (host does not know the arms, debug path): two launches per pass, the window depth follows
    // the data.  When no arm of this direction exceeds small_L (e.g. noise-like images) the small-ring variant runs at
    // 32 waves/CU and the full-ring variant exits at once, otherwise the other way round (armmax[0] = max horizontal
    // arm, armmax[1] = max vertical arm, from k_build_arms).  small_variant < 0: the host has read armmax and
    // launches only the variant that applies.
    // small_variant == 2: the host ASSUMED (from the previous Match of the handle) that no arm of this direction exceeds
    // small_L = the ring depth of this launch; when the assumption is wrong the pass is skipped and armmax[3] raised --
    // adc_wait then redoes the Match with the full ring.
    // small_variant 3 / 4: two plans enqueued back to back, exactly one of them works (agg_gate_skip, k_aggregate_rr.h)
    if (agg_gate_skip(armmax, small_variant, small_L, VERT)) return;
    extern __shared__ __attribute__((aligned(16))) float ring_all[];
    const int R = 2 * L + 1;
    const int lane = threadIdx.x;
    V* ring = reinterpret_cast<V*>(ring_all) + lane; // entry s at ring[s*64]; each lane only reads what it wrote

    const int chunks = Dp / (64 * VPL);         // (64*VPL)-float chunks per pixel
    const int N = VERT ? H : W;                 // length of a line
    const int nlines = (VERT ? W : H) * chunks; // independent lines
    // XCD-aware mapping: hardware places block b on XCD b % 8; give each XCD a contiguous band of lines
    const int b = (int)blockIdx.x;
    const int gw = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || gw >= nlines * nseg) return;
    const int seg = gw / nlines;
    const int line = gw - seg * nlines;
    const int fixed = line / chunks; // x (V pass) or y (H pass)
    const int chunk = line - fixed * chunks;

    // [s0, s1) = the outputs this wave delivers; [m0, m1) = the outputs of the (first) pass it has to compute for them
    const int s0 = seg * seg_len;
    const int s1 = adc_imin(N, s0 + seg_len);
    if (s0 >= s1) return;
    const int m0 = PAIR ? adc_imax(0, s0 - L) : s0;
    const int m1 = PAIR ? adc_imin(N, s1 + L) : s1;
    const int lo = adc_imax(0, m0 - L);
    const int hi = adc_imin(N, m1 + L);

    // element m of this line
    const long long pix_step = VERT ? (long long)W : 1LL;
    const long long pix0 = VERT ? (long long)fixed : (long long)fixed * W;
    const long long fstep = pix_step * Dp;
    const float* sp = src + pix0 * Dp + chunk * (64 * VPL) + lane * VPL;
    float* dp = dst + pix0 * Dp + chunk * (64 * VPL) + lane * VPL;
    const uint32_t* rp = rec + (long long)fixed * N; // records of this line, contiguous along m

    // ---- fused cost state (COSTIN)
    float* lutA = ring_all + (REGRING ? 0 : R * 64); // A[766] then C[64] behind the LDS ring (COSTIN: VPL == 1)
    float* lutC = lutA + 768;
    uint32_t wB = 0, wC0 = 0, wC1 = 0;          // this lane's right-image pixel {bgrx, census} for the current entry
    const uint4* rrow = nullptr;                 // rrow[x] = right record of column x - d_first (lane 0's column)
    const uint4* lrow = nullptr;
    bool pad_lane = false;
    if constexpr (COSTIN) {
        for (int i = lane; i < 766; i += 64) lutA[i] = ci.lut_ad[i];
        lutC[lane] = ci.lut_census[lane];
        const int d_first = chunk * 64 + ci.dmin;
        pad_lane = chunk * 64 + lane >= ci.D;
        rrow = ci.rrec + (size_t)fixed * ci.rpitch + ci.padl - d_first;
        lrow = ci.lrec + (size_t)fixed * W;
        // window of the entry BEFORE the first one (column lo-1-d); the first real step shifts it into place
        const uint4* rbase = ci.rrec + (size_t)fixed * ci.rpitch;
        int gi = ci.padl - d_first + lo - 1 - lane;
        gi = gi < 0 ? 0 : (gi >= ci.rpitch ? ci.rpitch - 1 : gi); // index 0 is a marker column (padl >= 1)
        const uint4 g0 = rbase[gi];
        wB = g0.x; wC0 = g0.y; wC1 = g0.z;
    }
// matching cost of the next entry: (RB, RC0, RC1) = new right pixel {bgrx, census lo, census hi} entering at lane 0,
// (LB, LC0, LC1) = left pixel of the entry (both wave-uniform)
#define AGG_COST(RB, RC0, RC1, LB, LC0, LC1, OUT)                                                                  \
    do {                                                                                                           \
        wB = (uint32_t)__builtin_amdgcn_update_dpp((int)(RB), (int)wB, 0x138, 0xf, 0xf, false);                    \
        wC0 = (uint32_t)__builtin_amdgcn_update_dpp((int)(RC0), (int)wC0, 0x138, 0xf, 0xf, false);                 \
        wC1 = (uint32_t)__builtin_amdgcn_update_dpp((int)(RC1), (int)wC1, 0x138, 0xf, 0xf, false);                 \
        const uint32_t ad_ = __builtin_amdgcn_sad_u8(wB, (uint32_t)(LB), 0u);                                      \
        const uint32_t hm_ = __popc(wC0 ^ (uint32_t)(LC0)) + __popc(wC1 ^ (uint32_t)(LC1));                        \
        float cv_ = lutA[ad_ < 766u ? ad_ : 765u] - lutC[hm_ & 63u]; /* == ((1 - ea) + 1) - ec, cost_computor.cpp:117 */ \
        cv_ = wB == 0xFFFFFFFFu ? 1.0f : cv_; /* right pixel outside the image (:101-104) */                        \
        OUT = pad_lane ? 0.0f : cv_;                                                                               \
    } while (0)

    int slot_w = 0;          // ring slot of the next entry to be written
    int slot_m = m0 - lo;    // ring slot of entry m0 (<= L < R)
    float* dpn = dp + (long long)s0 * fstep; // outputs leave in increasing order, starting at s0
    // ---- second stage (PAIR): ring of first-pass outputs + their records, behind the first ring
    V* ring2 = reinterpret_cast<V*>(ring_all) + R * 64 + lane;
    uint32_t* recring = reinterpret_cast<uint32_t*>(reinterpret_cast<V*>(ring_all) + 2 * R * 64);
    int slot2_w = 0;           // slot of the next first-pass output to be written
    int slot2_s = s0 - m0;     // slot of first-pass output s0 (the next second-pass output)
    int mcur = m0;             // index of the next first-pass output
// second-pass output: ordered sum of first-pass outputs over the pixel's own arm span, no division
#define AGG_EMIT2()                                                                               \
    do {                                                                                          \
        const uint32_t r2_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)recring[slot2_s]);     \
        const int b_lo_ = (int)(r2_ & 255u), b_hi_ = (int)((r2_ >> 8) & 255u);                    \
        int i2_ = slot2_s - b_lo_;                                                                \
        if (i2_ < 0) i2_ += R;                                                                    \
        const int k_ = b_lo_ + b_hi_ + 1;                                                         \
        V acc2_;                                                                                  \
        if (k_ == 1) {                                                                            \
            acc2_ = vzero + ring2[i2_ * 64];                                                      \
        } else {                                                                                  \
            const int k1_ = adc_imin(k_, R - i2_);                                                \
            acc2_ = agg_sum<SMALL, V>(vzero, ring2 + i2_ * 64, k1_);                              \
            if (k_ > k1_) acc2_ = agg_sum<SMALL, V>(acc2_, ring2, k_ - k1_);                      \
        }                                                                                         \
        ADC_VOL_STORE(reinterpret_cast<V*>(dpn), acc2_);                                          \
        dpn += fstep;                                                                             \
        slot2_s = slot2_s + 1 == R ? 0 : slot2_s + 1;                                             \
    } while (0)
// what happens to a finished first-pass output
#define AGG_OUT(ACC, RECV)                                                                        \
    do {                                                                                          \
        if constexpr (PAIR) {                                                                     \
            ring2[slot2_w * 64] = (ACC);                                                          \
            recring[slot2_w] = (RECV); /* wave-uniform value, same address in every lane */       \
            slot2_w = slot2_w + 1 == R ? 0 : slot2_w + 1;                                         \
            const int s_ = mcur - L; /* its look-ahead (<= L) is complete now */                  \
            mcur++;                                                                               \
            if (s_ >= s0 && s_ < s1) AGG_EMIT2();                                                 \
        } else {                                                                                  \
            ADC_VOL_STORE(reinterpret_cast<V*>(dpn), (ACC));                                      \
            dpn += fstep;                                                                         \
        }                                                                                         \
    } while (0)

#define AGG_PUSH(V)                                \
    do {                                           \
        if constexpr (REGRING) agg_reg_push(slot_w, (V)); \
        else ring[slot_w * 64] = (V);              \
        slot_w = slot_w + 1 == R ? 0 : slot_w + 1; \
    } while (0)

#define AGG_EMIT(M, REC)                                                                          \
    do {                                                                                          \
        const uint32_t r_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)(REC));                \
        const int a_lo_ = (int)(r_ & 255u), a_hi_ = (int)((r_ >> 8) & 255u);                      \
        int idx_ = slot_m - a_lo_;                                                                \
        if (idx_ < 0) idx_ += R;                                                                  \
        const int n_ = ADC_K4_DIAG == 1 ? 1 : (ADC_K4_DIAG == 3 ? adc_imin(4, a_lo_ + a_hi_ + 1) : a_lo_ + a_hi_ + 1); \
        V acc_;                                                                                   \
        if constexpr (REGRING) {                                                                  \
            const int n1_ = adc_imin(n_, R - idx_);                                               \
            acc_ = agg_reg_sum(vzero, idx_, n1_);                                                 \
            if (n_ > n1_) acc_ = agg_reg_sum(acc_, 0, n_ - n1_);                                  \
        } else if (n_ == 1) {                                                                     \
            acc_ = vzero + ring[idx_ * 64]; /* arms 0/0: the sum is the pixel itself */           \
        } else {                                                                                  \
            const int n1_ = adc_imin(n_, R - idx_);                                               \
            acc_ = agg_sum<SMALL, V>(vzero, ring + idx_ * 64, n1_); /* t = -arm .. +arm */         \
            if (n_ > n1_) acc_ = agg_sum<SMALL, V>(acc_, ring, n_ - n1_); /* wrapped part */      \
        }                                                                                         \
        if (DIVIDE && ADC_K4_DIAG != 2) {                                                         \
            const uint32_t c_ = r_ >> 16;                                                         \
            if (c_ != 1u) acc_ = acc_ / (float)c_; /* cross_aggregator.cpp:389 (x/1 == x) */     \
        }                                                                                         \
        AGG_OUT(acc_, r_);                                                                        \
        (void)(M);                                                                                \
        slot_m = slot_m + 1 == R ? 0 : slot_m + 1;                                                \
    } while (0)

    // ---- phase A: entries lo .. jB-1 that precede the first output's look-ahead (no output yet)
    const int jB = adc_imin(hi, m0 + L);
    if constexpr (COSTIN) {
        for (int j = lo; j < jB; j++) {
            const uint4 rn = rrow[j], ln = lrow[j];
            float v;
            AGG_COST(rn.x, rn.y, rn.z, ln.x, ln.y, ln.z, v);
            AGG_PUSH(v);
        }
    } else {
    for (int j = lo; j < jB; j += AGG_PF) {
        V t[AGG_PF];
#pragma unroll
        for (int u = 0; u < AGG_PF; u++) t[u] = *reinterpret_cast<const V*>(sp + (long long)adc_imin(j + u, jB - 1) * fstep);
#pragma unroll
        for (int u = 0; u < AGG_PF; u++)
            if (j + u < jB) AGG_PUSH(t[u]);
    }
    }

    // ---- phase B (steady state): entry jj arrives, output m = jj - L leaves.
    // All compiler-tracked VMEM traffic of phase A is drained first, so the manual vmcnt bookkeeping
    // below starts from an empty queue.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int j = jB;
    if constexpr (COSTIN) {
        // Same pipeline without the data load: VMEM ops per steady-state step = [arm record load] ... [output store].
        // The pixel records arrive in BULK: every 64 entries lane l loads the records of entry block_start + l (six
        // dword loads per block, issued a whole block ahead), and each step picks its entry's six dwords with
        // v_readlane (wave-uniform values).  The bulk loads make the hand-counted waits conservative (never unsafe).
        if (j + 2 * AGG_PF <= hi) {
            uint32_t cR0, cR1, cR2, cL0, cL1, cL2; // current block: records of entries bx0 + lane
            uint32_t nR0, nR1, nR2, nL0, nL1, nL2; // next block (in flight)
            uint32_t pr[AGG_PF];
            const uint4* rbase_ = ci.rrec + (size_t)fixed * ci.rpitch;
            const int roff_ = (int)(rrow - rbase_); // rrow[x] == rbase_[x + roff_]
            int bx0 = j, bpos = 0;
#define AGG_BULK_ISSUE(X0)                                                                                        \
    do {                                                                                                          \
        int ir_ = (X0) + lane + roff_;                                                                            \
        ir_ = ir_ < 0 ? 0 : (ir_ >= ci.rpitch ? ci.rpitch - 1 : ir_);                                             \
        const int il_ = adc_imin((X0) + lane, W - 1);                                                             \
        const uint4* pr_ = rbase_ + ir_;                                                                          \
        const uint4* pl_ = lrow + il_;                                                                            \
        asm volatile("global_load_dword %0, %1, off" : "=v"(nR0) : "v"(pr_) : "memory");                          \
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(nR1) : "v"(pr_) : "memory");                 \
        asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(nR2) : "v"(pr_) : "memory");                 \
        asm volatile("global_load_dword %0, %1, off" : "=v"(nL0) : "v"(pl_) : "memory");                          \
        asm volatile("global_load_dword %0, %1, off offset:4" : "=v"(nL1) : "v"(pl_) : "memory");                 \
        asm volatile("global_load_dword %0, %1, off offset:8" : "=v"(nL2) : "v"(pl_) : "memory");                 \
    } while (0)
#define AGG_BULK_TAKE(WAIT)                                                                                       \
    asm volatile(WAIT "v_mov_b32 %0, %6\n\tv_mov_b32 %1, %7\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %9\n\t"           \
                      "v_mov_b32 %4, %10\n\tv_mov_b32 %5, %11"                                                    \
                 : "=&v"(cR0), "=&v"(cR1), "=&v"(cR2), "=&v"(cL0), "=&v"(cL1), "=&v"(cL2)                         \
                 : "v"(nR0), "v"(nR1), "v"(nR2), "v"(nL0), "v"(nL1), "v"(nL2) : "memory")
            AGG_BULK_ISSUE(bx0);
            AGG_BULK_TAKE("s_waitcnt vmcnt(0)\n\t");
            AGG_BULK_ISSUE(bx0 + 64);
            const uint32_t* rpn = rp + (j - L);    // record of the output that entry triggers (>= m0)
#pragma unroll
            for (int u = 0; u < AGG_PF; u++) {
                asm volatile("global_load_dword %0, %1, off" : "=v"(pr[u]) : "v"(rpn) : "memory");
                rpn += 1;
            }
// next block: its loads were issued >= 64 steps (>= 128 VMEM operations) ago and at most 63 can be outstanding
#define AGG_BULK_NEXT()                                                                                           \
    if (bpos == 64) {                                                                                             \
        AGG_BULK_TAKE("");                                                                                        \
        bx0 += 64;                                                                                                \
        bpos = 0;                                                                                                 \
        AGG_BULK_ISSUE(bx0 + 64);                                                                                 \
    }
#define AGG_STEPC(U, WAITN, REFILL)                                                                               \
    do {                                                                                                          \
        uint32_t rr_;                                                                                             \
        asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=&v"(rr_) : "v"(pr[U]), "n"(WAITN) : "memory"); \
        if (REFILL) {                                                                                             \
            asm volatile("global_load_dword %0, %1, off" : "=v"(pr[U]) : "v"(rpn) : "memory");                    \
            rpn += 1;                                                                                             \
        }                                                                                                         \
        const int li_ = bpos + (U);                                                                               \
        const uint32_t rb_ = (uint32_t)__builtin_amdgcn_readlane((int)cR0, li_);                                  \
        const uint32_t rc0_ = (uint32_t)__builtin_amdgcn_readlane((int)cR1, li_);                                 \
        const uint32_t rc1_ = (uint32_t)__builtin_amdgcn_readlane((int)cR2, li_);                                 \
        const uint32_t lb_ = (uint32_t)__builtin_amdgcn_readlane((int)cL0, li_);                                  \
        const uint32_t lc0_ = (uint32_t)__builtin_amdgcn_readlane((int)cL1, li_);                                 \
        const uint32_t lc1_ = (uint32_t)__builtin_amdgcn_readlane((int)cL2, li_);                                 \
        float v_;                                                                                                 \
        AGG_COST(rb_, rc0_, rc1_, lb_, lc0_, lc1_, v_);                                                           \
        AGG_PUSH(v_);                                                                                             \
        AGG_EMIT(j + (U)-L, rr_); /* exactly one compiler-issued VMEM op (the store) */                           \
    } while (0)
            // first iteration: younger ops = 1 per not yet taken prologue slot + 2 per finished step
            static_assert(AGG_PF == 8, "the peeled first iteration below is written for AGG_PF == 8");
            AGG_STEPC(0, 7, true); AGG_STEPC(1, 8, true); AGG_STEPC(2, 9, true); AGG_STEPC(3, 10, true);
            AGG_STEPC(4, 11, true); AGG_STEPC(5, 12, true); AGG_STEPC(6, 13, true); AGG_STEPC(7, 14, true);
            j += AGG_PF;
            bpos += AGG_PF;
            for (; j + 2 * AGG_PF <= hi; j += AGG_PF) { // steady state: 2 ops per younger step (+ a bulk issue at times)
                AGG_BULK_NEXT();
                AGG_STEPC(0, 14, true); AGG_STEPC(1, 14, true); AGG_STEPC(2, 14, true); AGG_STEPC(3, 14, true);
                AGG_STEPC(4, 14, true); AGG_STEPC(5, 14, true); AGG_STEPC(6, 14, true); AGG_STEPC(7, 14, true);
                bpos += AGG_PF;
            }
            // drain: the AGG_PF records still in flight belong to entries j .. j+AGG_PF-1 (all < hi)
            AGG_BULK_NEXT();
            AGG_STEPC(0, 0, false); AGG_STEPC(1, 0, false); AGG_STEPC(2, 0, false); AGG_STEPC(3, 0, false);
            AGG_STEPC(4, 0, false); AGG_STEPC(5, 0, false); AGG_STEPC(6, 0, false); AGG_STEPC(7, 0, false);
            j += AGG_PF;
            // The bulk loads of the block that is never used are still in flight: wait for them and only THEN let their
            // destination registers die -- otherwise the compiler reuses those registers for the values of the
            // steps above and the late-landing loads overwrite them (found as one wrong entry per row segment).
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(nR0), "v"(nR1), "v"(nR2), "v"(nL0), "v"(nL1), "v"(nL2) : "memory");
#undef AGG_BULK_ISSUE
#undef AGG_BULK_TAKE
#undef AGG_BULK_NEXT
#undef AGG_STEPC
        }
    } else {
    V pf[AGG_PF];
    uint32_t pr[AGG_PF];
// data load of one entry (VPL floats per lane) into a prefetch slot
#define AGG_LDV(DST, PTR)                                                                                         \
    do {                                                                                                          \
        if constexpr (VPL == 2) asm volatile("global_load_dwordx2 %0, %1, off" ADC_VOL_NT_STR : "=v"(DST) : "v"(PTR) : "memory");  \
        else asm volatile("global_load_dword %0, %1, off" ADC_VOL_NT_STR : "=v"(DST) : "v"(PTR) : "memory");                     \
    } while (0)
    // The asm-prefetch loop needs every refill index valid without clamping: j + 2*AGG_PF <= hi.
    if (j + 2 * AGG_PF <= hi) {
        const float* spn = sp + (long long)j * fstep;       // next entry to prefetch
        const uint32_t* rpn = rp + (j - L);                 // record of the output that entry triggers (>= m0)
#pragma unroll
        for (int u = 0; u < AGG_PF; u++) {
            AGG_LDV(pf[u], spn);
            asm volatile("global_load_dword %0, %1, off" : "=v"(pr[u]) : "v"(rpn) : "memory");
            spn += fstep;
            rpn += 1;
        }
// one steady-state step; WAITN = number of VMEM ops younger than slot U's two loads that may stay in flight
#define AGG_STEP(U, WAITN)                                                                                       \
    do {                                                                                                         \
        V v_;                                                                                                    \
        uint32_t rr_;                                                                                            \
        /* wait and read the landed registers in ONE statement: a separate "+v" wait lets hipcc copy the      */ \
        /* (not yet landed) registers ABOVE the wait (tied-operand copies), i.e. read garbage                 */ \
        if constexpr (VPL == 2)                                                                                  \
            asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b64 %0, %2\n\tv_mov_b32 %1, %3"                          \
                         : "=&v"(v_), "=&v"(rr_) : "v"(pf[U]), "v"(pr[U]), "n"(WAITN) : "memory");               \
        else                                                                                                     \
            asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"                          \
                         : "=&v"(v_), "=&v"(rr_) : "v"(pf[U]), "v"(pr[U]), "n"(WAITN) : "memory");               \
        AGG_LDV(pf[U], spn);                                                                                     \
        asm volatile("global_load_dword %0, %1, off" : "=v"(pr[U]) : "v"(rpn) : "memory");                       \
        spn += fstep;                                                                                            \
        rpn += 1;                                                                                                \
        AGG_PUSH(v_);                                                                                            \
        AGG_EMIT(j + (U)-L, rr_); /* exactly one compiler-issued VMEM op (the store) */                          \
    } while (0)
        // first iteration: the younger ops are the prologue loads of slots U+1.. (2 each) and the U steps
        // already done (3 each): 2*(AGG_PF-1-U) + 3*U = 2*AGG_PF - 2 + U
        static_assert(AGG_PF == 8, "the peeled first iteration below is written for AGG_PF == 8");
        // (PAIR: a step may issue no store at all, so only the loads are counted -- a lower bound is always safe)
        AGG_STEP(0, 14); AGG_STEP(1, PAIR ? 14 : 15); AGG_STEP(2, PAIR ? 14 : 16); AGG_STEP(3, PAIR ? 14 : 17);
        AGG_STEP(4, PAIR ? 14 : 18); AGG_STEP(5, PAIR ? 14 : 19); AGG_STEP(6, PAIR ? 14 : 20); AGG_STEP(7, PAIR ? 14 : 21);
        j += AGG_PF;
        // steady state: younger ops = this slot's own store + 3 per younger step = 3*(AGG_PF-1)+1; we wait for
        // <= 3*(AGG_PF-1) outstanding (one stricter).  vmcnt retires in order (loads and stores) on gfx9-family.
        for (; j + 2 * AGG_PF <= hi; j += AGG_PF) {
#pragma unroll
            for (int u = 0; u < AGG_PF; u++) AGG_STEP(u, PAIR ? 2 * (AGG_PF - 1) : 3 * (AGG_PF - 1));
        }
#undef AGG_STEP
        // the AGG_PF entries still in flight are entries j .. j+AGG_PF-1 (all < hi)
        // drain: the AGG_PF entries still in flight are entries j .. j+AGG_PF-1 (all < hi); read them inside the
        // same statement as the wait (see AGG_STEP)
        V df[AGG_PF];
        uint32_t dr[AGG_PF];
#define AGG_DRAIN(MOVD)                                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)\n\t"                                                                                   \
                     MOVD " %0, %16\n\t" MOVD " %1, %17\n\t" MOVD " %2, %18\n\t" MOVD " %3, %19\n\t"                            \
                     MOVD " %4, %20\n\t" MOVD " %5, %21\n\t" MOVD " %6, %22\n\t" MOVD " %7, %23\n\t"                            \
                     "v_mov_b32 %8, %24\n\tv_mov_b32 %9, %25\n\tv_mov_b32 %10, %26\n\tv_mov_b32 %11, %27\n\t"                   \
                     "v_mov_b32 %12, %28\n\tv_mov_b32 %13, %29\n\tv_mov_b32 %14, %30\n\tv_mov_b32 %15, %31"                      \
                     : "=&v"(df[0]), "=&v"(df[1]), "=&v"(df[2]), "=&v"(df[3]), "=&v"(df[4]), "=&v"(df[5]), "=&v"(df[6]), "=&v"(df[7]), \
                       "=&v"(dr[0]), "=&v"(dr[1]), "=&v"(dr[2]), "=&v"(dr[3]), "=&v"(dr[4]), "=&v"(dr[5]), "=&v"(dr[6]), "=&v"(dr[7])  \
                     : "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(pf[5]), "v"(pf[6]), "v"(pf[7]),         \
                       "v"(pr[0]), "v"(pr[1]), "v"(pr[2]), "v"(pr[3]), "v"(pr[4]), "v"(pr[5]), "v"(pr[6]), "v"(pr[7])          \
                     : "memory")
        if constexpr (VPL == 2) AGG_DRAIN("v_mov_b64");
        else AGG_DRAIN("v_mov_b32");
#undef AGG_DRAIN
#pragma unroll
        for (int u = 0; u < AGG_PF; u++) {
            AGG_PUSH(df[u]);
            AGG_EMIT(j + u - L, dr[u]);
        }
        j += AGG_PF;
    }
    }
    // ---- tail of phase B (< 2*AGG_PF entries): plain compiler-scheduled loads
    for (; j < hi; j++) {
        V v;
        if constexpr (COSTIN) {
            const uint4 rn = rrow[j], ln = lrow[j];
            AGG_COST(rn.x, rn.y, rn.z, ln.x, ln.y, ln.z, v);
        } else {
            v = *reinterpret_cast<const V*>(sp + (long long)j * fstep);
        }
        AGG_PUSH(v);
        const int m = j - L;
        if (m >= m0) AGG_EMIT(m, rp[m]);
    }
    // ---- phase C: outputs whose +L look-ahead ends beyond the last entry (image end)
    for (int m = adc_imax(m0, hi - L); m < m1; m++) AGG_EMIT(m, rp[m]);
    // ---- second pass: outputs whose look-ahead ends beyond the last first-pass output (image end)
    if constexpr (PAIR) {
        for (int s = adc_imax(s0, mcur - L); s < s1; s++) AGG_EMIT2();
    }
#undef AGG_LDV
#undef AGG_EMIT2
#undef AGG_OUT
#undef AGG_PUSH
#undef AGG_EMIT
#undef AGG_COST
}

template <bool VERT, bool DIVIDE, bool SMALL, bool COSTIN, bool PAIR, int VPL = 1>
__global__ __launch_bounds__(64) void k_agg_march(const float* __restrict__ src, float* __restrict__ dst,
                                                  const uint32_t* __restrict__ rec, int W, int H, int Dp, int L, int seg_len,
                                                  int nseg, int per_xcd, const int* __restrict__ armmax, int small_variant,
                                                  int small_L, AggCostIn ci)
{
    agg_march_body<VERT, DIVIDE, SMALL, COSTIN, PAIR, VPL, false>(src, dst, rec, W, H, Dp, L, seg_len, nseg, per_xcd, armmax,
                                                                  small_variant, small_L, ci);
}

// full ring in registers (2L+1 <= AGG_RING_REGS): the compiler keeps to v0..v55, ring 1 owns v56..v127, ring 2 (pairs)
// v128..v199.  Body: k_aggregate_rr.h.
template <bool VERT, bool DIVIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(AGG_RING_V0))) void k_agg_regring(
    const float* __restrict__ src, float* __restrict__ dst, const uint2* __restrict__ rec, int W, int H, int Dp, int L,
    int seg_len, int nseg, int per_xcd, const int* __restrict__ armmax, int small_variant, int small_L, float* __restrict__ sink)
{
    agg_rr_body<VERT, DIVIDE, false>(src, dst, rec, W, H, Dp, L, seg_len, nseg, per_xcd, armmax, small_variant, small_L, sink);
}
// pass pair on two register rings: dividing pass + the next iteration's first pass
template <bool VERT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(AGG_RING_V0))) void k_agg_regring_pair(
    const float* __restrict__ src, float* __restrict__ dst, const uint2* __restrict__ rec, int W, int H, int Dp, int L,
    int seg_len, int nseg, int per_xcd, const int* __restrict__ armmax, int small_variant, int small_L, float* __restrict__ sink)
{
    agg_rr_body<VERT, true, true>(src, dst, rec, W, H, Dp, L, seg_len, nseg, per_xcd, armmax, small_variant, small_L, sink);
}

// the same for the first pass of the pipeline (fused matching cost; its two tables stay in LDS)
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(AGG_RING_V0))) void k_agg_regring_cost(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ rec, int W, int H, int Dp, int L,
    int seg_len, int nseg, int per_xcd, const int* __restrict__ armmax, int small_variant, int small_L, AggCostIn ci)
{
    agg_march_body<false, false, false, true, false, 1, true>(src, dst, rec, W, H, Dp, L, seg_len, nseg, per_xcd, armmax,
                                                               small_variant, small_L, ci);
}

// third generation (k_aggregate_rr2.h): two disparities per lane, ring slots = VGPR pairs v96..v239, packed adds; a wave =
// one chunk of the flattened (line-major) output index space
template <bool VERT, bool DIVIDE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(RR2_V0))) void k_agg_rr2(
    const float* __restrict__ src, float* __restrict__ dst, const uint2* __restrict__ rec, int W, int H, int Dp, int L,
    int chunk_len, int nwaves, int per_xcd, const int* __restrict__ armmax, int small_variant, int small_L)
{
    const AggCostIn none = {};
    agg_rr2_body<VERT, DIVIDE, false>(src, dst, rec, W, H, Dp, L, chunk_len, nwaves, per_xcd, armmax, small_variant, small_L, none);
}
// first pass of the pipeline on the same body: the matching cost is computed in registers (two lane windows)
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(RR2_V0))) void k_agg_rr2_cost(
    float* __restrict__ dst, const uint2* __restrict__ rec, int W, int H, int Dp, int L, int chunk_len, int nwaves, int per_xcd,
    const int* __restrict__ armmax, int small_variant, int small_L, AggCostIn ci)
{
    agg_rr2_body<false, false, true>(nullptr, dst, rec, W, H, Dp, L, chunk_len, nwaves, per_xcd, armmax, small_variant, small_L, ci);
}

static int env_int(const char* name, int dflt)
{
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

// arm length up to which the small-ring variant is used (ADC_AGG_SMALL_L, 0 disables it)
int adc_agg_small_L(const adc_handle* h)
{
    static const int small_L_env = env_int("ADC_AGG_SMALL_L", 8);
    const int L = adc_imax(0, adc_imin(h->p.opt.cross_L1, 255));
    return adc_imin(small_L_env, L);
}

// Picks the number of line segments: all waves of a "round" run concurrently (9 per CU), a pass costs
// rounds x (segment length + halo) steps.
static int pick_nseg(long long nlines, int N, int L, int slots)
{
    int best = 1;
    long long best_cost = -1;
    for (int ns = 1; ns <= 16; ns++) {
        const int seg = (N + ns - 1) / ns;
        if (ns > 1 && seg < 2 * L) break;
        const long long rounds = (nlines * ns + slots - 1) / slots;
        const long long cost = rounds * (seg + (ns > 1 ? 2 * L : 0));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
    }
    return best;
}

// Chunk length of the pair-register-ring kernels (a wave = chunk_len consecutive outputs of the line-major index space,
// k_aggregate_rr2.h).  Candidates: whole-line segmentations N / k and equal shares of the whole pass per wave slot (1x, 2x, 3x
// the slots); cost model = rounds x (steps + 2L halo entries + a fixed price per piece for its prologue / slow tail).
static int pick_chunk(long long nlines, int N, int L, int slots)
{
    const long long total = nlines * N;
    long long best_cost = -1;
    int best = N;
    auto consider = [&](long long c) {
        if (c < 1) c = 1;
        if (c < N && c < 4 * (long long)L) return; // halo-dominated
        if (c > total) c = total;
        const long long waves = (total + c - 1) / c;
        const long long rounds = (waves + slots - 1) / slots;
        const bool aligned = c >= N ? (c % N == 0) : (N % c == 0);
        const long long pieces = aligned ? (c >= N ? c / N : 1) : (c >= N ? c / N + 2 : 2);
        const long long halo = (c < N || !aligned) ? 2LL * L : 0;
        const long long cost = rounds * (c + halo + 60 * pieces);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = (int)c; }
    };
    for (int k = 1; k <= 16; k++) consider((N + k - 1) / k);
    for (int k = 1; k <= 3; k++) consider((total + (long long)slots * k - 1) / ((long long)slots * k));
    return best;
}

// Ring depth of a small-ring launch along one direction when the host works with arm maxima (armmax_host): the longest arm,
// plus one entry of margin when the maxima are ASSUMED from an earlier Match of the handle (armmax_valid == 2; the next image of a
// similar stream may have a longest arm of 3 after 2, and a wrong depth costs a redo or the full-ring plan: ADC_AGG_ASSUME_MARGIN).
static int agg_assumed_depth(const adc_handle* h, bool vert)
{
    const int assume_margin = env_int("ADC_AGG_ASSUME_MARGIN", 1); // (read per call: the tests vary it within one process)
    const int Lknown = adc_imax(1, h->armmax_host[vert ? 1 : 0]) + (h->armmax_valid == 2 ? assume_margin : 0);
    return adc_imin(adc_agg_small_L(h), Lknown);
}

// which: 0 = the host does not know the arms: launch the full-ring and the small-ring variant, the kernel decides;
//        1 = small ring only, 2 = full ring only (the host has read armmax).  PAIR needs which == 1.
template <bool VERT, bool DIVIDE, bool COSTIN = false, bool PAIR = false>
static hipError_t launch_pass(adc_handle* h, const float* src, float* dst, int which = 0)
{
    const AdcParams& p = h->p;
    const int L = adc_imax(0, adc_imin(p.opt.cross_L1, 255)); // (a ring of 2 * 255 + 1 entries of 256 bytes fits the 160 KiB of LDS: every arm limit marches)
    const int N = VERT ? p.H : p.W;
    const int small_L = adc_agg_small_L(h);
    // two disparities per lane with the small ring (ADC_AGG_VPL2=0 switches it off)
    static const int vpl2_env = env_int("ADC_AGG_VPL2", 1); // 1 = every small-ring launch, 2 = pass pairs only
    for (int variant = 0; variant < 2; variant++) { // 0: full ring, 1: small ring (exits unless every arm <= small_L)
        if (variant == 1 && (small_L <= 0 || small_L >= L)) break;
        if ((which == 1 && variant == 0 && small_L > 0 && small_L < L) || (which == 2 && variant == 1)) continue;
        // the ring only has to be as deep as the longest arm of this direction when the host knows it (which == 1)
        const int Lv = variant ? ((which == 1 && h->armmax_valid) ? agg_assumed_depth(h, VERT) : small_L) : L;
        // the fused-cost variant keeps the two cost tables (768 + 64 floats) behind the ring, the pair variant a second
        // ring and a record ring
        // full ring of a plain pass: in registers when it fits (ADC_AGG_REGRING=0: LDS ring)
        static const bool regring_env = env_int("ADC_AGG_REGRING", 1) != 0;
        const bool regring = variant == 0 && regring_env && Lv >= 1 && 2 * Lv + 1 <= AGG_RING_REGS;
        // ... as VGPR pairs, two disparities per lane (k_aggregate_rr2.h; ADC_AGG_RR2=0: the one-float register ring)
        static const bool rr2_env = env_int("ADC_AGG_RR2", 1) != 0;
        const bool rr2 = regring && rr2_env && !PAIR && p.Dp % 128 == 0 && 2 * Lv + 1 <= RR2_SLOTS;
        const int vpl = rr2 ? 2 : ((variant == 1 && !COSTIN && (vpl2_env == 1 || (vpl2_env == 2 && PAIR)) && p.Dp % 128 == 0) ? 2 : 1);
        const long long nlines = (long long)(VERT ? p.W : p.H) * (p.Dp / (64 * vpl));
        const size_t ring_bytes = regring ? 0 : (size_t)(2 * Lv + 1) * 64 * sizeof(float) * vpl;
        const size_t ldsv = ring_bytes + (COSTIN ? (768 + 64) * sizeof(float) : 0) + ((PAIR && !regring) ? ring_bytes + (2 * Lv + 1) * 4 + 64 : 0);
        // register rings: 128 VGPRs -> 4 waves per SIMD; a pair (two rings, 200 VGPRs) or a ring of pairs (240) -> 2
        const int waves_per_cu = regring ? ((PAIR || rr2) ? 8 : 16) : adc_imax(1, adc_imin(32, (int)((160 * 1024) / ((ldsv + 511) / 512 * 512))));
        int nseg = env_int(VERT ? "ADC_AGG_VSEG" : "ADC_AGG_HSEG", 0);
        if (nseg < 1) nseg = pick_nseg(nlines, N, PAIR ? 2 * Lv : Lv, 256 * waves_per_cu);
        int seg_len = (N + nseg - 1) / nseg;
        if (seg_len < 1) seg_len = 1;
        nseg = (N + seg_len - 1) / seg_len;
        const long long waves = nlines * nseg;
        const int per_xcd = (int)((waves + 7) / 8);
        const bool both = which == 0 && small_L > 0 && small_L < L;
        const bool verify = variant == 1 && which == 1 && h->armmax_valid == 2; // ring depth assumed from the previous Match
        int sv = both ? variant : (verify ? 2 : -1), sl = both ? small_L : (verify ? Lv : 0x7fffffff);
        if (h->agg_gate) { sv = h->agg_gate; sl = h->agg_gate_thr; } // one of two plans enqueued back to back (adc_launch_aggregate)
        AggCostIn ci;
        ci.rrec = reinterpret_cast<const uint4*>(h->cost_rrec);
        ci.lrec = reinterpret_cast<const uint4*>(h->cost_lrec);
        ci.lut_ad = h->lut_ad;
        ci.lut_census = h->lut_census;
        ci.rpitch = h->rrec_pitch; ci.padl = h->rrec_padl; ci.dmin = p.dmin; ci.D = p.D;
        if (!COSTIN) // (which == 0, debug path: both variants are launched and the kernels decide; the label is then the last one)
            h->agg_kernel = rr2 ? "k_agg_rr2 (register ring of VGPR pairs, 2 disparities per lane, one pass per launch)"
                          : regring ? (PAIR ? "k_agg_regring_pair (two register rings, dividing pass + next first pass per launch)"
                                            : "k_agg_regring (register ring, 1 disparity per lane, one pass per launch)")
                          : variant ? (PAIR ? "k_agg_march<.., PAIR> (LDS small rings: dividing pass + next first pass per launch)"
                                            : "k_agg_march<.., SMALL> (LDS small ring, one pass per launch)")
                                    : "k_agg_march (LDS full ring, one pass per launch)";
        if (rr2) {
            int chunk_len = env_int(VERT ? "ADC_AGG_VCHUNK" : "ADC_AGG_HCHUNK", 0);
            if (chunk_len < 1) chunk_len = pick_chunk(nlines, N, Lv, 256 * waves_per_cu);
            const long long total = nlines * N;
            const int nwaves = (int)((total + chunk_len - 1) / chunk_len);
            const int pxc = (nwaves + 7) / 8;
            if constexpr (COSTIN)
                hipLaunchKernelGGL(k_agg_rr2_cost, dim3((unsigned)pxc * 8), dim3(64), ldsv, h->heavy, dst,
                                   reinterpret_cast<const uint2*>(h->rec2_h), p.W, p.H, p.Dp, Lv, chunk_len, nwaves, pxc, h->armmax, sv, sl, ci);
            else if constexpr (!PAIR)
                hipLaunchKernelGGL((k_agg_rr2<VERT, DIVIDE>), dim3((unsigned)pxc * 8), dim3(64), 0, h->heavy, src, dst,
                                   reinterpret_cast<const uint2*>(VERT ? h->rec2_v : h->rec2_h), p.W, p.H, p.Dp, Lv, chunk_len, nwaves,
                                   pxc, h->armmax, sv, sl);
        } else if (regring) {
            if constexpr (COSTIN)
                hipLaunchKernelGGL(k_agg_regring_cost, dim3((unsigned)per_xcd * 8), dim3(64), ldsv, h->heavy, src, dst,
                                   VERT ? h->rec_v : h->rec_h, p.W, p.H, p.Dp, Lv, seg_len, nseg, per_xcd, h->armmax, sv, sl, ci);
            else if constexpr (PAIR)
                hipLaunchKernelGGL((k_agg_regring_pair<VERT>), dim3((unsigned)per_xcd * 8), dim3(64), 0, h->heavy, src, dst,
                                   reinterpret_cast<const uint2*>(VERT ? h->rec2_v : h->rec2_h), p.W, p.H, p.Dp, Lv, seg_len, nseg,
                                   per_xcd, h->armmax, sv, sl, h->agg_sink);
            else
                hipLaunchKernelGGL((k_agg_regring<VERT, DIVIDE>), dim3((unsigned)per_xcd * 8), dim3(64), 0, h->heavy, src, dst,
                                   reinterpret_cast<const uint2*>(VERT ? h->rec2_v : h->rec2_h), p.W, p.H, p.Dp, Lv, seg_len, nseg,
                                   per_xcd, h->armmax, sv, sl, h->agg_sink);
        } else if (variant && vpl == 2) {
            if constexpr (!COSTIN)
                hipLaunchKernelGGL((k_agg_march<VERT, DIVIDE, true, false, PAIR, 2>), dim3((unsigned)per_xcd * 8), dim3(64), ldsv, h->heavy, src, dst,
                                   VERT ? h->rec_v : h->rec_h, p.W, p.H, p.Dp, Lv, seg_len, nseg, per_xcd, h->armmax, sv, sl, ci);
        } else if (variant)
            hipLaunchKernelGGL((k_agg_march<VERT, DIVIDE, true, COSTIN, PAIR>), dim3((unsigned)per_xcd * 8), dim3(64), ldsv, h->heavy, src, dst,
                               VERT ? h->rec_v : h->rec_h, p.W, p.H, p.Dp, Lv, seg_len, nseg, per_xcd, h->armmax, sv, sl, ci);
        else
            hipLaunchKernelGGL((k_agg_march<VERT, DIVIDE, false, COSTIN, PAIR>), dim3((unsigned)per_xcd * 8), dim3(64), ldsv, h->heavy, src, dst,
                               VERT ? h->rec_v : h->rec_h, p.W, p.H, p.Dp, Lv, seg_len, nseg, per_xcd, h->armmax, sv, sl, ci);
    }
    return hipGetLastError();
}

// One plan of the aggregation: the launch sequence for the ring choice (which_h, which_v) the handle's state implies.
//   dry            only count (no launch, no event, nothing of the handle changes)
//   first_into_cur the first launch writes the volume it would have READ (only with the fused cost, which has no input volume):
//                  flips which of the two volumes the plan ends in
//   marks          record the profiling events / launch statistics of this plan
struct AggSeq { int launches, passes; float* result; bool first_fused; };
static hipError_t agg_sequence(adc_handle* h, int iterations, bool dry, bool first_into_cur, bool marks, AggSeq* out)
{
    static const bool pair_env = env_int("ADC_AGG_PAIR", 1) != 0;
    // pairs with the full ring: 0 (default) = never, 1 = when both rings fit into registers (k_agg_regring_pair), 2 = also
    // as two 17 KiB LDS rings per wave.  Measured on MI355X (structured 1080p pair, rocprofv3): a register-ring pair
    // launch takes 0.95-1.02 ms against 2 x 0.42 ms for two single passes -- 200 VGPRs leave 2 waves per SIMD, and this
    // kernel family runs at ~8.7 cycles per instruction and wave whatever the occupancy, so halving the waves doubles
    // the time per step while the saved HBM round trip (0.2 ms at the copy rate) does not pay for it; two LDS rings were
    // 3x slower.  The single pass itself now runs at the device copy rate.
    static const int pair_full = env_int("ADC_AGG_PAIR_FULL", 0);
    static const bool regring_on = env_int("ADC_AGG_REGRING", 1) != 0;
    hipError_t e = hipSuccess;
    const int Lfull = adc_imax(0, adc_imin(h->p.opt.cross_L1, 255));
    const bool regring_fits = regring_on && Lfull >= 1 && 2 * Lfull + 1 <= AGG_RING_REGS;
    const bool lds_fits = (size_t)(2 * Lfull + 1) * 64 * sizeof(float) + (768 + 64) * sizeof(float) <= 150 * 1024;
    const bool marching = true; // (every arm limit fits an LDS ring; round 6 dropped the one-thread-per-element fallback no geometry selected)
    const bool prof = marks && !dry && h->profiling;
    // armmax_host (valid when the pipeline / caller read the maximum arms back): pick the ring on the host
    const int small_L = adc_agg_small_L(h);
    const bool small_ok = small_L > 0 && small_L < Lfull;
    int which_h = 0, which_v = 0; // 0 = let the kernels decide (two launches per pass)
    if (h->armmax_valid == 3 && marching) { // nothing known about this image: the full ring is valid for every image
        which_h = which_v = 2;
    } else if (h->armmax_valid && marching) {
        which_h = (small_ok && h->armmax_host[0] <= small_L) ? 1 : 2;
        which_v = (small_ok && h->armmax_host[1] <= small_L) ? 1 : 2;
    }
    // Pass sequence (cross_aggregator.cpp:100-118): iteration k = [first direction][second direction, divided by the
    // support count]; the direction order alternates, so the dividing pass of iteration k and the first pass of
    // iteration k+1 run along the SAME direction and can share one launch (PAIR) when the small ring is in use.
    float* cur = h->vol_a; // holds the input of the next launch
    float* oth = h->vol_b;
    bool horizontal_first = true; // cross_aggregator.cpp:100
    int launch = 0;
    int passes = 0; // algorithmic passes (cross_aggregator.cpp: 2 per iteration) covered by the launches so far
    bool second_done = false; // the first pass of this iteration was already computed by the previous pair launch
    bool first_fused = false;
    for (int k = 0; k < iterations && e == hipSuccess; k++) {
        const bool hf = horizontal_first;
        if (!second_done) {
            if (prof && launch < 2) hipEventRecord(h->ev_agg[launch], h->heavy); // (only the marks adc_wait reads: start of the first / first regular launch)
            bool swap = true;
            if (hf) {
                // first pass of the pipeline: the matching cost is computed inside the pass (no input volume)
                const bool fused = k == 0 && h->fuse_cost && lds_fits;
                if (k == 0) first_fused = fused;
                if (fused) {
                    swap = !first_into_cur;
                    if (!dry) e = launch_pass<false, false, true>(h, cur, swap ? oth : cur, which_h);
                } else if (!dry) e = launch_pass<false, false>(h, cur, oth, which_h);
            } else if (!dry) {
                e = launch_pass<true, false>(h, cur, oth, which_v);
            }
            if (swap) { float* t = cur; cur = oth; oth = t; }
            launch++;
            passes++;
        }
        second_done = false;
        if (e != hipSuccess) break;
        if (prof && launch < 2) hipEventRecord(h->ev_agg[launch], h->heavy); // (only the marks adc_wait reads: start of the first / first regular launch)
        // second pass of the iteration (dividing): vertical after a horizontal first pass and vice versa
        const int wsec = hf ? which_v : which_h;
        // The LAST pass (horizontal, dividing) of a short-arm image moves into the first scanline pass (k_scanline_seg_agg:
        // one launch and 2 V of traffic less): arms up to 4, assumed or known; the other horizontal passes verify the depth.
        if (!hf && k + 1 == iterations && iterations == 4 && h->fuse_agg_so && !h->agg_gate && marching && which_h == 1 &&
            (h->armmax_valid == 1 || h->armmax_valid == 2) && agg_assumed_depth(h, false) <= 4 && adc_so_can_fuse_agg(h)) {
            if (!dry) { h->so_agg_fused = 1; h->agg_so_fusions++; }
            break;
        }
        const bool pair = pair_env && marching && k + 1 < iterations &&
                          (wsec == 1 || (wsec == 2 && (pair_full >= 2 || (pair_full == 1 && regring_fits))));
        if (!dry) {
            if (hf) {
                if (pair) e = launch_pass<true, true, false, true>(h, cur, oth, wsec);
                else e = launch_pass<true, true>(h, cur, oth, which_v); // / sup_h
            } else {
                if (pair) e = launch_pass<false, true, false, true>(h, cur, oth, wsec);
                else e = launch_pass<false, true>(h, cur, oth, which_h); // / sup_v
            }
        }
        { float* t = cur; cur = oth; oth = t; }
        launch++;
        passes += pair ? 2 : 1;
        second_done = pair;
        horizontal_first = !horizontal_first;
    }
    if (prof) hipEventRecord(h->ev_agg[launch < 8 ? launch : 8], h->heavy);
    if (marks && !dry) {
        h->agg_first_fused = first_fused ? 1 : 0;
        h->agg_launches = launch < 8 ? launch : 8;
        h->agg_passes = passes;
    }
    if (out) { out->launches = launch; out->passes = passes; out->result = cur; out->first_fused = first_fused; }
    return e;
}

// vol_a -> (H,V | V,H alternating) -> vol_a.  Every iteration is two launches: a -> b -> a.
hipError_t adc_launch_aggregate(adc_handle* h, int iterations)
{
    if ((h->paper & ADC_PAPER_RIGHT_ARMS) && h->arms_r) return adc_paper_aggregate(h, iterations); // opt-in paper mode (k_paper.hip)
    {   // allow > 64 KiB dynamic LDS for the ring (large cross_L1): a per-DEVICE function attribute -- set once for every
        // device this process drives (a farm on device 1 after one on device 0), under a lock (handles on several threads)
        static std::mutex attr_mu;
        static bool attr_set[64] = {false};
        std::lock_guard<std::mutex> lk(attr_mu);
        const int dv = (h->device >= 0 && h->device < 64) ? h->device : 0;
        if (!attr_set[dv]) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<false, false, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<false, false, false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<false, true, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<true, false, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<true, true, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<true, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<false, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set[dv] = true;
        }
    }
    h->agg_first_fused = 0;
    h->agg_dual_last = 0;
    h->so_agg_fused = 0; // (round-5 advisor finding: a Match that failed between this stage and the scanline stage must not leave it set)
    hipError_t e = hipSuccess;
    AggSeq seq;
    // Two plans (a stream that alternates between short-arm and long-arm images, h->agg_dual > 0; pipeline only: the arm maxima
    // are not known on the host): plan S = small rings of the depth the last short-arm image needed (+ margin) with pass pairs,
    // plan F = the full ring; every kernel of S runs iff both directions fit the assumed depths, every kernel of F iff not
    // (agg_gate_skip).  Needs the fused cost: its first pass has no input volume, so plan F can start by writing vol_a
    // instead of vol_b when that makes both plans END in the same volume (S: 5 launches, F: 8).
    const bool dual_env = env_int("ADC_AGG_DUAL", 1) != 0; // (read per call: the tests switch it within one process)
    const int small_L = adc_agg_small_L(h);
    const int Lfull = adc_imax(0, adc_imin(h->p.opt.cross_L1, 255));
    if (dual_env && h->agg_dual > 0 && h->armmax_valid >= 2 && h->fuse_cost && iterations >= 1 && small_L > 0 && small_L < Lfull) {
        const int keep_host[2] = {h->armmax_host[0], h->armmax_host[1]}, keep_valid = h->armmax_valid;
        h->armmax_host[0] = h->armmax_small[0] > 0 ? h->armmax_small[0] : small_L;
        h->armmax_host[1] = h->armmax_small[1] > 0 ? h->armmax_small[1] : small_L;
        h->armmax_valid = 2;
        AggSeq s_dry, f_dry;
        h->agg_gate = 3; // (also during the dry runs: a plan of a two-plan run never moves its last pass into the scanline stage)
        e = agg_sequence(h, iterations, true, false, false, &s_dry);
        h->armmax_valid = 3;
        h->agg_gate = 4;
        if (e == hipSuccess) e = agg_sequence(h, iterations, true, false, false, &f_dry);
        h->agg_gate = 0;
        const bool usable = e == hipSuccess && s_dry.first_fused && f_dry.first_fused;
        if (usable) {
            h->armmax_valid = 2;
            h->agg_gate_thr = agg_assumed_depth(h, false) | (agg_assumed_depth(h, true) << 16);
            h->agg_gate = 3;
            e = agg_sequence(h, iterations, false, false, true, &seq);
            h->armmax_valid = 3;
            h->agg_gate = 4;
            AggSeq f;
            if (e == hipSuccess) e = agg_sequence(h, iterations, false, f_dry.result != s_dry.result, false, &f);
            if (e == hipSuccess && f.result != seq.result) e = hipErrorUnknown; // (cannot happen: the flip above aligns them)
            h->agg_gate = 0;
            h->agg_dual_last = 1;
            h->agg_dual_runs++;
        }
        h->armmax_host[0] = keep_host[0]; h->armmax_host[1] = keep_host[1]; h->armmax_valid = keep_valid;
        if (!usable && e == hipSuccess) e = agg_sequence(h, iterations, false, false, true, &seq);
    } else {
        e = agg_sequence(h, iterations, false, false, true, &seq);
    }
    // the result must be in vol_a (cost_aggr_): swap the two volume pointers if it ended up in the other one
    if (e == hipSuccess && seq.result != h->vol_a) { h->vol_b = h->vol_a; h->vol_a = seq.result; }
    return e;
}

// The dividing H pass of the last iteration that adc_launch_aggregate left to the scanline stage (so_agg_fused), as a launch of
// its own: what run_so falls back to when its segment plan at scanline time cannot take the pass over after all (round-5 advisor
// finding: that used to fail the Match).  Full ring: valid whatever the arms are.
hipError_t adc_launch_aggregate_tail(adc_handle* h)
{
    const int keep = h->armmax_valid;
    h->armmax_valid = 0;
    const hipError_t e = launch_pass<false, true>(h, h->vol_a, h->vol_b, 2);
    h->armmax_valid = keep;
    if (e == hipSuccess) { float* t = h->vol_a; h->vol_a = h->vol_b; h->vol_b = t; }
    return e;
}
