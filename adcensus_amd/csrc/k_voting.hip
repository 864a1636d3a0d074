// k_voting.hip -- K8 iterative region voting (MultiStepRefiner::IterativeRegionVoting, multistep_refiner.cpp:153-227),
// DEVICE-DRIVEN: the host enqueues a fixed chain of identical kernels and never looks at the data; which round runs and when the
// iteration has converged is decided on the device.
//
// Semantics kept exactly: 5 iterations x {mismatches, occlusions}; inside a pass the reference fills the
// still-invalid pixels of the list in raster order IN PLACE, so a vote sees the fills of the list pixels that precede it, and
// later passes see the fills of earlier ones.  Round 5: all ten passes are ONE fixed-point iteration on ONE 16-bit state per
// pixel (irv_plan.h, top: bin | iteration of the fill | list) -- the system is triangular in the order (iteration, list, raster
// position), so every chaotic in-place iteration converges to the sequential result and a whole round without a change proves
// it.  Votes only ever need lround(d) - dmin of a region pixel, so the state map is 2 bytes per pixel (L2-resident at 1080p).
// A vote (one wave per entry) reads the cross region as 16-byte row blocks (8 pixels per lane and load, 16 region rows
// x 4 blocks per trip: one trip for a typical region of 13 rows x 13 pixels) into the LDS histograms of the iterations from
// which the pixels count, and takes the first iteration whose CUMULATIVE histogram passes the vote.  From round 1 on only
// entries whose region saw a change in the previous round are re-evaluated (8x8 change tiles over the region's bounding box).
//
// The chain.  Kernel k reads the state its predecessor wrote (slot k & 1) and the predecessor's accumulator
// acc[(k-1) & 63] (list length / "something changed"), derives its own action -- every block derives the same one -- and
// block 0 publishes the new state into slot (k+1) & 1 and clears the accumulators kernel k+2 will use.  No host round
// trip, no grid barrier (measured: 4.2-4.8 us on this chip against 2.4-2.9 us for a dependent launch,
// tools/ubench/grid_sync.hip), no ticket atomics:
//     BEGIN  seed the working copy of the map, build the state map and the work list (listed pixels whose region is too
//            small to ever pass the vote are left out)
//     ROUND  ONE kernel per round: every wave holds (up to) 64 list entries, one per lane, decides per lane whether the
//            entry has to be re-evaluated and evaluates its dirty entries one after the other.
//     FINAL  write the fills back
// The chain length is a BUDGET (adc_handle::irv_budget, adapted from the kernels the last Matches of the handle needed);
// when it is exhausted before the state machine reaches DONE, adc_wait continues the same chain synchronously and redoes
// the stages behind it -- a performance cliff, never a different result.
#include "adc_internal.h"
#include "adc_device_fn.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

#include "irv_plan.h"

#ifdef IRV_TIMING // diagnosis build (tools/build_variant.sh): where does a kernel of the chain spend its time?
// wave 0 of every 64th workgroup stamps s_memtime at the stage boundaries (drained loads) and s_memrealtime at entry / exit
__device__ long long g_irv_t[8][8192][12]; // [sampled workgroup][kernel][8 cycle stamps, entry / exit real time, pool items, votes of wave 0]
#define IRV_TSEL ((blockIdx.x & 63) == 0 && blockIdx.x < 512 && threadIdx.x < 64)
#define IRV_T(slot) do { if (IRV_TSEL) { long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (threadIdx.x == 0 && k < 8192) g_irv_t[blockIdx.x >> 6][k][slot] = t_; } } while (0)
#define IRV_TV(slot, val) do { if (IRV_TSEL && threadIdx.x == 0 && k < 8192) g_irv_t[blockIdx.x >> 6][k][slot] = (long long)(val); } while (0)
#define IRV_TR(slot) do { if (IRV_TSEL) { long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (threadIdx.x == 0 && k < 8192) g_irv_t[blockIdx.x >> 6][k][slot] = t_; } } while (0)
extern "C" int adc_debug_irv_timing(long long* out, int which, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_irv_t), (size_t)n * 12 * sizeof(long long), (size_t)which * 8192 * 12 * sizeof(long long));
}
#else
#define IRV_T(slot) do { } while (0)
#define IRV_TV(slot, val) do { } while (0)
#define IRV_TR(slot) do { } while (0)
#endif

// Box of a pixel's vote: bbox[p] = {-, -, ml_all, mr_all} = the widest H arms over ALL region rows -- the rectangle whose state
// blocks a vote requests together with the row arms, in the same memory round trip (the row arms then only mask), and whose change
// tiles decide whether the entry is looked at again.  Written by k_sup_counts (k_arms.hip), whose walk over the region rows yields it
// for free; it becomes the high half of word z of the pixel's work-list entry.
// Did a pixel of the tile box [tx0, tx1] x [ty0, ty1] change in the round whose stamp is want4 (replicated byte)?  A
// tile row of the box = 16 bytes from a dword-aligned column; bytes outside the box are forced non-zero (nk = ~byte
// mask per dword, from a 16-bit byte-valid mask) before the any-zero-byte test of (word ^ stamp); three tile rows are
// in flight (rows past ty1 repeat row ty1: harmless duplicates).  One iteration of each loop for arm limits <= 34 and
// boxes of <= 17 rows.  Not inlined: the round kernel has to stay at 64 VGPRs (8 waves per SIMD, the whole grid
// co-resident), and inlined the compiler interleaves this with the surrounding code at 80-110.
__device__ __forceinline__ bool irv_box_dirty(const uint8_t* __restrict__ chg_rd, int tpitch, int tx0, int tx1, int ty0, int ty1,
                                                        uint32_t want4)
{
    bool dirty = false;
    // (interleave(disable): the loop vectoriser otherwise runs two iterations of these OR-reductions side by side)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int txb = tx0; txb <= tx1; txb += 12) {
        const int cb = txb & ~3, last = adc_imin(tx1, txb + 11);
        uint32_t nk[4];
        irv_tile_row_masks(txb, last, nk);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
        for (int tyb = ty0; tyb <= ty1; tyb += 3) {
            const uint4 v0 = *reinterpret_cast<const uint4*>(chg_rd + (uint32_t)(adc_imin(tyb + 0, ty1) * tpitch + cb));
            const uint4 v1 = *reinterpret_cast<const uint4*>(chg_rd + (uint32_t)(adc_imin(tyb + 1, ty1) * tpitch + cb));
            const uint4 v2 = *reinterpret_cast<const uint4*>(chg_rd + (uint32_t)(adc_imin(tyb + 2, ty1) * tpitch + cb));
            const uint32_t hit = irv_tile_hit(v0.x, nk[0], want4) | irv_tile_hit(v0.y, nk[1], want4) | irv_tile_hit(v0.z, nk[2], want4) |
                                 irv_tile_hit(v0.w, nk[3], want4) | irv_tile_hit(v1.x, nk[0], want4) | irv_tile_hit(v1.y, nk[1], want4) |
                                 irv_tile_hit(v1.z, nk[2], want4) | irv_tile_hit(v1.w, nk[3], want4) | irv_tile_hit(v2.x, nk[0], want4) |
                                 irv_tile_hit(v2.y, nk[1], want4) | irv_tile_hit(v2.z, nk[2], want4) | irv_tile_hit(v2.w, nk[3], want4);
            dirty |= hit != 0u;
        }
    }
    return dirty;
}

// Slack budgets (irv_plan.h, bottom): bits [lo, hi) of a 64-bit word, 0 <= lo, hi <= 64 -- one half of a row span inside the 128-bit
// window of the per-pixel change bitmap
__device__ __forceinline__ unsigned long long irv_mask64(int lo, int hi)
{
    if (hi <= lo) return 0ull;
    const unsigned long long up = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    return up & ~((1ull << lo) - 1ull); // (lo < hi <= 64: the shift is defined)
}

// ... and, per LANE (phase 1 of a round: one entry per lane), the changed pixels of an entry's whole region in one plane of the bitmap:
// per row ONE 16-byte load of the bitmap and the row's arms (both addresses known up front: no load depends on another), IRV_RC_ROWS rows
// per trip (more rows in flight cost registers: the round kernel must stay at 64 VGPRs without scratch).  The entry's own pixel does
// not count (a pixel does not vote for itself).  Lanes without work pass ya > yb.
#ifndef IRV_RC_ROWS
#define IRV_RC_ROWS 4
#endif
#ifndef IRV_RC_INLINE
#define IRV_RC_INLINE __forceinline__
#endif
#ifndef IRV_RC_EXACT
#define IRV_RC_EXACT 0 // 1: every row with its own arms (fewer entries survive the filter, but an arms load and two 64-bit masks per row)
#endif
__device__ IRV_RC_INLINE int irv_region_changes(const uint32_t* __restrict__ px, int pitch, const uint32_t* __restrict__ arms32, int W,
                                                  int x, int y, int xa, int xb, int ya, int yb)
{
    const uint32_t base = (uint32_t)(xa >> 5);
    const int x0 = (int)(base << 5); // pixel of bit 0 of the window
    // (default: the bounding RECTANGLE of the region -- a superset, so the count is an upper bound: still exact, a few more entries
    // vote; the masks are the same for every row and no arms are fetched)
    const int rlo = xa - x0, rhi = xb - x0 + 1;
    const unsigned long long rm0 = irv_mask64(adc_imin(rlo, 64), adc_imin(rhi, 64)), rm1 = irv_mask64(adc_imax(rlo - 64, 0), adc_imax(rhi - 64, 0));
    int cnt = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int r = ya; r <= yb; r += IRV_RC_ROWS) {
        uint4 v[IRV_RC_ROWS];
        uint32_t a[IRV_RC_ROWS];
#pragma unroll
        for (int u = 0; u < IRV_RC_ROWS; u++) {
            const int rr = adc_imin(r + u, yb);
            v[u] = *reinterpret_cast<const uint4*>(px + (uint32_t)(rr * pitch) + base);
            if (IRV_RC_EXACT) a[u] = arms32[(uint32_t)(rr * W + x)];
        }
#pragma unroll
        for (int u = 0; u < IRV_RC_ROWS; u++) {
            unsigned long long m0 = rm0, m1 = rm1;
            if (IRV_RC_EXACT) {
                const int lo = x - (int)(a[u] & 255u) - x0, hi = x + (int)((a[u] >> 8) & 255u) - x0 + 1; // bits [lo, hi) of the 128-bit window
                m0 = irv_mask64(adc_imin(lo, 64), adc_imin(hi, 64));
                m1 = irv_mask64(adc_imax(lo - 64, 0), adc_imax(hi - 64, 0));
            }
            const int c = __popcll(((unsigned long long)v[u].x | ((unsigned long long)v[u].y << 32)) & m0) +
                          __popcll(((unsigned long long)v[u].z | ((unsigned long long)v[u].w << 32)) & m1);
            cnt += r + u <= yb ? c : 0;
        }
    }
    // the entry's own pixel does not count (a pixel does not vote for itself)
    if (ya <= yb) cnt -= (int)((px[(uint32_t)(y * pitch) + (uint32_t)(x >> 5)] >> (x & 31)) & 1u);
    return cnt;
}

// Wave-wide maximum / sum of non-negative ints in 6 DPP steps (row rotations, then the two row broadcasts of gfx9); the
// result is valid in lane 63 and returned wave-uniform.  (__shfl_xor is a ds_bpermute: ~100 cycles per step.)
#define IRV_DPP(V, CTRL, RMASK) __builtin_amdgcn_update_dpp(0, (V), (CTRL), (RMASK), 0xf, false)
__device__ __forceinline__ int irv_wave_max(int v)
{
    v = adc_imax(v, IRV_DPP(v, 0x121, 0xf)); // row_ror:1
    v = adc_imax(v, IRV_DPP(v, 0x122, 0xf));
    v = adc_imax(v, IRV_DPP(v, 0x124, 0xf));
    v = adc_imax(v, IRV_DPP(v, 0x128, 0xf));
    v = adc_imax(v, IRV_DPP(v, 0x142, 0xa)); // row_bcast:15 into rows 1, 3
    v = adc_imax(v, IRV_DPP(v, 0x143, 0xc)); // row_bcast:31 into rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int irv_wave_sum(int v)
{
    v += IRV_DPP(v, 0x121, 0xf);
    v += IRV_DPP(v, 0x122, 0xf);
    v += IRV_DPP(v, 0x124, 0xf);
    v += IRV_DPP(v, 0x128, 0xf);
    v += IRV_DPP(v, 0x142, 0xa);
    v += IRV_DPP(v, 0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum within each row of 16 lanes (row_shr with zero fill)
__device__ __forceinline__ int irv_row_prefix(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    return v;
}
#undef IRV_DPP

// A 16-byte block of the state map, read PAST the CU's L1 (sc1: served by the XCD's L2).  What the other workgroups of this XCD
// have written in THIS kernel so far is there -- the sweep down a band (irv_plan.h) lives on that; exactness does not.  (An
// acquire fence per vote -- buffer_inv sc1, which drops the whole L1 -- made the votes 7x slower, measured.)
#ifndef IRV_LOAD_CPOL
#define IRV_LOAD_CPOL 16 /* sc1 */
#endif
typedef unsigned int irv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 irv_ld_state(__amdgpu_buffer_rsrc_t rs, uint32_t elem)
{
    const irv_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(elem * 2u), 0, IRV_LOAD_CPOL);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// ------------------------------------------------------------------------------------------- the kernel of the chain
// Work-list entry: {pixel, arms of the pixel, max left | max right << 8 of its dependency box, row y}.
// Workgroups of up to 16 waves (blockDim.x = 64 * waves): the dirty entries of a workgroup's waves are
// pooled in LDS and dealt out to its waves round-robin, so a round takes ceil(pool / waves) votes per wave -- without
// the pooling a tail round is as slow as the unluckiest wave (3-4 votes of ~3 us each where the average is 0.2).
#define IRV_MAXW 16
// What only the BEGIN / BEGIN2 / FINAL kernels of a chain touch lives in a small block in device memory (IrvCold, filled per launch
// of a chain): as kernel arguments these six pointers stayed in scalar registers through the ROUND kernels' vote loop, which then
// spilled scalars into vector lanes -- 279 v_readlane in the hot loop, refine stage +0.5 ms (round 6, measured).
struct IrvCold {
    const uint8_t* label;
    float* disp;     // the chain's working copy of the map
    float* disp_io;  // the pipeline's map: read by the first BEGIN, written by FINAL (no copies)
    const uint16_t* sup_h;
    const uint32_t* bbox32;
    unsigned long long* listed_bits; // bit p: pixel p goes on the work list (BEGIN -> BEGIN2)
    int min_region, xcd_mode;
};
// (Register budget: 64 vector registers = 8 waves per SIMD, the whole grid co-resident.  `__launch_bounds__(1024, 8)` also caps the SCALAR
// registers at 80 -- the slack bounds of round 6 then made the vote loop spill scalars into vector lanes, 37 reloads per vote; with the
// caps spelled out separately the kernel may use 102 scalars and still runs 8 waves per SIMD.)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 8), amdgpu_num_vgpr(64), amdgpu_num_sgpr(102))) void k_irv_u(int32_t* __restrict__ ctrl, int k, const IrvCold* cold, uint16_t* st16, int4* list, uint8_t* chg,
                                                const uint32_t* __restrict__ arms32, int W, int H, int SP, int dmin,
                                                int D, int chg_bytes, int tpitch, int irv_ts, float irv_th,
                                                int32_t* __restrict__ evals_arr, int seg_cap, int32_t* wg_n /* entries per workgroup segment */,
                                                uint32_t* px_chg /* per-pixel change bitmap, IRV_PX_PLANES planes (slack budgets) */, int px_pitch,
                                                int use_slack)
{
    IRV_TR(8);
    IRV_T(0);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, WPB = blockDim.x >> 6, T = blockDim.x; // (wave: uniform, kept in a scalar register)
    const int gw = blockIdx.x * WPB + wave, NW = gridDim.x * WPB;
    // Everything the first phase of a ROUND needs is known at launch time: this wave's entries of batch 0, and -- the change
    // tiles being stamped and double-buffered by KERNEL index -- which plane and which stamp to look for.  So the entry,
    // state and tile loads are issued before the chain state (written by another XCD's workgroup 0 of the previous kernel:
    // the slowest load of the kernel) has arrived.  Slots beyond the list hold older entries or zeros: always in-bounds.
    // (The state is fetched with a VECTOR load -- lane l reads word l, lane 8 the accumulator -- because scalar loads return
    // out of order: the kernel-argument loads would wait for it.)
    int4* seg = list + (size_t)blockIdx.x * seg_cap; // this workgroup's segment of the work list (irv_plan.h)
    const int4 spec = seg[threadIdx.x];
    const int cword = ctrl[lane < 8 ? 16 * (k & 1) + lane : IRV_ACC + ((k + 63) & 63)];
    const int ngv = wg_n[blockIdx.x]; // (a vector load like the state: scalar loads return out of order)
    const uint32_t want4 = ((uint32_t)((k + 254) % 255) + 1u) * 0x01010101u; // stamp of kernel k - 1
    const uint32_t stamp = (uint32_t)(k % 255) + 1u;
    const uint8_t* chg_rd = chg + (size_t)((k + 1) & 1) * chg_bytes;
    uint8_t* chg_wr = chg + (size_t)(k & 1) * chg_bytes;
    const uint32_t px_words = (uint32_t)px_pitch * (uint32_t)H; // dwords of one bitmap plane
    uint32_t* px_wr = px_chg + (size_t)(k % IRV_PX_PLANES) * px_words;
    // Slots behind the end of the list hold IRV_LIST_END (see irv_plan.h): a wave that finds nothing else skips this phase.
    uint32_t spec_state = 0u;
    bool spec_box = false;
    if (__ballot(spec.x != IRV_LIST_END) != 0ull) {
        const bool have = spec.x != IRV_LIST_END;
        const int p = have ? spec.x : 0, y = have ? spec.w : 0, x = p - y * W;
        spec_state = st16[(uint32_t)(y * SP + x)]; // (32-bit offsets from a uniform base: saddr addressing, no 64-bit VGPR pairs)
        // the bounding box of the WHOLE region: every region pixel is an input of some iteration's vote
        const int top = (int)(((uint32_t)spec.y >> 16) & 255u), bot = (int)((uint32_t)spec.y >> 24);
        const int ml = (int)(((uint32_t)spec.z >> 16) & 255u), mr = (int)((uint32_t)spec.z >> 24);
        // (a lane without an entry gets an empty tile range: no loads)
        spec_box = irv_box_dirty(chg_rd, tpitch, adc_imax(0, x - ml) / IRV_TILE, have ? adc_imin(W - 1, x + mr) / IRV_TILE : -1,
                                 adc_imax(0, y - top) / IRV_TILE, adc_imin(H - 1, y + bot) / IRV_TILE, want4);
    }
    const IrvState sprev = {__builtin_amdgcn_readlane(cword, 0), __builtin_amdgcn_readlane(cword, 1), __builtin_amdgcn_readlane(cword, 2),
                            __builtin_amdgcn_readlane(cword, 3), __builtin_amdgcn_readlane(cword, 4), __builtin_amdgcn_readlane(cword, 5),
                            __builtin_amdgcn_readlane(cword, 6), __builtin_amdgcn_readlane(cword, 7)};
    const IrvPlan pl = irv_plan_from(sprev, k > 0 ? __builtin_amdgcn_readlane(cword, 8) : 0, k);
    IRV_T(1);
    if (pl.act != IRV_FINAL_WB && blockIdx.x == 0 && threadIdx.x == 0) irv_publish(ctrl, k, pl.s);
    if (pl.act == IRV_DONE) return;
    int32_t* acc = ctrl + IRV_ACC + (k & 63);
    if (pl.act == IRV_BEGIN || pl.act == IRV_FINAL_WB) {
        const uint8_t* label = cold->label;
        float *disp = cold->disp, *disp_io = cold->disp_io;
        const uint16_t* sup_h = cold->sup_h;
        unsigned long long* listed_bits = cold->listed_bits;
        const int min_region = cold->min_region;
        // evaluation statistics: every wave counts in its own slot (a same-address atomic per wave and round cost more
        // than the votes of a tail round: they retire at ~8 ns each); cleared by the first kernel, summed by the last
        if (pl.act == IRV_BEGIN)
            for (int t = blockIdx.x * T + threadIdx.x; t < NW; t += gridDim.x * T) evals_arr[t] = 0;
        if (pl.act == IRV_FINAL_WB && blockIdx.x == 0) {
            __shared__ int esum[IRV_MAXW];
            int e = 0;
            for (int t = threadIdx.x; t < NW; t += T) e += evals_arr[t];
            e = irv_wave_sum(e);
            if (lane == 0) esum[wave] = e;
            __syncthreads();
            if (threadIdx.x == 0) {
                IrvState fs = pl.s;
                fs.evals = 0;
                for (int w = 0; w < WPB; w++) fs.evals += esum[w];
                irv_publish(ctrl, k, fs);
            }
        }
        // the no-op kernels behind the end of the chain (the budget's surplus) then find no entry and skip the state / tile
        // round trip: the first batch of the segment becomes end markers
        if (pl.act == IRV_FINAL_WB) seg[threadIdx.x].x = IRV_LIST_END;
        if (pl.act == IRV_BEGIN) { // clear both change-tile planes (bytes, written as dwords) and the per-pixel change planes
            for (int t = blockIdx.x * T + threadIdx.x; t < chg_bytes / 2; t += gridDim.x * T) reinterpret_cast<uint32_t*>(chg)[t] = 0u;
            for (uint32_t t = blockIdx.x * T + threadIdx.x; t < IRV_PX_PLANES * px_words; t += gridDim.x * T) px_chg[t] = 0u;
        }
        // one coalesced pass over the image: state map / working copy (BEGIN), fills (FINAL)
        const int P = W * H;
        bool any_listed = false; // (wave-uniform)
        for (int c0 = blockIdx.x * T; c0 < P; c0 += gridDim.x * T) {
            const int p = c0 + threadIdx.x;
            bool li = false;
            if (p < P) {
                const int y = p / W, x = p - y * W;
                const size_t i16 = (size_t)y * SP + x;
                if (pl.act == IRV_FINAL_WB) { // fills: the vote result is best_bin + min_disparity (multistep_refiner.cpp:211)
                    float dv = disp[p];
                    const uint32_t sv = st16[i16];
                    if ((sv >> IRV_LIST_SHIFT) != 0u && (sv & IRV_BIN_MASK) != IRV_BIN_MASK) dv = (float)((int)(sv & IRV_BIN_MASK) + dmin);
                    disp_io[p] = dv; // the result goes straight back into the pipeline's map
                } else {
                    const float dv = disp_io[p]; // the LR-checked map itself ...
                    disp[p] = dv;                // ... seeds the working copy (what a continued chain's FINAL starts from)
                    const uint32_t lab = label[p];
                    const bool e = (lab == ADC_LABEL_MISMATCH || lab == ADC_LABEL_OCCLUSION) && (dv == ADC_INVALID_FLOAT);
                    // the vote needs count > irv_ts and count <= region size == horizontal-first support count, so
                    // pixels with sup_h <= irv_ts stay invalid whatever happens: on their list (they are invalid in every
                    // iteration), but not on the work list
                    li = e && ((int)sup_h[p] > min_region);
                    uint32_t bin = IRV_BIN_MASK;
                    if (dv != ADC_INVALID_FLOAT) {
                        const long b = lroundf(dv) - dmin; // multistep_refiner.cpp:193-196
                        if (b >= 0 && b < D) bin = (uint32_t)b; // (outside the histogram: never counted)
                    }
                    st16[i16] = (uint16_t)(bin | (e ? lab << IRV_LIST_SHIFT : 0u));
                }
            }
            if (pl.act == IRV_BEGIN) { // (the 64 pixels of a wave are consecutive and start at a multiple of 64)
                const unsigned long long m = __ballot(li);
                if (lane == 0 && c0 + wave * 64 < P) listed_bits[(c0 + wave * 64) >> 6] = m;
                any_listed |= m != 0ull;
            }
        }
        // an empty work list ends the chain (irv_plan_from): ONE store per wave that found a listed pixel (the same value: no atomic)
        if (pl.act == IRV_BEGIN && lane == 0 && any_listed) *acc = 1;
        return;
    }
    if (pl.act == IRV_BEGIN2) {
        const uint32_t* bbox32 = cold->bbox32;
        const unsigned long long* listed_bits = cold->listed_bits;
        const int xcd_mode = cold->xcd_mode;
        // The workgroup walks ITS tiles (irv_plan.h: tile t belongs to workgroup t % G) row by row and compacts the listed pixels
        // into its segment in the order (row inside the band, tile, column) -- the order in which the rounds evaluate them.
        __shared__ int wcnt[IRV_MAXW];
        __shared__ int base;
        if (threadIdx.x == 0) base = 0;
        __syncthreads();
        const int my_tiles = irv_wg_tiles(W, H, (int)gridDim.x, (int)blockIdx.x, xcd_mode), per_step = T / IRV_TCOLS;
        const int tk = threadIdx.x / IRV_TCOLS, xi = threadIdx.x % IRV_TCOLS;
#pragma unroll 1
        for (int j = 0; j < IRV_BAND; j++) {
#pragma unroll 1
            for (int k0 = 0; k0 < my_tiles; k0 += per_step) {
                bool li = false;
                int p = 0;
                if (k0 + tk < my_tiles) {
                    int band, tx;
                    irv_wg_tile(W, H, (int)gridDim.x, (int)blockIdx.x, k0 + tk, xcd_mode, &band, &tx);
                    const int y = band * IRV_BAND + j, x = tx * IRV_TCOLS + xi;
                    if (y < H && x < W) {
                        p = y * W + x;
                        li = (listed_bits[p >> 6] >> (p & 63)) & 1ull;
                    }
                }
                const unsigned long long m = __ballot(li);
                if (lane == 0) wcnt[wave] = __popcll(m);
                __syncthreads();
                int mine = base, tot = 0;
                for (int w = 0; w < WPB; w++) {
                    const int c = wcnt[w];
                    mine += w < wave ? c : 0;
                    tot += c;
                }
                if (li) // everything a round needs to know about the entry in ONE 16-byte load
                    seg[mine + __popcll(m & ((1ull << lane) - 1ull))] = make_int4(p, (int)arms32[p], (int)(bbox32[p] & 0xFFFF0000u), p / W); // (low half of z: the slack budget, 0)
                __syncthreads();
                if (threadIdx.x == 0) base += tot;
                __syncthreads();
            }
        }
        const int ng = base;
        for (int t = ng + threadIdx.x; t < ((ng + T - 1) / T) * T && t < seg_cap; t += T) seg[t].x = IRV_LIST_END; // the rest of the last batch
        if (threadIdx.x == 0) {
            wg_n[blockIdx.x] = ng;
            if (ng) atomicAdd(acc, ng); // (the chain state's n: statistics only)
        }
        return;
    }
    // ROUND.  Phase 1 (one entry per lane; done above for batch 0): did a pixel of the entry's region box change in the previous
    // round?  Change tiles are BYTES holding the stamp of the last kernel that changed a pixel of the 8x8 tile (0 = never; a
    // stamp aliasing a kernel 510 launches earlier can only cause a redundant evaluation, never a missed one).  Phase 2: the
    // dirty entries of the workgroup are pooled in LDS.  Phase 3: wave w evaluates pool entries w, w + waves, ...
    const int round = pl.s.round;
    extern __shared__ int lds_dyn[]; // [waves][IRV_LEVELS][D] histograms, then the pool: [waves][64] int4
    int* hist = lds_dyn + wave * (IRV_LEVELS * D);
    int4* pool = reinterpret_cast<int4*>(lds_dyn + ((WPB * IRV_LEVELS * D + 3) & ~3)); // (16-byte aligned whatever D is)
    __shared__ int pcount[IRV_MAXW];
    __shared__ int pidx[IRV_MAXW * 64]; // segment index of a pool item (its new slack budget is written back there)
    const int sub = lane >> 2, bslot = lane & 3;
    if (use_slack) { // the per-pixel change plane the NEXT kernel writes (read by this kernel's predecessor: free now)
        uint32_t* px_clr = px_chg + (size_t)((k + 1) % IRV_PX_PLANES) * px_words;
        for (uint32_t t = blockIdx.x * T + threadIdx.x; t < px_words; t += gridDim.x * T) px_clr[t] = 0u;
    }
    const __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc(st16, 0, (SP * H + 64) * 2, 0x00020000);
    const int ng = __builtin_amdgcn_readfirstlane(ngv); // entries of this workgroup's segment, in evaluation order
    int evals = 0;
    if (ng > 0) // the wave's histograms start empty; every vote clears the rows it has touched behind itself
        for (int b = lane; b < IRV_LEVELS * D; b += 64) hist[b] = 0;
    for (int b0 = 0; b0 < ng; b0 += T) {
        const int i = b0 + (int)threadIdx.x;
        int4 ent = spec;
        if (b0 != 0) {
            ent = seg[i];
            __syncthreads(); // the previous batch's pool has been consumed
        }
        uint32_t mystate = spec_state;
        bool box = spec_box;
        if (b0 != 0) { // (later batches are full up to ng: entries of this chain, or older ones / end markers behind ng)
            const bool have = ent.x != IRV_LIST_END;
            const int p = have ? ent.x : 0, y = have ? ent.w : 0, x = p - y * W;
            mystate = st16[(uint32_t)(y * SP + x)];
            const int top = (int)(((uint32_t)ent.y >> 16) & 255u), bot = (int)((uint32_t)ent.y >> 24);
            const int ml = (int)(((uint32_t)ent.z >> 16) & 255u), mr = (int)((uint32_t)ent.z >> 24);
            box = irv_box_dirty(chg_rd, tpitch, adc_imax(0, x - ml) / IRV_TILE, have ? adc_imin(W - 1, x + mr) / IRV_TILE : -1,
                                adc_imax(0, y - top) / IRV_TILE, adc_imin(H - 1, y + bot) / IRV_TILE, want4);
        }
        bool dirty = i < ng && (round == 0 || box);
        if (use_slack > 0 && round != 0 && __popcll(__ballot(dirty)) >= use_slack) { // (wave-uniform) slack budgets, irv_plan.h: an entry whose tiles were hit counts, one entry per lane, the pixels
            // of its region's bounding rectangle that changed in the previous kernel.  Budget used up: the entry votes again.  Otherwise it
            // is not looked at in this round (pool items are what a heavy round's time is made of) and the count comes off its budget.
            // `use_slack` = hit entries per wave from which the wave filters (ADC_IRV_SLACK; 0 = no budgets at all).  Default 1: letting
            // waves with few hits skip the walk below (a chain of up to 18 dependent trips the workgroup waits for at the pool barrier) was
            // measured -- thresholds 1 and 4 the same, 8 / 16 / 32 slower (refine 3.22 / 3.21 / 3.28 / 3.34 / 3.53 ms).
            // (Measured and dropped, profiles/r6_k8_experiments.txt: entries with a small remaining budget kept as "maybes" that count this
            // kernel's changes at their turn in the sweep -- fewer rounds, but every maybe is a pool item and the vote loop ran out of
            // scalar registers; the same at the turn for everything: 48 % fewer votes, no shorter rounds.)
            const bool cand = dirty;
            const int p = cand ? ent.x : 0, y = cand ? ent.w : 0, x = p - y * W;
            const int top = (int)(((uint32_t)ent.y >> 16) & 255u), bot = (int)((uint32_t)ent.y >> 24);
            const int ml = (int)(((uint32_t)ent.z >> 16) & 255u), mr = (int)((uint32_t)ent.z >> 24);
            const int used = irv_region_changes(px_chg + (size_t)((k + 2) % IRV_PX_PLANES) * px_words, px_pitch, arms32, W, x, y, x - ml, x + mr,
                                                cand ? y - top : 1, cand ? y + bot : 0);
            const int rem = (int)((uint32_t)ent.z & 0xFFFFu) - used;
            dirty = cand && rem < 0;
            if (cand && rem >= 0 && used > 0) seg[i].z = (int)(((uint32_t)ent.z & 0xFFFF0000u) | (uint32_t)rem);
        }
        IRV_T(2);
        // pool: every wave puts its dirty entries into its own 64 slots and publishes the count -- ONE barrier; the
        // consumers find pool item t by a prefix sum over the (<= 16) counts
        const unsigned long long dm = __ballot(dirty);
        if (lane == 0) pcount[wave] = __popcll(dm);
        if (dirty) { // {pixel, arms, state | read box << 16, row}
            pool[wave * 64 + __popcll(dm & ((1ull << lane) - 1ull))] = make_int4(ent.x, ent.y, (int)((mystate & 0xFFFFu) | ((uint32_t)ent.z & 0xFFFF0000u)), ent.w);
            pidx[wave * 64 + __popcll(dm & ((1ull << lane) - 1ull))] = i;
        }
        __syncthreads();
        const int cnt_l = lane < WPB ? pcount[lane] : 0;
        const int incl = irv_row_prefix(cnt_l); // inclusive prefix over the first 16 lanes
        const int excl = incl - cnt_l;
        const int total = __builtin_amdgcn_readlane(incl, 15);
        IRV_T(3);
        IRV_TV(10, total);
        for (int t = wave; t < total; t += WPB) {
            const int sw = __popcll(__ballot(lane < WPB && incl <= t)); // the wave whose slots hold item t
            const int4 pe = pool[sw * 64 + (t - __builtin_amdgcn_readlane(excl, sw))]; // (one LDS address for the whole wave: a broadcast read)
            const int eidx = __builtin_amdgcn_readfirstlane(pidx[sw * 64 + (t - __builtin_amdgcn_readlane(excl, sw))]);
            const int p = __builtin_amdgcn_readfirstlane(pe.x), armsp = __builtin_amdgcn_readfirstlane(pe.y), y = __builtin_amdgcn_readfirstlane(pe.w);
            const uint32_t pz = (uint32_t)__builtin_amdgcn_readfirstlane(pe.z);
            const uint32_t cur = pz & 0xFFFFu; // (only this wave writes the entry in this round)
            const int lp = (int)(cur >> IRV_LIST_SHIFT); // the entry's list: 1 mismatch, 2 occlusion
            const int x = p - y * W;
            uint32_t lvls = 0u; // iterations from which a pixel seen by this lane starts to count
            const int top = (int)(((uint32_t)armsp >> 16) & 255u), nrows = top + (int)((uint32_t)armsp >> 24) + 1; // region rows y-top .. y+bottom
            const uint32_t own = (uint32_t)(y * SP + x) & ~7u; // the entry's own block: address of masked-out loads
            // the read box: blocks blkL .. blkR cover the widest row of the region
            const int blkL = (x - (int)((pz >> 16) & 255u)) >> 3, blkR = (x + (int)(pz >> 24)) >> 3;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (int rbase = 0; rbase < nrows; rbase += 64) {
                // the H arms of (up to 64) region rows (lane r holds row rbase + r; handed to the row slots with a shuffle) AND
                // the first 16 rows x 4 blocks of the read box in ONE memory round trip: the arms only decide which pixels of a
                // block belong to the region (round 4; before, the blocks were requested when the arms had arrived)
                const int myr = rbase + lane;
                uint32_t a2 = 0;
                if (myr < nrows) a2 = arms32[(uint32_t)((y - top + myr) * W + x)];
                const int rend = adc_imin(nrows - rbase, 64);
                uint4 vfirst;
                {
                    const bool in0 = sub < rend && blkL + bslot <= blkR;
                    vfirst = irv_ld_state(st_rs, in0 ? (uint32_t)((y - top + rbase + sub) * SP + (blkL + bslot) * 8) : own);
                }
                IRV_T(4);
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                for (int r0 = 0; r0 < rend; r0 += 16) {
                    const int r = r0 + sub;
                    const uint32_t arm2 = (uint32_t)__shfl((int)a2, r & 63, 64);
                    const bool rowok = r < rend;
                    const int yt = y - top + rbase + r;
                    const int xl = x - (int)(arm2 & 255u), xr = x + (int)((arm2 >> 8) & 255u);
                    const int b0x = xl >> 3, b1x = xr >> 3;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                    for (int bo = 0;; bo += 4) { // one iteration unless the read box spans more than 4 blocks
                        const int blk = blkL + bo + bslot;
                        const bool use = rowok && blk >= b0x && blk <= b1x;
                        uint4 v = vfirst;
                        if (r0 + bo != 0) { // (uniform) loads stay unconditional: masked-out lanes read the entry's own block
                            const uint32_t addr = rowok && blk <= blkR ? (uint32_t)(yt * SP + blk * 8) : own;
                            v = irv_ld_state(st_rs, addr);
                        }
                        IRV_T(5);
                        // The 8 pixels of the block, decoded as packed halfwords (irv_plan.h: irv_decode_block): which pixels
                        // count from which iteration on, and which of them share the first one's key (iteration | bin)?
                        if (use) {
                            const IrvBlock bd = irv_decode_block(v.x, v.y, v.z, v.w, blk * 8, xl, xr, yt, y, x, lp);
                            uint32_t okm = bd.okm;
                            // One LDS atomic per DISTINCT key of the block (same-address LDS atomics serialise: never one per pixel).  The
                            // pixels of the first key come with the decode; a region row that straddles a disparity edge holds a second
                            // surface (24-27 % of the blocks of a natural image hold several bins, tools/irv_block_stats.py), rarely a
                            // third: the loop takes the lowest remaining pixel's key and all pixels that share it.
                            if (okm != 0u) {
                                atomicAdd(&hist[(int)(bd.first >> IRV_F_SHIFT) * D + (int)(bd.first & IRV_BIN_MASK)], __popc(bd.same));
                                lvls |= 1u << (bd.first >> IRV_F_SHIFT);
                                okm &= ~bd.same;
                            }
#pragma clang loop unroll(disable)
                            while (okm != 0u) {
                                uint32_t key2;
                                const uint32_t same2 = irv_same_key_mask(bd.k0, bd.k1, bd.k2, bd.k3, okm, &key2);
                                atomicAdd(&hist[(int)(key2 >> IRV_F_SHIFT) * D + (int)(key2 & IRV_BIN_MASK)], __popc(same2));
                                lvls |= 1u << (key2 >> IRV_F_SHIFT);
                                okm &= ~same2;
                            }
                        }
                        if (!__any(rowok && (blkL + bo + 4 <= b1x))) break;
                    }
                }
            }
            evals++;
            IRV_T(6);
            // The votes of the iterations, lowest first, on the CUMULATIVE histograms (row `it` += the last non-empty row below it);
            // an iteration that adds no pixel repeats the previous decision (which failed).  First maximum (lowest bin on ties)
            // and total count (multistep_refiner.cpp:199-209): key = count << 11 | (2047 - bin).
            uint32_t present = 0u;
#pragma unroll
            for (int it = 0; it < IRV_LEVELS; it++) present |= (__ballot((lvls >> it) & 1u) != 0ull ? 1u : 0u) << it;
            uint32_t ns = IRV_BIN_MASK | ((uint32_t)lp << IRV_LIST_SHIFT); // no iteration's vote passes: invalid
            int below = -1;
            // slack budget of this outcome (irv_plan.h): the minimum over the levels up to the deciding one; levels without new pixels
            // repeat their predecessor's histogram (same slack), leading empty levels are failing levels with c = m = 0
            // (computed in VECTOR registers on purpose -- every lane the same values: as scalar code these bounds cost the vote loop ~40 scalar
            // registers it does not have: 37 spill reloads per vote, refine stage +0.27 ms, measured)
            float thv = irv_th;
            asm volatile("" : "+v"(thv));
            const IrvSlackK sq = irv_slack_consts(thv);
            int K = (present & 1u) ? 0xFFFF : irv_level_slack(false, 0, 0, 0, irv_ts, sq);
#pragma clang loop unroll(disable)
            for (int it = 0; it < IRV_LEVELS; it++) {
                if (!((present >> it) & 1u)) continue; // (uniform)
                int key = 0, cnt = 0;
                for (int b = lane; b < D; b += 64) {
                    int hv = hist[it * D + b];
                    if (below >= 0) { hv += hist[below * D + b]; hist[it * D + b] = hv; }
                    cnt += hv;
                    key = adc_imax(key, hv > 0 ? ((hv << 11) | (0x7FF - b)) : 0);
                }
                below = it;
                key = irv_wave_max(key);
                cnt = irv_wave_sum(cnt);
                const int bh = key >> 11, bbin = 0x7FF - (key & 0x7FF);
                const bool pass = adc_vote_decide(bbin, bh, cnt, dmin, irv_ts, irv_th) != ADC_INVALID_FLOAT;
                if (use_slack) { // (runner-up <= everything outside the top bin)
                    int cv = cnt, bv = bh;
                    asm volatile("" : "+v"(cv), "+v"(bv));
                    K = adc_imin(K, irv_level_slack(pass, cv, bv, cv - bv, irv_ts, sq));
                }
                if (pass) {
                    ns = (uint32_t)bbin | ((uint32_t)it << IRV_F_SHIFT) | ((uint32_t)lp << IRV_LIST_SHIFT);
                    break;
                }
            }
            // leave the histograms empty: only the rows of iterations that occurred were written (1-2 of 5 for most votes)
#pragma clang loop unroll(disable)
            for (int it = 0; it < IRV_LEVELS; it++)
                if ((present >> it) & 1u) // (uniform)
                    for (int b = lane; b < D; b += 64) hist[it * D + b] = 0;
            IRV_T(7);
            if (lane == 0 && use_slack) seg[eidx].z = (int)((pz & 0xFFFF0000u) | (uint32_t)K);
            if (lane == 0 && ns != cur) { // bin and iteration in ONE store; both matter to the votes that see this pixel
                st16[(uint32_t)(y * SP + x)] = (uint16_t)ns;
                chg_wr[(uint32_t)((y / IRV_TILE) * tpitch + x / IRV_TILE)] = (uint8_t)stamp;
                if (use_slack) atomicOr(&px_wr[(uint32_t)(y * px_pitch + (x >> 5))], 1u << (x & 31));
                *acc = 1;
            }
        }
    }
    if (lane == 0 && evals) evals_arr[gw] += evals;
    IRV_TV(11, evals);
    IRV_TR(9);
}

// --------------------------------------------------------------------------------------------------------- host side
static int irv_min_region(const adc_handle* h)
{
    const int Lmax = adc_imax(0, adc_imin(h->p.opt.cross_L1, 255));
    return Lmax <= 127 ? h->p.opt.irv_ts : -1; // u16 support counts cannot wrap for L <= 127
}
// Launch shape of the chain: 8 waves per workgroup (fewer when the histograms + pool would not fit into 64 KB of LDS);
// workgroups: one per 1024 pixels, rounded up to a power of two, between 128 and 1024 (four per CU of an MI355X: the whole grid
// co-resident).  Round 5, structured 1080p pairs, refine stage (profiles/r5_k8_workgroup_shapes.txt): 16 waves x 512 workgroups
// 3.58 / 4.03 ms, 8 x 1024 3.27 / 3.65 (fewer rounds AND fewer evaluations: a workgroup's waves take CONSECUTIVE entries of its
// segment, so smaller workgroups follow the sweep order more closely), 8 x 768 3.50 / 3.67, 8 x 2048 3.86 / 4.18, 4 x 1024
// 3.90 / 3.68; KITTI size: 16 x 256 = 8 x 512.  ADC_IRV_GRID / ADC_IRV_WPB override.
int adc_irv_grid(size_t pixels)
{
    static const int g_env = [] { const char* e = getenv("ADC_IRV_GRID"); return e ? atoi(e) : 0; }();
    if (g_env > 0) return g_env;
    int g = 128;
    while (g < 1024 && (size_t)g * 1024 < pixels) g *= 2;
    return g;
}
static int irv_wpb(int D)
{
    static const int wmax = [] { const char* e = getenv("ADC_IRV_WPB"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= IRV_MAXW ? v : 8; }();
    int w = 1;
    while (2 * w <= wmax) w *= 2;
    while (w > 1 && (size_t)w * IRV_LEVELS * D * 4 + (size_t)w * 64 * 16 > 60 * 1024) w >>= 1;
    return w;
}
size_t adc_irv_waves(int grid) { return (size_t)IRV_MAXW * grid + (size_t)grid; } // per-wave evaluation counters + per-workgroup segment lengths
// entries the work list must hold: one segment of whole batches per workgroup of the chain's grid (irv_plan.h: irv_seg_cap)
// The band -> XCD sweep (irv_plan.h: irv_wg_tile) assumes that workgroup g of a launch runs on XCD g % 8.  Exactness never depends
// on it, but under another partition mode / XCD count / dispatch order the in-kernel sweep would silently degrade to Jacobi rounds
// (round-5 advisor finding).  So the assumption is PROBED once per device: 1024 workgroups report the XCD they run on
// (HW_REG_XCC_ID); the sweep layout is used iff all workgroups with the same g % 8 share one XCD.  ADC_IRV_XCD=0 / 1 overrides.
__global__ __launch_bounds__(64) void k_irv_xcc_probe(int* __restrict__ out)
{
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(id & 15u);
}
int adc_irv_probe_xcd_mode(int device)
{
    static const int env = [] { const char* e = getenv("ADC_IRV_XCD"); return e ? atoi(e) : -1; }();
    if (env >= 0) return env;
    static std::mutex mu;
    static int cached[64];
    static bool have[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    const int dv = (device >= 0 && device < 64) ? device : 0;
    if (have[dv]) return cached[dv];
    int mode = 0;
    int* d = nullptr;
    int hst[1024];
    if (hipMalloc(&d, sizeof(hst)) == hipSuccess) {
        if (hipMemset(d, 0xFF, sizeof(hst)) == hipSuccess) {
            hipLaunchKernelGGL(k_irv_xcc_probe, dim3(1024), dim3(64), 0, 0, d);
            if (hipGetLastError() == hipSuccess && hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost) == hipSuccess) {
                mode = 1;
                for (int g = 0; g < 1024; g++)
                    if (hst[g] < 0 || hst[g] != hst[g & 7]) mode = 0;
            }
        }
        hipFree(d);
    }
    (void)hipGetLastError();
    cached[dv] = mode;
    have[dv] = true;
    return mode;
}
size_t adc_irv_list_entries(int W, int H, int D, int grid) // (room for either layout)
{
    return (size_t)grid * (size_t)adc_imax((int)irv_seg_cap(W, H, grid, irv_wpb(D), 0), (int)irv_seg_cap(W, H, grid, irv_wpb(D), 1));
}
static int irv_use_slack()
{
    static const int v = [] { const char* e = getenv("ADC_IRV_SLACK"); return e ? atoi(e) : 1; }(); // hit entries per wave from which the wave filters (measured: 1 = 4 < 8 < 16 < 32: every wave filters); 0 = off
    return v;
}
size_t adc_irv_px_words(int W, int H) { return (size_t)IRV_PX_PLANES * irv_px_pitch(W) * H + 16; }
static hipError_t irv_launch(adc_handle* h, int k0, int count)
{
    if (k0 == 0) { // the block of rarely used arguments of this chain (stream-ordered: in front of the chain's first kernel).  Staged in
        // PINNED memory: a copy from pageable host memory is synchronous inside the runtime (+0.19 ms per Match, measured)
        IrvCold c;
        c.label = h->label; c.disp = h->disp_vote; c.disp_io = h->disp_l; c.sup_h = h->sup_h;
        c.bbox32 = reinterpret_cast<const uint32_t*>(h->irv_bbox); c.listed_bits = reinterpret_cast<unsigned long long*>(h->elig);
        c.min_region = irv_min_region(h); c.xcd_mode = h->irv_xcd_mode;
        static_assert(sizeof(IrvCold) <= 64, "adc_handle::irv_cold and pin_flags[32..47] hold 64 bytes");
        if (!h->pin_flags) return hipErrorInvalidValue;
        if (!h->irv_cold_valid || memcmp(h->irv_cold_host, &c, sizeof(c)) != 0) { // (the same block Match after Match unless buffers swapped roles)
            // two staging slots, alternating: a slot is rewritten two changes later at the earliest -- a handle has one Match in flight,
            // so the copy that read it has long drained
            int32_t* slot = h->pin_flags + 32 + 16 * (h->irv_cold_flip & 1);
            h->irv_cold_flip ^= 1;
            memcpy(slot, &c, sizeof(c));
            const hipError_t e = hipMemcpyAsync(h->irv_cold, slot, sizeof(IrvCold), hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) return e;
            memcpy(h->irv_cold_host, &c, sizeof(c));
            h->irv_cold_valid = 1;
        }
    }
    const AdcParams& p = h->p;
    const int tpitch = h->chg_pitch, chg_bytes = tpitch * ((p.H + IRV_TILE - 1) / IRV_TILE);
    const int wpb = irv_wpb(p.D);
    const size_t lds = (size_t)((wpb * IRV_LEVELS * p.D + 3) & ~3) * 4 + (size_t)wpb * 64 * 16;
    for (int i = 0; i < count; i++)
        hipLaunchKernelGGL(k_irv_u, dim3((unsigned)h->irv_grid), dim3(64 * wpb), lds, h->stream, h->vote_counters, k0 + i,
                           reinterpret_cast<const IrvCold*>(h->irv_cold), h->st16, reinterpret_cast<int4*>(h->vote_list), h->chg_a,
                           reinterpret_cast<const uint32_t*>(h->arms), p.W, p.H, h->st16_pitch, p.dmin, p.D, chg_bytes, tpitch, p.opt.irv_ts, p.opt.irv_th,
                           h->vote_evals_arr, (int)irv_seg_cap(p.W, p.H, h->irv_grid, wpb, h->irv_xcd_mode), h->vote_evals_arr + (size_t)IRV_MAXW * h->irv_grid,
                           h->irv_px, irv_px_pitch(p.W), irv_use_slack());
    return hipGetLastError();
}

// Enqueue-only: bbox, the budgeted chain (its first kernel reads the LR-checked map disp_l and seeds the working copy
// disp_vote, its FINAL kernel writes the result back into disp_l), state read-back (pinned).
hipError_t adc_run_region_voting(adc_handle* h)
{
    const AdcParams& p = h->p;
    hipError_t e;
    if ((e = hipMemsetAsync(h->vote_counters, 0, IRV_CTRL_INTS * sizeof(int32_t), h->stream)) != hipSuccess) return e;
    if (h->irv_budget < 4) h->irv_budget = 4;
    if ((e = irv_launch(h, 0, h->irv_budget)) != hipSuccess) return e;
    h->irv_chain = h->irv_budget;
    // the state the last kernel published (slot chain & 1), read by adc_wait / adc_voting_finish
    if (h->pin_flags)
        e = hipMemcpyAsync(h->pin_flags + 16, h->vote_counters + 16 * (h->irv_chain & 1), 8 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    h->irv_pending = 1;
    return e;
}

// After the stream has drained: did the chain reach DONE?  If not, continue it synchronously (batches of 64 kernels) until
// it does, and copy the result to disp_l.  Adapts the budget of the next Match.  *continued = 1 when the stages behind
// the voting have to be redone.
hipError_t adc_voting_finish(adc_handle* h, int* continued)
{
    *continued = 0;
    if (!h->irv_pending || !h->pin_flags) return hipSuccess;
    h->irv_pending = 0;
    hipError_t e;
    int32_t* st = h->pin_flags + 16;
    int guard = 0;
    while (st[0] != IRV_DONE) {
        *continued = 1;
        if ((e = irv_launch(h, h->irv_chain, 64)) != hipSuccess) return e;
        h->irv_chain += 64;
        if ((e = hipMemcpyAsync(st, h->vote_counters + 16 * (h->irv_chain & 1), 8 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
        if (++guard > (1 << 16)) return hipErrorUnknown; // cannot happen: the triangular system converges in <= 10 n rounds
    }
    h->vote_rounds = st[5];
    h->vote_evals = st[6];
    if (*continued) h->irv_overflows++; // (the FINAL kernel of the continued chain has written the result into disp_l)
    // budget of the next Match: the kernels this one actually needed (st[7] = index of the first kernel that found
    // nothing left to do) + 40 % (at least 2) + 2
    const int used = st[7] + 1;
    static const int fixed = [] { const char* ev = getenv("ADC_IRV_BUDGET"); return ev ? atoi(ev) : 0; }();
    // (the longest chain of the last 8 Matches of the handle: the pairs of a stream differ -- at the KITTI size 3 of 23 distinct
    // structured pairs overran a budget taken from their predecessor alone; a surplus kernel is a 2.5 us no-op, an overrun a
    // synchronous continuation)
    h->irv_used_hist[h->irv_used_pos++ & 7] = used;
    int longest = 0;
    for (int i = 0; i < 8; i++) longest = adc_imax(longest, h->irv_used_hist[i]);
    // (round 5: the chain of a natural 1080p image is 50-75 kernels instead of ~350 and varies more from pair to pair, relatively:
    // a quarter of margin instead of an eighth -- 10 surplus kernels cost 0.05 ms, a continuation a synchronisation and the tail stages)
    // a short chain (an image with next to no listed pixel: 6 kernels) gets 4 surplus kernels, a long one 40 %
    // (measured on 24 distinct structured pairs: with a quarter of margin 1-2 of ~28 Matches still overran -- the chains of a stream
    // vary 48-75 kernels --, and a continuation costs ~1.5 ms where 6 more surplus kernels cost 0.03 ms: 40 %)
    // (an EMPTY work list -- BEGIN, then DONE: no round whose count could vary -- needs no margin beyond the floor of 4 kernels:
    // two surplus kernels less per Match of a noise-like stream, 10 us)
    h->irv_budget = fixed > 0 ? fixed : (longest <= 2 ? 4 : adc_imin(1 << 16, longest + adc_imax(2 * longest / 5, 2) + 2));
    return hipSuccess;
}
