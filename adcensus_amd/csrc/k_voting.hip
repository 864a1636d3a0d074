// k_voting.hip -- K8 iterative region voting (MultiStepRefiner::IterativeRegionVoting, multistep_refiner.cpp:153-227),
// DEVICE-DRIVEN: the host enqueues a fixed chain of kernels A, B, A, B, ... and never looks at the data; which list is
// being filled, which round runs and when a pass has converged is decided on the device.
//
// Semantics kept exactly: 5 iterations x {mismatches, occlusions}; inside a pass the reference fills the
// still-invalid pixels of the list in raster order IN PLACE, so a vote sees the fills of the list pixels that precede
// it.  Per pass we iterate "value(p) = vote(p | values of the eligible pixels that precede p in raster order)" to its
// fixed point: the system is triangular, every (chaotic, in-place) iteration converges to the sequential result, and a
// whole round without a change proves the fixed point.  Votes only ever need lround(d) - dmin of a region pixel, so the
// pass works on a 16-bit STATE MAP (2 bytes per pixel, L2-resident at 1080p):
//     bits 0..10  histogram bin (0x7FF = invalid / never counted)
//     bit  14     final   (the value can no longer change in this pass: set together with the value, ONE 16-bit store)
//     bit  15     eligible (pixel of the current list that was invalid at pass start)
// A vote (one wave per entry) reads the cross region as 16-byte row blocks (8 pixels per lane and load, 16 region rows
// x 4 blocks per trip: one trip for a typical region of 13 rows x 13 pixels) into an LDS histogram.  Entries whose
// eligible predecessors were all final get the final bit and are never evaluated again; from round 1 on only entries
// whose dependency box (k_irv_bbox) saw a change in the previous round are re-evaluated (8x8 change tiles).
//
// The chain.  Kernel k (even = A, odd = B) reads the state its predecessor wrote (slot k & 1) and the predecessor's
// accumulator acc[(k-1) & 63] (list length / number of dirty entries / "something changed"), derives its own action --
// every block derives the same one -- and block 0 publishes the new state into slot (k+1) & 1 and clears the
// accumulator kernel k+2 will use.  No host round trip, no grid barrier, no ticket atomics:
//     A: BEGIN  write the previous pass's fills back to the float map, mark the eligible pixels of the next list, compact
//               the work list (pixels whose region is too small to ever pass the vote are left out)
//        CHECK  compact the entries that must be re-evaluated in this round
//        FINAL  write the last pass's fills back
//     B: VOTE   evaluate the work list (round 0) or the dirty list
// The chain length is a BUDGET (adc_handle::irv_budget, adapted from the rounds the previous Match of the handle needed);
// when it is exhausted before the state machine reaches DONE, adc_wait continues the same chain synchronously and redoes
// the stages behind it -- a performance cliff, never a different result.
#include "adc_internal.h"
#include "adc_device_fn.h"

#include <stdio.h>
#include <stdlib.h>

#include "irv_plan.h"

// Dependency box of a pixel's vote: the cross region of p spans rows y-top..y+bottom, but only pixels that PRECEDE p in
// raster order can influence it, i.e. rows y-top..y; its horizontal extent is the widest H arm of those rows.
// bbox[p] = {top, max left arm, max right arm} (computed once per Match; arms do not change).
__global__ __launch_bounds__(256) void k_irv_bbox(const uchar4* __restrict__ arms, uchar4* __restrict__ bbox, int W, int H)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uchar4 a = arms[(size_t)y * W + x];
    int ml = 0, mr = 0;
    for (int t = -(int)a.z; t <= 0; t++) {
        const uchar4 q = arms[(size_t)(y + t) * W + x];
        ml = adc_imax(ml, (int)q.x);
        mr = adc_imax(mr, (int)q.y);
    }
    bbox[(size_t)y * W + x] = make_uchar4(a.z, (unsigned char)ml, (unsigned char)mr, 0);
}

// mask of the bytes [lo, hi) of a dword, 0 <= lo, hi <= 4
__device__ __forceinline__ uint32_t irv_byte_range_mask(int lo, int hi)
{
    const uint32_t a = hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
    const uint32_t b = lo >= 4 ? 0xFFFFFFFFu : ((1u << (8 * lo)) - 1u);
    return hi > lo ? (a & ~b) : 0u;
}

// ------------------------------------------------------------------------------------------------------- kernel A
__global__ __launch_bounds__(256) void k_irv_a(int32_t* __restrict__ ctrl, int k, const uint8_t* __restrict__ label, float* __restrict__ disp,
                                               float* disp_io, /* the pipeline's map: read by the first BEGIN, written by FINAL (no copies) */
                                               const uint16_t* __restrict__ sup_h, uint16_t* __restrict__ st16, int2* __restrict__ list,
                                               int2* __restrict__ dlist, uint8_t* __restrict__ chg, const uchar4* __restrict__ bbox,
                                               const uint32_t* __restrict__ arms32, int W, int H, int SP, int dmin, int D, int min_region,
                                               int chg_bytes, int tpitch)
{
    const IrvPlan pl = irv_plan(ctrl, k);
    if (blockIdx.x == 0 && threadIdx.x == 0) irv_publish(ctrl, k, pl.s);
    if (pl.act == IRV_DONE) return;
    int32_t* acc = ctrl + IRV_ACC + (k & 63);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int P = W * H;
    if (pl.act == IRV_BEGIN || pl.act == IRV_FINAL_WB) {
        const bool have_state = !(pl.act == IRV_BEGIN && pl.s.pass == 0); // pass 0: the state map is not initialised yet
        const int which = (pl.s.pass & 1) ? ADC_LABEL_OCCLUSION : ADC_LABEL_MISMATCH; // mismatches, then occlusions (:170-171)
        if (pl.act == IRV_BEGIN) // clear the change-tile map of the pass (bytes, written as dwords)
            for (int t = blockIdx.x * 256 + threadIdx.x; t < chg_bytes / 4; t += gridDim.x * 256) reinterpret_cast<uint32_t*>(chg)[t] = 0u;
        __shared__ int wcnt[IRV_PPT][4];
        __shared__ int base;
        for (int c0 = blockIdx.x * (256 * IRV_PPT); c0 < P; c0 += gridDim.x * (256 * IRV_PPT)) {
            unsigned long long m[IRV_PPT];
            bool listed[IRV_PPT];
#pragma unroll
            for (int q = 0; q < IRV_PPT; q++) {
                const int p = c0 + q * 256 + threadIdx.x;
                listed[q] = false;
                if (p < P) {
                    const int y = p / W, x = p - y * W;
                    const size_t i16 = (size_t)y * SP + x;
                    float dv = have_state ? disp[p] : disp_io[p]; // pass 0 starts from the LR-checked map itself
                    if (have_state) { // fills of the previous pass: the vote result is best_bin + min_disparity (:211)
                        const uint32_t s = st16[i16];
                        if ((s & IRV_ELIG) && (s & IRV_BIN_MASK) != IRV_BIN_MASK) {
                            dv = (float)((int)(s & IRV_BIN_MASK) + dmin);
                            disp[p] = dv;
                        }
                    }
                    if (!have_state) disp[p] = dv;              // ... and seeds the working copy
                    if (pl.act == IRV_FINAL_WB) disp_io[p] = dv; // the result goes straight back into the pipeline's map
                    if (pl.act == IRV_BEGIN) {
                        const bool e = (label[p] == which) && (dv == ADC_INVALID_FLOAT);
                        // the vote needs count > irv_ts and count <= region size == horizontal-first support count, so
                        // pixels with sup_h <= irv_ts stay invalid whatever happens: eligible (they order the pass) but
                        // final from the start and not on the work list
                        listed[q] = e && ((int)sup_h[p] > min_region);
                        uint32_t bin = IRV_BIN_MASK;
                        if (dv != ADC_INVALID_FLOAT) {
                            const long b = lroundf(dv) - dmin; // multistep_refiner.cpp:193-196
                            if (b >= 0 && b < D) bin = (uint32_t)b; // (outside the histogram: never counted)
                        }
                        st16[i16] = (uint16_t)(bin | (e ? IRV_ELIG : 0u) | (listed[q] ? 0u : IRV_FINAL));
                    }
                }
                m[q] = __ballot(listed[q]);
                if (lane == 0) wcnt[q][wave] = __popcll(m[q]);
            }
            if (pl.act != IRV_BEGIN) continue; // (uniform)
            __syncthreads();
            if (threadIdx.x == 0) {
                int tot = 0;
#pragma unroll
                for (int q = 0; q < IRV_PPT; q++) tot += wcnt[q][0] + wcnt[q][1] + wcnt[q][2] + wcnt[q][3];
                base = tot ? atomicAdd(acc, tot) : 0; // one same-address atomic per 2048 pixels (they retire at ~8 ns each)
            }
            __syncthreads();
            int off = base;
#pragma unroll
            for (int q = 0; q < IRV_PPT; q++) {
                const int p = c0 + q * 256 + threadIdx.x;
                int mine = off;
                for (int w = 0; w < wave; w++) mine += wcnt[q][w];
                // entry = {pixel, its arms}: the vote starts from ONE 8-byte load (no dependent arms[p] round trip)
                if (listed[q]) list[mine + __popcll(m[q] & ((1ull << lane) - 1ull))] = make_int2(p, (int)arms32[p]);
                off += wcnt[q][0] + wcnt[q][1] + wcnt[q][2] + wcnt[q][3];
            }
            __syncthreads();
        }
        return;
    }
    // CHECK: one thread per open list entry decides whether some pixel of its dependency box changed in the previous
    // round; dirty entries are compacted into dlist.  Change tiles are BYTES holding (round % 255) + 1 of the last round
    // that changed a pixel of the 8x8 tile (0 = never; a stamp aliasing a round 255 rounds earlier can only cause a
    // redundant evaluation, never a missed one), so one 16-byte load covers a tile row of the box (<= 11 tiles for arms
    // <= 34) and the 6 rows of a box are in flight together -- the 30 dependent dword loads of the first version were
    // what this kernel spent its time on.
    const int round = pl.s.round, n = pl.nwork;
    const uint32_t want4 = ((uint32_t)((round - 1) % 255) + 1u) * 0x01010101u;
    __shared__ int cw[4];
    __shared__ int cbase;
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const int i = i0 + threadIdx.x;
        bool dirty = false;
        int2 ent = make_int2(-1, 0);
        if (i < n) {
            ent = list[i];
            const int p = ent.x;
            const int y = p / W, x = p - y * W;
            if (!(st16[(size_t)y * SP + x] & IRV_FINAL)) { // final values are never re-evaluated
                const uchar4 bb = bbox[p];
                const int tx0 = adc_imax(0, x - (int)bb.y) / IRV_TILE, tx1 = adc_imin(W - 1, x + (int)bb.z) / IRV_TILE;
                const int ty0 = adc_imax(0, y - (int)bb.x) / IRV_TILE, ty1 = y / IRV_TILE;
#pragma unroll 1
                for (int tyb = ty0; tyb <= ty1; tyb += 6)
#pragma unroll 1
                    for (int txb = tx0; txb <= tx1; txb += 12) { // (one iteration each for arm limits <= 34)
                        const int cb = txb & ~3, last = adc_imin(tx1, txb + 11); // 16 bytes from a dword-aligned column
                        uint4 v0, v1, v2, v3, v4, v5; // the (up to) 6 tile rows of the box, all in flight
                        const uint8_t* cp = chg + cb;
                        v0 = *reinterpret_cast<const uint4*>(cp + (size_t)adc_imin(tyb + 0, ty1) * tpitch);
                        v1 = *reinterpret_cast<const uint4*>(cp + (size_t)adc_imin(tyb + 1, ty1) * tpitch);
                        v2 = *reinterpret_cast<const uint4*>(cp + (size_t)adc_imin(tyb + 2, ty1) * tpitch);
                        v3 = *reinterpret_cast<const uint4*>(cp + (size_t)adc_imin(tyb + 3, ty1) * tpitch);
                        v4 = *reinterpret_cast<const uint4*>(cp + (size_t)adc_imin(tyb + 4, ty1) * tpitch);
                        v5 = *reinterpret_cast<const uint4*>(cp + (size_t)adc_imin(tyb + 5, ty1) * tpitch);
                        // (rows past ty1 repeat row ty1: harmless duplicates.)  Bytes outside [txb, last] are forced non-zero
                        // before the "any zero byte" test of (word ^ want4)
                        uint32_t hit = 0u;
#define IRV_ANYZ(WORD, CW)                                                                                          \
    do {                                                                                                            \
        const int lo_ = adc_imax(0, adc_imin(4, txb - (cb + 4 * (CW)))), hi_ = adc_imax(0, adc_imin(4, last + 1 - (cb + 4 * (CW)))); \
        const uint32_t vm_ = irv_byte_range_mask(lo_, hi_);                                                          \
        const uint32_t x_ = ((WORD) ^ want4) | ~vm_;                                                                \
        hit |= (x_ - 0x01010101u) & ~x_ & 0x80808080u;                                                               \
    } while (0)
#define IRV_ROW(V) do { IRV_ANYZ((V).x, 0); IRV_ANYZ((V).y, 1); IRV_ANYZ((V).z, 2); IRV_ANYZ((V).w, 3); } while (0)
                        IRV_ROW(v0); IRV_ROW(v1); IRV_ROW(v2); IRV_ROW(v3); IRV_ROW(v4); IRV_ROW(v5);
#undef IRV_ROW
#undef IRV_ANYZ
                        dirty |= hit != 0u;
                    }
            }
        }
        const unsigned long long mm = __ballot(dirty);
        if (lane == 0) cw[wave] = __popcll(mm);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = cw[0] + cw[1] + cw[2] + cw[3];
            cbase = tot ? atomicAdd(acc, tot) : 0;
        }
        __syncthreads();
        if (dirty) {
            int off = cbase;
            for (int w = 0; w < wave; w++) off += cw[w];
            dlist[off + __popcll(mm & ((1ull << lane) - 1ull))] = ent;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------- kernel B
// One wave per entry.  Trip = 16 region rows x 4 row blocks of 8 pixels (lane = row slot * 4 + block slot).
__global__ __launch_bounds__(256) void k_irv_b(int32_t* __restrict__ ctrl, int k, const int2* __restrict__ list,
                                               const int2* __restrict__ dlist, uint16_t* st16, const uchar4* __restrict__ arms,
                                               uint8_t* __restrict__ chg, int W, int H, int SP, int dmin, int D, int irv_ts, float irv_th,
                                               int tpitch, int list_cap)
{
    // the first entry of this wave from BOTH lists, issued before the plan is known (one dependent round trip less)
    const int e_first = adc_imin((int)(blockIdx.x * 4 + (threadIdx.x >> 6)), list_cap - 1);
    const int2 spec_l = list[e_first], spec_d = dlist[e_first];
    const IrvPlan pl = irv_plan(ctrl, k);
    if (blockIdx.x == 0 && threadIdx.x == 0) irv_publish(ctrl, k, pl.s);
    if (pl.act != IRV_VOTE) return;
    const int round = pl.s.round, n = pl.nwork;
    const int2* work = round == 0 ? list : dlist;
    const uint32_t stamp = (uint32_t)(round % 255) + 1u;
    int32_t* acc = ctrl + IRV_ACC + (k & 63);
    extern __shared__ int hist_all[]; // [4][D]: one histogram per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* hist = hist_all + wave * D;
    const int nwaves = gridDim.x * 4;
    const int sub = lane >> 2, bslot = lane & 3;
    for (int e = blockIdx.x * 4 + wave; e < n; e += nwaves) {
        int2 ent;
        if (e == e_first) { ent.x = round == 0 ? spec_l.x : spec_d.x; ent.y = round == 0 ? spec_l.y : spec_d.y; }
        else ent = work[e];
        const int p = ent.x;
        const int y = p / W, x = p - y * W;
        for (int b = lane; b < D; b += 64) hist[b] = 0;
        bool deps_open = false;
        const int top = (int)(((uint32_t)ent.y >> 16) & 255u), nrows = top + (int)((uint32_t)ent.y >> 24) + 1; // region rows y-top .. y+bottom
        const uint32_t cur = st16[(size_t)y * SP + x]; // (only this wave writes the entry in this round)
        const size_t own = ((size_t)y * SP + x) & ~(size_t)7;     // the entry's own block: address of masked-out loads
        for (int rbase = 0; rbase < nrows; rbase += 64) {
            // the H arms of (up to 64) region rows in ONE round trip (lane r holds row rbase + r), handed to the row
            // slots with a shuffle
            const int myr = rbase + lane;
            uint32_t a2 = 0;
            if (myr < nrows) a2 = reinterpret_cast<const uint32_t*>(arms)[(size_t)(y - top + myr) * W + x];
            const int rend = adc_imin(nrows - rbase, 64);
            for (int r0 = 0; r0 < rend; r0 += 16) {
                const int r = r0 + sub;
                const uint32_t arm2 = (uint32_t)__shfl((int)a2, r & 63, 64);
                const bool rowok = r < rend;
                const int yt = y - top + rbase + r;
                const int xl = x - (int)(arm2 & 255u), xr = x + (int)((arm2 >> 8) & 255u);
                const int b0 = xl >> 3, b1 = xr >> 3;
                for (int bo = 0;; bo += 4) { // one iteration unless a row spans more than 4 blocks
                    const int blk = b0 + bo + bslot;
                    const bool use = rowok && blk <= b1;
                    const size_t addr = use ? (size_t)yt * SP + (size_t)blk * 8 : own;
                    const uint4 v = *reinterpret_cast<const uint4*>(st16 + addr); // loads stay unconditional
                    if (use) {
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const int px = blk * 8 + q;
                            const uint32_t wq = q < 2 ? v.x : (q < 4 ? v.y : (q < 6 ? v.z : v.w)); // (q is a constant after unrolling)
                            const uint32_t s = (wq >> (16 * (q & 1))) & 0xffffu;
                            const bool in = px >= xl && px <= xr;
                            const bool el = (s & IRV_ELIG) != 0;
                            const bool pre = yt < y || (yt == y && px < x); // precedes p in raster order
                            const uint32_t bin = s & IRV_BIN_MASK;
                            // eligible pixels of this pass are visible only if they precede p (already processed by the
                            // sequential scan); otherwise they are still invalid
                            if (in && bin != IRV_BIN_MASK && (!el || pre)) atomicAdd(&hist[bin], 1);
                            // an eligible predecessor that is not final yet: this vote may still change
                            if (in && el && pre && !(s & IRV_FINAL)) deps_open = true;
                        }
                    }
                    if (!__any(rowok && (b0 + bo + 4 <= b1))) break;
                }
            }
        }
        // first maximum (lowest bin on ties) and total count (multistep_refiner.cpp:199-209)
        int bh = 0, bbin = 0x7fffffff, cnt = 0;
        for (int b = lane; b < D; b += 64) {
            const int hv = hist[b];
            cnt += hv;
            if (hv > bh) { bh = hv; bbin = b; }
        }
#pragma unroll
        for (int mk = 32; mk >= 1; mk >>= 1) {
            const int oh = __shfl_xor(bh, mk, 64), ob = __shfl_xor(bbin, mk, 64);
            cnt += __shfl_xor(cnt, mk, 64);
            const bool take = (oh > bh) || (oh == bh && ob < bbin);
            bh = take ? oh : bh;
            bbin = take ? ob : bbin;
        }
        const bool fill = adc_vote_decide(bbin, bh, cnt, dmin, irv_ts, irv_th) != ADC_INVALID_FLOAT;
        const bool all_final = __ballot(deps_open) == 0ull; // every eligible predecessor in the region was already final
        if (lane == 0) {
            const size_t i16 = (size_t)y * SP + x;
            const uint32_t nb = fill ? (uint32_t)bbin : IRV_BIN_MASK;
            const uint32_t ns = nb | IRV_ELIG | (all_final ? IRV_FINAL : 0u);
            if (ns != cur) st16[i16] = (uint16_t)ns; // value and final bit in ONE store
            if (nb != (cur & IRV_BIN_MASK)) {
                chg[(size_t)(y / IRV_TILE) * tpitch + x / IRV_TILE] = (uint8_t)stamp;
                *acc = 1;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------- host side
static int irv_min_region(const adc_handle* h)
{
    const int Lmax = adc_imax(0, adc_imin(h->p.opt.cross_L1, 255));
    return Lmax <= 127 ? h->p.opt.irv_ts : -1; // u16 support counts cannot wrap for L <= 127
}
static hipError_t irv_launch_pair(adc_handle* h, int k0, int npairs)
{
    const AdcParams& p = h->p;
    const int P = p.W * p.H;
    const int tpitch = h->chg_pitch, chg_bytes = tpitch * ((p.H + IRV_TILE - 1) / IRV_TILE);
    const unsigned ga = (unsigned)adc_imax(1, adc_imin((P + 256 * IRV_PPT - 1) / (256 * IRV_PPT), 1024));
    static const unsigned gb = [] { const char* e = getenv("ADC_IRV_GRID"); const int v = e ? atoi(e) : 2048; return (unsigned)(v > 0 ? v : 2048); }();
    uint8_t* chg = h->chg_a;
    int2* list = reinterpret_cast<int2*>(h->vote_list);
    int2* dlist = reinterpret_cast<int2*>(h->vote_dirty);
    for (int i = 0; i < npairs; i++) {
        const int k = k0 + 2 * i;
        hipLaunchKernelGGL(k_irv_a, dim3(ga), dim3(256), 0, h->stream, h->vote_counters, k, h->label, h->disp_vote, h->disp_l, h->sup_h, h->st16,
                           list, dlist, chg, reinterpret_cast<const uchar4*>(h->irv_bbox), reinterpret_cast<const uint32_t*>(h->arms), p.W, p.H,
                           h->st16_pitch, p.dmin, p.D, irv_min_region(h), chg_bytes, tpitch);
        hipLaunchKernelGGL(k_irv_b, dim3(gb), dim3(256), (size_t)4 * p.D * sizeof(int), h->stream, h->vote_counters, k + 1, list, dlist,
                           h->st16, reinterpret_cast<const uchar4*>(h->arms), chg, p.W, p.H, h->st16_pitch, p.dmin, p.D, p.opt.irv_ts,
                           p.opt.irv_th, tpitch, P);
    }
    return hipGetLastError();
}

// Enqueue-only: bbox, the budgeted chain (its first kernel reads the LR-checked map disp_l and seeds the working copy
// disp_vote, its FINAL kernel writes the result back into disp_l), state read-back (pinned).
hipError_t adc_run_region_voting(adc_handle* h)
{
    const AdcParams& p = h->p;
    hipError_t e;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_irv_bbox, grid, block, 0, h->stream, reinterpret_cast<const uchar4*>(h->arms),
                       reinterpret_cast<uchar4*>(h->irv_bbox), p.W, p.H);
    if ((e = hipMemsetAsync(h->vote_counters, 0, 160 * sizeof(int32_t), h->stream)) != hipSuccess) return e;
    if (h->irv_budget < 4) h->irv_budget = 4;
    if ((e = irv_launch_pair(h, 0, h->irv_budget)) != hipSuccess) return e;
    h->irv_chain = 2 * h->irv_budget;
    // the state the last kernel published (slot chain & 1), read by adc_wait / adc_voting_finish
    if (h->pin_flags)
        e = hipMemcpyAsync(h->pin_flags + 16, h->vote_counters + 16 * (h->irv_chain & 1), 8 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    h->irv_pending = 1;
    return e;
}

// After the stream has drained: did the chain reach DONE?  If not, continue it synchronously (batches of 32 rounds) until
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
        if ((e = irv_launch_pair(h, h->irv_chain, 32)) != hipSuccess) return e;
        h->irv_chain += 64;
        if ((e = hipMemcpyAsync(st, h->vote_counters + 16 * (h->irv_chain & 1), 8 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
        if (++guard > (1 << 16)) return hipErrorUnknown; // cannot happen: a pass converges in <= n rounds
    }
    h->vote_rounds = st[5];
    h->vote_evals = st[6];
    if (*continued) h->irv_overflows++; // (the FINAL kernel of the continued chain has written the result into disp_l)
    // budget of the next Match: the kernel pairs this one actually needed (st[7] = index of the first kernel that found
    // nothing left to do) + 12 % + 2
    const int used = (st[7] + 2) / 2;
    static const int fixed = [] { const char* ev = getenv("ADC_IRV_BUDGET"); return ev ? atoi(ev) : 0; }();
    h->irv_budget = fixed > 0 ? fixed : adc_imin(1 << 15, used + used / 8 + 2);
    return hipSuccess;
}
