// irv_plan.h -- the device-side state machine of the region-voting chain (k_voting.hip), shared with the CPU emulation
// (tests/emul/emul_irv.cpp): constants of the 16-bit state map, the control-block layout, the list layout and the pure
// function every block of kernel k evaluates to find out what this kernel has to do.
#pragma once
#include <stdint.h>
#include "adc_device_fn.h"

#define IRV_TILE 8
#define IRV_BIN_MASK 0x7FFu
#define IRV_FINAL 0x4000u
#define IRV_ELIG 0x8000u
#define IRV_PPT 4 // pixels per thread and block iteration of the BEGIN phase (one list-length atomic per 4 x blockDim pixels)

enum { IRV_NONE = 0, IRV_BEGIN, IRV_ROUND, IRV_FINAL_WB, IRV_DONE };
// ctrl layout (int32): state slot s at ctrl[16*s ..]: {did, pass, round, filled_any, n, rounds_total, evals, kdone};
// accumulator ring at ctrl[IRV_ACC + (k & 63)] (BEGIN: list length; ROUND: "a value changed")
#define IRV_ACC 64
#define IRV_CTRL_INTS 160
struct IrvState { int did, pass, round, filled_any, n, rounds, evals, kdone; }; // kdone: index of the kernel that found the chain finished
struct IrvPlan { int act; IrvState s; };

// One kernel type: kernel k reads the state its predecessor published (slot k & 1) and the predecessor's accumulator.
//   BEGIN    write the previous pass's fills back, mark the eligible pixels of the next list, build the work list
//   ROUND    round r of the pass: every open entry whose dependency box changed in round r-1 (round 0: every entry) votes.
//            Change tiles: kernel k stamps (k % 255) + 1 into plane k & 1 and reads the stamps of kernel k-1 in the other
//            plane -- both known at launch time, so the check can run before the state has arrived.
//   FINAL_WB write the last pass's fills back (and sum the per-wave evaluation counters into the state's evals)
ADC_HD IrvPlan irv_plan_from(IrvState s, int prev, int k) // s: the published state, prev: the predecessor's accumulator
{
    IrvPlan p;
    if (s.did == IRV_NONE) {
        p.act = IRV_BEGIN;
        s.pass = 0; s.round = 0; s.filled_any = 0; s.n = 0;
    } else if (s.did == IRV_BEGIN) {
        p.act = IRV_ROUND;
        s.n = prev; // list length
        s.round = 0;
    } else if (s.did == IRV_ROUND) {
        int fa = s.filled_any | ((s.round == 0 && prev != 0) ? 1 : 0); // a pass that fills anything does so in round 0
        if (prev != 0) { // the round changed something: next round
            p.act = IRV_ROUND;
            s.round++;
            s.filled_any = fa;
        } else { // a whole round without a change: the pass has converged
            bool fin = false;
            if (s.pass & 1) { // end of an iteration (multistep_refiner.cpp:167-171): nothing filled -> the rest are no-ops
                if (!fa) fin = true;
                fa = 0;
            }
            s.pass++;
            if (s.pass >= 10) fin = true;
            p.act = fin ? IRV_FINAL_WB : IRV_BEGIN;
            s.round = 0; s.filled_any = fa; s.n = 0;
        }
    } else {
        p.act = IRV_DONE;
    }
    if (p.act == IRV_ROUND) s.rounds++;
    if (p.act == IRV_DONE && s.did != IRV_DONE) s.kdone = k; // first kernel with nothing left to do
    s.did = p.act;
    p.s = s;
    return p;
}
ADC_HD IrvPlan irv_plan(const int32_t* ctrl, int k)
{
    const int32_t* in = ctrl + 16 * (k & 1);
    const IrvState s = {in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7]};
    return irv_plan_from(s, k > 0 ? ctrl[IRV_ACC + ((k - 1) & 63)] : 0, k);
}
ADC_HD void irv_publish(int32_t* ctrl, int k, const IrvState& s)
{
    int32_t* out = ctrl + 16 * ((k + 1) & 1);
    out[0] = s.did; out[1] = s.pass; out[2] = s.round; out[3] = s.filled_any; out[4] = s.n; out[5] = s.rounds; out[6] = s.evals; out[7] = s.kdone;
    ctrl[IRV_ACC + ((k + 2) & 63)] = 0;
}

// Work-list layout.  The chain's grid has G workgroups of WPB waves; a batch is B = 64 * WPB * G entries.  Entry i (in the
// order the BEGIN phase compacts them: raster order inside chunks of pixels) belongs to WORKGROUP (i % B) % G: consecutive
// entries -- which tend to be dirty in the same rounds (a fill front is a few hundred adjacent pixels) -- land in different
// workgroups, so every workgroup's pool of dirty entries holds about dirty / G of them.  INSIDE the workgroup the entries
// are packed densely: t = (i % B) / G sits in wave t / 64, lane t % 64 (round 4; before: wave t % WPB, lane t / WPB).  A
// list of n entries then occupies the first ceil(n / G / 64) waves of every workgroup with all 64 lanes, and the other waves
// find nothing but end markers (IRV_LIST_END, written over the unused slots by the first round of a pass and by the FINAL
// kernel) and skip the state / change-tile phase altogether: in the long tail of a pass, where a round is a chain of memory
// round trips plus the instruction issue of 8192 waves each checking a handful of lanes, a quarter (pass 0) to a tenth
// (later passes) of the waves do the checking.  The votes are dealt out over ALL waves of the workgroup from the LDS pool.
#define IRV_LIST_END (-1)
ADC_HD long irv_list_slot(long i, int G, int WPB)
{
    const long B = 64L * WPB * G, r = i % B;
    const long blk = r % G, t = r / G;
    return (i / B) * B + blk * (64L * WPB) + t;
}
// the list index held by lane `lane` of wave `wave` of workgroup `blk` in the batch that starts at b0 (inverse of irv_list_slot)
ADC_HD long irv_list_index(long b0, int blk, int wave, int lane, int G) { return b0 + (long)(wave * 64 + lane) * G + blk; }
// Entry = {pixel, arms of the pixel (left | right << 8 | top << 16 | bottom << 24), boxes, row}; boxes = max left | max right << 8
// over the rows y - top .. y (the dependency box: only pixels that precede p can influence its vote) | max left << 16 |
// max right << 24 over ALL region rows (the rectangle a vote starts to read before the row arms have arrived).

// ---------------------------------------------------------------------------------------------------------------------
// Packed-halfword decode of a region row block (8 pixels of the 16-bit state map in four dwords) and of a change-tile row,
// shared with the CPU tests (tests/emul/emul_irv.cpp checks them against per-pixel loops).
// bits 0..3 of n -> bytes 0..3 (0xFF where the bit is set)
ADC_HD uint32_t irv_expand_nibble(uint32_t n) { return ((n * 0x00204081u) & 0x01010101u) * 0xFFu; }
// bit k of the words t0..t3 -> bits 0, 2, 4, 6 and bit k + 16 -> bits 1, 3, 5, 7 (k >= 6): the same flag of the 8 packed
// halfwords of a 16-byte block as one 8-bit mask
ADC_HD uint32_t irv_gather8(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, int k)
{
    const uint32_t sel = 0x00010001u;
    const uint32_t u = ((t0 >> k) & sel) | ((t1 >> (k - 2)) & (sel << 2)) | ((t2 >> (k - 4)) & (sel << 4)) | ((t3 >> (k - 6)) & (sel << 6));
    return (u | (u >> 15)) & 0xffu;
}
// ~(byte mask) of the four dwords of a 16-byte tile row for the tiles [txb, last] of a row that starts at tile cb
// (cb = txb & ~3, last <= txb + 11): bytes outside the range are forced non-zero before the any-zero-byte test
ADC_HD void irv_tile_row_masks(int txb, int last, uint32_t* nk)
{
    const int cb = txb & ~3;
    const uint32_t m16 = ((1u << (last + 1 - cb)) - 1u) & ~((1u << (txb - cb)) - 1u);
    nk[0] = ~irv_expand_nibble(m16 & 15u); nk[1] = ~irv_expand_nibble((m16 >> 4) & 15u);
    nk[2] = ~irv_expand_nibble((m16 >> 8) & 15u); nk[3] = ~irv_expand_nibble(m16 >> 12);
}
// non-zero iff a byte of `word` that is not masked out by nk equals the stamp replicated in want4
ADC_HD uint32_t irv_tile_hit(uint32_t word, uint32_t nk, uint32_t want4)
{
    const uint32_t x = (word ^ want4) | nk;
    return (x - 0x01010101u) & ~x & 0x80808080u;
}
struct IrvBlock {
    uint32_t okm;   // pixels that count in the vote (bit q = pixel px0 + q)
    uint32_t first; // bin of the lowest counted pixel
    uint32_t same;  // the counted pixels that fall into `first` (== okm when single)
    bool single;    // all counted pixels fall into `first`
    bool open;      // an eligible predecessor of p in this block is not final yet
};
// The pixels of `rem` (a non-empty subset of a block's counted pixels) that share the bin of the lowest one; *bin = that bin.
ADC_HD uint32_t irv_same_bin_mask(uint32_t vx, uint32_t vy, uint32_t vz, uint32_t vw, uint32_t rem, uint32_t* bin)
{
    const uint32_t b0 = vx & 0x07FF07FFu, b1 = vy & 0x07FF07FFu, b2 = vz & 0x07FF07FFu, b3 = vw & 0x07FF07FFu;
    const int q0 = __builtin_ffs((int)rem) - 1;
    const uint32_t wsel = q0 < 2 ? b0 : (q0 < 4 ? b1 : (q0 < 6 ? b2 : b3));
    const uint32_t f = (wsel >> (16 * (q0 & 1))) & IRV_BIN_MASK, f2 = f * 0x00010001u;
    // halfwords that differ from that bin: (d + 0x7FF) carries into bit 11 iff d != 0
    const uint32_t difm = irv_gather8((b0 ^ f2) + 0x07FF07FFu, (b1 ^ f2) + 0x07FF07FFu, (b2 ^ f2) + 0x07FF07FFu, (b3 ^ f2) + 0x07FF07FFu, 11);
    *bin = f;
    return rem & ~difm;
}
// Block of 8 pixels px0 .. px0 + 7 of region row yt, of which [xl, xr] belong to the region of p = (x, y).
ADC_HD IrvBlock irv_decode_block(uint32_t vx, uint32_t vy, uint32_t vz, uint32_t vw, int px0, int xl, int xr, int yt, int y, int x)
{
    IrvBlock r;
    const uint32_t inm = (((2u << adc_imin(xr - px0, 7)) - 1u) & ~((1u << adc_imax(xl - px0, 0)) - 1u)) & 0xffu;
    // pixels that precede p in raster order
    const uint32_t prem = yt < y ? 0xffu : (yt == y ? ((1u << adc_imax(0, adc_imin(x - px0, 8))) - 1u) : 0u);
    const uint32_t b0 = vx & 0x07FF07FFu, b1 = vy & 0x07FF07FFu, b2 = vz & 0x07FF07FFu, b3 = vw & 0x07FF07FFu; // bins
    const uint32_t elm = irv_gather8(vx, vy, vz, vw, 15);  // eligible
    const uint32_t finm = irv_gather8(vx, vy, vz, vw, 14); // final
    // bin == 0x7FF (invalid / never counted): 0x7FF + 1 carries into bit 11 of the halfword
    const uint32_t invm = irv_gather8(b0 + 0x00010001u, b1 + 0x00010001u, b2 + 0x00010001u, b3 + 0x00010001u, 11);
    // eligible pixels of this pass are visible only if they precede p (already processed by the sequential scan);
    // otherwise they are still invalid
    r.okm = inm & ~invm & (~elm | prem);
    // an eligible predecessor that is not final yet: this vote may still change
    r.open = (inm & elm & prem & ~finm) != 0u;
    r.first = 0u;
    r.same = 0u;
    r.single = false;
    if (r.okm != 0u) {
        const int q0 = __builtin_ffs((int)r.okm) - 1;
        const uint32_t wsel = q0 < 2 ? b0 : (q0 < 4 ? b1 : (q0 < 6 ? b2 : b3));
        r.first = (wsel >> (16 * (q0 & 1))) & IRV_BIN_MASK;
        const uint32_t f2 = r.first * 0x00010001u;
        // halfwords that differ from the first counted bin: (d + 0x7FF) carries into bit 11 iff d != 0
        const uint32_t difm = irv_gather8((b0 ^ f2) + 0x07FF07FFu, (b1 ^ f2) + 0x07FF07FFu, (b2 ^ f2) + 0x07FF07FFu, (b3 ^ f2) + 0x07FF07FFu, 11);
        r.same = r.okm & ~difm;
        r.single = r.same == r.okm;
    }
    return r;
}
