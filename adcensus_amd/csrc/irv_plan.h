// irv_plan.h -- the device-side state machine of the region-voting chain (k_voting.hip), shared with the CPU emulation
// (tests/emul/emul_irv.cpp): constants of the 16-bit state map, the control-block layout, the list layout and the pure
// function every block of kernel k evaluates to find out what this kernel has to do.
#pragma once
#include <stdint.h>
#include "adc_device_fn.h"

#define IRV_TILE 8
// 16-bit state of a pixel: bin (11 bits; 0x7FF = invalid, never counted) | fill iteration f (3 bits) | list (2 bits: 0 = not on a
// list, 1 = mismatch, 2 = occlusion == the outlier label).  bits 0..13 = the KEY f << 11 | bin.
#define IRV_BIN_MASK 0x7FFu
#define IRV_F_SHIFT 11
#define IRV_KEY_MASK 0x3FFFu
#define IRV_LIST_SHIFT 14
#define IRV_LEVELS 5 // iterations of the reference (multistep_refiner.cpp:166)
#define IRV_PPT 4 // pixels per thread and block iteration of the BEGIN phase (one list-length atomic per 4 x blockDim pixels)

// THE FORMULATION (round 5).  The reference runs 5 iterations x {mismatches, occlusions}, every pass in place in raster order
// (multistep_refiner.cpp:153-227).  A filled pixel stays filled, so the value of a listed pixel over the five iterations is
// INVALID, .., INVALID, b, .., b: ONE state (f, b) = (iteration of the fill, bin) per pixel.  Region pixel q counts in the vote of p
// at iteration `it` iff it >= t(q):
//     q not on a list           t = 0                (its LR-checked value)
//     same list, before p       t = f_q              (the in-place scan has already passed q in this iteration)
//     same list, behind p       t = f_q + 1          (... passes it later: p sees the previous iteration's value)
//     other list                t = f_q + 1 for a mismatch p (the mismatch pass of an iteration runs first), f_q for an occlusion p
// and the state of p is (first iteration whose vote passes, its bin) -- the five votes of a pixel are the five CUMULATIVE
// histograms of one gather of its region.  This is a system of equations that is triangular in the order (iteration, list,
// raster position): any chaotic in-place iteration converges to its unique solution = the sequential result, and a whole
// round without a change proves the fixed point.  Rounds 1-4 iterated the ten passes one after the other (~255 rounds at 1080p:
// every pass has a long tail of rounds that are latency chains); iterating all of them at once lets the tails overlap:
// tools/irv_joint_rounds.py, 960x540 structured pair: 58 rounds instead of 195, 1.2x the evaluations.
enum { IRV_NONE = 0, IRV_BEGIN, IRV_ROUND, IRV_FINAL_WB, IRV_DONE, IRV_BEGIN2 };
// ctrl layout (int32): state slot s at ctrl[16*s ..]: {did, -, round, -, n, rounds_total, evals, kdone};
// accumulator ring at ctrl[IRV_ACC + (k & 63)] (BEGIN: "a pixel is listed"; BEGIN2: list length; ROUND: "a state changed")
#define IRV_ACC 64
#define IRV_CTRL_INTS 160
struct IrvState { int did, pass, round, filled_any, n, rounds, evals, kdone; }; // kdone: index of the kernel that found the chain finished (pass / filled_any: unused since round 5)
struct IrvPlan { int act; IrvState s; };

// One kernel type: kernel k reads the state its predecessor published (slot k & 1) and the predecessor's accumulator.
//   BEGIN    seed the working copy of the map, build the state map and the bitmap of the pixels that go on the work list (every
//            listed pixel whose region is large enough to ever pass a vote): one coalesced pass over the image
//   BEGIN2   every workgroup compacts the listed pixels of ITS tiles into its segment of the work list, in evaluation order
//   ROUND    every entry whose region saw a change in round r-1 (round 0: every entry) is evaluated.
//            Change tiles: kernel k stamps (k % 255) + 1 into plane k & 1 and reads the stamps of kernel k-1 in the other
//            plane -- both known at launch time, so the check can run before the state has arrived.
//   FINAL_WB write the fills back (and sum the per-wave evaluation counters into the state's evals)
ADC_HD IrvPlan irv_plan_from(IrvState s, int prev, int k) // s: the published state, prev: the predecessor's accumulator
{
    IrvPlan p;
    if (s.did == IRV_NONE) {
        p.act = IRV_BEGIN;
        s.pass = 0; s.round = 0; s.filled_any = 0; s.n = 0;
    } else if (s.did == IRV_BEGIN) {
        // (round 6) BEGIN raises its accumulator when a pixel goes on the work list; with an EMPTY list -- a noise-like image: every
        // invalid pixel's region is too small to ever pass a vote -- the chain is done: no list to build, no round, nothing to write back
        p.act = prev != 0 ? IRV_BEGIN2 : IRV_DONE;
    } else if (s.did == IRV_BEGIN2) {
        p.act = IRV_ROUND;
        s.n = prev; // list length
        s.round = 0;
    } else if (s.did == IRV_ROUND) {
        if (prev != 0) { // the round changed something: next round
            p.act = IRV_ROUND;
            s.round++;
        } else { // a whole round without a change: the fixed point
            p.act = IRV_FINAL_WB;
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

// Work-list layout and evaluation ORDER (round 5).  A fill front moves down the image: most inputs of a vote lie in the rows
// above the pixel (and the votes of later iterations need the earlier iterations' values of the pixels below).  Evaluated all at
// once (one Jacobi round per kernel) a front advances one row per kernel and every entry near it is evaluated again and again
// (measured: 12 evaluations per entry); evaluated top-down IN PLACE it runs through in a few sweeps.  So the image is cut into
// BANDS of IRV_BAND rows and tiles of IRV_BAND x IRV_TCOLS pixels; the tiles of a band are dealt out over the workgroups of ONE
// XCD (irv_wg_tile below; neighbouring tiles -- which tend to be busy in the same rounds -- land in different workgroups: with
// tiles 32 columns wide the busiest workgroup of a round held 3x the average and set the round's time, measured), and a
// workgroup's segment of the list holds the entries of ITS tiles sorted by (row inside the band, tile, column).  The workgroup works through its segment in that order (its waves
// take consecutive dirty entries), all workgroups side by side: at any time the chip evaluates about the same row of every band,
// with the rows above it already settled in this very kernel (as far as they belong to the same band).  Nothing depends on that pace -- the iteration converges to the same
// fixed point under any schedule (irv_plan.h, top) -- only the number of rounds does (tools/irv_joint_rounds.py, 960x540: 19
// rounds and 0.38 M evaluations with bands of 16 rows against 58 rounds / 0.84 M all at once, 195 / 0.61 M pass after pass).
// Segment of workgroup g: list[g * cap ..], cap = irv_seg_cap (whole batches of 64 * WPB entries); n_g entries, then
// IRV_LIST_END up to the end of the batch.
#define IRV_LIST_END (-1)
#ifndef IRV_BAND
#define IRV_BAND 8 // (measured at 1080p, refine stage of two structured pairs: 4 rows 3.61 / 4.10 ms, 8: 3.43 / 3.76, 16: 3.76 / 4.25, 32: 4.34 / 5.16)
#endif
#ifndef IRV_TCOLS
#define IRV_TCOLS 1
#endif
ADC_HD int irv_tiles_x(int W) { return (W + IRV_TCOLS - 1) / IRV_TCOLS; }
ADC_HD int irv_bands(int H) { return (H + IRV_BAND - 1) / IRV_BAND; }
// Which workgroup owns which tiles.  A vote sees what other workgroups have written IN THE SAME KERNEL only through the L2 of its
// own XCD (the L2s of different XCDs are not coherent inside a kernel), and workgroup g runs on XCD g % 8: so a whole BAND belongs
// to one XCD -- band b to XCD b % 8 -- and its tiles are dealt out over the G / 8 workgroups of that XCD.  (G % 8 != 0: plain
// round-robin; still exact, the sweep just sees less of itself.)
//   band index inside the XCD: bi = b / 8;   workgroup = (b % 8) + 8 * ((bi * (tiles_x + IRV_SKEW) + tx) % (G / 8))
// IRV_SKEW (round 6, measured, default 0): without it a workgroup owns the SAME columns in every band of its XCD (1920 columns =
// 15 x 128), and the invalid regions of an image are tall -- occlusion bands along depth edges -- so the workgroups whose columns run
// through them hold up to twice the average list (1080p structured pair: 360 entries against a mean of 184; 242 with the columns
// shifted by 37 from band to band).  Balanced lists made every heavy round ~35 % shorter -- and the chain 37 -> 49 rounds long: the
// workgroups then run at the same pace, and a vote finds fewer of this kernel's fills in its rows (the slow workgroups of the
// unbalanced layout see their neighbours' work finished).  Refine stage 3.48 -> 3.72 ms (profiles/r6_k8_experiments.txt).
#ifndef IRV_SKEW
#define IRV_SKEW 0
#endif
ADC_HD int irv_xcd_bands(int H, int xcd) { return (irv_bands(H) - xcd + 7) / 8; } // bands b with b % 8 == xcd
// first column of workgroup m (inside its XCD, `per` workgroups) in band bi, and how many columns it has there
ADC_HD int irv_band_first(int tiles_x, int per, int m, int bi)
{
    const long v = (long)bi * (tiles_x + IRV_SKEW);
    return (int)((((long)m - v) % per + per) % per);
}
ADC_HD int irv_band_count(int tiles_x, int per, int r) { return r < tiles_x ? (tiles_x - 1 - r) / per + 1 : 0; }
ADC_HD int irv_wg_tiles(int W, int H, int G, int g, int xcd)
{
    if (!xcd || G % 8 != 0) { const int n = irv_tiles_x(W) * irv_bands(H); return g < n ? (n - g + G - 1) / G : 0; }
#if IRV_SKEW == 0 // closed form (BEGIN2 asks for every tile of every row: the band loop below cost ~0.2 ms per Match, measured)
    { const int per = G / 8, m = g / 8, n = irv_xcd_bands(H, g % 8) * irv_tiles_x(W); return m < n ? (n - m + per - 1) / per : 0; }
#endif
    const int per = G / 8, m = g / 8, tiles_x = irv_tiles_x(W), nb = irv_xcd_bands(H, g % 8);
    int n = 0;
    for (int bi = 0; bi < nb; bi++) n += irv_band_count(tiles_x, per, irv_band_first(tiles_x, per, m, bi));
    return n;
}
// the k-th tile of workgroup g: *band, *tx  (k < irv_wg_tiles; bands in ascending order, columns ascending inside a band)
ADC_HD void irv_wg_tile(int W, int H, int G, int g, int k, int xcd, int* band, int* tx)
{
    const int tiles_x = irv_tiles_x(W);
    if (!xcd || G % 8 != 0) { const int t = g + k * G; *band = t / tiles_x; *tx = t % tiles_x; return; }
#if IRV_SKEW == 0 // tile index inside the XCD: u = bi * tiles_x + tx; workgroup = (b % 8) + 8 * (u % (G / 8))
    { const int u = g / 8 + k * (G / 8); *band = (u / tiles_x) * 8 + g % 8; *tx = u % tiles_x; (void)H; return; }
#endif
    const int per = G / 8, m = g / 8, nb = irv_xcd_bands(H, g % 8);
    for (int bi = 0; bi < nb; bi++) {
        const int r = irv_band_first(tiles_x, per, m, bi), n = irv_band_count(tiles_x, per, r);
        if (k < n) { *band = bi * 8 + g % 8; *tx = r + k * per; return; }
        k -= n;
    }
    *band = 0; *tx = 0; // (not reached for k < irv_wg_tiles)
}
ADC_HD long irv_seg_cap(int W, int H, int G, int WPB, int xcd)
{
    const long most = (!xcd || G % 8 != 0) ? ((long)irv_tiles_x(W) * irv_bands(H) + G - 1) / G
                                 : (long)((irv_bands(H) + 7) / 8) * ((irv_tiles_x(W) + G / 8 - 1) / (G / 8)); // tiles of the busiest workgroup (upper bound)
    const long per = most * IRV_BAND * IRV_TCOLS, batch = 64L * WPB;
    return ((per + batch - 1) / batch) * batch;
}
// Entry = {pixel, arms of the pixel (left | right << 8 | top << 16 | bottom << 24), boxes, row}; boxes = max left << 16 |
// max right << 24 over ALL region rows: the rectangle a vote starts to read before the row arms have arrived, and -- every
// region pixel being an input of some iteration's vote -- the box whose change tiles decide whether the entry is evaluated again
// (bits 0..15: the same maxima over the rows y - top .. y only, unused since round 5).

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
    uint32_t okm;        // pixels that count in SOME iteration's vote (bit q = pixel px0 + q)
    uint32_t k0, k1, k2, k3; // their keys t << 11 | bin, packed like the state halfwords (t = iteration from which the pixel counts)
    uint32_t first;      // key of the lowest counted pixel
    uint32_t same;       // the counted pixels that share that key
};
// The pixels of `rem` (a non-empty subset of a block's counted pixels) that share the key of the lowest one; *key = that key.
ADC_HD uint32_t irv_same_key_mask(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t rem, uint32_t* key)
{
    const int q0 = __builtin_ffs((int)rem) - 1;
    const uint32_t wsel = q0 < 2 ? k0 : (q0 < 4 ? k1 : (q0 < 6 ? k2 : k3));
    const uint32_t f = (wsel >> (16 * (q0 & 1))) & IRV_KEY_MASK, f2 = f * 0x00010001u;
    // halfwords that differ from that key: (d + 0x3FFF) carries into bit 14 iff d != 0 (keys are 14 bits wide)
    const uint32_t difm = irv_gather8(((k0 ^ f2) & 0x3FFF3FFFu) + 0x3FFF3FFFu, ((k1 ^ f2) & 0x3FFF3FFFu) + 0x3FFF3FFFu,
                                      ((k2 ^ f2) & 0x3FFF3FFFu) + 0x3FFF3FFFu, ((k3 ^ f2) & 0x3FFF3FFFu) + 0x3FFF3FFFu, 14);
    *key = f;
    return rem & ~difm;
}
// bits 2j, 2j+1 of an 8-bit pixel mask -> bit 11 of the low / high halfword of word j (the key's t field += 1)
ADC_HD uint32_t irv_adj_word(uint32_t m, int j) { return (((m >> (2 * j)) & 1u) << IRV_F_SHIFT) | (((m >> (2 * j + 1)) & 1u) << (IRV_F_SHIFT + 16)); }
// Block of 8 pixels px0 .. px0 + 7 of region row yt, of which [xl, xr] belong to the region of p = (x, y), a pixel of list lp.
ADC_HD IrvBlock irv_decode_block(uint32_t vx, uint32_t vy, uint32_t vz, uint32_t vw, int px0, int xl, int xr, int yt, int y, int x, int lp)
{
    IrvBlock r;
    const uint32_t inm = (((2u << adc_imin(xr - px0, 7)) - 1u) & ~((1u << adc_imax(xl - px0, 0)) - 1u)) & 0xffu;
    // pixels that precede p in raster order, and p itself (a pixel does not vote for itself: it is invalid while it is voted for)
    const uint32_t prem = yt < y ? 0xffu : (yt == y ? ((1u << adc_imax(0, adc_imin(x - px0, 8))) - 1u) : 0u);
    const uint32_t selfm = (yt == y && x >= px0 && x < px0 + 8) ? (1u << (x - px0)) : 0u;
    const uint32_t b0 = vx & 0x07FF07FFu, b1 = vy & 0x07FF07FFu, b2 = vz & 0x07FF07FFu, b3 = vw & 0x07FF07FFu; // bins
    const uint32_t l1m = irv_gather8(vx, vy, vz, vw, 14); // on the mismatch list
    const uint32_t l2m = irv_gather8(vx, vy, vz, vw, 15); // on the occlusion list
    // bin == 0x7FF (invalid / never counted): 0x7FF + 1 carries into bit 11 of the halfword
    const uint32_t invm = irv_gather8(b0 + 0x00010001u, b1 + 0x00010001u, b2 + 0x00010001u, b3 + 0x00010001u, 11);
    // pixels that count one iteration later than they were filled (irv_plan.h, top)
    const uint32_t adjm = lp == 1 ? ((l1m & ~prem) | l2m) : (l2m & ~prem);
    r.k0 = (vx & 0x3FFF3FFFu) + irv_adj_word(adjm, 0);
    r.k1 = (vy & 0x3FFF3FFFu) + irv_adj_word(adjm, 1);
    r.k2 = (vz & 0x3FFF3FFFu) + irv_adj_word(adjm, 2);
    r.k3 = (vw & 0x3FFF3FFFu) + irv_adj_word(adjm, 3);
    // t >= IRV_LEVELS (filled in the last iteration and only visible in the next one): never counted.  t + 3 >= 8 carries into bit 14
    const uint32_t latem = irv_gather8((r.k0 & 0x38003800u) + 0x18001800u, (r.k1 & 0x38003800u) + 0x18001800u,
                                       (r.k2 & 0x38003800u) + 0x18001800u, (r.k3 & 0x38003800u) + 0x18001800u, 14);
    r.okm = inm & ~invm & ~selfm & ~latem;
    r.first = 0u;
    r.same = 0u;
    if (r.okm != 0u) r.same = irv_same_key_mask(r.k0, r.k1, r.k2, r.k3, r.okm, &r.first);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// SLACK BUDGETS (round 6).  In the heavy rounds of the chain four out of five re-evaluations end in the state the entry already
// had: something in its region changed, but not enough to flip a vote (tools/irv_joint_rounds.py: 0.61 M evaluations for 0.1 M state
// changes).  An evaluation therefore also computes how many region pixels would have to change before its outcome CAN change.  One
// pixel's state change moves the cumulative histogram of a level by at most -1 in one bin and +1 in another, so after at most k
// pixel changes a level with count c, top bin m and runner-up m2 has c' in [c - k, c + k], every bin within +-k, and
//     a FAILING level keeps failing  if  c + k <= ts                         (count > ts stays false)
//                                    or  c - k >= 1 and fl((m + k) / (c - k)) <= th   (the ratio test stays false: float division is monotone)
//     the PASSING level keeps its bin if  c - k > ts,  m - k >= 1,  fl((m - k) / (c + k)) > th  and  m - k > m2 + k
// (multistep_refiner.cpp:199-214 is `max > 0 && count > ts && max * 1.0f / count > th`, first maximum = lowest bin on ties).  The
// entry's budget K = the largest such k over the levels up to the deciding one; K = 0 ("any change -> evaluate again") is always
// valid, and every value below the true bound is (the tests are monotone in k): the closed forms below round DOWN with a margin
// (tests/emul/emul_irv.cpp: emul_irv_slack_check proves them against the reference's own float expression).  The change tiles say WHERE something changed in the previous kernel; a
// bit per pixel says which pixels did, and an entry whose tiles were hit counts the changed pixels inside its region (row by row:
// the bitmap words of the row and the row's arms, every load independent of the others), subtracts them from its budget and is only
// re-evaluated when the budget is used up.  Exactness: every state change of a region pixel since the entry's last evaluation is
// counted at least once (changes of the kernel the entry was evaluated in are counted in the next one, whether its gather saw
// them or not), so an entry that is skipped would vote what it voted.
// Division-free form (a vote computes it for every level it decides): thresholds moved by 2^-20 relative -- 16 times the rounding of
// the reference's float division -- so that the REAL inequalities  (m + k) <= tl (c - k)  /  (m - k) >= th (c + k)  imply the
// float tests; solved for k with a precomputed 1 / (1 + t); the float evaluation of the bound is off by < 0.01 for counts up to
// 69 x 69, one is subtracted.  K = 0 is always valid, and so is every value below the true bound (the tests are monotone in k).
struct IrvSlackK { float tl, rl, th, rh; int ok; };
ADC_HD IrvSlackK irv_slack_consts(float irv_th)
{
    IrvSlackK s;
    s.tl = irv_th * (1.0f - 9.5367431640625e-07f); // 1 - 2^-20
    s.th = irv_th * (1.0f + 9.5367431640625e-07f);
    s.ok = irv_th >= 0.0f && irv_th <= 4.0f;       // (anything else -- incl. NaN: no ratio-based slack at all)
    // (a reciprocal accurate to an ulp is plenty: the bounds subtract one for errors of < 0.01)
#if defined(__HIP_DEVICE_COMPILE__)
    s.rl = s.ok ? __builtin_amdgcn_rcpf(1.0f + s.tl) : 0.0f;
    s.rh = s.ok ? __builtin_amdgcn_rcpf(1.0f + s.th) : 0.0f;
#else
    s.rl = s.ok ? 1.0f / (1.0f + s.tl) : 0.0f;
    s.rh = s.ok ? 1.0f / (1.0f + s.th) : 0.0f;
#endif
    return s;
}
// (Branch-free on purpose: as nested conditionals inside the vote loop of k_irv_u this function cost the kernel the scalar registers
// of its execution masks -- 37 spill reloads per vote, measured.)
ADC_HD int irv_level_slack(bool pass, int c, int m, int m2, int ts, const IrvSlackK& q)
{
    const float cf = (float)c, mf = (float)m;
    // a failing level: c + k <= ts, or the ratio test stays false: (m + k) <= tl (c - k)
    const float xf = q.ok ? (q.tl * cf - mf) * q.rl : 0.0f;
    const int kf = (c >= 1 && xf >= 1.0f) ? adc_imin((int)xf - 1, c - 1) : -1;
    const int kfail = adc_imax(ts - c, kf);
    // the passing level: c - k > ts, m - k >= 1, m - k > m2 + k, and the ratio test stays true: (m - k) >= th (c + k)
    const float xp = q.ok ? (mf - q.th * cf) * q.rh : 0.0f;
    const int kc = xp >= 1.0f ? (int)xp - 1 : 0;
    const int kpass = adc_imin(adc_imin(c - ts - 1, (m - m2 - 1) >> 1), adc_imin(kc, m - 1));
    return adc_imax(0, adc_imin(pass ? kpass : kfail, 0xFFFF));
}
// per-pixel change bitmap: IRV_PX_PLANES planes of H rows x pitch dwords (bit x & 31 of dword x >> 5; the padding dwords stay 0).
// Kernel k sets bits in plane k % 3, reads plane (k + 2) % 3 (its predecessor's) and clears plane (k + 1) % 3 for its successor.
#define IRV_PX_PLANES 3
ADC_HD int irv_px_pitch(int W) { return ((W + 31) >> 5) + 4; } // (+4: a 16-byte load that starts at a row's last word stays inside the row)
