// CPU model of the device-driven region-voting chain (adcensus_amd/csrc/k_voting.hip): the SAME state machine
// (irv_plan.h: every kernel derives its action from the state and the accumulators its predecessor left) and the SAME
// work-list layout (per-workgroup segments of tiles, sorted by row inside the band: irv_seg_cap / irv_wg_tiles) drive plain-loop versions of the kernel's phases on the same 16-bit state map
// (bin | iteration of the fill | list: ONE fixed-point iteration over all ten passes of the reference, irv_plan.h) and the two
// change-tile planes, with the waves of a round visited in a shuffled order against the in-place map (arbitrary scheduling).
// The vote itself is written per pixel here (the kernel decodes packed halfwords: emul_irv_swar_check).  Test infrastructure.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../adcensus_amd/csrc/irv_plan.h"

// use_slack (0 = off, else the number of hit entries per wave from which the wave filters; the product's default is 1): the slack budgets of round 6 (irv_plan.h, bottom): per-pixel change planes, a budget per entry (low half of the entry's
// box word), changed pixels counted over the region -- as in the kernel
extern "C" long emul_irv_chain2(float* disp, const uint8_t* label, const uint8_t* arms, const uint16_t* sup_h, int W, int H, int dmin,
                                int D, int irv_ts, float irv_th, int min_region, unsigned seed, int groups, int wpb, int use_slack, long* out_stats)
{
    const int P = W * H, SP = (W + 7) & ~7, T = IRV_TILE;
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    const int G = groups > 0 ? groups : 2, WPB = wpb > 0 ? wpb : 4; // the (emulated) grid: decides the list layout
    const int XCD = 1; // (the kernel's default; a grid that is not a multiple of 8 falls back to round-robin inside the helpers)
    const long cap = irv_seg_cap(W, H, G, WPB, XCD), BT = 64L * WPB;
    struct Ent { int p, arms, box, y; };
    std::vector<uint16_t> st((size_t)SP * H + 64, 0xFFFF);
    std::vector<int32_t> ctrl(IRV_CTRL_INTS, 0), hist((size_t)IRV_LEVELS * D);
    std::vector<int32_t> chg(2 * (size_t)tiles_x * tiles_y, 0); // two planes (round parity)
    std::vector<uint8_t> px((size_t)IRV_PX_PLANES * P, 0);       // per-pixel change planes (a byte per pixel here, a bit in the kernel)
    std::vector<Ent> list((size_t)cap * G, Ent{IRV_LIST_END, 0, 0, 0});
    std::vector<int> wg_n(G, 0);
    std::vector<uint8_t> listed_bit(P, 0);
    std::vector<float> work(P);
    std::vector<uint8_t> bb((size_t)P * 2);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) { // widest H arms over all region rows (k_irv_bbox)
            const uint8_t* a = arms + ((size_t)y * W + x) * 4;
            int ml = 0, mr = 0;
            for (int t = -(int)a[2]; t <= (int)a[3]; t++) {
                const uint8_t* q = arms + ((size_t)(y + t) * W + x) * 4;
                ml = std::max(ml, (int)q[0]);
                mr = std::max(mr, (int)q[1]);
            }
            bb[((size_t)y * W + x) * 2] = (uint8_t)ml;
            bb[((size_t)y * W + x) * 2 + 1] = (uint8_t)mr;
        }
    srand(seed);
    long kernels = 0, total_evals = 0;
    const int32_t* fin = nullptr;
    for (int k = 0;; k++) {
        if (ctrl[16 * (k & 1)] == IRV_DONE) { fin = &ctrl[16 * (k & 1)]; break; } // the state the previous kernel published
        const IrvPlan pl = irv_plan(ctrl.data(), k);
        int32_t* acc = &ctrl[IRV_ACC + (k & 63)];
        kernels++;
        if (pl.act == IRV_BEGIN || pl.act == IRV_FINAL_WB) {
            if (pl.act == IRV_BEGIN) { std::fill(chg.begin(), chg.end(), 0); std::fill(px.begin(), px.end(), 0); }
            for (int p = 0; p < P; p++) { // one pass over the image (any order)
                const int y = p / W, x = p - y * W;
                const size_t i16 = (size_t)y * SP + x;
                if (pl.act == IRV_FINAL_WB) {
                    float dv = work[p];
                    const uint32_t s = st[i16];
                    if ((s >> IRV_LIST_SHIFT) != 0u && (s & IRV_BIN_MASK) != IRV_BIN_MASK) dv = (float)((int)(s & IRV_BIN_MASK) + dmin);
                    disp[p] = dv;
                } else {
                    const float dv = disp[p];
                    work[p] = dv;
                    const uint32_t lab = label[p];
                    const bool e = (lab == ADC_LABEL_MISMATCH || lab == ADC_LABEL_OCCLUSION) && dv == ADC_INVALID_FLOAT;
                    listed_bit[p] = e && (int)sup_h[p] > min_region;
                    if (listed_bit[p]) *acc = 1; // (an empty work list ends the chain: irv_plan_from)
                    uint32_t bin = IRV_BIN_MASK;
                    if (dv != ADC_INVALID_FLOAT) { const long b = lroundf(dv) - dmin; if (b >= 0 && b < D) bin = (uint32_t)b; }
                    st[i16] = (uint16_t)(bin | (e ? lab << IRV_LIST_SHIFT : 0u));
                }
            }
        } else if (pl.act == IRV_BEGIN2) {
            // every workgroup walks its own tiles row by row (irv_plan.h); workgroups in a shuffled order
            std::vector<int> wgs(G);
            for (int g = 0; g < G; g++) wgs[g] = g;
            for (int g = G; g > 1; g--) std::swap(wgs[g - 1], wgs[rand() % g]);
            for (int g : wgs) {
                int ng = 0;
                const int my_tiles = irv_wg_tiles(W, H, G, g, XCD);
                for (int j = 0; j < IRV_BAND; j++)
                    for (int kk = 0; kk < my_tiles; kk++)
                        for (int xi = 0; xi < IRV_TCOLS; xi++) {
                            int band, tx;
                            irv_wg_tile(W, H, G, g, kk, XCD, &band, &tx);
                            const int y = band * IRV_BAND + j, x = tx * IRV_TCOLS + xi;
                            if (y >= H || x >= W) continue;
                            const int p = y * W + x;
                            if (!listed_bit[p]) continue;
                            const uint8_t* a = arms + (size_t)p * 4;
                            const int arms32 = (int)((uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)a[2] << 16) | ((uint32_t)a[3] << 24));
                            list[(size_t)g * cap + ng++] = Ent{p, arms32, (int)(((uint32_t)bb[(size_t)p * 2] << 16) | ((uint32_t)bb[(size_t)p * 2 + 1] << 24)), y};
                        }
                if (ng > cap) return -3; // (cannot happen: a segment holds every pixel of the workgroup's tiles)
                for (long t = ng; t < ((ng + BT - 1) / BT) * BT && t < cap; t++) list[(size_t)g * cap + t].p = IRV_LIST_END;
                wg_n[g] = ng;
                *acc += ng;
            }
        } else if (pl.act == IRV_ROUND) {
            // every wave: phase 1 (one entry per lane: dirty?), phase 2 (evaluate the dirty ones); waves in a shuffled order
            // against the in-place map (arbitrary scheduling)
            const int round = pl.s.round;
            const int32_t want = ((k + 254) % 255) + 1, stamp = (k % 255) + 1; // stamps and planes go by KERNEL index
            const int32_t* chg_rd = chg.data() + (size_t)((k + 1) & 1) * tiles_x * tiles_y;
            int32_t* chg_wr = chg.data() + (size_t)(k & 1) * tiles_x * tiles_y;
            const uint8_t* px_rd = px.data() + (size_t)((k + 2) % IRV_PX_PLANES) * P;
            uint8_t* px_wr = px.data() + (size_t)(k % IRV_PX_PLANES) * P;
            if (use_slack) std::fill(px.begin() + (size_t)((k + 1) % IRV_PX_PLANES) * P, px.begin() + (size_t)((k + 1) % IRV_PX_PLANES + 1) * P, 0);
            // workgroups in a shuffled order; a workgroup takes its segment batch by batch, pools the dirty entries of the batch
            // in list order and its WPB waves take consecutive pool items: groups of WPB items, each group in a shuffled order
            std::vector<int> wgs(G);
            for (int g = 0; g < G; g++) wgs[g] = g;
            for (int g = G; g > 1; g--) std::swap(wgs[g - 1], wgs[rand() % g]);
            for (int g : wgs)
                for (long b0 = 0; b0 < wg_n[g]; b0 += BT) {
                    std::vector<int> todo;
                    std::vector<uint16_t> seen;
                    // phase 1 is done wave by wave (64 consecutive entries of the batch): a wave filters its hit entries against their slack
                    // budgets only when at least `use_slack` of them were hit (k_voting.hip: the tail rounds skip the walk)
                    std::vector<uint8_t> hit;
                    for (long i = b0; i < std::min((long)wg_n[g], b0 + BT); i++) {
                        const Ent& e = list[(size_t)g * cap + i];
                        const int p = e.p, y = e.y, x = p - y * W;
                        bool dirty = round == 0;
                        if (!dirty) {
                            const int top = (e.arms >> 16) & 255, bot = (e.arms >> 24) & 255, ml = (e.box >> 16) & 255, mr = (e.box >> 24) & 255;
                            const int tx0 = std::max(0, x - ml) / T, tx1 = std::min(W - 1, x + mr) / T;
                            const int ty0 = std::max(0, y - top) / T, ty1 = std::min(H - 1, y + bot) / T;
                            for (int ty = ty0; ty <= ty1; ty++)
                                for (int tx = tx0; tx <= tx1; tx++) dirty |= chg_rd[ty * tiles_x + tx] == want;
                        }
                        hit.push_back(dirty);
                    }
                    for (long i = b0; i < std::min((long)wg_n[g], b0 + BT); i++) {
                        Ent& e = list[(size_t)g * cap + i];
                        const int p = e.p, y = e.y, x = p - y * W;
                        bool dirty = round == 0;
                        if (!dirty) {
                            const int top = (e.arms >> 16) & 255, bot = (e.arms >> 24) & 255, ml = (e.box >> 16) & 255, mr = (e.box >> 24) & 255;
                            const int tx0 = std::max(0, x - ml) / T, tx1 = std::min(W - 1, x + mr) / T;
                            const int ty0 = std::max(0, y - top) / T, ty1 = std::min(H - 1, y + bot) / T;
                            for (int ty = ty0; ty <= ty1; ty++)
                                for (int tx = tx0; tx <= tx1; tx++) dirty |= chg_rd[ty * tiles_x + tx] == want; // byte stamps (k_voting.hip)
                        }
                        int nh = 0;
                        {
                            const long w0 = b0 + ((i - b0) / 64) * 64;
                            for (long j = w0; j < std::min(w0 + 64, std::min((long)wg_n[g], b0 + BT)); j++) nh += hit[j - b0];
                        }
                        if (dirty && use_slack > 0 && round != 0 && nh >= use_slack) { // phase 1 (k_voting.hip): changed pixels of the region's bounding RECTANGLE in the
                            // previous kernel's plane (a superset of the region: an upper bound; no arm lookups), without the entry's own pixel,
                            // against the entry's budget
                            const uint8_t* pa = arms + (size_t)p * 4;
                            const int ml = (e.box >> 16) & 255, mr = (e.box >> 24) & 255;
                            int used = 0;
                            for (int yy = y - (int)pa[2]; yy <= y + (int)pa[3]; yy++)
                                for (int xx = x - ml; xx <= x + mr; xx++)
                                    if (!(yy == y && xx == x)) used += px_rd[(size_t)yy * W + xx];
                            const int rem = (e.box & 0xFFFF) - used;
                            dirty = rem < 0;
                            if (rem >= 0 && used > 0) e.box = (int)(((uint32_t)e.box & 0xFFFF0000u) | (uint32_t)rem);
                        }
                        if (dirty) { todo.push_back((int)i); seen.push_back(st[(size_t)y * SP + x]); } // (phase 1 reads the entry's own state)
                    }
                    for (size_t c0 = 0; c0 < todo.size(); c0 += WPB) {
                        std::vector<size_t> grp;
                        for (size_t c = c0; c < std::min(todo.size(), c0 + WPB); c++) grp.push_back(c);
                        for (size_t c = grp.size(); c > 1; c--) std::swap(grp[c - 1], grp[rand() % c]);
                        for (size_t t : grp) {
                            const Ent& e = list[(size_t)g * cap + todo[t]];
                            const int p = e.p, y = e.y, x = p - y * W;
                            const uint32_t cur = seen[t];
                            const int lp = (int)(cur >> IRV_LIST_SHIFT);
                            total_evals++;
                            std::fill(hist.begin(), hist.end(), 0);
                            const uint8_t* arm = arms + (size_t)p * 4;
                            for (int dy = -(int)arm[2]; dy <= (int)arm[3]; dy++) {
                                const int yt = y + dy;
                                const uint8_t* a2 = arms + ((size_t)yt * W + x) * 4;
                                for (int px = x - (int)a2[0]; px <= x + (int)a2[1]; px++) {
                                    if (yt == y && px == x) continue; // a pixel does not vote for itself
                                    const uint32_t s = st[(size_t)yt * SP + px];
                                    const uint32_t bin = s & IRV_BIN_MASK;
                                    if (bin == IRV_BIN_MASK) continue;
                                    const int lq = (int)(s >> IRV_LIST_SHIFT), fq = (int)((s >> IRV_F_SHIFT) & 7u);
                                    const bool pre = yt < y || (yt == y && px < x);
                                    int tq; // the iteration from which q counts (irv_plan.h, top)
                                    if (lq == 0) tq = fq;
                                    else if (lq == lp) tq = pre ? fq : fq + 1;
                                    else tq = lp == 1 ? fq + 1 : fq;
                                    if (tq < IRV_LEVELS) hist[(size_t)tq * D + bin]++;
                                }
                            }
                            uint32_t ns = IRV_BIN_MASK | ((uint32_t)lp << IRV_LIST_SHIFT);
                            int K = 0xFFFF;
                            for (int it = 0; it < IRV_LEVELS; it++) {
                                if (it) for (int b = 0; b < D; b++) hist[(size_t)it * D + b] += hist[(size_t)(it - 1) * D + b];
                                // the kernel's key: count << 11 | (2047 - bin), maximum = highest count, lowest bin on ties
                                int key = 0, cnt = 0;
                                for (int b = 0; b < D; b++) { const int hv = hist[(size_t)it * D + b]; cnt += hv; if (hv > 0) key = std::max(key, (hv << 11) | (0x7FF - b)); }
                                const int bh = key >> 11, bbin = 0x7FF - (key & 0x7FF);
                                // (every level here, the kernel only those that add pixels: a level without new pixels repeats its
                                // predecessor's histogram and slack; leading empty levels have c = m = 0 in both)
                                K = std::min(K, irv_level_slack(adc_vote_decide(bbin, bh, cnt, dmin, irv_ts, irv_th) != ADC_INVALID_FLOAT, cnt, bh, cnt - bh, irv_ts, irv_slack_consts(irv_th)));
                                if (adc_vote_decide(bbin, bh, cnt, dmin, irv_ts, irv_th) != ADC_INVALID_FLOAT) {
                                    ns = (uint32_t)bbin | ((uint32_t)it << IRV_F_SHIFT) | ((uint32_t)lp << IRV_LIST_SHIFT);
                                    break;
                                }
                            }
                            if (use_slack) list[(size_t)g * cap + todo[t]].box = (int)(((uint32_t)e.box & 0xFFFF0000u) | (uint32_t)K);
                            if (ns != cur) {
                                st[(size_t)y * SP + x] = (uint16_t)ns;
                                chg_wr[(y / T) * tiles_x + x / T] = stamp;
                                if (use_slack) px_wr[(size_t)y * W + x] = 1;
                                *acc = 1;
                            }
                        }
                    }
                }
        }
        IrvState ps = pl.s;
        if (pl.act == IRV_FINAL_WB) ps.evals = (int)total_evals; // (summed from the per-wave counters by the kernel)
        irv_publish(ctrl.data(), k, ps);
        if (kernels > 4000000) return -1;
    }
    if (out_stats) { out_stats[0] = fin[5]; out_stats[1] = fin[6]; out_stats[2] = kernels; }
    return fin[0] == IRV_DONE ? fin[5] : -2;
}

extern "C" long emul_irv_chain(float* disp, const uint8_t* label, const uint8_t* arms, const uint16_t* sup_h, int W, int H, int dmin,
                               int D, int irv_ts, float irv_th, int min_region, unsigned seed, int groups, int wpb, long* out_stats)
{
    return emul_irv_chain2(disp, label, arms, sup_h, W, H, dmin, D, irv_ts, irv_th, min_region, seed, groups, wpb, 1, out_stats);
}

// The closed forms of irv_level_slack (irv_plan.h) against the definition: for random (pass, c, m, m2, ts, th) the returned K must
// satisfy the level's own tests at K (the budget is VALID), and K be at most 3 below the largest valid
// value (the budget is not wastefully small).  Returns invalid * 1000000 + loose.
static bool slack_holds(bool pass, int c, int m, int m2, int ts, float th, int k)
{
    if (!pass) return (c + k <= ts) || (c - k >= 1 && !((float)(m + k) * 1.0f / (float)(c - k) > th));
    return (c - k > ts) && (m - k >= 1) && ((float)(m - k) * 1.0f / (float)(c + k) > th) && (m - k > m2 + k);
}
extern "C" long emul_irv_slack_check(unsigned seed, long trials)
{
    srand(seed);
    long invalid = 0, loose = 0;
    for (long t = 0; t < trials; t++) {
        const int ts = rand() % 60 - 5;
        const float th = (float)(rand() % 1000) / 1000.0f;
        const int c = rand() % (t % 4 == 0 ? 5000 : 400), m = c ? 1 + rand() % c : 0, m2 = std::min(m, c - m > 0 ? rand() % (c - m + 1) : 0);
        const bool pass = m > 0 && c > ts && (float)m * 1.0f / (float)c > th;
        const int K = irv_level_slack(pass, c, m, m2, ts, irv_slack_consts(th));
        if (K > 0 && !slack_holds(pass, c, m, m2, ts, th, K)) invalid++;
        for (int k = 1; k < K; k++) if (!slack_holds(pass, c, m, m2, ts, th, k)) { invalid++; break; } // (monotone: everything below K holds too)
        int best = 0;
        while (best < 5000 && slack_holds(pass, c, m, m2, ts, th, best + 1)) best++;
        if (K > best) invalid++;
        if (best - K > 3 && K < 0xFFFF) loose++;
    }
    return invalid * 1000000 + loose;
}

// ---------------------------------------------------------------------------------------------------------------------
// The packed-halfword helpers of irv_plan.h against per-pixel loops on random inputs (values drawn so that equal keys,
// invalid bins, both lists, late fills and range edges all occur often).  Returns the number of disagreements.
extern "C" long emul_irv_swar_check(unsigned seed, long trials)
{
    srand(seed);
    long bad = 0;
    for (long t = 0; t < trials; t++) {
        // ---- region row block
        uint16_t s16[8];
        const int nb = 1 + rand() % 4;                   // few distinct bins -> "single key" happens
        const int pool[4] = {rand() % 2047, rand() % 2047, IRV_BIN_MASK, rand() % 2047};
        for (int q = 0; q < 8; q++) {
            uint32_t v = (uint32_t)pool[rand() % nb];
            const int lq = rand() % 3;                    // not listed / mismatch / occlusion
            v |= (uint32_t)lq << IRV_LIST_SHIFT;
            if (lq != 0 && (v & IRV_BIN_MASK) != IRV_BIN_MASK) v |= (uint32_t)(rand() % IRV_LEVELS) << IRV_F_SHIFT; // filled in iteration 0..4
            s16[q] = (uint16_t)v;
        }
        uint32_t w[4];
        for (int j = 0; j < 4; j++) w[j] = (uint32_t)s16[2 * j] | ((uint32_t)s16[2 * j + 1] << 16);
        const int px0 = 8 * (rand() % 50), y = 5 + rand() % 5, yt = y - 2 + rand() % 5, x = px0 - 6 + rand() % 20, lp = 1 + rand() % 2;
        int xl = px0 - 4 + rand() % 12, xr = xl + rand() % 14;
        if (xr < px0) xr = px0;                           // (the kernel only decodes blocks that intersect [xl, xr])
        if (xl > px0 + 7) xl = px0 + 7;
        if (xr < xl) xr = xl;
        const IrvBlock got = irv_decode_block(w[0], w[1], w[2], w[3], px0, xl, xr, yt, y, x, lp);
        uint32_t okm = 0;
        uint32_t keys[8];
        for (int q = 0; q < 8; q++) {
            const int px = px0 + q;
            const uint32_t sv = s16[q], bin = sv & IRV_BIN_MASK;
            const int lq = (int)(sv >> IRV_LIST_SHIFT), fq = (int)((sv >> IRV_F_SHIFT) & 7u);
            const bool in = px >= xl && px <= xr, pre = yt < y || (yt == y && px < x), self = yt == y && px == x;
            int tq;
            if (lq == 0) tq = fq;
            else if (lq == lp) tq = pre ? fq : fq + 1;
            else tq = lp == 1 ? fq + 1 : fq;
            keys[q] = ((uint32_t)tq << IRV_F_SHIFT) | bin;
            if (in && !self && bin != IRV_BIN_MASK && tq < IRV_LEVELS) okm |= 1u << q;
        }
        if (got.okm != okm) bad++;
        else if (okm) { // the pixels of the first key, and -- on what is left -- the pixels that share the key of the lowest one
            const uint32_t gk[4] = {got.k0, got.k1, got.k2, got.k3};
            bool keys_ok = true;
            for (int q = 0; q < 8; q++)
                if (((okm >> q) & 1u) && ((gk[q >> 1] >> (16 * (q & 1))) & IRV_KEY_MASK) != keys[q]) keys_ok = false;
            if (!keys_ok) { bad++; continue; }
            uint32_t rem = okm;
            bool first = true;
            while (rem) {
                uint32_t key = 0xFFFFu, want_mask = 0;
                const uint32_t got_mask = first ? got.same : irv_same_key_mask(got.k0, got.k1, got.k2, got.k3, rem, &key);
                if (first) key = got.first;
                const uint32_t kq = keys[__builtin_ctz(rem)];
                for (int q = 0; q < 8; q++)
                    if (((rem >> q) & 1u) && keys[q] == kq) want_mask |= 1u << q;
                if (key != kq || got_mask != want_mask || !got_mask) { bad++; break; }
                rem &= ~got_mask;
                first = false;
            }
        }
        // ---- change-tile row
        uint8_t tb[16];
        const uint32_t stamp = 1 + rand() % 255;
        for (int i = 0; i < 16; i++) tb[i] = rand() % 5 == 0 ? (uint8_t)stamp : (uint8_t)(rand() % 256 == (int)stamp ? 0 : rand() % 256);
        const int cb = 4 * (rand() % 30), txb = cb + rand() % 4, last = txb + rand() % 12;
        uint32_t nk[4], tw[4], hit = 0;
        irv_tile_row_masks(txb, last, nk);
        for (int j = 0; j < 4; j++) {
            tw[j] = (uint32_t)tb[4 * j] | ((uint32_t)tb[4 * j + 1] << 8) | ((uint32_t)tb[4 * j + 2] << 16) | ((uint32_t)tb[4 * j + 3] << 24);
            hit |= irv_tile_hit(tw[j], nk[j], stamp * 0x01010101u);
        }
        bool want = false;
        for (int tx = txb; tx <= last; tx++) want = want || tb[tx - cb] == stamp;
        if ((hit != 0) != want) bad++;
    }
    return bad;
}

// The tiles of all workgroups partition the image, and a workgroup's tiles fit its segment (irv_plan.h: irv_wg_tiles / irv_wg_tile /
// irv_seg_cap).  Returns the number of violations.
extern "C" long emul_irv_tile_partition(int W, int H, int G, int WPB, int XCD)
{
    std::vector<int> owner((size_t)W * H, -1);
    long bad = 0;
    const long cap = irv_seg_cap(W, H, G, WPB, XCD);
    for (int g = 0; g < G; g++) {
        const int nt = irv_wg_tiles(W, H, G, g, XCD);
        long pixels = 0;
        for (int k = 0; k < nt; k++) {
            int band, tx;
            irv_wg_tile(W, H, G, g, k, XCD, &band, &tx);
            if (band < 0 || band >= irv_bands(H) || tx < 0 || tx >= irv_tiles_x(W)) { bad++; continue; }
            if (XCD && G % 8 == 0 && band % 8 != g % 8) bad++; // a band belongs to ONE XCD
            for (int j = 0; j < IRV_BAND; j++)
                for (int xi = 0; xi < IRV_TCOLS; xi++) {
                    const int y = band * IRV_BAND + j, x = tx * IRV_TCOLS + xi;
                    if (y >= H || x >= W) continue;
                    if (owner[(size_t)y * W + x] != -1) bad++;
                    owner[(size_t)y * W + x] = g;
                    pixels++;
                }
        }
        if (pixels > cap) bad++;
    }
    for (int v : owner) bad += v < 0;
    return bad;
}
