// CPU model of the device-driven region-voting chain (adcensus_amd/csrc/k_voting.hip): the SAME state machine
// (irv_plan.h: every kernel derives its action from the state and the accumulators its predecessor left) and the SAME
// work-list layout (irv_list_slot) drive plain-loop versions of the kernel's phases on the same 16-bit state map
// (bin | final | eligible) and the two change-tile planes, with the waves of a round visited in a shuffled order against
// the in-place map (arbitrary scheduling).  Test infrastructure.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../adcensus_amd/csrc/irv_plan.h"

extern "C" long emul_irv_chain(float* disp, const uint8_t* label, const uint8_t* arms, const uint16_t* sup_h, int W, int H, int dmin,
                               int D, int irv_ts, float irv_th, int min_region, unsigned seed, int groups, int wpb, long* out_stats)
{
    const int P = W * H, SP = (W + 7) & ~7, T = IRV_TILE;
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    const int G = groups > 0 ? groups : 2, WPB = wpb > 0 ? wpb : 4, NW = G * WPB; // the (emulated) grid: decides the list layout (irv_list_slot)
    const long B = 64L * NW, cap = ((P + B - 1) / B) * B;
    struct Ent { int p, arms, mlmr, y; };
    std::vector<uint16_t> st((size_t)SP * H + 64, 0xFFFF);
    std::vector<int32_t> ctrl(IRV_CTRL_INTS, 0), hist(D);
    std::vector<int32_t> chg(2 * (size_t)tiles_x * tiles_y, 0); // two planes (round parity)
    std::vector<Ent> list(cap, Ent{0, 0, 0, 0});
    std::vector<uint8_t> bb((size_t)P * 3);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t* a = arms + ((size_t)y * W + x) * 4;
            int ml = 0, mr = 0;
            for (int t = -(int)a[2]; t <= 0; t++) {
                const uint8_t* q = arms + ((size_t)(y + t) * W + x) * 4;
                ml = std::max(ml, (int)q[0]);
                mr = std::max(mr, (int)q[1]);
            }
            uint8_t* o = &bb[((size_t)y * W + x) * 3];
            o[0] = a[2]; o[1] = (uint8_t)ml; o[2] = (uint8_t)mr;
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
            const bool have_state = !(pl.act == IRV_BEGIN && pl.s.pass == 0);
            const int which = (pl.s.pass & 1) ? ADC_LABEL_OCCLUSION : ADC_LABEL_MISMATCH;
            if (pl.act == IRV_BEGIN) std::fill(chg.begin(), chg.end(), 0);
            // the blocks of the kernel claim list ranges with an atomic in arbitrary order: model it by visiting the
            // chunks of pixels (one per workgroup iteration) in a shuffled order
            const int CH = 1024 * IRV_PPT;
            std::vector<int> chunks((P + CH - 1) / CH);
            for (size_t c = 0; c < chunks.size(); c++) chunks[c] = (int)c;
            for (size_t c = chunks.size(); c > 1; c--) std::swap(chunks[c - 1], chunks[rand() % c]);
            for (int c : chunks)
                for (int p = c * CH; p < std::min(P, (c + 1) * CH); p++) {
                    const int y = p / W, x = p - y * W;
                    const size_t i16 = (size_t)y * SP + x;
                    float dv = disp[p];
                    if (have_state) {
                        const uint32_t s = st[i16];
                        if ((s & IRV_ELIG) && (s & IRV_BIN_MASK) != IRV_BIN_MASK) { dv = (float)((int)(s & IRV_BIN_MASK) + dmin); disp[p] = dv; }
                    }
                    if (pl.act == IRV_BEGIN) {
                        const bool e = label[p] == which && dv == ADC_INVALID_FLOAT;
                        const bool listed = e && (int)sup_h[p] > min_region;
                        uint32_t bin = IRV_BIN_MASK;
                        if (dv != ADC_INVALID_FLOAT) { const long b = lroundf(dv) - dmin; if (b >= 0 && b < D) bin = (uint32_t)b; }
                        st[i16] = (uint16_t)(bin | (e ? IRV_ELIG : 0u) | (listed ? 0u : IRV_FINAL));
                        if (listed) {
                            const uint8_t* o = &bb[(size_t)p * 3];
                            const uint8_t* a = arms + (size_t)p * 4;
                            const int arms32 = (int)((uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)a[2] << 16) | ((uint32_t)a[3] << 24));
                            list[irv_list_slot((*acc)++, G, WPB)] = Ent{p, arms32, (int)o[1] | ((int)o[2] << 8), y};
                        }
                    }
                }
        } else if (pl.act == IRV_ROUND) {
            // every wave: phase 1 (one entry per lane: open and dirty?), phase 2 (evaluate the dirty ones); waves in a
            // shuffled order against the in-place map (arbitrary scheduling)
            const int round = pl.s.round, n = pl.s.n;
            const int32_t want = ((k + 254) % 255) + 1, stamp = (k % 255) + 1; // stamps and planes go by KERNEL index
            const int32_t* chg_rd = chg.data() + (size_t)((k + 1) & 1) * tiles_x * tiles_y;
            int32_t* chg_wr = chg.data() + (size_t)(k & 1) * tiles_x * tiles_y;
            std::vector<int> waves(NW);
            for (int w = 0; w < NW; w++) waves[w] = w;
            for (int w = NW; w > 1; w--) std::swap(waves[w - 1], waves[rand() % w]);
            for (long b0 = 0; b0 < n; b0 += B)
                for (int gw : waves) {
                    int todo[64], nt = 0;
                    for (int lane = 0; lane < 64; lane++) {
                        const long i = irv_list_index(b0, gw / WPB, gw % WPB, lane, G);
                        if (i >= n) continue;
                        const Ent& e = list[b0 + (size_t)gw * 64 + lane];
                        const int p = e.p, y = e.y, x = p - y * W;
                        if (st[(size_t)y * SP + x] & IRV_FINAL) continue;
                        bool dirty = round == 0;
                        if (!dirty) {
                            const int top = (e.arms >> 16) & 255, ml = e.mlmr & 255, mr = (e.mlmr >> 8) & 255;
                            const int tx0 = std::max(0, x - ml) / T, tx1 = std::min(W - 1, x + mr) / T;
                            const int ty0 = std::max(0, y - top) / T, ty1 = y / T;
                            for (int ty = ty0; ty <= ty1; ty++)
                                for (int tx = tx0; tx <= tx1; tx++) dirty |= chg_rd[ty * tiles_x + tx] == want; // byte stamps (k_voting.hip)
                        }
                        if (dirty) todo[nt++] = lane;
                    }
                    total_evals += nt;
                    for (int t = 0; t < nt; t++) {
                        const Ent& e = list[b0 + (size_t)gw * 64 + todo[t]];
                        const int p = e.p, y = e.y, x = p - y * W;
                        std::fill(hist.begin(), hist.end(), 0);
                        bool deps_open = false;
                        const uint8_t* arm = arms + (size_t)p * 4;
                        for (int dy = -(int)arm[2]; dy <= (int)arm[3]; dy++) {
                            const int yt = y + dy;
                            const uint8_t* a2 = arms + ((size_t)yt * W + x) * 4;
                            for (int px = x - (int)a2[0]; px <= x + (int)a2[1]; px++) {
                                const uint32_t s = st[(size_t)yt * SP + px];
                                const bool el = (s & IRV_ELIG) != 0, pre = yt < y || (yt == y && px < x);
                                const uint32_t bin = s & IRV_BIN_MASK;
                                if (bin != IRV_BIN_MASK && (!el || pre)) hist[bin]++;
                                if (el && pre && !(s & IRV_FINAL)) deps_open = true;
                            }
                        }
                        // the kernel's key: count << 11 | (2047 - bin), maximum = highest count, lowest bin on ties
                        int key = 0, cnt = 0;
                        for (int b = 0; b < D; b++) { cnt += hist[b]; if (hist[b] > 0) key = std::max(key, (hist[b] << 11) | (0x7FF - b)); }
                        const int bh = key >> 11, bbin = 0x7FF - (key & 0x7FF);
                        const bool fill = adc_vote_decide(bbin, bh, cnt, dmin, irv_ts, irv_th) != ADC_INVALID_FLOAT;
                        const size_t i16 = (size_t)y * SP + x;
                        const uint32_t cur = st[i16], nb = fill ? (uint32_t)bbin : IRV_BIN_MASK;
                        st[i16] = (uint16_t)(nb | IRV_ELIG | (deps_open ? 0u : IRV_FINAL));
                        if (nb != (cur & IRV_BIN_MASK)) { chg_wr[(y / T) * tiles_x + x / T] = stamp; *acc = 1; }
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

// ---------------------------------------------------------------------------------------------------------------------
// The packed-halfword helpers of irv_plan.h against per-pixel loops on random inputs (values drawn so that equal bins,
// invalid bins, eligible / final flags and range edges all occur often).  Returns the number of disagreements.
extern "C" long emul_irv_swar_check(unsigned seed, long trials)
{
    srand(seed);
    long bad = 0;
    for (long t = 0; t < trials; t++) {
        // ---- region row block
        uint16_t s16[8];
        const int nb = 1 + rand() % 4;                   // few distinct bins -> "single bin" happens
        const int pool[4] = {rand() % 2047, rand() % 2047, IRV_BIN_MASK, rand() % 2047};
        for (int q = 0; q < 8; q++) {
            uint32_t v = (uint32_t)pool[rand() % nb];
            if (rand() % 3 == 0) v |= IRV_ELIG;
            if (rand() % 2 == 0) v |= IRV_FINAL;
            if (rand() % 4 == 0) v |= 0x3800u;           // bits 11..13 are unused: must not matter
            s16[q] = (uint16_t)v;
        }
        uint32_t w[4];
        for (int j = 0; j < 4; j++) w[j] = (uint32_t)s16[2 * j] | ((uint32_t)s16[2 * j + 1] << 16);
        const int px0 = 8 * (rand() % 50), y = 5 + rand() % 5, yt = y - 2 + rand() % 5, x = px0 - 6 + rand() % 20;
        int xl = px0 - 4 + rand() % 12, xr = xl + rand() % 14;
        if (xr < px0) xr = px0;                           // (the kernel only decodes blocks that intersect [xl, xr])
        if (xl > px0 + 7) xl = px0 + 7;
        if (xr < xl) xr = xl;
        const IrvBlock got = irv_decode_block(w[0], w[1], w[2], w[3], px0, xl, xr, yt, y, x);
        uint32_t okm = 0, first = 0;
        bool single = true, open = false, have = false;
        for (int q = 0; q < 8; q++) {
            const int px = px0 + q;
            const uint32_t sv = s16[q], bin = sv & IRV_BIN_MASK;
            const bool in = px >= xl && px <= xr, el = (sv & IRV_ELIG) != 0, pre = yt < y || (yt == y && px < x);
            if (in && bin != IRV_BIN_MASK && (!el || pre)) {
                okm |= 1u << q;
                if (!have) { first = bin; have = true; }
                single = single && bin == first;
            }
            if (in && el && pre && !(sv & IRV_FINAL)) open = true;
        }
        if (got.okm != okm || got.open != open) bad++;
        else if (okm && (got.first != first || got.single != single)) bad++;
        else if (okm) { // the pixels of the first bin, and -- on what is left -- the pixels that share the bin of the lowest one
            uint32_t same = 0;
            for (int q = 0; q < 8; q++)
                if (((okm >> q) & 1u) && (s16[q] & IRV_BIN_MASK) == first) same |= 1u << q;
            if (got.same != same) bad++;
            uint32_t rem = okm & ~same;
            while (rem) {
                uint32_t bin = 0xFFFFu, want_mask = 0;
                const uint32_t got_mask = irv_same_bin_mask(w[0], w[1], w[2], w[3], rem, &bin);
                const uint32_t b = s16[__builtin_ctz(rem)] & IRV_BIN_MASK;
                for (int q = 0; q < 8; q++)
                    if (((rem >> q) & 1u) && (s16[q] & IRV_BIN_MASK) == b) want_mask |= 1u << q;
                if (bin != b || got_mask != want_mask || !got_mask) { bad++; break; }
                rem &= ~got_mask;
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
