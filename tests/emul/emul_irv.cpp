// CPU model of the device-driven region-voting chain (adcensus_amd/csrc/k_voting.hip): the SAME state machine
// (irv_plan.h: every kernel derives its action from the state and the accumulator its predecessor left) drives plain-loop
// versions of the kernels' phases on the same 16-bit state map (bin | final | eligible), with the votes of a round
// evaluated in a shuffled order against the in-place map (arbitrary wave scheduling).  Test infrastructure.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../adcensus_amd/csrc/irv_plan.h"

extern "C" long emul_irv_chain(float* disp, const uint8_t* label, const uint8_t* arms, const uint16_t* sup_h, int W, int H, int dmin,
                               int D, int irv_ts, float irv_th, int min_region, unsigned seed, int budget_pairs, long* out_stats)
{
    const int P = W * H, SP = (W + 7) & ~7, T = IRV_TILE;
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    std::vector<uint16_t> st((size_t)SP * H + 64, 0xFFFF);
    std::vector<int32_t> ctrl(160, 0), list(P), dlist(P), chg(tiles_x * tiles_y, 0), hist(D);
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
    long kernels = 0;
    const int32_t* fin = nullptr;
    (void)budget_pairs;
    for (int k = 0;; k++) {
        if (ctrl[16 * (k & 1)] == IRV_DONE) { fin = &ctrl[16 * (k & 1)]; break; } // the state the previous kernel published
        const IrvPlan pl = irv_plan(ctrl.data(), k);
        int32_t* acc = &ctrl[IRV_ACC + (k & 63)];
        kernels++;
        if ((k & 1) == 0) { // kernel A
            if (pl.act == IRV_BEGIN || pl.act == IRV_FINAL_WB) {
                const bool have_state = !(pl.act == IRV_BEGIN && pl.s.pass == 0);
                const int which = (pl.s.pass & 1) ? ADC_LABEL_OCCLUSION : ADC_LABEL_MISMATCH;
                if (pl.act == IRV_BEGIN) std::fill(chg.begin(), chg.end(), 0);
                for (int p = 0; p < P; p++) {
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
                        if (listed) list[(*acc)++] = p;
                    }
                }
            } else if (pl.act == IRV_CHECK) {
                for (int i = 0; i < pl.nwork; i++) {
                    const int p = list[i], y = p / W, x = p - y * W;
                    if (st[(size_t)y * SP + x] & IRV_FINAL) continue;
                    const uint8_t* o = &bb[(size_t)p * 3];
                    const int tx0 = std::max(0, x - (int)o[1]) / T, tx1 = std::min(W - 1, x + (int)o[2]) / T;
                    const int ty0 = std::max(0, y - (int)o[0]) / T, ty1 = y / T;
                    bool dirty = false;
                    for (int ty = ty0; ty <= ty1; ty++)
                        for (int tx = tx0; tx <= tx1; tx++) dirty |= chg[ty * tiles_x + tx] == ((pl.s.round - 1) % 255) + 1; // byte stamps (k_voting.hip)
                    if (dirty) dlist[(*acc)++] = p;
                }
            }
        } else if (pl.act == IRV_VOTE) { // kernel B
            std::vector<int32_t> work(pl.s.round == 0 ? list.begin() : dlist.begin(), (pl.s.round == 0 ? list.begin() : dlist.begin()) + pl.nwork);
            for (size_t i = work.size(); i > 1; i--) std::swap(work[i - 1], work[rand() % i]);
            for (int p : work) {
                const int y = p / W, x = p - y * W;
                std::fill(hist.begin(), hist.end(), 0);
                bool deps_open = false;
                const uint8_t* arm = arms + (size_t)p * 4;
                for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                    const int yt = y + t;
                    const uint8_t* a2 = arms + ((size_t)yt * W + x) * 4;
                    for (int px = x - (int)a2[0]; px <= x + (int)a2[1]; px++) {
                        const uint32_t s = st[(size_t)yt * SP + px];
                        const bool el = (s & IRV_ELIG) != 0, pre = yt < y || (yt == y && px < x);
                        const uint32_t bin = s & IRV_BIN_MASK;
                        if (bin != IRV_BIN_MASK && (!el || pre)) hist[bin]++;
                        if (el && pre && !(s & IRV_FINAL)) deps_open = true;
                    }
                }
                int bh = 0, bbin = 0x7fffffff, cnt = 0;
                for (int b = 0; b < D; b++) { cnt += hist[b]; if (hist[b] > bh) { bh = hist[b]; bbin = b; } }
                const bool fill = adc_vote_decide(bbin, bh, cnt, dmin, irv_ts, irv_th) != ADC_INVALID_FLOAT;
                const size_t i16 = (size_t)y * SP + x;
                const uint32_t cur = st[i16], nb = fill ? (uint32_t)bbin : IRV_BIN_MASK;
                st[i16] = (uint16_t)(nb | IRV_ELIG | (deps_open ? 0u : IRV_FINAL));
                if (nb != (cur & IRV_BIN_MASK)) { chg[(y / T) * tiles_x + x / T] = (pl.s.round % 255) + 1; *acc = 1; }
            }
        }
        irv_publish(ctrl.data(), k, pl.s);
        if (kernels > 4000000) return -1;
    }
    if (out_stats) { out_stats[0] = fin[5]; out_stats[1] = fin[6]; out_stats[2] = kernels; }
    return fin[0] == IRV_DONE ? fin[5] : -2;
}
