// Region voting (K8), analysis tool (CPU only, see tools/irv_chunk_sweep.py): ORDERED sweeps of list chunks.
// The work list of a pass is in raster order; chunk c = entries [c * C, (c + 1) * C) belongs to one wave.  One "kernel" = every
// chunk that holds a dirty entry is swept ONCE, in list order, in place: an entry is evaluated if it is dirty (round 0: every
// entry; later: a pixel of its dependency box -- 8x8 change tiles, like the product -- changed in the previous kernel) or if a
// pixel this sweep has changed so far lies in its dependency box (exact test).  Dependencies only point backwards in the list, so
// one ordered sweep settles everything INSIDE a chunk; what crosses chunks waits for the next kernel.  C = 1 is the present
// scheme (one wave per dirty entry and round).  Chunks run in random order (in-place, chaotic across chunks).
// Per kernel: votes, chunks swept, the longest sweep (votes of one wave = the critical path of the kernel).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "../adcensus_amd/csrc/adc_device_fn.h"

extern "C" long irv_chunk_sweep(float* disp, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D, int irv_ts, float irv_th,
                                int C, unsigned seed, long* stats /* [kernels][4]: pass, chunks swept, votes, longest sweep */, long max_kernels)
{
    const int P = W * H, T = 8;
    std::vector<uint8_t> elig(P);
    std::vector<int> hist(D), list;
    // dependency box per pixel: rows y-top..y, widest H arm of those rows
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
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    std::vector<uint8_t> chg_a(tiles_x * tiles_y), chg_b(tiles_x * tiles_y);
    long total = 0;
    srand(seed);
    for (int it = 0; it < 5; it++)
        for (int k = 0; k < 2; k++) {
            const int pass = it * 2 + k, which = k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION;
            list.clear();
            for (int p = 0; p < P; p++) {
                elig[p] = (label[p] == which && disp[p] == ADC_INVALID_FLOAT) ? 1 : 0;
                if (elig[p]) list.push_back(p);
            }
            if (list.empty()) continue;
            const int nch = ((int)list.size() + C - 1) / C;
            std::vector<int> order(nch);
            for (int i = 0; i < nch; i++) order[i] = i;
            std::fill(chg_a.begin(), chg_a.end(), 0);
            for (int round = 0;; round++) {
                std::fill(chg_b.begin(), chg_b.end(), 0);
                bool changed = false;
                long votes = 0, swept = 0, longest = 0;
                for (int i = nch - 1; i > 0; i--) std::swap(order[i], order[rand() % (i + 1)]);
                std::vector<int> mine; // pixels this sweep has changed
                for (int oi = 0; oi < nch; oi++) {
                    const int c = order[oi];
                    mine.clear();
                    long v = 0;
                    for (int e = c * C; e < std::min((int)list.size(), (c + 1) * C); e++) {
                        const int p = list[e], y = p / W, x = p - y * W;
                        const uint8_t* o = &bb[(size_t)p * 3];
                        const int bx0 = std::max(0, x - (int)o[1]), bx1 = std::min(W - 1, x + (int)o[2]), by0 = std::max(0, y - (int)o[0]);
                        bool dirty = round == 0;
                        if (!dirty)
                            for (int ty = by0 / T; ty <= y / T && !dirty; ty++)
                                for (int tx = bx0 / T; tx <= bx1 / T; tx++) dirty |= chg_a[ty * tiles_x + tx] != 0;
                        if (!dirty)
                            for (int q : mine) {
                                const int qy = q / W, qx = q - qy * W;
                                if (qy >= by0 && qy <= y && qx >= bx0 && qx <= bx1) { dirty = true; break; }
                            }
                        if (!dirty) continue;
                        std::fill(hist.begin(), hist.end(), 0);
                        const uint8_t* arm = arms + (size_t)p * 4;
                        for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                            const int yt = y + t;
                            const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                            for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                                const int q = yt * W + x + s;
                                float val = disp[q];
                                if (elig[q] && q >= p) val = ADC_INVALID_FLOAT;
                                if (val != ADC_INVALID_FLOAT) {
                                    const long b = lroundf(val) - dmin;
                                    if (b >= 0 && b < D) hist[b]++;
                                }
                            }
                        }
                        int bh = 0, bbn = 0x7fffffff, cnt = 0;
                        for (int b = 0; b < D; b++) { cnt += hist[b]; if (hist[b] > bh) { bh = hist[b]; bbn = b; } }
                        const float nv = adc_vote_decide(bbn, bh, cnt, dmin, irv_ts, irv_th);
                        v++;
                        if (memcmp(&disp[p], &nv, 4)) {
                            disp[p] = nv;
                            chg_b[(y / T) * tiles_x + x / T] = 1;
                            mine.push_back(p);
                            changed = true;
                        }
                    }
                    if (v) { swept++; votes += v; longest = std::max(longest, v); }
                }
                if (total < max_kernels) {
                    stats[4 * total] = pass; stats[4 * total + 1] = swept; stats[4 * total + 2] = votes; stats[4 * total + 3] = longest;
                }
                total++;
                chg_a.swap(chg_b);
                if (!changed) break;
            }
        }
    return total;
}
