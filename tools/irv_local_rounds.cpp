// Region voting (K8) with tile-local rounds: analysis tool, CPU only (see tools/irv_local_rounds.py).
// One "kernel" = every tile of S x S pixels once, in random order: the tile takes a snapshot of the map around it (halo = the
// longest arm), runs up to T rounds over ITS eligible pixels on the snapshot and writes their values back.  jacobi = 1: a local
// round evaluates every entry of the tile from the previous local iterate (what a workgroup does with all its waves in
// parallel); jacobi = 0: in place in list (raster) order -- a sequential sweep, which resolves every dependency INSIDE the
// tile at once (a lower bound of the kernel count for this tile size, not a GPU schedule).  A pass ends with the first kernel that changes nothing (then the map is the fixed point = the
// reference's in-place result).  T = 1 is the present scheme (one round per kernel).  Prints kernels and vote evaluations.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "../adcensus_amd/csrc/adc_device_fn.h"

extern "C" long irv_local(float* disp, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D, int irv_ts, float irv_th,
                          int S, int T, int jacobi, unsigned seed, long* kernels_per_pass /*[10]*/, long* evals_out)
{
    const int P = W * H, HALO = 34;
    std::vector<uint8_t> elig(P);
    std::vector<int> hist(D);
    long total = 0, evals = 0;
    srand(seed);
    const int tw = (W + S - 1) / S, th = (H + S - 1) / S;
    std::vector<std::vector<int>> tile_list(tw * th);
    std::vector<float> loc;
    int pass = 0;
    for (int it = 0; it < 5; it++)
        for (int k = 0; k < 2; k++, pass++) {
            const int which = k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION;
            for (auto& v : tile_list) v.clear();
            long n = 0;
            for (int p = 0; p < P; p++) {
                elig[p] = (label[p] == which && disp[p] == ADC_INVALID_FLOAT) ? 1 : 0;
                if (elig[p]) { tile_list[((p / W) / S) * tw + (p % W) / S].push_back(p); n++; }
            }
            kernels_per_pass[pass] = 0;
            if (!n) continue;
            std::vector<int> order(tw * th);
            for (int i = 0; i < tw * th; i++) order[i] = i;
            while (true) {
                bool changed = false;
                for (int i = tw * th - 1; i > 0; i--) std::swap(order[i], order[rand() % (i + 1)]);
                for (int oi = 0; oi < tw * th; oi++) {
                    const std::vector<int>& L = tile_list[order[oi]];
                    if (L.empty()) continue;
                    const int ty = (order[oi] / tw) * S, tx = (order[oi] % tw) * S;
                    const int y0 = std::max(0, ty - HALO), y1 = std::min(H, ty + S + HALO), x0 = std::max(0, tx - HALO), x1 = std::min(W, tx + S + HALO);
                    const int lw = x1 - x0;
                    loc.resize((size_t)(y1 - y0) * lw);
                    for (int y = y0; y < y1; y++) memcpy(&loc[(size_t)(y - y0) * lw], &disp[(size_t)y * W + x0], lw * sizeof(float));
                    std::vector<float> pend(L.size());
                    for (int r = 0; r < T; r++) {
                        bool lchg = false;
                        size_t li = 0;
                        for (int p : L) {
                            const int y = p / W, x = p - y * W;
                            std::fill(hist.begin(), hist.end(), 0);
                            const uint8_t* arm = arms + (size_t)p * 4;
                            for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                                const int yt = y + t;
                                const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                                for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                                    const int q = yt * W + x + s;
                                    float v = loc[(size_t)(yt - y0) * lw + (x + s - x0)];
                                    if (elig[q] && q >= p) v = ADC_INVALID_FLOAT;
                                    if (v != ADC_INVALID_FLOAT) {
                                        const long b = lroundf(v) - dmin;
                                        if (b >= 0 && b < D) hist[b]++;
                                    }
                                }
                            }
                            int bh = 0, bb = 0x7fffffff, cnt = 0;
                            for (int b = 0; b < D; b++) { cnt += hist[b]; if (hist[b] > bh) { bh = hist[b]; bb = b; } }
                            const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                            evals++;
                            float& cur = loc[(size_t)(y - y0) * lw + (x - x0)];
                            if (jacobi) pend[li++] = nv;
                            else if (memcmp(&cur, &nv, 4)) { cur = nv; lchg = true; }
                        }
                        if (jacobi) {
                            li = 0;
                            for (int p : L) {
                                const int y = p / W, x = p - y * W;
                                float& cur = loc[(size_t)(y - y0) * lw + (x - x0)];
                                if (memcmp(&cur, &pend[li], 4)) { cur = pend[li]; lchg = true; }
                                li++;
                            }
                        }
                        if (!lchg) break;
                    }
                    for (int p : L) {
                        const int y = p / W, x = p - y * W;
                        const float nv = loc[(size_t)(y - y0) * lw + (x - x0)];
                        if (memcmp(&disp[p], &nv, 4)) { disp[p] = nv; changed = true; }
                    }
                }
                kernels_per_pass[pass]++;
                total++;
                if (!changed) break;
            }
        }
    if (evals_out) *evals_out = evals;
    return total;
}


// Same scheme with what a kernel would really do: (a) a tile runs in kernel k of a pass only if it or one of its 8 neighbours
// changed a value in kernel k-1 (the first kernel of a pass runs every tile that has entries); (b) inside a tile an entry is
// re-evaluated in local round r > 0 only if a value inside its dependency box changed in local round r-1 (the box test of the
// present kernel, on pixels instead of 8x8 tiles).  Per kernel: active tiles, votes, and the votes of the busiest tile (the
// critical path of the kernel: a tile's votes are shared by the 16 waves of its workgroup).
extern "C" long irv_local_stats(float* disp, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D, int irv_ts,
                                float irv_th, int S, int T, unsigned seed, long* stats /* [kernels][4]: pass, active tiles, votes, max votes per tile */,
                                long max_kernels)
{
    const int P = W * H, HALO = 34;
    std::vector<uint8_t> elig(P);
    std::vector<int> hist(D);
    long total = 0;
    srand(seed);
    const int tw = (W + S - 1) / S, th = (H + S - 1) / S;
    std::vector<std::vector<int>> tile_list(tw * th);
    std::vector<float> loc;
    std::vector<uint8_t> lchg_map, lchg_prev;
    int pass = 0;
    for (int it = 0; it < 5; it++)
        for (int k = 0; k < 2; k++, pass++) {
            const int which = k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION;
            for (auto& v : tile_list) v.clear();
            long n = 0;
            for (int p = 0; p < P; p++) {
                elig[p] = (label[p] == which && disp[p] == ADC_INVALID_FLOAT) ? 1 : 0;
                if (elig[p]) { tile_list[((p / W) / S) * tw + (p % W) / S].push_back(p); n++; }
            }
            if (!n) continue;
            std::vector<uint8_t> tchg_prev(tw * th, 1), tchg(tw * th, 0);
            std::vector<int> order(tw * th);
            for (int i = 0; i < tw * th; i++) order[i] = i;
            for (int kern = 0;; kern++) {
                bool changed = false;
                long active = 0, votes = 0, maxv = 0;
                std::fill(tchg.begin(), tchg.end(), 0);
                for (int i = tw * th - 1; i > 0; i--) std::swap(order[i], order[rand() % (i + 1)]);
                for (int oi = 0; oi < tw * th; oi++) {
                    const int t = order[oi];
                    const std::vector<int>& L = tile_list[t];
                    if (L.empty()) continue;
                    const int tyi = t / tw, txi = t % tw;
                    bool run = kern == 0;
                    for (int dy = -1; dy <= 1 && !run; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            const int yy = tyi + dy, xx = txi + dx;
                            if (yy >= 0 && yy < th && xx >= 0 && xx < tw && tchg_prev[yy * tw + xx]) run = true;
                        }
                    if (!run) continue;
                    active++;
                    const int ty = tyi * S, tx = txi * S;
                    const int y0 = std::max(0, ty - HALO), y1 = std::min(H, ty + S + HALO), x0 = std::max(0, tx - HALO), x1 = std::min(W, tx + S + HALO);
                    const int lw = x1 - x0, lh = y1 - y0;
                    loc.resize((size_t)lh * lw);
                    for (int y = y0; y < y1; y++) memcpy(&loc[(size_t)(y - y0) * lw], &disp[(size_t)y * W + x0], lw * sizeof(float));
                    lchg_prev.assign((size_t)lh * lw, 1);
                    std::vector<float> pend(L.size());
                    std::vector<uint8_t> evald(L.size());
                    long tv = 0;
                    for (int r = 0; r < T; r++) {
                        bool lchg = false;
                        lchg_map.assign((size_t)lh * lw, 0);
                        size_t li = 0;
                        for (int p : L) {
                            const int y = p / W, x = p - y * W;
                            const uint8_t* arm = arms + (size_t)p * 4;
                            bool dirty = r == 0;
                            if (!dirty) // any value of the cross region changed in the previous local round?
                                for (int tt = -(int)arm[2]; tt <= (int)arm[3] && !dirty; tt++) {
                                    const uint8_t* arm2 = arms + ((size_t)(y + tt) * W + x) * 4;
                                    for (int ss = -(int)arm2[0]; ss <= (int)arm2[1]; ss++)
                                        if (lchg_prev[(size_t)(y + tt - y0) * lw + (x + ss - x0)]) { dirty = true; break; }
                                }
                            evald[li] = dirty;
                            if (dirty) {
                                std::fill(hist.begin(), hist.end(), 0);
                                for (int tt = -(int)arm[2]; tt <= (int)arm[3]; tt++) {
                                    const int yt = y + tt;
                                    const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                                    for (int ss = -(int)arm2[0]; ss <= (int)arm2[1]; ss++) {
                                        const int q = yt * W + x + ss;
                                        float v = loc[(size_t)(yt - y0) * lw + (x + ss - x0)];
                                        if (elig[q] && q >= p) v = ADC_INVALID_FLOAT;
                                        if (v != ADC_INVALID_FLOAT) {
                                            const long b = lroundf(v) - dmin;
                                            if (b >= 0 && b < D) hist[b]++;
                                        }
                                    }
                                }
                                int bh = 0, bb = 0x7fffffff, cnt = 0;
                                for (int b = 0; b < D; b++) { cnt += hist[b]; if (hist[b] > bh) { bh = hist[b]; bb = b; } }
                                pend[li] = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                                tv++;
                            }
                            li++;
                        }
                        li = 0;
                        for (int p : L) {
                            if (evald[li]) {
                                const int y = p / W, x = p - y * W;
                                float& cur = loc[(size_t)(y - y0) * lw + (x - x0)];
                                if (memcmp(&cur, &pend[li], 4)) { cur = pend[li]; lchg = true; lchg_map[(size_t)(y - y0) * lw + (x - x0)] = 1; }
                            }
                            li++;
                        }
                        lchg_prev.swap(lchg_map);
                        if (!lchg) break;
                    }
                    votes += tv;
                    maxv = std::max(maxv, tv);
                    for (int p : L) {
                        const int y = p / W, x = p - y * W;
                        const float nv = loc[(size_t)(y - y0) * lw + (x - x0)];
                        if (memcmp(&disp[p], &nv, 4)) { disp[p] = nv; changed = true; tchg[t] = 1; }
                    }
                }
                if (total < max_kernels) { stats[total * 4] = pass; stats[total * 4 + 1] = active; stats[total * 4 + 2] = votes; stats[total * 4 + 3] = maxv; }
                total++;
                tchg_prev.swap(tchg);
                if (!changed) break;
            }
        }
    return total;
}
