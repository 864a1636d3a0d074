// Region voting (K8) as ONE fixed-point iteration over all ten passes: analysis tool, CPU only (tools/irv_joint_rounds.py).
//
// The reference (multistep_refiner.cpp:153-227) runs 5 iterations x {mismatches, occlusions} one after the other, each pass
// in place in raster order.  Written as a system of equations over "levels" l = 2 * it + list:
//     value(p, it) = value(p, it - 1)                                   if that one is valid (a filled pixel stays filled)
//                  = vote(p | q of the same list before p:  value(q, it)
//                           | q of the same list from p on: value(q, it - 1)
//                           | q of the other list:          value(q, it - 1) for a mismatch p, value(q, it) for an occlusion p
//                           | unlisted q:                   the LR-checked map)
// The system is triangular in the order (level, raster position), so a Jacobi iteration over ALL levels at once converges to
// the unique solution = the sequential result; a round in which nothing changes proves it.  Passes overlap like a
// pipeline: level l works on a region as soon as level l - 1 has settled there.  This tool counts the rounds of that
// iteration against the sum of the per-pass rounds of the present chain, and the vote evaluations with a per-entry dirty test
// (an entry is re-evaluated iff one of its inputs changed in the previous round).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "../adcensus_amd/csrc/adc_device_fn.h"

namespace {
struct Ctx {
    int W, H, dmin, D, irv_ts;
    float irv_th;
    const float* d0;
    const uint8_t* label;
    const uint8_t* arms;
};
// value of pixel q as seen at level `lvl` (map after level lvl), lvl = -1: the LR-checked map
inline float seen(const Ctx& c, const std::vector<float>& v, int q, int lvl)
{
    const int lab = c.label[q];
    if (lab == 0) return c.d0[q];
    const int list = lab == ADC_LABEL_MISMATCH ? 0 : 1;
    if (lvl < list) return c.d0[q];
    return v[(size_t)q * 5 + (lvl - list) / 2];
}
}

// returns the number of Jacobi rounds (incl. the final round without a change); out = value(p, 4) merged into the map
extern "C" long irv_joint(float* out, const float* d0, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D, int irv_ts,
                          float irv_th, int max_levels, long* evals_per_round /*[max_rounds]*/, long* changes_per_round, long max_rounds)
{
    const Ctx c = {W, H, dmin, D, irv_ts, irv_th, d0, label, arms};
    const int P = W * H;
    std::vector<float> cur((size_t)P * 5, ADC_INVALID_FLOAT), old, nxt;
    std::vector<int> listed;
    for (int p = 0; p < P; p++)
        if (label[p] != 0) {
            listed.push_back(p);
            for (int it = 0; it < 5; it++) cur[(size_t)p * 5 + it] = d0[p]; // (invalid for every listed pixel)
        }
    old = cur;
    nxt = cur;
    std::vector<int> hist(D);
    long rounds = 0;
    const int iters = max_levels / 2;
    for (;; rounds++) {
        long evals = 0, changes = 0;
        for (int p : listed) {
            const int list = label[p] == ADC_LABEL_MISMATCH ? 0 : 1;
            const int y = p / W, x = p - y * W;
            const uint8_t* arm = arms + (size_t)p * 4;
            for (int it = 0; it < iters; it++) {
                const int lvl = 2 * it + list;
                const float prevv = it ? cur[(size_t)p * 5 + it - 1] : d0[p];
                float nv;
                if (prevv != ADC_INVALID_FLOAT) nv = prevv;
                else {
                    // dirty test: did any input differ between the last two iterates?
                    bool dirty = rounds == 0 || (it && memcmp(&cur[(size_t)p * 5 + it - 1], &old[(size_t)p * 5 + it - 1], 4));
                    std::fill(hist.begin(), hist.end(), 0);
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                        const int yt = y + t;
                        const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                        for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                            const int q = yt * W + x + s;
                            const int ql = label[q];
                            int ql_lvl; // level at which q is seen
                            if (ql == 0) ql_lvl = -1;
                            else {
                                const int qlist = ql == ADC_LABEL_MISMATCH ? 0 : 1;
                                if (qlist == list) ql_lvl = q < p ? lvl : lvl - 2;
                                else ql_lvl = lvl - 1;
                            }
                            const float vq = seen(c, cur, q, ql_lvl);
                            if (!dirty && ql != 0) {
                                const float vo = seen(c, old, q, ql_lvl);
                                if (memcmp(&vq, &vo, 4)) dirty = true;
                            }
                            if (vq != ADC_INVALID_FLOAT) {
                                const long b = lroundf(vq) - dmin;
                                if (b >= 0 && b < D) hist[b]++;
                            }
                        }
                    }
                    int bh = 0, bb = 0x7fffffff, cnt = 0;
                    for (int b = 0; b < D; b++) { cnt += hist[b]; if (hist[b] > bh) { bh = hist[b]; bb = b; } }
                    nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                    if (dirty) evals++;
                }
                float& dst = nxt[(size_t)p * 5 + it];
                if (memcmp(&cur[(size_t)p * 5 + it], &nv, 4)) changes++;
                dst = nv;
            }
        }
        if (rounds < max_rounds) { evals_per_round[rounds] = evals; changes_per_round[rounds] = changes; }
        old.swap(cur);   // old = iterate r - 1
        cur.swap(nxt);   // cur = iterate r
        nxt = cur;
        if (changes == 0) { rounds++; break; }
    }
    memcpy(out, d0, (size_t)P * 4);
    for (int p : listed) out[p] = cur[(size_t)p * 5 + iters - 1];
    return rounds;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same system with ONE state per pixel.  A filled pixel stays filled, so value(p, 0..4) = INVALID, .., INVALID, b, .., b:
// the vector is (f, b) = (iteration of the fill, bin), f = 5: never filled.  Region pixel q counts in the vote of p at
// iteration `it` iff it >= t(q):
//     unlisted q               t = 0 (if valid)
//     same list, before p      t = f_q          (the in-place scan has already passed q in this iteration)
//     same list, from p on     t = f_q + 1      (... will only pass it later: p sees the previous iteration's value)
//     other list               t = f_q + 1 for a mismatch p (mismatches run first), f_q for an occlusion p
// so ONE gather of the region yields the five cumulative histograms, and evaluating a pixel = finding the first iteration
// whose vote passes.  jacobi = 1: every evaluation of a round reads the previous iterate; 0: in place (a GPU round is in between).
// dirty_mode 0: exact (a region pixel's state changed in the previous round); 1: an 8x8 tile of the region's bounding box
// saw a change.  Returns rounds; evals = pixel evaluations (dirty pixels), votes = histogram decisions taken.
extern "C" long irv_joint_px(float* out, const float* d0, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D,
                             int irv_ts, float irv_th, int jacobi, int dirty_mode, long* evals_per_round, long* changes_per_round,
                             long* votes_total, long max_rounds)
{
    const int P = W * H;
    std::vector<uint8_t> f(P, 5), fn;
    std::vector<int> b(P, -1), bn;
    std::vector<int> listed;
    for (int p = 0; p < P; p++) {
        if (label[p] != 0) listed.push_back(p);
        else if (d0[p] != ADC_INVALID_FLOAT) { f[p] = 0; b[p] = (int)(lroundf(d0[p]) - dmin); }
    }
    fn = f; bn = b;
    const int TW = (W + 7) / 8, TH = (H + 7) / 8;
    std::vector<uint8_t> chg_prev((size_t)P, 1), chg_now((size_t)P, 0), tchg_prev((size_t)TW * TH, 1), tchg_now((size_t)TW * TH, 0);
    // dirty_mode 2: a tile remembers the LOWEST fill iteration involved in a change (min of old and new f over its changed pixels,
    // 9 = no change); p, currently filled at f_p, only depends on inputs that count at iterations <= f_p, i.e. with f_q <= f_p
    std::vector<uint8_t> tlvl_prev((size_t)TW * TH, 0), tlvl_now((size_t)TW * TH, 9);
    std::vector<int> add((size_t)5 * D);
    long rounds = 0, votes = 0;
    for (;; rounds++) {
        long evals = 0, changes = 0;
        std::fill(chg_now.begin(), chg_now.end(), 0);
        std::fill(tchg_now.begin(), tchg_now.end(), 0);
        std::fill(tlvl_now.begin(), tlvl_now.end(), 9);
        for (int p : listed) {
            const int list = label[p] == ADC_LABEL_MISMATCH ? 0 : 1;
            const int y = p / W, x = p - y * W;
            const uint8_t* arm = arms + (size_t)p * 4;
            bool dirty = rounds == 0;
            if (!dirty && dirty_mode == 0) {
                for (int t = -(int)arm[2]; t <= (int)arm[3] && !dirty; t++) {
                    const uint8_t* arm2 = arms + ((size_t)(y + t) * W + x) * 4;
                    for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++)
                        if (chg_prev[(size_t)(y + t) * W + x + s]) { dirty = true; break; }
                }
            } else if (!dirty) {
                int ml = 0, mr = 0;
                for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                    const uint8_t* arm2 = arms + ((size_t)(y + t) * W + x) * 4;
                    ml = std::max(ml, (int)arm2[0]); mr = std::max(mr, (int)arm2[1]);
                }
                for (int ty = (y - arm[2]) / 8; ty <= (y + arm[3]) / 8 && !dirty; ty++)
                    for (int tx = (x - ml) / 8; tx <= (x + mr) / 8; tx++)
                        if (dirty_mode == 2 ? tlvl_prev[(size_t)ty * TW + tx] <= f[p] : tchg_prev[(size_t)ty * TW + tx]) { dirty = true; break; }
            }
            if (!dirty) continue;
            evals++;
            std::fill(add.begin(), add.end(), 0);
            const std::vector<uint8_t>& fr = jacobi ? f : fn;
            const std::vector<int>& br = jacobi ? b : bn;
            for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                const int yt = y + t;
                const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                    const int q = yt * W + x + s;
                    int fq = fr[q];
                    if (fq >= 5) continue;
                    const int ql = label[q];
                    int tq;
                    if (ql == 0) tq = 0;
                    else {
                        const int qlist = ql == ADC_LABEL_MISMATCH ? 0 : 1;
                        if (qlist == list) tq = q < p ? fq : fq + 1;
                        else tq = list == 0 ? fq + 1 : fq;
                    }
                    if (tq >= 5) continue;
                    const int bq = br[q];
                    if (bq >= 0 && bq < D) add[(size_t)tq * D + bq]++;
                }
            }
            int nf = 5, nb = -1;
            for (int it = 0; it < 5; it++) {
                if (it) for (int d = 0; d < D; d++) add[(size_t)it * D + d] += add[(size_t)(it - 1) * D + d];
                const int* hist = &add[(size_t)it * D];
                int bh = 0, bb = 0x7fffffff, cnt = 0;
                for (int d = 0; d < D; d++) { cnt += hist[d]; if (hist[d] > bh) { bh = hist[d]; bb = d; } }
                votes++;
                const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                if (nv != ADC_INVALID_FLOAT) { nf = it; nb = (int)(lroundf(nv) - dmin); break; }
            }
            if (nf != fn[p] || nb != bn[p]) {
                changes++;
                chg_now[p] = 1;
                tchg_now[(size_t)(y / 8) * TW + x / 8] = 1;
                uint8_t& tl = tlvl_now[(size_t)(y / 8) * TW + x / 8];
                tl = std::min(tl, (uint8_t)std::min((int)fn[p], nf));
            }
            fn[p] = (uint8_t)nf; bn[p] = nb;
        }
        if (rounds < max_rounds) { evals_per_round[rounds] = evals; changes_per_round[rounds] = changes; }
        f = fn; b = bn;
        chg_prev.swap(chg_now);
        tchg_prev.swap(tchg_now);
        tlvl_prev.swap(tlvl_now);
        if (changes == 0) { rounds++; break; }
    }
    memcpy(out, d0, (size_t)P * 4);
    for (int p : listed) out[p] = f[p] < 5 ? (float)(b[p] + dmin) : ADC_INVALID_FLOAT;
    if (votes_total) *votes_total = votes;
    return rounds;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same iteration with a GAUSS-SEIDEL schedule a GPU can run: the work list is sorted COLUMN-major (x, then y), a wave owns a
// chunk of C consecutive entries (a vertical run) and evaluates its dirty entries top-down, one after the other, IN PLACE; all
// chunks run side by side (modelled in lock step: at step j every chunk evaluates its j-th dirty entry against the map as it was
// after step j - 1).  Most of a vote's inputs lie in the rows above the pixel, which the wave itself (and its neighbours, one
// column over, at the same pace) have just settled -- a fill front runs through a whole chunk in ONE round instead of one row.
// row_major = 1: chunks along rows instead (for comparison).  Returns rounds.
extern "C" long irv_joint_chunks(float* out, const float* d0, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D,
                                 int irv_ts, float irv_th, int C, int row_major, long* evals_per_round, long* changes_per_round,
                                 long* steps_per_round, long max_rounds)
{
    const int P = W * H;
    std::vector<uint8_t> f(P, 5);
    std::vector<int> b(P, -1);
    std::vector<int> listed;
    if (row_major) { for (int p = 0; p < P; p++) if (label[p] != 0) listed.push_back(p); }
    else for (int x = 0; x < W; x++) for (int y = 0; y < H; y++) if (label[y * W + x] != 0) listed.push_back(y * W + x);
    for (int p = 0; p < P; p++)
        if (label[p] == 0 && d0[p] != ADC_INVALID_FLOAT) { f[p] = 0; b[p] = (int)(lroundf(d0[p]) - dmin); }
    const int TW = (W + 7) / 8, TH = (H + 7) / 8;
    std::vector<uint8_t> tchg_prev((size_t)TW * TH, 1), tchg_now((size_t)TW * TH, 0);
    std::vector<int> add((size_t)5 * D);
    const long nchunks = ((long)listed.size() + C - 1) / C;
    std::vector<std::vector<int>> todo(nchunks);
    struct Upd { int p; uint8_t f; int b; };
    std::vector<Upd> upd;
    long rounds = 0;
    for (;; rounds++) {
        long evals = 0, changes = 0, steps = 0;
        std::fill(tchg_now.begin(), tchg_now.end(), 0);
        for (long c = 0; c < nchunks; c++) { // phase 1: which entries are dirty (stamps of the previous round)
            todo[c].clear();
            for (long i = c * C; i < std::min((long)listed.size(), (c + 1) * C); i++) {
                const int p = listed[i], y = p / W, x = p - y * W;
                const uint8_t* arm = arms + (size_t)p * 4;
                bool dirty = rounds == 0;
                if (!dirty) {
                    int ml = 0, mr = 0;
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                        const uint8_t* arm2 = arms + ((size_t)(y + t) * W + x) * 4;
                        ml = std::max(ml, (int)arm2[0]); mr = std::max(mr, (int)arm2[1]);
                    }
                    for (int ty = (y - arm[2]) / 8; ty <= (y + arm[3]) / 8 && !dirty; ty++)
                        for (int tx = (x - ml) / 8; tx <= (x + mr) / 8; tx++)
                            if (tchg_prev[(size_t)ty * TW + tx]) { dirty = true; break; }
                }
                if (dirty) todo[c].push_back(p);
            }
            steps = std::max(steps, (long)todo[c].size());
        }
        for (long j = 0; j < steps; j++) { // lock step j: every chunk's j-th dirty entry, reading the map after step j - 1
            upd.clear();
            for (long c = 0; c < nchunks; c++) {
                if ((long)todo[c].size() <= j) continue;
                const int p = todo[c][j];
                const int list = label[p] == ADC_LABEL_MISMATCH ? 0 : 1;
                const int y = p / W, x = p - y * W;
                const uint8_t* arm = arms + (size_t)p * 4;
                evals++;
                std::fill(add.begin(), add.end(), 0);
                for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                    const int yt = y + t;
                    const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                    for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                        const int q = yt * W + x + s;
                        if (q == p) continue;
                        const int fq = f[q];
                        if (fq >= 5) continue;
                        const int ql = label[q];
                        int tq;
                        if (ql == 0) tq = 0;
                        else {
                            const int qlist = ql == ADC_LABEL_MISMATCH ? 0 : 1;
                            if (qlist == list) tq = q < p ? fq : fq + 1;
                            else tq = list == 0 ? fq + 1 : fq;
                        }
                        if (tq >= 5) continue;
                        const int bq = b[q];
                        if (bq >= 0 && bq < D) add[(size_t)tq * D + bq]++;
                    }
                }
                int nf = 5, nb = -1;
                for (int it = 0; it < 5; it++) {
                    if (it) for (int d = 0; d < D; d++) add[(size_t)it * D + d] += add[(size_t)(it - 1) * D + d];
                    const int* hist = &add[(size_t)it * D];
                    int bh = 0, bb = 0x7fffffff, cnt = 0;
                    for (int d = 0; d < D; d++) { cnt += hist[d]; if (hist[d] > bh) { bh = hist[d]; bb = d; } }
                    const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                    if (nv != ADC_INVALID_FLOAT) { nf = it; nb = (int)(lroundf(nv) - dmin); break; }
                }
                if (nf != f[p] || nb != b[p]) upd.push_back(Upd{p, (uint8_t)nf, nb});
            }
            for (const Upd& u : upd) {
                f[u.p] = u.f; b[u.p] = u.b;
                changes++;
                tchg_now[(size_t)((u.p / W) / 8) * TW + (u.p % W) / 8] = 1;
            }
        }
        if (rounds < max_rounds) { evals_per_round[rounds] = evals; changes_per_round[rounds] = changes; steps_per_round[rounds] = steps; }
        tchg_prev.swap(tchg_now);
        if (changes == 0) { rounds++; break; }
    }
    memcpy(out, d0, (size_t)P * 4);
    for (int p : listed) out[p] = f[p] < 5 ? (float)(b[p] + dmin) : ADC_INVALID_FLOAT;
    return rounds;
}

// Row-synchronised variant of the Gauss-Seidel schedule: the image is cut into BANDS of R rows; all bands run side by side, and
// inside a band the dirty entries are evaluated ROW BY ROW (step j = row r0 + j of every band; all pixels of that row at once,
// against the map as it was after step j - 1).  R = H: one top-down sweep of the whole image per round.
extern "C" long irv_joint_bands(float* out, const float* d0, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D,
                                int irv_ts, float irv_th, int R, int level_aware, long* evals_per_round, long* changes_per_round, long max_rounds)
{
    const int P = W * H;
    std::vector<uint8_t> f(P, 5);
    std::vector<int> b(P, -1);
    for (int p = 0; p < P; p++)
        if (label[p] == 0 && d0[p] != ADC_INVALID_FLOAT) { f[p] = 0; b[p] = (int)(lroundf(d0[p]) - dmin); }
    const int TW = (W + 7) / 8, TH = (H + 7) / 8;
    std::vector<uint8_t> tchg_prev((size_t)TW * TH, 1), tchg_now((size_t)TW * TH, 0), dirty(P);
    // level_aware: a second tile map for changes that involve iteration 0 (min of old and new fill iteration == 0); a pixel that is
    // filled at iteration 0 depends only on inputs that count at iteration 0 -- a mismatch pixel only on the rows above it
    std::vector<uint8_t> t0_prev((size_t)TW * TH, 1), t0_now((size_t)TW * TH, 0);
    std::vector<int> add((size_t)5 * D);
    struct Upd { int p; uint8_t f; int b; };
    std::vector<Upd> upd;
    long rounds = 0;
    for (;; rounds++) {
        long evals = 0, changes = 0;
        std::fill(tchg_now.begin(), tchg_now.end(), 0);
        std::fill(t0_now.begin(), t0_now.end(), 0);
        for (int p = 0; p < P; p++) {
            dirty[p] = 0;
            if (label[p] == 0) continue;
            const int y = p / W, x = p - y * W;
            const uint8_t* arm = arms + (size_t)p * 4;
            bool dt = rounds == 0;
            if (!dt) {
                int ml = 0, mr = 0;
                for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                    const uint8_t* arm2 = arms + ((size_t)(y + t) * W + x) * 4;
                    ml = std::max(ml, (int)arm2[0]); mr = std::max(mr, (int)arm2[1]);
                }
                const bool lvl0 = level_aware && f[p] == 0;
                const int ybot = (lvl0 && label[p] == ADC_LABEL_MISMATCH) ? y : y + arm[3];
                for (int ty = (y - arm[2]) / 8; ty <= ybot / 8 && !dt; ty++)
                    for (int tx = (x - ml) / 8; tx <= (x + mr) / 8; tx++)
                        if (lvl0 ? t0_prev[(size_t)ty * TW + tx] : tchg_prev[(size_t)ty * TW + tx]) { dt = true; break; }
            }
            dirty[p] = dt;
        }
        for (int j = 0; j < R; j++) {
            upd.clear();
            for (int r0 = 0; r0 < H; r0 += R) {
                const int y = r0 + j;
                if (y >= H) continue;
                for (int x = 0; x < W; x++) {
                    const int p = y * W + x;
                    if (!dirty[p]) continue;
                    const int list = label[p] == ADC_LABEL_MISMATCH ? 0 : 1;
                    const uint8_t* arm = arms + (size_t)p * 4;
                    evals++;
                    std::fill(add.begin(), add.end(), 0);
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                        const int yt = y + t;
                        const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                        for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                            const int q = yt * W + x + s;
                            if (q == p) continue;
                            const int fq = f[q];
                            if (fq >= 5) continue;
                            const int ql = label[q];
                            int tq;
                            if (ql == 0) tq = 0;
                            else {
                                const int qlist = ql == ADC_LABEL_MISMATCH ? 0 : 1;
                                if (qlist == list) tq = q < p ? fq : fq + 1;
                                else tq = list == 0 ? fq + 1 : fq;
                            }
                            if (tq >= 5) continue;
                            const int bq = b[q];
                            if (bq >= 0 && bq < D) add[(size_t)tq * D + bq]++;
                        }
                    }
                    int nf = 5, nb = -1;
                    for (int it = 0; it < 5; it++) {
                        if (it) for (int d = 0; d < D; d++) add[(size_t)it * D + d] += add[(size_t)(it - 1) * D + d];
                        const int* hist = &add[(size_t)it * D];
                        int bh = 0, bb = 0x7fffffff, cnt = 0;
                        for (int d = 0; d < D; d++) { cnt += hist[d]; if (hist[d] > bh) { bh = hist[d]; bb = d; } }
                        const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                        if (nv != ADC_INVALID_FLOAT) { nf = it; nb = (int)(lroundf(nv) - dmin); break; }
                    }
                    if (nf != f[p] || nb != b[p]) upd.push_back(Upd{p, (uint8_t)nf, nb});
                }
            }
            for (const Upd& u : upd) {
                if (f[u.p] == 0 || u.f == 0) t0_now[(size_t)((u.p / W) / 8) * TW + (u.p % W) / 8] = 1;
                f[u.p] = u.f; b[u.p] = u.b;
                changes++;
                tchg_now[(size_t)((u.p / W) / 8) * TW + (u.p % W) / 8] = 1;
            }
        }
        if (rounds < max_rounds) { evals_per_round[rounds] = evals; changes_per_round[rounds] = changes; }
        tchg_prev.swap(tchg_now);
        t0_prev.swap(t0_now);
        if (changes == 0) { rounds++; break; }
    }
    memcpy(out, d0, (size_t)P * 4);
    for (int p = 0; p < P; p++) if (label[p] != 0) out[p] = f[p] < 5 ? (float)(b[p] + dmin) : ADC_INVALID_FLOAT;
    return rounds;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: SLACK BUDGETS.  In the heavy rounds ~80 % of the re-evaluations end in the decision the entry already had: something in the
// region changed, but not enough to flip a vote.  An evaluation therefore also computes how many region pixels would have to change
// before its decision CAN change: per level with cumulative histogram (count c, top bin m, runner-up m2), after at most k pixel changes
//     a failing level keeps failing  if  c + k <= ts  or  (c - k >= 1 and fl((m + k) / (c - k)) <= th)
//     the passing level keeps its bin if  c - k > ts  and fl((m - k) / (c + k)) > th  and  m - k > m2 + k
// (every pixel change moves a level's histogram by at most -1 in one bin and +1 in another; float division is monotone).  K = the
// largest k for which every level up to the deciding one holds.  Change tiles COUNT the changes of a round; an entry subtracts the
// counts of its bounding box from its budget round by round and is only re-evaluated when the budget is used up.  Same band schedule
// as irv_joint_bands.  slack_mode 0 = off (any change -> dirty), 1 = on with the true runner-up, 2 = on with m2 := c - m (no second
// reduction).  cap = largest budget an entry can hold (8 bits on the GPU: 255).
namespace {
inline bool lvl_pass(int m, int c, int ts, float th) { return m > 0 && c > ts && (float)m * 1.0f / (float)c > th; }
inline int lvl_slack(bool pass, int c, int m, int m2, int ts, float th, int cap)
{
    int k = 0;
    for (; k < cap; k++) { // (largest k that still holds: test k + 1)
        const int kk = k + 1;
        bool ok;
        if (!pass) ok = (c + kk <= ts) || (c - kk >= 1 && (float)(m + kk) * 1.0f / (float)(c - kk) <= th);
        else ok = (c - kk > ts) && (m - kk >= 1) && ((float)(m - kk) * 1.0f / (float)(c + kk) > th) && (m - kk > m2 + kk);
        if (!ok) break;
    }
    return k;
}
}
extern "C" long irv_joint_bands_slack(float* out, const float* d0, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D,
                                      int irv_ts, float irv_th, int R, int slack_mode, int cap, long* evals_per_round, long* changes_per_round,
                                      long* slack_hist /*[cap + 1]*/, long max_rounds)
{
    const int P = W * H;
    std::vector<uint8_t> f(P, 5);
    std::vector<int> b(P, -1);
    for (int p = 0; p < P; p++)
        if (label[p] == 0 && d0[p] != ADC_INVALID_FLOAT) { f[p] = 0; b[p] = (int)(lroundf(d0[p]) - dmin); }
    const int TW = (W + 7) / 8, TH = (H + 7) / 8;
    std::vector<int> tcnt_prev((size_t)TW * TH, 0), tcnt_now((size_t)TW * TH, 0), budget(P, -1);
    std::vector<uint8_t> dirty(P), pchg_prev(P, 0), pchg_now(P, 0); // (slack_mode 3: per-pixel change map, counted over the exact region)
    std::vector<int> add((size_t)5 * D);
    struct Upd { int p; uint8_t f; int b; };
    std::vector<Upd> upd;
    long rounds = 0;
    for (;; rounds++) {
        long evals = 0, changes = 0;
        std::fill(tcnt_now.begin(), tcnt_now.end(), 0);
        std::fill(pchg_now.begin(), pchg_now.end(), 0);
        for (int p = 0; p < P; p++) {
            dirty[p] = 0;
            if (label[p] == 0) continue;
            const int y = p / W, x = p - y * W;
            const uint8_t* arm = arms + (size_t)p * 4;
            bool dt = rounds == 0;
            if (!dt) {
                int ml = 0, mr = 0;
                for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                    const uint8_t* arm2 = arms + ((size_t)(y + t) * W + x) * 4;
                    ml = std::max(ml, (int)arm2[0]); mr = std::max(mr, (int)arm2[1]);
                }
                int s = 0;
                if (slack_mode == 5 || slack_mode == 6) { // exact count over the region's bounding RECTANGLE (no arm lookups: all loads independent)
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++)
                        for (int q = -ml; q <= mr; q++) s += pchg_prev[(size_t)(y + t) * W + x + q];
                } else if (slack_mode >= 3) {
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                        const uint8_t* arm2 = arms + ((size_t)(y + t) * W + x) * 4;
                        for (int q = -(int)arm2[0]; q <= (int)arm2[1]; q++) s += pchg_prev[(size_t)(y + t) * W + x + q];
                    }
                } else
                for (int ty = (y - arm[2]) / 8; ty <= (y + arm[3]) / 8; ty++)
                    for (int tx = (x - ml) / 8; tx <= (x + mr) / 8; tx++) s += tcnt_prev[(size_t)ty * TW + tx];
                if (s > 0) {
                    if (!slack_mode) dt = true;
                    else { budget[p] -= s; dt = budget[p] < 0; }
                }
            }
            dirty[p] = dt;
        }
        for (int j = 0; j < R; j++) {
            upd.clear();
            for (int r0 = 0; r0 < H; r0 += R) {
                const int y = r0 + j;
                if (y >= H) continue;
                for (int x = 0; x < W; x++) {
                    const int p = y * W + x;
                    if (!dirty[p]) continue;
                    const int list = label[p] == ADC_LABEL_MISMATCH ? 0 : 1;
                    const uint8_t* arm = arms + (size_t)p * 4;
                    evals++;
                    std::fill(add.begin(), add.end(), 0);
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                        const int yt = y + t;
                        const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                        for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                            const int q = yt * W + x + s;
                            if (q == p) continue;
                            const int fq = f[q];
                            if (fq >= 5) continue;
                            const int ql = label[q];
                            int tq;
                            if (ql == 0) tq = 0;
                            else {
                                const int qlist = ql == ADC_LABEL_MISMATCH ? 0 : 1;
                                if (qlist == list) tq = q < p ? fq : fq + 1;
                                else tq = list == 0 ? fq + 1 : fq;
                            }
                            if (tq >= 5) continue;
                            const int bq = b[q];
                            if (bq >= 0 && bq < D) add[(size_t)tq * D + bq]++;
                        }
                    }
                    int nf = 5, nb = -1, K = cap;
                    for (int it = 0; it < 5; it++) {
                        if (it) for (int d = 0; d < D; d++) add[(size_t)it * D + d] += add[(size_t)(it - 1) * D + d];
                        const int* hist = &add[(size_t)it * D];
                        int bh = 0, bb = 0x7fffffff, cnt = 0, second = 0;
                        for (int d = 0; d < D; d++) { cnt += hist[d]; if (hist[d] > bh) { bh = hist[d]; bb = d; } }
                        for (int d = 0; d < D; d++) if (d != bb && hist[d] > second) second = hist[d];
                        if (slack_mode == 2 || slack_mode == 4 || slack_mode == 6) second = cnt - bh;
                        const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                        const bool pass = nv != ADC_INVALID_FLOAT;
                        K = std::min(K, lvl_slack(pass, cnt, bh, second, irv_ts, irv_th, cap));
                        if (pass) { nf = it; nb = (int)(lroundf(nv) - dmin); break; }
                    }
                    budget[p] = K;
                    if (slack_hist) slack_hist[K]++;
                    if (nf != f[p] || nb != b[p]) upd.push_back(Upd{p, (uint8_t)nf, nb});
                }
            }
            for (const Upd& u : upd) {
                f[u.p] = u.f; b[u.p] = u.b;
                changes++;
                tcnt_now[(size_t)((u.p / W) / 8) * TW + (u.p % W) / 8]++;
                pchg_now[u.p] = 1;
            }
        }
        if (rounds < max_rounds) { evals_per_round[rounds] = evals; changes_per_round[rounds] = changes; }
        tcnt_prev.swap(tcnt_now);
        pchg_prev.swap(pchg_now);
        if (changes == 0) { rounds++; break; }
    }
    memcpy(out, d0, (size_t)P * 4);
    for (int p = 0; p < P; p++) if (label[p] != 0) out[p] = f[p] < 5 ? (float)(b[p] + dmin) : ADC_INVALID_FLOAT;
    return rounds;
}
