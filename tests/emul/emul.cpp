// tests/emul/emul.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Scalar CPU emulation of the *parallel formulations* the HIP kernels use for the reference's
// order-dependent steps, sharing adcensus_amd/csrc/adc_device_fn.h with the device code.  It lets
// the CPU-only test tier (no GPU in the build container) check, against the oracle, that
//   * the marching-ring index logic of k_agg_march visits exactly the reference's summation order,
//   * the closed-form "sticky d2" + diff-map indexing of k_scanline equals the sequential code,
//   * the two-phase LR check, the fixed-point region voting (with dirty tiles and an arbitrary
//     evaluation order), the Jacobi interpolation and the level-synchronous median equal the
//     reference's in-place raster scans.
// It is NOT a product path: nothing in adcensus_amd/ links it; it mirrors kernel control flow one
// lane at a time and is far too slow for anything but small images.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../adcensus_amd/csrc/adc_device_fn.h"
#include "../../include/adcensus_c_api.h"

extern "C" {

// ------------------------------------------------------------------ colour-difference maps (k_arms.hip)
void emul_color_diffs(const uint8_t* img, uint8_t* dh, uint8_t* dv, int W, int H)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint8_t* p = img + ((size_t)y * W + x) * 3;
            dh[(size_t)y * W + x] = x > 0 ? (uint8_t)adc_color_dist_max(p, p - 3) : 0;
            dv[(size_t)y * W + x] = y > 0 ? (uint8_t)adc_color_dist_max(p, p - (size_t)W * 3) : 0;
        }
}

// ------------------------------------------------------------------ k_agg_march, one lane of one line
// vol layout [H][W][D] (reference layout; a "lane" is one d).  Mirrors the kernel's loop structure:
// prefetch registers, ring slots, 8-wide chunks, epilogue.
// ordered partial sum over cnt consecutive ring entries starting at p (8 / 4 / <4 wide chunks)
static float agg_run(float acc, const float* p, int cnt)
{
    while (cnt >= 8) {
        for (int k = 0; k < 8; k++) acc += p[k];
        p += 8;
        cnt -= 8;
    }
    if (cnt >= 4) {
        for (int k = 0; k < 4; k++) acc += p[k];
        p += 4;
        cnt -= 4;
    }
    if (cnt > 0) {
        acc += p[0];
        if (cnt > 1) acc += p[1];
        if (cnt > 2) acc += p[2];
    }
    return acc;
}

// rec = packed {lo, hi, count16} records of this line (contiguous along m).  Mirrors k_agg_march:
// phase A (warm-up pushes), phase B steady state (register prefetch of PF entries + records, unclamped
// refills while j + 2*PF <= hi), plain tail, phase C (outputs at the image end).
static void agg_line(const float* src, float* dst, const uint32_t* rp, int W, int H, int D, int d, bool vert, bool divide,
                     int fixed, int L, int m0, int m1, int PF)
{
    const int R = 2 * L + 1;
    const int N = vert ? H : W;
    const int lo = std::max(0, m0 - L), hi = std::min(N, m1 + L);
    std::vector<float> ring(R, NAN), pf(PF, NAN), t(PF, NAN);
    std::vector<uint32_t> pr(PF);
    auto pix_of = [&](int m) -> size_t { return vert ? (size_t)m * W + fixed : (size_t)fixed * W + m; };
    int slot_w = 0, slot_m = m0 - lo;
    auto push = [&](float v) { ring[slot_w] = v; slot_w = slot_w + 1 == R ? 0 : slot_w + 1; };
    auto emit = [&](int m, uint32_t r) {
        const int a_lo = r & 255u, a_hi = (r >> 8) & 255u;
        int idx = slot_m - a_lo;
        if (idx < 0) idx += R;
        const int n = a_lo + a_hi + 1;
        const int n1 = std::min(n, R - idx);
        float acc = agg_run(0.0f, ring.data() + idx, n1);
        if (n > n1) acc = agg_run(acc, ring.data(), n - n1);
        if (divide) {
            const uint32_t c = r >> 16;
            if (c != 1u) acc = acc / (float)c;
        }
        dst[pix_of(m) * D + d] = acc;
        slot_m = slot_m + 1 == R ? 0 : slot_m + 1;
    };
    const int jB = std::min(hi, m0 + L);
    for (int j = lo; j < jB; j += PF) {
        for (int u = 0; u < PF; u++) t[u] = src[pix_of(std::min(j + u, jB - 1)) * D + d];
        for (int u = 0; u < PF; u++)
            if (j + u < jB) push(t[u]);
    }
    int j = jB;
    if (j + 2 * PF <= hi) {
        int nxt = j;
        for (int u = 0; u < PF; u++, nxt++) { pf[u] = src[pix_of(nxt) * D + d]; pr[u] = rp[nxt - L]; }
        for (; j + 2 * PF <= hi; j += PF)
            for (int u = 0; u < PF; u++, nxt++) {
                const float v = pf[u];
                const uint32_t r = pr[u];
                pf[u] = src[pix_of(nxt) * D + d];
                pr[u] = rp[nxt - L];
                push(v);
                emit(j + u - L, r);
            }
        for (int u = 0; u < PF; u++) { push(pf[u]); emit(j + u - L, pr[u]); }
        j += PF;
    }
    for (; j < hi; j++) {
        push(src[pix_of(j) * D + d]);
        const int m = j - L;
        if (m >= m0) emit(m, rp[m]);
    }
    for (int m = std::max(m0, hi - L); m < m1; m++) emit(m, rp[m]);
}

void emul_aggregate_pass(const float* src, float* dst, const uint8_t* arms, const uint16_t* sup, int W, int H, int D,
                         int vert, int divide, int L, int nseg, int PF)
{
    const int N = vert ? H : W;
    int seg_len = (N + nseg - 1) / nseg;
    if (seg_len < 1) seg_len = 1;
    nseg = (N + seg_len - 1) / seg_len;
    const int nfixed = vert ? W : H;
    // packed records as k_make_records builds them (`sup` = the divisor map of this pass)
    std::vector<uint32_t> rec((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const uint8_t* a = arms + p * 4;
            if (vert) rec[(size_t)x * H + y] = (uint32_t)a[2] | ((uint32_t)a[3] << 8) | ((uint32_t)sup[p] << 16);
            else rec[p] = (uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)sup[p] << 16);
        }
    for (int seg = 0; seg < nseg; seg++) {
        const int m0 = seg * seg_len, m1 = std::min(N, m0 + seg_len);
        for (int f = 0; f < nfixed; f++)
            for (int d = 0; d < D; d++)
                agg_line(src, dst, rec.data() + (size_t)f * N, W, H, D, d, vert != 0, divide != 0, f, L, m0, m1, PF);
    }
}

// Pass PAIR (k_agg_march<.., PAIR>): the dividing pass of an iteration and the non-dividing first pass of the next one
// run along the same direction and share one launch.  Mirrors the kernel: the first pass's outputs go to a second ring
// (their records to a record ring) instead of memory; when first-pass output m exists, second-pass output m - L is
// summed from the second ring.  A segment [s0, s1) of second-pass outputs needs first-pass outputs [s0-L, s1+L) and
// entries [s0-2L, s1+2L) (clipped to the line).
static void agg_line_pair(const float* src, float* dst, const uint32_t* rp, int W, int H, int D, int d, bool vert, int fixed,
                          int L, int s0, int s1, int PF)
{
    const int R = 2 * L + 1;
    const int N = vert ? H : W;
    const int m0 = std::max(0, s0 - L), m1 = std::min(N, s1 + L);
    const int lo = std::max(0, m0 - L), hi = std::min(N, m1 + L);
    std::vector<float> ring(R, NAN), ring2(R, NAN), pf(PF, NAN), t(PF, NAN);
    std::vector<uint32_t> recring(R, 0), pr(PF);
    auto pix_of = [&](int m) -> size_t { return vert ? (size_t)m * W + fixed : (size_t)fixed * W + m; };
    int slot_w = 0, slot_m = m0 - lo;
    int slot2_w = 0, slot2_s = s0 - m0, mcur = m0, snext = s0;
    auto push = [&](float v) { ring[slot_w] = v; slot_w = slot_w + 1 == R ? 0 : slot_w + 1; };
    auto emit2 = [&]() {
        const uint32_t r2 = recring[slot2_s];
        const int b_lo = r2 & 255u, b_hi = (r2 >> 8) & 255u;
        int i2 = slot2_s - b_lo;
        if (i2 < 0) i2 += R;
        const int k = b_lo + b_hi + 1;
        const int k1 = std::min(k, R - i2);
        float acc2 = agg_run(0.0f, ring2.data() + i2, k1);
        if (k > k1) acc2 = agg_run(acc2, ring2.data(), k - k1);
        dst[pix_of(snext) * D + d] = acc2; // outputs leave in increasing order, starting at s0
        snext++;
        slot2_s = slot2_s + 1 == R ? 0 : slot2_s + 1;
    };
    auto emit = [&](uint32_t r) {
        const int a_lo = r & 255u, a_hi = (r >> 8) & 255u;
        int idx = slot_m - a_lo;
        if (idx < 0) idx += R;
        const int n = a_lo + a_hi + 1;
        const int n1 = std::min(n, R - idx);
        float acc = agg_run(0.0f, ring.data() + idx, n1);
        if (n > n1) acc = agg_run(acc, ring.data(), n - n1);
        const uint32_t c = r >> 16;
        if (c != 1u) acc = acc / (float)c; // the first pass of the pair is the dividing one
        ring2[slot2_w] = acc;
        recring[slot2_w] = r;
        slot2_w = slot2_w + 1 == R ? 0 : slot2_w + 1;
        const int s = mcur - L; // its look-ahead (<= L) is complete now
        mcur++;
        if (s >= s0 && s < s1) emit2();
        slot_m = slot_m + 1 == R ? 0 : slot_m + 1;
    };
    const int jB = std::min(hi, m0 + L);
    for (int j = lo; j < jB; j += PF) {
        for (int u = 0; u < PF; u++) t[u] = src[pix_of(std::min(j + u, jB - 1)) * D + d];
        for (int u = 0; u < PF; u++)
            if (j + u < jB) push(t[u]);
    }
    int j = jB;
    if (j + 2 * PF <= hi) {
        int nxt = j;
        for (int u = 0; u < PF; u++, nxt++) { pf[u] = src[pix_of(nxt) * D + d]; pr[u] = rp[nxt - L]; }
        for (; j + 2 * PF <= hi; j += PF)
            for (int u = 0; u < PF; u++, nxt++) {
                const float v = pf[u];
                const uint32_t r = pr[u];
                pf[u] = src[pix_of(nxt) * D + d];
                pr[u] = rp[nxt - L];
                push(v);
                emit(r);
            }
        for (int u = 0; u < PF; u++) { push(pf[u]); emit(pr[u]); }
        j += PF;
    }
    for (; j < hi; j++) {
        push(src[pix_of(j) * D + d]);
        const int m = j - L;
        if (m >= m0) emit(rp[m]);
    }
    for (int m = std::max(m0, hi - L); m < m1; m++) emit(rp[m]);
    for (int sx = std::max(s0, mcur - L); sx < s1; sx++) emit2(); // look-ahead cut by the end of the line
}

void emul_aggregate_pair(const float* src, float* dst, const uint8_t* arms, const uint16_t* sup, int W, int H, int D, int vert,
                         int L, int nseg, int PF)
{
    const int N = vert ? H : W;
    int seg_len = (N + nseg - 1) / nseg;
    if (seg_len < 1) seg_len = 1;
    nseg = (N + seg_len - 1) / seg_len;
    const int nfixed = vert ? W : H;
    std::vector<uint32_t> rec((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const uint8_t* a = arms + p * 4;
            if (vert) rec[(size_t)x * H + y] = (uint32_t)a[2] | ((uint32_t)a[3] << 8) | ((uint32_t)sup[p] << 16);
            else rec[p] = (uint32_t)a[0] | ((uint32_t)a[1] << 8) | ((uint32_t)sup[p] << 16);
        }
    for (int seg = 0; seg < nseg; seg++) {
        const int s0 = seg * seg_len, s1 = std::min(N, s0 + seg_len);
        for (int f = 0; f < nfixed; f++)
            for (int d = 0; d < D; d++)
                agg_line_pair(src, dst, rec.data() + (size_t)f * N, W, H, D, d, vert != 0, f, L, s0, s1, PF);
    }
}

// Fused matching cost (k_agg_march<.., COSTIN>): model of the lane window.  A wave owns 64 disparities d_first + lane
// of one image row; lane l needs the right-image pixel of column x - d.  Marching in x, that column moves one lane
// per step: the window {bgrx, census} is shifted by one lane (DPP wave_shr:1) and the one new column x - d_first
// enters at lane 0.  Right rows are stored with padl marker records (bgrx = 0xFFFFFFFF -> cost 1.0) in front and
// markers behind, so no bounds logic exists; the window of the entry BEFORE the first one is gathered (clamped) at the
// start of a segment.  cost = A[sad_u8] - C[popcount(census xor)] with the host-built tables.
void emul_cost_window(const uint8_t* img_l, const uint8_t* img_r, const uint64_t* cen_l, const uint64_t* cen_r, float* cost,
                      int W, int H, int dmin, int D, int lambda_ad, int lambda_census, int seg_len)
{
    float A[768] = {0}, C[64];
    for (int k = 0; k <= 765; k++) A[k] = (1.0f - expf(-((float)k / 3.0f) / (float)lambda_ad)) + 1.0f;
    for (int hm = 0; hm < 64; hm++) C[hm] = expf(-(float)hm / (float)lambda_census);
    // the product's padded range (capi.hip): 64 * VPL, VPL in {1,2,4,8,16} -- for 128 < D <= 192 the last chunk is all padding
    const int Dp = D <= 64 ? 64 : (D <= 128 ? 128 : (D <= 256 ? 256 : (D <= 512 ? 512 : 1024)));
    const int padl = (dmin + Dp - 1 > 0 ? dmin + Dp - 1 : 0) + 1; // sized from Dp: padding chunks march too
    const int pitch = padl + W + (dmin < 0 ? -dmin : 0) + 1;
    struct Rec { uint32_t b, c0, c1; };
    auto pack = [](const uint8_t* px) { return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16); };
    auto sad_u8 = [](uint32_t a, uint32_t b) {
        uint32_t s_ = 0;
        for (int i = 0; i < 4; i++) { const int x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255; s_ += x > y ? x - y : y - x; }
        return s_;
    };
    std::vector<Rec> rrow(pitch), lrow(W);
    if (seg_len < 1) seg_len = W;
    for (int y = 0; y < H; y++) {
        for (int i = 0; i < pitch; i++) {
            const int c = i - padl;
            rrow[i] = Rec{0xFFFFFFFFu, 0u, 0u};
            if (c >= 0 && c < W) {
                const size_t p = (size_t)y * W + c;
                rrow[i] = Rec{pack(img_r + 3 * p), (uint32_t)cen_r[p], (uint32_t)(cen_r[p] >> 32)};
            }
        }
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            lrow[x] = Rec{pack(img_l + 3 * p), (uint32_t)cen_l[p], (uint32_t)(cen_l[p] >> 32)};
        }
        for (int chunk = 0; chunk < Dp / 64; chunk++) {
            const int d_first = chunk * 64 + dmin;
            for (int lo = 0; lo < W; lo += seg_len) { // a segment restarts the window (the kernel's entry range [lo, hi))
                const int hi = std::min(W, lo + seg_len);
                Rec win[64];
                for (int lane = 0; lane < 64; lane++) { // window of entry lo - 1
                    int gi = padl - d_first + lo - 1 - lane;
                    gi = gi < 0 ? 0 : (gi >= pitch ? pitch - 1 : gi); // index 0 is a marker column (padl >= 1)
                    win[lane] = rrow[gi];
                }
                for (int x = lo; x < hi; x++) {
                    for (int lane = 63; lane > 0; lane--) win[lane] = win[lane - 1]; // wave_shr:1
                    win[0] = rrow.at(padl - d_first + x);                              // column x - d_first (bounds-checked: the kernel does not clamp)
                    for (int lane = 0; lane < 64; lane++) {
                        const int d = chunk * 64 + lane;
                        if (d >= D) continue; // padding lanes write 0 into the padded volume
                        const uint32_t ad = sad_u8(win[lane].b, lrow[x].b);
                        const uint32_t hm = (uint32_t)__builtin_popcount(win[lane].c0 ^ lrow[x].c0) +
                                            (uint32_t)__builtin_popcount(win[lane].c1 ^ lrow[x].c1);
                        float cv = A[ad < 766u ? ad : 765u] - C[hm & 63u];
                        if (win[lane].b == 0xFFFFFFFFu) cv = 1.0f;
                        cost[((size_t)y * W + x) * D + d] = cv;
                    }
                }
            }
        }
    }
}

// Register ring (k_agg_regring): model of the span addressing of agg_reg_sum.  vgpr[] stands for the wave's VGPR file
// (one lane); the ring occupies vgpr[56 .. 56+R).  A block of 16 adds names the registers v40+p (p = 0..15), the
// hardware adds M0 to the register number; with M0 = idx + c and entry at position 16 - c the adds read
// vgpr[56+idx] .. vgpr[56+idx+c-1] in this order.  Returns the ordered sum and, in *lowest / *highest, the lowest and
// highest register number actually read (must stay inside the ring).
float emul_regring_span(const float* vgpr, int idx, int cnt, int* lowest, int* highest)
{
    float acc = 0.0f;
    *lowest = 1 << 30;
    *highest = -1;
    while (cnt > 0) {
        const int c = cnt < 16 ? cnt : 16;
        const int m0reg = idx + c; // M0[7:0]
        for (int p = 16 - c; p < 16; p++) { // computed jump: skip the first 16 - c adds
            const int reg = 40 + p + m0reg;
            acc += vgpr[reg];
            *lowest = reg < *lowest ? reg : *lowest;
            *highest = reg > *highest ? reg : *highest;
        }
        idx += c;
        cnt -= c;
    }
    return acc;
}

// ------------------------------------------------------------------ k_scanline (per disparity, no lanes)
void emul_scanline_pass(const float* src, float* dst, const uint8_t* cd_left, const uint8_t* cd_right, int W, int H, int dmin,
                        int D, int vert, int dir, int tso, float p1, float p2)
{
    const float P1c[3] = {p1, p1 / 4, p1 / 10}, P2c[3] = {p2, p2 / 4, p2 / 10};
    const int npaths = vert ? W : H, plen = vert ? H : W;
    std::vector<float> Lp(D), out(D);
    for (int path = 0; path < npaths; path++) {
        auto coord = [&](int i, int& x, int& y) {
            const int m = dir > 0 ? i : plen - 1 - i;
            if (vert) { x = path; y = m; } else { x = m; y = path; }
        };
        int x, y;
        coord(0, x, y);
        float minLp = ADC_LARGE_FLOAT;
        for (int d = 0; d < D; d++) {
            const float c = src[((size_t)y * W + x) * D + d];
            dst[((size_t)y * W + x) * D + d] = c;
            Lp[d] = c;
            minLp = c < minLp ? c : minLp;
        }
        for (int i = 1; i < plen; i++) {
            coord(i, x, y);
            const int sx = vert ? x : (dir > 0 ? x : x + 1);
            const int sy = vert ? (dir > 0 ? y : y + 1) : y;
            const int d1 = cd_left[(size_t)sy * W + sx];
            const uint8_t* row = cd_right + (size_t)sy * W;
            const int shift = vert ? 0 : (dir > 0 ? 0 : 1);
            float omin = ADC_LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                const int col = adc_so_d2_column(x, dmin, d, W);
                const int d2 = col >= 0 ? (int)row[col + shift] : d1;
                const int cls = adc_so_penalty_class(d1, d2, tso);
                const float P1 = P1c[cls], P2 = P2c[cls];
                const float lm1 = d > 0 ? Lp[d - 1] : ADC_LARGE_FLOAT;
                const float lp1 = d < D - 1 ? Lp[d + 1] : ADC_LARGE_FLOAT;
                const float l1 = Lp[d], l2 = lm1 + P1, l3 = lp1 + P1, l4 = minLp + P2;
                const float m12 = l2 < l1 ? l2 : l1, m34 = l4 < l3 ? l4 : l3;
                const float mm = m34 < m12 ? m34 : m12;
                float cs = src[((size_t)y * W + x) * D + d] + mm;
                cs = cs / 2;
                out[d] = cs;
                omin = cs < omin ? cs : omin;
            }
            for (int d = 0; d < D; d++) { dst[((size_t)y * W + x) * D + d] = out[d]; Lp[d] = out[d]; }
            minLp = omin;
        }
    }
}

// ------------------------------------------------------------------ scanline paths cut into verified segments
// (tools/so_merge_length.py).  A path of plen elements is cut at nseg - 1 seams (multiples of 16).  Segment k > 0
// starts `warm` elements before its seam with the state of a FIRST element (the raw costs), runs to the seam without storing,
// and its state at the last warm-up element is compared bit for bit with what the predecessor stored there: equal => the
// segment continues from its own state (which is the true one), different => it continues from the predecessor's output (the
// re-run of the real kernel; counted).  Either way the result is the full pass; returns the number of seams that failed.
long emul_scanline_pass_segments(const float* src, float* dst, const uint8_t* cd_left, const uint8_t* cd_right, int W, int H, int dmin,
                                 int D, int vert, int dir, int tso, float p1, float p2, int nseg, int warm)
{
    const float P1c[3] = {p1, p1 / 4, p1 / 10}, P2c[3] = {p2, p2 / 4, p2 / 10};
    const int npaths = vert ? W : H, plen = vert ? H : W;
    std::vector<float> Lp(D), out(D);
    long failed = 0;
    for (int path = 0; path < npaths; path++) {
        auto coord = [&](int i, int& x, int& y) {
            const int m = dir > 0 ? i : plen - 1 - i;
            if (vert) { x = path; y = m; } else { x = m; y = path; }
        };
        auto start = [&](int i, float& minLp, bool store) { // element i as a first element
            int x, y;
            coord(i, x, y);
            minLp = ADC_LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                const float c = src[((size_t)y * W + x) * D + d];
                if (store) dst[((size_t)y * W + x) * D + d] = c;
                Lp[d] = c;
                minLp = c < minLp ? c : minLp;
            }
        };
        auto step = [&](int i, float& minLp, bool store) {
            int x, y;
            coord(i, x, y);
            const int sx = vert ? x : (dir > 0 ? x : x + 1);
            const int sy = vert ? (dir > 0 ? y : y + 1) : y;
            const int d1 = cd_left[(size_t)sy * W + sx];
            const uint8_t* row = cd_right + (size_t)sy * W;
            const int shift = vert ? 0 : (dir > 0 ? 0 : 1);
            float omin = ADC_LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                const int col = adc_so_d2_column(x, dmin, d, W);
                const int d2 = col >= 0 ? (int)row[col + shift] : d1;
                const int cls = adc_so_penalty_class(d1, d2, tso);
                const float P1 = P1c[cls], P2 = P2c[cls];
                const float lm1 = d > 0 ? Lp[d - 1] : ADC_LARGE_FLOAT;
                const float lp1 = d < D - 1 ? Lp[d + 1] : ADC_LARGE_FLOAT;
                const float l1 = Lp[d], l2 = lm1 + P1, l3 = lp1 + P1, l4 = minLp + P2;
                const float m12 = l2 < l1 ? l2 : l1, m34 = l4 < l3 ? l4 : l3;
                const float mm = m34 < m12 ? m34 : m12;
                float cs = src[((size_t)y * W + x) * D + d] + mm;
                cs = cs / 2;
                out[d] = cs;
                omin = cs < omin ? cs : omin;
            }
            for (int d = 0; d < D; d++) { if (store) dst[((size_t)y * W + x) * D + d] = out[d]; Lp[d] = out[d]; }
            minLp = omin;
        };
        std::vector<int> seam(nseg + 1);
        for (int k = 0; k <= nseg; k++) seam[k] = k == nseg ? plen : (int)((long)plen * k / nseg) / 16 * 16;
        for (int k = 0; k < nseg; k++) { // (on the GPU the segments run concurrently; the comparison is what orders them)
            if (seam[k + 1] <= seam[k]) continue;
            float minLp;
            if (k == 0 || seam[k] == 0) start(seam[k], minLp, true);
            else {
                const int s = std::max(seam[k] - warm, 0);
                start(s, minLp, false);
                for (int i = s + 1; i < seam[k]; i++) step(i, minLp, false);
                int x, y;
                coord(seam[k] - 1, x, y);
                const float* truth = dst + ((size_t)y * W + x) * D; // the predecessor's output at the last warm-up element
                if (s > 0 && memcmp(Lp.data(), truth, D * sizeof(float))) {
                    failed++;
                    minLp = ADC_LARGE_FLOAT;
                    for (int d = 0; d < D; d++) { Lp[d] = truth[d]; minLp = Lp[d] < minLp ? Lp[d] : minLp; }
                }
                step(seam[k], minLp, true);
            }
            for (int i = seam[k] + 1; i < seam[k + 1]; i++) step(i, minLp, true);
        }
    }
    return failed;
}

// The same pass in the form the kernel runs it: segment bounds from adc_so_seg_start, warm + 1 elements of overlap, the seam
// check behind the pass; a failed seam is only COUNTED (the product then redoes the Match with whole rows).
int emul_so_seg_ok(int plen, int nseg, int warm) { return adc_so_seg_ok(plen, nseg, warm) ? 1 : 0; }
int emul_so_seg_start(int plen, int nseg, int warm, int s) { return adc_so_seg_start(plen, nseg, warm, s); }
long emul_scanline_pass_kernel_segments(const float* src, float* dst, const uint8_t* cd_left, const uint8_t* cd_right, int W, int H, int dmin,
                                 int D, int vert, int dir, int tso, float p1, float p2, int nseg, int warm)
{
    const float P1c[3] = {p1, p1 / 4, p1 / 10}, P2c[3] = {p2, p2 / 4, p2 / 10};
    const int npaths = vert ? W : H, plen = vert ? H : W;
    std::vector<float> Lp(D), out(D);
    long failed = 0;
    std::vector<std::vector<float>> slots;
    std::vector<size_t> slot_pix;
    for (int path = 0; path < npaths; path++) {
        auto coord = [&](int i, int& x, int& y) {
            const int m = dir > 0 ? i : plen - 1 - i;
            if (vert) { x = path; y = m; } else { x = m; y = path; }
        };
        auto start = [&](int i, float& minLp, bool store) { // element i as a first element
            int x, y;
            coord(i, x, y);
            minLp = ADC_LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                const float c = src[((size_t)y * W + x) * D + d];
                if (store) dst[((size_t)y * W + x) * D + d] = c;
                Lp[d] = c;
                minLp = c < minLp ? c : minLp;
            }
        };
        auto step = [&](int i, float& minLp, bool store) {
            int x, y;
            coord(i, x, y);
            const int sx = vert ? x : (dir > 0 ? x : x + 1);
            const int sy = vert ? (dir > 0 ? y : y + 1) : y;
            const int d1 = cd_left[(size_t)sy * W + sx];
            const uint8_t* row = cd_right + (size_t)sy * W;
            const int shift = vert ? 0 : (dir > 0 ? 0 : 1);
            float omin = ADC_LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                const int col = adc_so_d2_column(x, dmin, d, W);
                const int d2 = col >= 0 ? (int)row[col + shift] : d1;
                const int cls = adc_so_penalty_class(d1, d2, tso);
                const float P1 = P1c[cls], P2 = P2c[cls];
                const float lm1 = d > 0 ? Lp[d - 1] : ADC_LARGE_FLOAT;
                const float lp1 = d < D - 1 ? Lp[d + 1] : ADC_LARGE_FLOAT;
                const float l1 = Lp[d], l2 = lm1 + P1, l3 = lp1 + P1, l4 = minLp + P2;
                const float m12 = l2 < l1 ? l2 : l1, m34 = l4 < l3 ? l4 : l3;
                const float mm = m34 < m12 ? m34 : m12;
                float cs = src[((size_t)y * W + x) * D + d] + mm;
                cs = cs / 2;
                out[d] = cs;
                omin = cs < omin ? cs : omin;
            }
            for (int d = 0; d < D; d++) { if (store) dst[((size_t)y * W + x) * D + d] = out[d]; Lp[d] = out[d]; }
            minLp = omin;
        };
        // the kernel's form (k_scanline.hip, SEG): segment s visits the elements [e0, e1) with e0 = a - warm - 1 for s > 0, the
        // first of them as a FIRST element; steps 1 .. warm (relative to e0) store to the seam slot, the later ones to the volume;
        // k_so_seam_check compares the slot with the predecessor's output at element a - 1 (only d < D)
        std::vector<float> slot(D);
        for (int k = 0; k < nseg; k++) {
            const int a = adc_so_seg_start(plen, nseg, warm, k), e1 = adc_so_seg_start(plen, nseg, warm, k + 1);
            const int e0 = k == 0 ? 0 : a - warm - 1;
            if (e0 < 0 || (e0 & 3) || e1 - e0 < 2) return -1;
            float minLp;
            start(e0, minLp, k == 0);
            for (int i = e0 + 1; i < e1; i++) {
                const bool real = k == 0 || i - e0 > warm;
                step(i, minLp, real);
                if (!real) slot = Lp; // (the warm-up steps overwrite the slot; the last one stays)
            }
            if (k > 0) {
                int x, y;
                coord(a - 1, x, y);
                slots.push_back(slot);
                slot_pix.push_back((size_t)y * W + x);
            }
        }
    }
    // the seam check runs behind the pass (all segments of a pass run concurrently)
    for (size_t q = 0; q < slots.size(); q++)
        if (memcmp(slots[q].data(), dst + slot_pix[q] * D, D * sizeof(float))) failed++;
    return failed;
}

// ------------------------------------------------------------------ k_scanline, lane structure of the penalty classes
// Same DP, but the (P1,P2) class of every disparity is derived the way the kernel does it: lane l owns VPL consecutive
// disparities, fetches VPL consecutive bytes of the right-image step map at column max(xr_last, 1) (+1 on R->L) and
// maps them with adc_so_class_offsets (closed form of the sticky-d2 rule).
} // extern "C"
// chunked != 0: additionally follows the control flow of the asm-prefetch kernels (k_scanline / k_scanline_pin, VPL <= 2):
// element e >= 1 is prefetched PF = 16 steps ahead by the chunk that stands on e - PF (prologue / first chunk: clamped form);
// a steady-state chunk that adc_so_chunk_interior accepts (and D == 64 * VPL) computes the rmap offsets of the elements it
// prefetches from ONE clamped offset plus a running step, and runs the interior class rule on its own elements without the
// per-step test.  Returns the number of chunks that took the short form.
template <int VPL>
static int scanline_pass_lanes(const float* src, float* dst, const uint8_t* cd_left, const uint8_t* cd_right, int W, int H,
                               int dmin, int D, int vert, int dir, int tso, float p1, float p2, int chunked = 0)
{
    constexpr int PF = 16;
    int short_chunks = 0;
    const float P1c[3] = {p1, p1 / 4, p1 / 10}, P2c[3] = {p2, p2 / 4, p2 / 10};
    const int npaths = vert ? W : H, plen = vert ? H : W, Dp = 64 * VPL;
    std::vector<uint8_t> rmap((size_t)W * H + 64, 0); // the kernel's map has slack behind the last element
    std::copy(cd_right, cd_right + (size_t)W * H, rmap.begin());
    std::vector<float> Lp(Dp), out(Dp);
    std::vector<int> cls(Dp);
    for (int path = 0; path < npaths; path++) {
        auto coord = [&](int i, int& x, int& y) {
            const int m = dir > 0 ? i : plen - 1 - i;
            if (vert) { x = path; y = m; } else { x = m; y = path; }
        };
        int x, y;
        coord(0, x, y);
        float minLp = ADC_LARGE_FLOAT;
        for (int d = 0; d < Dp; d++) {
            const float c = d < D ? src[((size_t)y * W + x) * D + d] : ADC_LARGE_FLOAT;
            if (d < D) dst[((size_t)y * W + x) * D + d] = c;
            Lp[d] = c;
            minLp = c < minLp ? c : minLp;
        }
        for (int i = 1; i < plen; i++) {
            coord(i, x, y);
            const int sx = vert ? x : (dir > 0 ? x : x + 1);
            const int sy = vert ? (dir > 0 ? y : y + 1) : y;
            const int d1 = cd_left[(size_t)sy * W + sx];
            const int shift = vert ? 0 : (dir > 0 ? 0 : 1);
            // chunk of the steady state this element belongs to / was prefetched by (see above)
            const int Dp_ = 64 * VPL;
            auto chunk_short = [&](int i0) {
                return chunked && VPL <= 2 && D == Dp_ && i0 >= 1 + PF && i0 + PF <= plen &&
                       adc_so_chunk_interior(i0, PF, plen, dir, vert != 0, path, W, dmin, Dp_);
            };
            const int i_own = i - ((i - 1) % PF);                               // first element of the chunk that steps on i
            const int i_pre = i > PF ? (i - PF) - ((i - PF - 1) % PF) : 0;      // ... of the chunk that prefetched i
            const bool own_short = chunk_short(i_own), pre_short = i_pre >= 1 + PF && chunk_short(i_pre);
            if (own_short && i == i_own) short_chunks++;
            for (int lane = 0; lane < 64; lane++) {
                const int cl_last = lane * VPL + VPL - 1 + dmin, xr_last = x - cl_last;
                size_t off = (size_t)sy * W + (xr_last > 1 ? xr_last : 1) + shift;
                if (chunked && VPL <= 2) {
                    const int m = dir > 0 ? i : plen - 1 - i;
                    if (pre_short) { // running offset: the clamped offset of the chunk's first prefetch + steps
                        const int e0 = i_pre + PF, m0 = dir > 0 ? e0 : plen - 1 - e0;
                        off = (size_t)(adc_so_rmap_offset(W, vert != 0, dir, path, m0, cl_last) + (i - e0) * ((vert ? W : 1) * dir));
                    } else
                        off = (size_t)adc_so_rmap_offset(W, vert != 0, dir, path, m, cl_last);
                }
                uint32_t rb[(VPL + 3) / 4] = {0};
                for (int j = 0; j < VPL; j++) rb[j >> 2] |= (uint32_t)rmap[off + j] << (8 * (j & 3));
                int o8[VPL];
                // the kernel takes the interior form whenever the wave-uniform test allows it
                const int Dpad = (D + VPL - 1) / VPL * VPL;
                if (own_short || adc_so_interior(x, W, dmin, Dpad)) adc_so_class_offsets_interior<VPL>(rb, d1, tso, o8);
                else adc_so_class_offsets<VPL>(rb, d1, xr_last, W, tso, W >= 3 && x - dmin >= 1, o8);
                for (int k = 0; k < VPL; k++) cls[lane * VPL + k] = o8[k] / 8;
            }
            float omin = ADC_LARGE_FLOAT;
            for (int d = 0; d < Dp; d++) {
                const float P1 = P1c[cls[d]], P2 = P2c[cls[d]];
                const float lm1 = d > 0 ? Lp[d - 1] : ADC_LARGE_FLOAT;
                const float lp1 = d < Dp - 1 ? Lp[d + 1] : ADC_LARGE_FLOAT;
                const float l1 = Lp[d], l2 = lm1 + P1, l3 = lp1 + P1, l4 = minLp + P2;
                const float m12 = l2 < l1 ? l2 : l1, m123 = l3 < m12 ? l3 : m12, mm = l4 < m123 ? l4 : m123;
                const float cs = ((d < D ? src[((size_t)y * W + x) * D + d] : 0.0f) + mm) * 0.5f;
                out[d] = d < D ? cs : ADC_LARGE_FLOAT; // padding lanes hold the sentinel
                omin = out[d] < omin ? out[d] : omin;
            }
            for (int d = 0; d < Dp; d++) { if (d < D) dst[((size_t)y * W + x) * D + d] = out[d]; Lp[d] = out[d]; }
            minLp = omin;
        }
    }
    return short_chunks;
}
extern "C" {
// exhaustive check of what the short form of a chunk relies on (tests/test_emul.py); returns the number of violations
int emul_so_chunk_predicate_check(void)
{
    int bad = 0;
    const int PF = 16;
    for (int VPL = 1; VPL <= 2; VPL++)
        for (int vert = 0; vert <= 1; vert++)
            for (int dir = -1; dir <= 1; dir += 2)
                for (int dmin = -70; dmin <= 40; dmin += 11)
                    for (int W = 1; W <= 260; W += (W < 8 ? 1 : 21))
                        for (int H = 1; H <= 120; H += (H < 4 ? 1 : 29)) {
                            const int Dp = 64 * VPL, plen = vert ? H : W, npaths = vert ? W : H;
                            const int ngr = (plen - 1 + 3) / 4 > 0 ? (plen - 1 + 3) / 4 : 1;
                            for (int path = 0; path < npaths; path += (npaths > 40 ? 7 : 1))
                                for (int i = 1 + PF; i + PF <= plen; i += PF) {
                                    if (!adc_so_chunk_interior(i, PF, plen, dir, vert != 0, path, W, dmin, Dp)) continue;
                                    for (int e = i; e < i + PF; e++) { // (i) the per-step test agrees
                                        const int m = dir > 0 ? e : plen - 1 - e, x = vert ? path : m;
                                        if (!adc_so_interior(x, W, dmin, Dp)) bad++;
                                    }
                                    if (i + 2 * PF - 1 > plen - 1) bad++; // prefetches stay inside the path
                                    if ((i - 1) / 4 + 2 * (PF / 4) - 1 > ngr - 1) bad++; // (iii) groups (i-1)/4 + 4 .. + 7 exist
                                    const int e0 = i + PF, m0 = dir > 0 ? e0 : plen - 1 - e0, rstep = (vert ? W : 1) * dir;
                                    for (int lane = 0; lane < 64; lane++) { // (ii) affine offsets, no clamp
                                        const int cl_last = lane * VPL + VPL - 1 + dmin;
                                        const int o0 = adc_so_rmap_offset(W, vert != 0, dir, path, m0, cl_last);
                                        for (int e = e0; e < e0 + PF; e++) {
                                            const int m = dir > 0 ? e : plen - 1 - e, x = vert ? path : m;
                                            if (adc_so_rmap_offset(W, vert != 0, dir, path, m, cl_last) != o0 + (e - e0) * rstep) bad++;
                                            const int xr = x - cl_last;
                                            if (xr < 1 || xr > W - 1) bad++;
                                        }
                                    }
                                }
                        }
    return bad;
}
int emul_scanline_pass_chunked(const float* src, float* dst, const uint8_t* cd_left, const uint8_t* cd_right, int W, int H,
                               int dmin, int D, int vert, int dir, int tso, float p1, float p2)
{
    if (D <= 64) return scanline_pass_lanes<1>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2, 1);
    if (D <= 128) return scanline_pass_lanes<2>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2, 1);
    return -1; // wider lanes have no asm-prefetch form
}
void emul_scanline_pass_lanes(const float* src, float* dst, const uint8_t* cd_left, const uint8_t* cd_right, int W, int H,
                              int dmin, int D, int vert, int dir, int tso, float p1, float p2)
{
    if (D <= 64) scanline_pass_lanes<1>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2);
    else if (D <= 128) scanline_pass_lanes<2>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2);
    else if (D <= 256) scanline_pass_lanes<4>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2);
    else if (D <= 512) scanline_pass_lanes<8>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2);
    else scanline_pass_lanes<16>(src, dst, cd_left, cd_right, W, H, dmin, D, vert, dir, tso, p1, p2);
}

// ------------------------------------------------------------------ k_wta_right_band
// A "wave" owns 64 consecutive right pixels; lane l walks the columns base + dmin + t, looking at disparity index t - l,
// with the reference's own sequential scan (strict '<', neighbours of the running minimum remembered).
void emul_wta_right_band(const float* vol, float* disp, int W, int H, int dmin, int D)
{
    for (int y = 0; y < H; y++)
        for (int base = 0; base < W; base += 64)
            for (int lane = 0; lane < 64; lane++) {
                const int x = base + lane;
                float minc = ADC_LARGE_FLOAT, prev = 0.f, c1 = 0.f, c2 = 0.f;
                int best = 0;
                bool capture = false;
                for (int t = 0; t < 63 + D; t++) {
                    const int c = base + dmin + t, di = t - lane;
                    if (di < 0 || di >= D) continue;
                    const float cost = (c >= 0 && c < W) ? vol[((size_t)y * W + c) * D + di] : ADC_LARGE_FLOAT;
                    if (capture) { c2 = cost; capture = false; }
                    if (cost < minc) { minc = cost; best = di + dmin; c1 = prev; capture = true; }
                    prev = cost;
                }
                if (x >= W) continue;
                float out;
                if (best == dmin || best == dmin + D - 1) out = (float)best;
                else if (best - 1 - dmin < 0 || best + 1 - dmin >= D) out = (float)best;
                else out = adc_subpixel(best, c1, c2, minc);
                disp[(size_t)y * W + x] = out;
            }
}

// ------------------------------------------------------------------ k_wta_right_march
// The launch plan (adc_wtam_plan / adc_wtam_unit: whole rows, the remainder cut into segments), the ring of 256 pixel vectors
// with a pitch of Dp + 1, steps of 64 vectors written by 8 "waves" of 8 vectors each, group s - lag scanned in 8 parts of
// Dp / 8 disparities (strict '<' inside a part), the parts combined in increasing d (strict '<'), the winner's neighbours read
// from the ring.  vol has a pitch of D floats per pixel (the oracle's layout).  Returns the number of units, -1 for D > 128.
long emul_wta_right_march(const float* vol, float* disp, int W, int H, int dmin, int D, int ncu, int force_nseg)
{
    if (D > 128) return -1;
    const int Dp = D <= 64 ? 64 : 128, P = Dp + 1, Dq = Dp / 8, R = ADC_WTAM_RING;
    const AdcWtamPlan pl = adc_wtam_plan(W, H, D, ncu, force_nseg);
    std::vector<float> ring((size_t)R * P);
    std::vector<char> covered((size_t)W * H, 0);
    for (int u = 0; u < pl.units; u++) {
        const AdcWtamUnit un = adc_wtam_unit(u, W, pl.rows_full, pl.nseg, pl.segw);
        if (un.x0 >= un.x1) continue;
        const int G = (un.x1 - un.x0 + 63) >> 6, lag = adc_wtam_lag(D), last = G - 1 + lag;
        std::fill(ring.begin(), ring.end(), -12345.0f); // (what was never written must never decide)
        for (int s = 0; s <= (last | 1); s++) {
            for (int wave = 0; wave < 8; wave++)
                for (int j = 0; j < 8; j++) {
                    const int i = s * 64 + wave * 8 + j, c = un.x0 + dmin + i;
                    for (int d = 0; d < Dp; d++)
                        ring[(size_t)(i & (R - 1)) * P + d] = (c >= 0 && c < W) ? (d < D ? vol[((size_t)un.y * W + c) * D + d] : 7.0f) : ADC_LARGE_FLOAT;
                }
            const int g = s - lag;
            if (g < 0 || g >= G) continue;
            for (int lane = 0; lane < 64; lane++) {
                const int b = g * 64 + lane;
                float pmin[8];
                int pbest[8];
                for (int wave = 0; wave < 8; wave++) {
                    float mc = ADC_LARGE_FLOAT;
                    int mb = -1;
                    for (int t = 0; t < Dq; t++) {
                        const int di = wave * Dq + t;
                        const float v = ring[(size_t)((b + di) & (R - 1)) * P + di];
                        const float cost = di < D ? v : ADC_LARGE_FLOAT;
                        if (cost < mc) { mc = cost; mb = di; }
                    }
                    pmin[wave] = mc;
                    pbest[wave] = mb;
                }
                float minc = ADC_LARGE_FLOAT;
                int bi = -1;
                for (int q = 0; q < 8; q++)
                    if (pmin[q] < minc) { minc = pmin[q]; bi = pbest[q]; }
                const int x = un.x0 + b;
                if (x >= un.x1) continue;
                const int best = bi < 0 ? 0 : bi + dmin, i1 = best - 1 - dmin, i2 = best + 1 - dmin;
                float out = (float)best;
                if (best != dmin && best != dmin + D - 1 && i1 >= 0 && i2 < D)
                    out = adc_subpixel(best, ring[(size_t)((b + i1) & (R - 1)) * P + i1], ring[(size_t)((b + i2) & (R - 1)) * P + i2], minc);
                disp[(size_t)un.y * W + x] = out;
                covered[(size_t)un.y * W + x]++;
            }
        }
    }
    for (char c : covered)
        if (c != 1) return -2; // every right pixel exactly once
    return pl.units;
}

// ------------------------------------------------------------------ k_wta
void emul_wta(const float* vol, float* disp, int W, int H, int dmin, int D, int right)
{
    const int dmax = dmin + D;
    std::vector<float> c(D);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float bc = ADC_LARGE_FLOAT;
            int bd = 0x7fffffff;
            for (int di = 0; di < D; di++) {
                float v = ADC_LARGE_FLOAT;
                bool cand = false;
                if (!right) { v = vol[((size_t)y * W + x) * D + di]; cand = true; }
                else {
                    const int col = x + di + dmin;
                    if (col >= 0 && col < W) { v = vol[((size_t)y * W + col) * D + di]; cand = true; }
                }
                c[di] = v;
                // lexicographic (cost, d) minimum among candidates strictly below Large_Float
                if (cand && v < ADC_LARGE_FLOAT && (v < bc || (v == bc && di + dmin < bd))) { bc = v; bd = di + dmin; }
            }
            int best = bd == 0x7fffffff ? 0 : bd;
            float out;
            if (best == dmin || best == dmax - 1) out = right ? (float)best : ADC_INVALID_FLOAT;
            else if (best - 1 - dmin < 0 || best + 1 - dmin >= D) out = (float)best;
            else out = adc_subpixel(best, c[best - 1 - dmin], c[best + 1 - dmin], bc);
            disp[(size_t)y * W + x] = out;
        }
}

// ------------------------------------------------------------------ k_lr_phase1/2
static bool lr_invalid(const float* dl, const float* dr, int W, int x, int y, float thres, int& col_right, float& disp_r)
{
    const float d = dl[(size_t)y * W + x];
    col_right = -1;
    disp_r = 0.f;
    if (d == ADC_INVALID_FLOAT) return true;
    const long cr = lroundf((float)x - d);
    if (cr < 0 || cr >= W) return true;
    col_right = (int)cr;
    disp_r = dr[(size_t)y * W + cr];
    return fabsf(d - disp_r) > thres;
}

void emul_lrcheck(const float* dl, const float* dr, float* out, uint8_t* label, int W, int H, float thres)
{
    std::vector<uint8_t> inv((size_t)W * H);
    int cr;
    float drv;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) inv[(size_t)y * W + x] = lr_invalid(dl, dr, W, x, y, thres, cr, drv) ? 1 : 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const float d = dl[p];
            float disp_r;
            const bool bad = lr_invalid(dl, dr, W, x, y, thres, cr, disp_r);
            uint8_t lab = ADC_LABEL_VALID;
            if (bad) {
                lab = ADC_LABEL_MISMATCH;
                if (d != ADC_INVALID_FLOAT && cr >= 0) {
                    const long col_rl = lroundf((float)cr + disp_r);
                    if (col_rl > 0 && col_rl < W) {
                        const float disp_l = (col_rl < x && inv[(size_t)y * W + col_rl]) ? ADC_INVALID_FLOAT : dl[(size_t)y * W + col_rl];
                        if (disp_l > d) lab = ADC_LABEL_OCCLUSION;
                    }
                }
            }
            label[p] = lab;
            out[p] = bad ? ADC_INVALID_FLOAT : d;
        }
}

// ------------------------------------------------------------------ k_irv_begin / k_irv_round
// Evaluation order inside a round is shuffled (seeded) to model arbitrary wave scheduling with
// in-place (chaotic) updates.  Returns total rounds; *evals gets the number of vote evaluations.
long emul_region_voting(float* disp, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D, int irv_ts,
                        float irv_th, int Lmax, unsigned seed, long* evals_out)
{
    const int P = W * H, T = 8;
    (void)Lmax;
    // dependency box per pixel (k_irv_bbox): rows y-top..y, widest H arm of those rows
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
    std::vector<uint8_t> elig(P), chg_a(tiles_x * tiles_y, 0), chg_b(tiles_x * tiles_y, 0);
    std::vector<int> list, hist(D);
    long rounds = 0, evals = 0;
    srand(seed);
    for (int it = 0; it < 5; it++)
        for (int k = 0; k < 2; k++) {
            const int which = k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION;
            list.clear();
            for (int p = 0; p < P; p++) {
                elig[p] = (label[p] == which && disp[p] == ADC_INVALID_FLOAT) ? 1 : 0;
                if (elig[p]) list.push_back(p);
            }
            if (list.empty()) continue;
            for (int round = 0;; round++) {
                std::fill(chg_b.begin(), chg_b.end(), 0);
                bool changed = false;
                for (size_t i = list.size(); i > 1; i--) std::swap(list[i - 1], list[rand() % i]);
                for (int p : list) {
                    const int y = p / W, x = p - y * W;
                    if (round > 0) {
                        const uint8_t* o = &bb[(size_t)p * 3];
                        const int tx0 = std::max(0, x - (int)o[1]) / T, tx1 = std::min(W - 1, x + (int)o[2]) / T;
                        const int ty0 = std::max(0, y - (int)o[0]) / T, ty1 = y / T;
                        bool dirty = false;
                        for (int ty = ty0; ty <= ty1; ty++)
                            for (int tx = tx0; tx <= tx1; tx++) dirty |= chg_a[ty * tiles_x + tx] != 0;
                        if (!dirty) continue;
                    }
                    std::fill(hist.begin(), hist.end(), 0);
                    const uint8_t* arm = arms + (size_t)p * 4;
                    for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
                        const int yt = y + t;
                        const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
                        for (int s = -(int)arm2[0]; s <= (int)arm2[1]; s++) {
                            const int q = yt * W + x + s;
                            float v = disp[q];
                            if (elig[q] && q >= p) v = ADC_INVALID_FLOAT;
                            if (v != ADC_INVALID_FLOAT) {
                                const long b = lroundf(v) - dmin;
                                if (b >= 0 && b < D) hist[b]++;
                            }
                        }
                    }
                    int bh = 0, bb = 0x7fffffff, cnt = 0;
                    for (int b = 0; b < D; b++) {
                        cnt += hist[b];
                        if (hist[b] > bh) { bh = hist[b]; bb = b; }
                    }
                    const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
                    evals++;
                    uint32_t a, b2;
                    memcpy(&a, &disp[p], 4);
                    memcpy(&b2, &nv, 4);
                    if (a != b2) {
                        disp[p] = nv;
                        chg_b[(y / T) * tiles_x + x / T] = 1;
                        changed = true;
                    }
                }
                rounds++;
                chg_a.swap(chg_b);
                if (!changed) break;
            }
        }
    if (evals_out) *evals_out = evals;
    return rounds;
}

// ------------------------------------------------------------------ k_interpolate (one list)
void emul_interpolate(const float* din, float* dout, const uint8_t* label, const uint8_t* img_l, int W, int H, int which,
                      int max_search)
{
    double sc[32];
    const float pi = 3.1415926f;
    double ang = 0.0;
    for (int s = 0; s < 16; s++) { sc[2 * s] = sin(ang); sc[2 * s + 1] = cos(ang); ang += pi / 16; }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const float d0 = din[p];
            if (!(label[p] == which && d0 == ADC_INVALID_FLOAT)) { dout[p] = d0; continue; }
            const uint8_t* c0 = img_l + p * 3;
            const bool mismatch = which == ADC_LABEL_MISMATCH;
            int min_dist = 9999;
            float best = mismatch ? 0.0f : ADC_LARGE_FLOAT;
            bool any = false;
            for (int s = 0; s < 16; s++) {
                const double sina = sc[2 * s], cosa = sc[2 * s + 1];
                for (int m = 1; m < max_search; m++) {
                    const int yy = (int)lround((double)y + (double)m * sina);
                    const int xx = (int)lround((double)x + (double)m * cosa);
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) break;
                    const float d = din[(size_t)yy * W + xx];
                    if (d != ADC_INVALID_FLOAT) {
                        any = true;
                        if (mismatch) {
                            const int dist = adc_color_dist_l1(c0, img_l + ((size_t)yy * W + xx) * 3);
                            if (min_dist > dist) { min_dist = dist; best = d; }
                        } else best = d < best ? d : best;
                        break;
                    }
                }
            }
            dout[p] = any ? best : 0.0f;
        }
}

// ------------------------------------------------------------------ k_interpolate_tab with empty-space skipping
// The list kernel's walk: per ray its own step counter m, trips of NS steps whose map values are requested together, a
// skip of adc_itp_skip(cell distance) steps at the start and after every trip (cell distance at the LAST position of the
// trip), on the cell maps built by the same three passes as k_itp_cells / k_itp_rowdist / k_itp_coldist.  Must equal
// emul_interpolate (the plain walk) on any input.  Returns the number of map look-ups (the plain walk's count goes to
// *plain_lookups) so that a test can also see that steps were really skipped.
long emul_interpolate_skip(const float* din, float* dout, const uint8_t* label, const uint8_t* img_l, int W, int H, int which,
                           int max_search, int NS, long* plain_lookups)
{
    double sc[32];
    const float pi = 3.1415926f;
    double ang = 0.0;
    for (int s = 0; s < 16; s++) { sc[2 * s] = sin(ang); sc[2 * s + 1] = cos(ang); ang += pi / 16; }
    const int cw = (W + ADC_ITP_CELL - 1) / ADC_ITP_CELL, ch = (H + ADC_ITP_CELL - 1) / ADC_ITP_CELL;
    std::vector<uint8_t> cell((size_t)cw * ch), rowd((size_t)cw * ch), cdist((size_t)cw * ch);
    for (int cy = 0; cy < ch; cy++)
        for (int cx = 0; cx < cw; cx++) {
            bool any = false;
            for (int r = 0; r < ADC_ITP_CELL; r++)
                for (int q = 0; q < ADC_ITP_CELL; q++) {
                    const int y = cy * ADC_ITP_CELL + r, x = cx * ADC_ITP_CELL + q;
                    if (y < H && x < W) any = any || din[(size_t)y * W + x] != ADC_INVALID_FLOAT;
                }
            cell[(size_t)cy * cw + cx] = any ? 1 : 0;
        }
    for (int c = 0; c < cw * ch; c++) rowd[c] = (uint8_t)adc_itp_rowdist(cell.data(), cw, c % cw, c / cw);
    for (int c = 0; c < cw * ch; c++) cdist[c] = (uint8_t)adc_itp_coldist(rowd.data(), cw, ch, c % cw, c / cw);
    long lookups = 0, plain = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const float d0 = din[p];
            if (!(label[p] == which && d0 == ADC_INVALID_FLOAT)) { dout[p] = d0; continue; }
            const uint8_t* c0 = img_l + p * 3;
            const bool mismatch = which == ADC_LABEL_MISMATCH;
            int min_dist = 9999;
            float best = mismatch ? 0.0f : ADC_LARGE_FLOAT;
            bool any = false;
            for (int s = 0; s < 16; s++) {
                const double sina = sc[2 * s], cosa = sc[2 * s + 1];
                for (int m = 1; m < max_search; m++) { // (look-ups of the plain walk, for the statistics only)
                    const int yy = (int)lround((double)y + (double)m * sina), xx = (int)lround((double)x + (double)m * cosa);
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) break;
                    plain++;
                    if (din[(size_t)yy * W + xx] != ADC_INVALID_FLOAT) break;
                }
                bool walking = true;
                float hit = ADC_INVALID_FLOAT;
                size_t hitq = p;
                int m = 1 + adc_itp_skip(cdist[(size_t)(y / ADC_ITP_CELL) * cw + x / ADC_ITP_CELL]);
                while (walking && m < max_search) {
                    int cl = 0;
                    for (int j = 0; j < NS && walking; j++) {
                        if (m + j >= max_search) { walking = false; break; }
                        const int yy = (int)lround((double)y + (double)(m + j) * sina), xx = (int)lround((double)x + (double)(m + j) * cosa);
                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) { walking = false; break; }
                        lookups++;
                        const float d = din[(size_t)yy * W + xx];
                        if (d != ADC_INVALID_FLOAT) { hit = d; hitq = (size_t)yy * W + xx; walking = false; break; }
                        if (j == NS - 1) cl = cdist[(size_t)(yy / ADC_ITP_CELL) * cw + xx / ADC_ITP_CELL];
                    }
                    m += NS + adc_itp_skip(cl);
                }
                if (hit != ADC_INVALID_FLOAT) {
                    any = true;
                    if (mismatch) {
                        const int dist = adc_color_dist_l1(c0, img_l + hitq * 3);
                        if (min_dist > dist) { min_dist = dist; best = hit; }
                    } else best = hit < best ? hit : best;
                }
            }
            dout[p] = any ? best : 0.0f;
        }
    if (plain_lookups) *plain_lookups = plain;
    return lookups;
}

// ------------------------------------------------------------------ k_interpolate_tab on the code map (round 6)
// The list kernel's walk as k_refine.hip runs it since round 6: ONE byte map of the image padded by the search range on the left, on
// the right and below (adc_device_fn.h: ADC_ITP_VALID / ADC_ITP_OUTSIDE / skip of the pixel's cell), linear ray offsets, trips of
// ns1 (first) / ns2 (following) steps without bounds tests, the skip taken from the code of the trip's last position.  The map and the table are built
// the way k_itp_cells .. k_itp_code and upload_tables (capi.hip) build them.  Must equal emul_interpolate (the plain walk) on any
// input.  Returns the number of map look-ups (< 0: an internal inconsistency).
long emul_interpolate_code(const float* din, float* dout, const uint8_t* label, const uint8_t* img_l, int W, int H, int which,
                           int max_search, int ns1, int ns2)
{
    const int ms = max_search;
    if (ns1 < 1 || ns2 < 1 || ns1 > ADC_ITP_NS || ns2 > ADC_ITP_NS) return -6;
    double sc[32];
    const float pi = 3.1415926f;
    double ang = 0.0;
    for (int s = 0; s < 16; s++) { sc[2 * s] = sin(ang); sc[2 * s + 1] = cos(ang); ang += pi / 16; }
    const int cw = (W + ADC_ITP_CELL - 1) / ADC_ITP_CELL, ch = (H + ADC_ITP_CELL - 1) / ADC_ITP_CELL;
    std::vector<uint8_t> cell((size_t)cw * ch), rowd((size_t)cw * ch), cdist((size_t)cw * ch);
    for (int cy = 0; cy < ch; cy++)
        for (int cx = 0; cx < cw; cx++) {
            bool any = false;
            for (int r = 0; r < ADC_ITP_CELL; r++)
                for (int q = 0; q < ADC_ITP_CELL; q++) {
                    const int y = cy * ADC_ITP_CELL + r, x = cx * ADC_ITP_CELL + q;
                    if (y < H && x < W) any = any || din[(size_t)y * W + x] != ADC_INVALID_FLOAT;
                }
            cell[(size_t)cy * cw + cx] = any ? 1 : 0;
        }
    for (int c = 0; c < cw * ch; c++) rowd[c] = (uint8_t)adc_itp_rowdist(cell.data(), cw, c % cw, c / cw);
    for (int c = 0; c < cw * ch; c++) cdist[c] = (uint8_t)adc_itp_coldist(rowd.data(), cw, ch, c % cw, c / cw);
    // the code map: everything ADC_ITP_OUTSIDE (adc_create), then one dword per 4 padded columns around the image's own (k_itp_code)
    const int pitch = adc_itp_code_pitch(W, ms), rows = adc_itp_code_rows(H, ms), gx = ms;
    std::vector<uint8_t> code((size_t)pitch * rows + 64, (uint8_t)ADC_ITP_OUTSIDE);
    const int q4 = (W + 3) >> 2;
    for (int y = 0; y < H; y++)
        for (int k = 0; k <= q4; k++) {
            const int c0 = (gx & ~3) + 4 * k;
            if (c0 + 3 >= pitch) return -4;
            for (int b = 0; b < 4; b++) {
                const int x = c0 + b - gx;
                uint8_t c = ADC_ITP_OUTSIDE;
                if (x >= 0 && x < W) c = din[(size_t)y * W + x] != ADC_INVALID_FLOAT ? (uint8_t)ADC_ITP_VALID : (uint8_t)adc_itp_skip(cdist[(size_t)(y / ADC_ITP_CELL) * cw + x / ADC_ITP_CELL]);
                code[(size_t)y * pitch + c0 + b] = c;
            }
        }
    // the tables: packed (dy << 16 | dx & 0xffff) [ms][16], linear [ms + ADC_ITP_LPAD][16]
    std::vector<int32_t> tab((size_t)ms * 16, 0), lin((size_t)(ms + ADC_ITP_LPAD) * 16, 0);
    for (int m = 1; m < ms; m++)
        for (int s = 0; s < 16; s++) {
            const long dy = lround((double)m * sc[2 * s]), dx = lround((double)m * sc[2 * s + 1]);
            if (dy < 0 || dy >= ms || dx <= -ms || dx >= ms) return -5;
            tab[(size_t)m * 16 + s] = (int32_t)(((uint32_t)(dy & 0xffff) << 16) | (uint32_t)(dx & 0xffff));
            lin[(size_t)m * 16 + s] = (int32_t)(dy * pitch + dx);
        }
    long lookups = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const float d0 = din[p];
            if (!(label[p] == which && d0 == ADC_INVALID_FLOAT)) { dout[p] = d0; continue; }
            const uint8_t* c0 = img_l + p * 3;
            const bool mismatch = which == ADC_LABEL_MISMATCH;
            int min_dist = 9999;
            float best = mismatch ? 0.0f : ADC_LARGE_FLOAT;
            bool any = false;
            const long pb = (long)y * pitch + x + gx;
            for (int s = 0; s < 16; s++) {
                int ho = 0x7fffffff; // (ITP_NO_HIT)
                bool walking = true;
                int m = 1 + (int)code[pb];
                for (int trip = 0; walking && m < ms; trip++) {
                    const int NS = trip == 0 ? ns1 : ns2; // (the kernel: ITP_NS1 steps in a ray's first trip, ITP_NS2 in the following ones)
                    uint32_t c[ADC_ITP_NS];
                    int o[ADC_ITP_NS];
                    for (int j = 0; j < NS; j++) {
                        o[j] = lin[(size_t)(m + j) * 16 + s];
                        const long q = pb + o[j];
                        if (q < 0 || q >= (long)pitch * rows) return -1; // (the padding covers every position of the range)
                        c[j] = code[q];
                        lookups++;
                    }
                    bool act = true;
                    const int left = ms - m;
                    for (int j = 0; j < NS; j++) {
                        const bool end = c[j] >= ADC_ITP_OUTSIDE || j >= left;
                        if (act && end && c[j] == ADC_ITP_VALID && j < left) { ho = o[j]; if (tab[(size_t)(m + j) * 16 + s] == 0 && m + j > 0 && o[j] == 0) return -7; }
                        act = act && !end;
                    }
                    walking = act;
                    m += act ? NS + (int)c[NS - 1] : 0;
                }
                if (ho != 0x7fffffff) {
                    // the kernel's way back from the linear offset: dy = (ho + gx) / pitch by a float reciprocal + one correction step
                    if ((long)ms * pitch >= (1L << 24)) return -8; // (the launch falls back to the f64 kernel there)
                    const float rcp_pitch = 1.0f / (float)pitch;
                    const int a = ho + gx;
                    int dy = (int)((float)a * rcp_pitch);
                    const int r = a - dy * pitch;
                    dy += (r >= pitch ? 1 : 0) - (r < 0 ? 1 : 0);
                    const int dx = ho - dy * pitch;
                    if (a < 0 || dy < 0 || dx <= -ms || dx >= ms) return -9;
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) return -2; // (a hit lies inside the image)
                    const float hit = din[(size_t)yy * W + xx];
                    if (hit == ADC_INVALID_FLOAT) return -3;
                    any = true;
                    if (mismatch) {
                        const int dist = adc_color_dist_l1(c0, img_l + ((size_t)yy * W + xx) * 3);
                        if (min_dist > dist) { min_dist = dist; best = hit; }
                    } else best = hit < best ? hit : best;
                }
            }
            dout[p] = any ? best : 0.0f;
        }
    return lookups;
}

// ------------------------------------------------------------------ k_median_wavefront
void emul_median_wavefront(const float* in, float* out, int W, int H)
{
    std::vector<float> ring((size_t)H * 4, NAN);
    const int nsteps = W + 2 * (H - 1);
    for (int t = 0; t < nsteps; t++) {
        std::vector<std::pair<int, float>> writes; // commit after the level, like the barrier does
        for (int y = H - 1; y >= 0; y--) {          // any order inside a level
            const int x = t - 2 * y;
            if (x < 0 || x >= W) continue;
            float v[9];
            int n = 0;
            for (int r = -1; r <= 1; r++)
                for (int c = -1; c <= 1; c++) {
                    const int row = y + r, col = x + c;
                    float val = ADC_INVALID_FLOAT;
                    if (row >= 0 && row < H && col >= 0 && col < W) {
                        n++;
                        const bool filtered = (r < 0) || (r == 0 && c < 0);
                        val = filtered ? ring[row * 4 + (col & 3)] : in[(size_t)row * W + col];
                    }
                    v[(r + 1) * 3 + (c + 1)] = val;
                }
            adc_sort9(v);
            const float res = v[n / 2];
            out[(size_t)y * W + x] = res;
            ring[y * 4 + (x & 3)] = res; // in-level write is safe: distinct slot from all reads of this level
        }
    }
}

// ------------------------------------------------------------------ k_median_banded (selection rule)
// The banded kernel keeps the raster-order (in-place) data flow but replaces "sort the n in-image values,
// take wnd[n/2]" by the median of nine with -inf / +inf padding (adc_device_fn.h).  Sequential restatement.
float emul_median9(const float* v) { return adc_median9(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }

void emul_median_padded(const float* in, float* out, int W, int H)
{
    std::vector<float> cur(in, in + (size_t)W * H); // filtered in place, like the reference
    const float PINF = ADC_INVALID_FLOAT, NINF = -ADC_INVALID_FLOAT;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float v[9];
            for (int r = -1; r <= 1; r++)
                for (int c = -1; c <= 1; c++) {
                    const int row = y + r, col = x + c, i = (r + 1) * 3 + (c + 1);
                    const bool corner = (r != 0) && (c != 0);
                    v[i] = (row >= 0 && row < H && col >= 0 && col < W) ? cur[(size_t)row * W + col] : (corner ? PINF : NINF);
                }
            // same triple grouping as the kernel
            cur[(size_t)y * W + x] = adc_median9(v[0], v[1], v[6], v[5], v[7], v[8], v[2], v[3], v[4]);
        }
    std::copy(cur.begin(), cur.end(), out);
}

// Speculative bands of the banded median (k_median_banded, spec = 1): bands of `rows` rows; the real band c + 1 (c >= 1) takes as
// its "filtered row above" the last row of a COPY of band c that was filtered starting from the RAW row above band c; band 1
// takes the real band 0's last row.  `out` = the map the real bands write; returns the number of copies whose last row differs
// from the last row the real band wrote (what k_median_seg_check counts) -- 0 means out is the true in-place filter.
static void emul_median_band_rows(const float* raw, float* dst_rows, const float* above /* filtered row y0 - 1 or nullptr */, int W, int H, int y0, int y1)
{
    // rows [y0, y1) filtered in raster order: row above = `above` (already filtered), rows below = raw
    const float PINF = ADC_INVALID_FLOAT, NINF = -ADC_INVALID_FLOAT;
    std::vector<float> prev(W), cur(W);
    if (above) std::copy(above, above + W, prev.begin());
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            float v[9];
            for (int r = -1; r <= 1; r++)
                for (int c = -1; c <= 1; c++) {
                    const int row = y + r, col = x + c, i = (r + 1) * 3 + (c + 1);
                    const bool corner = (r != 0) && (c != 0);
                    float val = corner ? PINF : NINF;
                    if (row >= 0 && row < H && col >= 0 && col < W) {
                        if (r < 0) val = prev[col];                       // filtered row above
                        else if (r == 0 && c < 0) val = cur[col];         // filtered left neighbour
                        else val = raw[(size_t)row * W + col];            // not filtered yet
                    }
                    v[i] = val;
                }
            cur[x] = adc_median9(v[0], v[1], v[6], v[5], v[7], v[8], v[2], v[3], v[4]);
        }
        std::copy(cur.begin(), cur.end(), dst_rows + (size_t)(y - y0) * W);
        prev = cur;
    }
}
long emul_median_spec_bands(const float* raw, float* out, int W, int H, int rows, int depth)
{
    // depth = run-in in bands (k_median_banded: spec): the real band b > depth takes its row above from the last link of a chain
    // of copies of the bands b - depth .. b - 1, whose first link starts from the RAW row above it; bands 1 .. depth chain from the
    // real band 0
    const int nb = (H + rows - 1) / rows;
    long fails = 0;
    std::vector<float> copy_rows((size_t)rows * W), handoff(W);
    for (int b = 0; b < nb; b++) {
        const int y0 = b * rows, y1 = std::min(H, y0 + rows);
        const float* above = nullptr;
        if (b >= 1 && b <= depth) above = out + (size_t)(y0 - 1) * W; // the real band above (exact chain from band 0)
        if (b > depth) {
            for (int j = 0; j < depth; j++) { // copy of band b - depth + j
                const int c0 = (b - depth + j) * rows;
                std::vector<float> prev(handoff);
                emul_median_band_rows(raw, copy_rows.data(), j == 0 ? raw + (size_t)(c0 - 1) * W : prev.data(), W, H, c0, c0 + rows);
                std::copy(copy_rows.begin() + (size_t)(rows - 1) * W, copy_rows.begin() + (size_t)rows * W, handoff.begin());
            }
            above = handoff.data();
        }
        emul_median_band_rows(raw, out + (size_t)y0 * W, above, W, H, y0, y1);
        if (b > depth && memcmp(handoff.data(), out + (size_t)(y0 - 1) * W, (size_t)W * sizeof(float))) fails++; // (the real band b - 1 is done by now)
    }
    return fails;
}

// Speculative COLUMN SEGMENTS of the banded median (round 6; k_median_banded with nseg > 1).  Every band link is cut into nseg
// segments [xs, xe) (boundaries multiples of 16); the waves of a chain (copies of the bands b - min(b, depth) .. b - 1 and the real band b)
// that serve segment s all run the SAME window of levels t = x + 2y:
//     ts = xs - warm + 2 * (first row of the chain's real band)   (0 for segment 0: the true left border)
//     te = xe + 2 * (last row of the chain's real band) + 48       (the link's own last row when xe == W: the true right border)
// below ts a lane passes the raw value through (the speculation: unfiltered instead of filtered, as for the raw row above a chain),
// from te on nothing is computed (NaN here, the hand-off sentinel on the device: must never be consumed).  A real link writes the
// columns [xs, xe) of its rows into `out` and keeps the column xs - 1 of its warm-up as its seam.  Checks (k_median_seg_check):
//   row seams     the hand-off a real band b >= 1 consumed over the columns xs - 1 .. xe == the map row above it
//   column seams  the seam column of a real segment s >= 1 == the map column xs - 1 (written by segment s - 1)
// Returns the number of failing (band, segment) pairs; 0 means `out` is the true in-place filter (induction over bands and segments).
static bool emul_med_isnan(float v) { return v != v; }
static void emul_median_link(const float* raw, const float* above /* W values of row y0 - 1 as the upstream published them; nullptr: no row above */,
                             float* vals /* [y1 - y0][W] */, int W, int H, int y0, int y1, int ts, int te, bool* nan_used)
{
    const float PINF = ADC_INVALID_FLOAT, NINF = -ADC_INVALID_FLOAT, NANV = NAN;
    for (int y = y0; y < y1; y++) {
        float* cur = vals + (size_t)(y - y0) * W;
        const float* prev = y == y0 ? above : vals + (size_t)(y - 1 - y0) * W;
        for (int x = 0; x < W; x++) {
            const int t = x + 2 * y;
            if (t < ts) { cur[x] = raw[(size_t)y * W + x]; continue; }
            if (t >= te) { cur[x] = NANV; continue; }
            float v[9];
            for (int r = -1; r <= 1; r++)
                for (int c = -1; c <= 1; c++) {
                    const int row = y + r, col = x + c, i = (r + 1) * 3 + (c + 1);
                    const bool corner = (r != 0) && (c != 0);
                    float val = corner ? PINF : NINF;
                    if (row >= 0 && row < H && col >= 0 && col < W) {
                        if (r < 0) val = prev[col];
                        else if (r == 0 && c < 0) val = cur[col];
                        else val = raw[(size_t)row * W + col];
                        if (emul_med_isnan(val)) *nan_used = true;
                    }
                    v[i] = val;
                }
            cur[x] = adc_median9(v[0], v[1], v[6], v[5], v[7], v[8], v[2], v[3], v[4]);
        }
    }
}
long emul_median_spec_segments(const float* raw, float* out, int W, int H, int rows, int depth, int nseg, int warm, long* nan_reads, int seg_shift)
{
    const int nb = (H + rows - 1) / rows;
    long fails = 0;
    bool nan_used = false;
    std::vector<float> seam((size_t)nb * nseg * rows, 0.f), consumed((size_t)nb * nseg * W, 0.f); // per real (band, segment): seam column, hand-off row consumed
    std::vector<float> a((size_t)rows * W), b2((size_t)rows * W);
    for (int b = 0; b < nb; b++) {
        const int y0 = b * rows, y1 = std::min(H, y0 + rows);
        for (int s = 0; s < nseg; s++) {
            const int xs = adc_med_seg_x(W, nseg, s, seg_shift), xe = adc_med_seg_x(W, nseg, s + 1, seg_shift);
            std::vector<float> up; // hand-off row the real link consumes
            // the window of the chain of band b: every link runs it (its own last row decides when xe == W: the true right border)
            auto window = [&](int link_ylast, int* ts, int* te) {
                *ts = s == 0 ? 0 : std::max(0, xs - warm + 2 * y0);
                *te = xe == W ? W + 2 * link_ylast + 48 : xe + 2 * (y1 - 1) + 48;
            };
            int ts, te;
            const int dpt = std::min(b, depth); // band b >= 1: a private chain of copies of the bands b - dpt .. b - 1 (a copy of band 0 has no row above)
            for (int j = 0; j < dpt; j++) {
                const int c0 = (b - dpt + j) * rows, c1 = c0 + rows;
                window(c1 - 1, &ts, &te);
                std::vector<float> prev(up);
                emul_median_link(raw, j == 0 ? (c0 > 0 ? raw + (size_t)(c0 - 1) * W : nullptr) : prev.data(), a.data(), W, H, c0, c1, ts, te, &nan_used);
                up.assign(a.begin() + (size_t)(rows - 1) * W, a.begin() + (size_t)rows * W);
            }
            window(y1 - 1, &ts, &te);
            b2.assign((size_t)rows * W, 0.f);
            emul_median_link(raw, b >= 1 ? up.data() : nullptr, b2.data(), W, H, y0, y1, ts, te, &nan_used);
            for (int y = y0; y < y1; y++)
                for (int x = xs; x < xe; x++) {
                    const float v = b2[(size_t)(y - y0) * W + x];
                    if (emul_med_isnan(v)) nan_used = true;
                    out[(size_t)y * W + x] = v;
                }
            if (s > 0) for (int y = y0; y < y1; y++) seam[((size_t)b * nseg + s) * rows + (y - y0)] = b2[(size_t)(y - y0) * W + xs - 1];
            if (b >= 1) std::copy(up.begin(), up.end(), consumed.begin() + ((size_t)b * nseg + s) * W);
        }
    }
    for (int b = 0; b < nb; b++) {
        const int y0 = b * rows, y1 = std::min(H, y0 + rows);
        for (int s = 0; s < nseg; s++) {
            const int xs = adc_med_seg_x(W, nseg, s, seg_shift), xe = adc_med_seg_x(W, nseg, s + 1, seg_shift);
            bool bad = false;
            if (b >= 1)
                for (int c = std::max(xs - 1, 0); c <= std::min(xe, W - 1); c++)
                    bad = bad || memcmp(&consumed[((size_t)b * nseg + s) * W + c], &out[(size_t)(y0 - 1) * W + c], sizeof(float)) != 0;
            if (s >= 1)
                for (int y = y0; y < y1; y++)
                    bad = bad || memcmp(&seam[((size_t)b * nseg + s) * rows + (y - y0)], &out[(size_t)y * W + xs - 1], sizeof(float)) != 0;
            if (bad) fails++;
        }
    }
    if (nan_reads) *nan_reads = nan_used ? 1 : 0;
    return fails;
}

// gray of a single pixel (device function check over all 2^24 triples is done from Python in chunks)
void emul_gray(const uint8_t* bgr, uint8_t* gray, size_t n)
{
    for (size_t i = 0; i < n; i++) gray[i] = adc_gray(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
}

} // extern "C"
