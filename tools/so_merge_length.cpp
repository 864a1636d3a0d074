// Scanline DP (K5): after how many path elements does a pass that STARTS IN THE MIDDLE of a path (with the wrong state: the
// raw costs of its first element, like the reference's first pixel) become bit-identical to the full pass?  Analysis tool, CPU
// only (tools/so_merge_length.py).  The recurrence halves every perturbation per step -- L = (C + min(...)) / 2 with 1-Lipschitz
// minima -- so the two states merge; this counts the steps until ALL disparities of the state are equal bit for bit, for
// restarts every `stride` elements on every path of one pass, and returns the histogram of those merge lengths.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../adcensus_amd/csrc/adc_device_fn.h"

extern "C" void so_merge(const float* src, const float* full /* result of the full pass */, const uint8_t* cd_left, const uint8_t* cd_right,
                         int W, int H, int dmin, int D, int vert, int dir, int tso, float p1, float p2, int stride, int max_len,
                         long* hist /* [max_len + 1]; [max_len] = not merged within max_len */)
{
    const float P1c[3] = {p1, p1 / 4, p1 / 10}, P2c[3] = {p2, p2 / 4, p2 / 10};
    const int npaths = vert ? W : H, plen = vert ? H : W;
    std::vector<float> Lp(D), out(D);
    memset(hist, 0, (max_len + 1) * sizeof(long));
    for (int path = 0; path < npaths; path++) {
        auto coord = [&](int i, int& x, int& y) {
            const int m = dir > 0 ? i : plen - 1 - i;
            if (vert) { x = path; y = m; } else { x = m; y = path; }
        };
        for (int i0 = stride; i0 + max_len < plen; i0 += stride) { // (only restarts with max_len elements left to merge in)
            int x, y;
            coord(i0, x, y);
            float minLp = ADC_LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                Lp[d] = src[((size_t)y * W + x) * D + d];
                minLp = Lp[d] < minLp ? Lp[d] : minLp;
            }
            int merged = max_len;
            for (int i = i0 + 1; i < plen && i - i0 <= max_len; i++) {
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
                Lp = out;
                minLp = omin;
                if (!memcmp(out.data(), full + ((size_t)y * W + x) * D, D * sizeof(float))) { merged = i - i0; break; }
            }
            hist[merged]++;
        }
    }
}
