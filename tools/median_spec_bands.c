#include <stdlib.h>
#include <string.h>
#include <stdio.h>
// in-place 3x3 median on rows [r0, r1) of an h x w map (rows outside stay as they are, but ARE read); adcensus_util.cpp:55-81 semantics
static void med_rows(float* disp, int w, int h, int r0, int r1)
{
    float wnd[9];
    for (int y = r0; y < r1; y++)
        for (int x = 0; x < w; x++) {
            int n = 0;
            for (int r = -1; r <= 1; r++)
                for (int cc = -1; cc <= 1; cc++) {
                    const int row = y + r, col = x + cc;
                    if (row >= 0 && row < h && col >= 0 && col < w) wnd[n++] = disp[(size_t)row * w + col];
                }
            for (int i = 1; i < n; i++) { const float v = wnd[i]; int j = i - 1; while (j >= 0 && wnd[j] > v) { wnd[j + 1] = wnd[j]; j--; } wnd[j + 1] = v; }
            if (n) disp[(size_t)y * w + x] = wnd[n / 2];
        }
}
// returns number of seams (band starts) whose speculative last warm-up row differs from the truth; worst = max differing pixels in a seam row
long spec_bands(const float* raw, const float* truth, int w, int h, int band, int wu, long* worst, long* seams)
{
    long bad = 0; *worst = 0; *seams = 0;
    float* tmp = malloc((size_t)w * h * sizeof(float));
    for (int s = band; s < h; s += band) {
        int r0 = s - wu; if (r0 < 0) r0 = 0;
        memcpy(tmp, raw, (size_t)w * h * sizeof(float));
        med_rows(tmp, w, h, r0, s);   // rows above r0 stay RAW (speculation), rows >= s raw (as in the true order)
        long diff = 0;
        for (int x = 0; x < w; x++) diff += memcmp(&tmp[(size_t)(s - 1) * w + x], &truth[(size_t)(s - 1) * w + x], 4) != 0;
        (*seams)++;
        if (diff) bad++;
        if (diff > *worst) *worst = diff;
    }
    free(tmp);
    return bad;
}
