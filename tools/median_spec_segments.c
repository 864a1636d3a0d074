// CPU model of K11's speculative COLUMN SEGMENTS (round 6): the in-place 3x3 median (adcensus_util.cpp:55-81 semantics) run on the
// region rows [y0, yl] x levels [ts, te) (level = x + 2y) of a copy of the raw map, in raster order, everything outside the region raw --
// what a chain of band links does for one segment when all of its links run the same window of levels.  Compared with the true filter's result: the column xs - 1 over the rows [yf, yl] (the state a real segment takes
// over from its warm-up) and the row yf - 1 over the columns [xs - 1, xe] (the hand-off it consumes).
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
// region: rows [y0, y1], levels ts <= x + 2y < te (what a chain of band links covers when every link runs the levels [ts, te):
// a row starts two columns to the left of the row above it; everything outside the region is raw)
static void med_region(float* disp, int w, int h, int y0, int y1, int ts, int te)
{
    float wnd[9];
    for (int y = y0; y <= y1; y++)
        for (int x = 0; x < w; x++) {
            if (x + 2 * y < ts || x + 2 * y >= te) continue;
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
// all (band, segment) pairs of an image: bands of `band` rows, nseg segments (boundaries multiples of 16), run-in of `vrun` rows and
// `warm` columns.  Returns the number of (band, segment) pairs with a failing check; *hbad / *vbad = failing H / V checks, *pairs = pairs checked
long spec_segments(const float* raw, const float* truth, int w, int h, int band, int nseg, int vrun, int warm, long* hbad, long* vbad, long* pairs)
{
    long bad = 0; *hbad = 0; *vbad = 0; *pairs = 0;
    float* tmp = malloc((size_t)w * h * sizeof(float));
    for (int yf = 0; yf < h; yf += band) {
        const int yl = (yf + band < h ? yf + band : h) - 1;
        const int y0 = yf - vrun > 0 ? yf - vrun : 0;
        for (int s = 0; s < nseg; s++) {
            int xs = (int)((long)w * s / nseg) & ~15, xe = s + 1 == nseg ? w : ((int)((long)w * (s + 1) / nseg) & ~15);
            const int ts = s == 0 ? 0 : xs - warm + 2 * yf; // the band's top row starts `warm` columns in front of the segment
            const int te = xe + 2 * yl + 2;
            if (s == 0 && yf == 0) continue; // (the true start)
            memcpy(tmp, raw, (size_t)w * h * sizeof(float));
            med_region(tmp, w, h, y0, yl, ts, te);
            long hd = 0, vd = 0;
            if (s > 0) for (int y = yf; y <= yl; y++) hd += memcmp(&tmp[(size_t)y * w + xs - 1], &truth[(size_t)y * w + xs - 1], 4) != 0;
            if (yf > 0) for (int x = (xs > 0 ? xs - 1 : 0); x <= (xe < w ? xe : w - 1); x++) vd += memcmp(&tmp[(size_t)(yf - 1) * w + x], &truth[(size_t)(yf - 1) * w + x], 4) != 0;
            (*pairs)++;
            if (hd) (*hbad)++;
            if (vd) (*vbad)++;
            if (hd || vd) bad++;
        }
    }
    free(tmp);
    return bad;
}
