/* Ray-walk statistics of the interpolation stage (analysis tool, CPU only; see tools/itp_ray_stats.py).
 * Walks the 16 rays of every target pixel like k_interpolate_tab does -- NS0 steps in the first round trip, NS in the
 * following ones, empty-space skipping on a map of 2x2-pixel cells with a distance cap of CAP cells -- and counts
 *   gathers      map values requested (NS per round trip and ray, finished rays of a wave keep re-reading their own pixel:
 *                not counted),
 *   trips        round trips per ray (sum),
 *   wave_trips   round trips of the waves: 4 consecutive targets x 16 rays share a wave, which iterates until its last ray ends.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

void itp_cdist(const uint8_t* valid, int W, int H, int cap, uint8_t* cdist)
{
    const int cw = (W + 1) / 2, ch = (H + 1) / 2;
    uint8_t* cell = (uint8_t*)calloc((size_t)cw * ch, 1);
    uint8_t* rowd = (uint8_t*)malloc((size_t)cw * ch);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            if (valid[(size_t)y * W + x]) cell[(size_t)(y / 2) * cw + x / 2] = 1;
    for (int cy = 0; cy < ch; cy++)
        for (int cx = 0; cx < cw; cx++) {
            int best = cap + 1;
            for (int dx = -cap; dx <= cap; dx++) {
                const int x = cx + dx;
                if (x >= 0 && x < cw && cell[(size_t)cy * cw + x]) best = imin(best, abs(dx));
            }
            rowd[(size_t)cy * cw + cx] = (uint8_t)best;
        }
    for (int cy = 0; cy < ch; cy++)
        for (int cx = 0; cx < cw; cx++) {
            int best = cap + 1;
            for (int dy = -cap; dy <= cap; dy++) {
                const int y = cy + dy;
                if (y >= 0 && y < ch) best = imin(best, imax((int)rowd[(size_t)y * cw + cx], abs(dy)));
            }
            cdist[(size_t)cy * cw + cx] = (uint8_t)best;
        }
    free(cell);
    free(rowd);
}

/* tab[m*16+s] = (dy << 16) | (dx & 0xffff) for m = 1 .. max_search-1; targets = pixel indices in list order */
void itp_walk(const uint8_t* valid, const uint8_t* cdist, int W, int H, const int32_t* tab, int max_search, const int32_t* targets,
              long n, int ns0, int ns, long long* gathers, long long* trips, long long* wave_trips, long long* hist /* [64]: trips per ray */)
{
    const int cw = (W + 1) / 2;
    *gathers = *trips = *wave_trips = 0;
    memset(hist, 0, 64 * sizeof(long long));
    for (long e0 = 0; e0 < n; e0 += 4) {
        int wave_max = 0;
        for (long e = e0; e < e0 + 4 && e < n; e++) {
            const int p = targets[e], y = p / W, x = p - y * W;
            for (int s = 0; s < 16; s++) {
                int m = 1, t = 0, walking = 1;
                const int c0 = cdist[(size_t)(y / 2) * cw + x / 2];
                m += c0 >= 2 ? (c0 - 1) * 2 - 1 : 0;
                int first = 1;
                while (walking && m < max_search) {
                    const int k = first ? ns0 : ns;
                    first = 0;
                    t++;
                    *gathers += k;
                    int lastc = 0;
                    for (int j = 0; j < k && walking; j++) {
                        if (m + j >= max_search) { walking = 0; break; }
                        const int o = tab[(m + j) * 16 + s];
                        const int yy = y + (o >> 16), xx = x + (int)(short)(o & 0xffff);
                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) { walking = 0; break; }
                        if (valid[(size_t)yy * W + xx]) { walking = 0; break; }
                        if (j == k - 1) lastc = cdist[(size_t)(yy / 2) * cw + xx / 2];
                    }
                    m += k;
                    m += lastc >= 2 ? (lastc - 1) * 2 - 1 : 0;
                }
                *trips += t;
                hist[t < 63 ? t : 63]++;
                wave_max = imax(wave_max, t);
            }
        }
        *wave_trips += wave_max;
    }
}
