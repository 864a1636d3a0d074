#include <stdint.h>
#include <stdlib.h>
// depth of the finality DAG of one voting pass: p depends on the eligible pixels of its cross region that precede it
int irv_depth(const uint8_t* arms, const uint8_t* elig, int W, int H, int* depth_out, long* edges)
{
    int maxd = 0; long e = 0;
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        const int p = y * W + x; depth_out[p] = 0;
        if (!elig[p]) continue;
        const uint8_t* a = arms + 4 * p; // l r t b
        int d = 0;
        for (int t = -(int)a[2]; t <= 0; t++) {
            const int yy = y + t; const uint8_t* ar = arms + 4 * (yy * W + x);
            for (int s = -(int)ar[0]; s <= (int)ar[1]; s++) {
                const int xx = x + s; if (t == 0 && xx >= x) break;
                const int q = yy * W + xx;
                if (elig[q]) { e++; if (depth_out[q] > d) d = depth_out[q]; }
            }
        }
        depth_out[p] = d + 1; if (d + 1 > maxd) maxd = d + 1;
    }
    *edges = e; return maxd;
}
