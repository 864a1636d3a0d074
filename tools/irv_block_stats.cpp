// Region voting (K8), analysis tool (CPU only, see tools/irv_block_stats.py): how many trips does the "pixels of several bins" loop
// of a vote take (k_voting.hip: one LDS atomic per counted pixel of a 16-byte block whose pixels do not share one bin, a wave runs
// as many trips as its worst lane), and how many would it take if the pixels of the block's FIRST bin (and of its second bin) went
// into one atomic each?  Evaluated for every work-list entry of the first two passes on the state at pass start.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "../adcensus_amd/csrc/adc_device_fn.h"

extern "C" long irv_block_stats(const float* disp, const uint8_t* label, const uint8_t* arms, int W, int H, int dmin, int D, int which,
                                double* out /* [0] votes, [1] votes with a mixed block, [2..4] sum of trips: now / first bin merged / two bins merged,
                                               [5] blocks, [6] mixed blocks */)
{
    const int P = W * H;
    std::vector<uint8_t> elig(P);
    for (int p = 0; p < P; p++) elig[p] = (label[p] == which && disp[p] == ADC_INVALID_FLOAT) ? 1 : 0;
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (int p = 0; p < P; p++) {
        if (!elig[p]) continue;
        const int y = p / W, x = p - y * W;
        const uint8_t* arm = arms + (size_t)p * 4;
        int t_now = 0, t_one = 0, t_two = 0;
        bool mixed_vote = false;
        for (int t = -(int)arm[2]; t <= (int)arm[3]; t++) {
            const int yt = y + t;
            const uint8_t* arm2 = arms + ((size_t)yt * W + x) * 4;
            const int xl = x - (int)arm2[0], xr = x + (int)arm2[1];
            for (int blk = xl >> 3; blk <= xr >> 3; blk++) {
                int bins[8], n = 0;
                for (int q = 0; q < 8; q++) {
                    const int xx = blk * 8 + q;
                    if (xx < xl || xx > xr || xx < 0 || xx >= W) continue;
                    const int qi = yt * W + xx;
                    float v = disp[qi];
                    if (elig[qi] && qi >= p) v = ADC_INVALID_FLOAT;
                    if (v == ADC_INVALID_FLOAT) continue;
                    const long b = lroundf(v) - dmin;
                    if (b >= 0 && b < D) bins[n++] = (int)b;
                }
                if (!n) continue;
                out[5] += 1;
                int nfirst = 0;
                for (int i = 0; i < n; i++) nfirst += bins[i] == bins[0];
                if (nfirst == n) continue; // single bin: one atomic, no trips
                out[6] += 1;
                mixed_vote = true;
                int second = -1, nsecond = 0;
                for (int i = 0; i < n; i++)
                    if (bins[i] != bins[0]) { if (second < 0) second = bins[i]; nsecond += bins[i] == second; }
                t_now = std::max(t_now, n);
                t_one = std::max(t_one, n - nfirst);
                t_two = std::max(t_two, n - nfirst - nsecond);
            }
        }
        out[0] += 1;
        out[1] += mixed_vote;
        out[2] += t_now; out[3] += t_one; out[4] += t_two;
    }
    return (long)out[0];
}
