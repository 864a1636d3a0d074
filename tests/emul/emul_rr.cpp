// CPU build of the register-ring aggregation body (adcensus_amd/csrc/k_aggregate_rr.h, RR_EMUL): the SAME control flow
// (segments, halos, ring slots, record positions, pair hand-over, tails) runs lane by lane against a modelled VGPR file
// with the M0-relative addressing of the indexed add blocks; only the asynchronous-load / inline-asm primitives are
// replaced.  Test infrastructure (tests/test_emul.py); shares no code with the oracle.
#define RR_EMUL
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
struct uint2 { uint32_t x, y; };
#include "../../adcensus_amd/csrc/adc_device_fn.h"
#include "../../adcensus_amd/csrc/k_aggregate_rr.h"

template <bool VERT, bool DIVIDE, bool PAIR>
static void run_grid(const float* src, float* dst, const uint2* rec, int W, int H, int Dp, int L, int seg_len, int nseg, int per_xcd,
                     float* sink)
{
    const int armmax[2] = {L, L};
    for (int b = 0; b < per_xcd * 8; b++)
        for (int lane = 0; lane < 64; lane++) {
            rr_emul.block = b;
            rr_emul.lane = lane;
            for (int i = 0; i < 256; i++) rr_emul.vgpr[i] = NAN; // anything read outside what was written poisons the sum
            agg_rr_body<VERT, DIVIDE, PAIR>(src, dst, rec, W, H, Dp, L, seg_len, nseg, per_xcd, armmax, -1, 0x7fffffff, sink);
        }
}

extern "C" int emul_rr_pass(const float* src_hwd, float* dst_hwd, const uint8_t* arms, const uint16_t* sup, int W, int H, int D,
                            int vert, int divide, int pair, int L, int nseg)
{
    if (2 * L + 1 > 72 || L < 1) return 1;
    const int Dp = (D + 63) / 64 * 64;
    const size_t P = (size_t)W * H;
    std::vector<float> a(P * Dp, 0.0f), bvol(P * Dp, -12345.0f), sink(1024 * 64);
    for (size_t p = 0; p < P; p++)
        for (int d = 0; d < D; d++) a[p * Dp + d] = src_hwd[p * D + d];
    // records exactly as k_make_records writes them (k_arms.hip)
    const int N = vert ? H : W;
    std::vector<uint2> rec(P);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const uint8_t* ar = arms + p * 4;
            const uint32_t alo = vert ? ar[2] : ar[0], ahi = vert ? ar[3] : ar[1], c = sup[p];
            uint2 r;
            r.x = ((alo + (uint32_t)L + 1u) & 255u) | (((alo + ahi + 1u) & 255u) << 8) | (c << 16);
            const float y_ = 1.0f / (float)c;
            memcpy(&r.y, &y_, 4);
            rec[vert ? (size_t)x * H + y : p] = r;
        }
    // launch geometry as launch_pass computes it
    int seg_len = (N + nseg - 1) / nseg;
    if (seg_len < 1) seg_len = 1;
    nseg = (N + seg_len - 1) / seg_len;
    const long long nlines = (long long)(vert ? W : H) * (Dp / 64);
    const int per_xcd = (int)((nlines * nseg + 7) / 8);
#define RUN(V, DV, PR) run_grid<V, DV, PR>(a.data(), bvol.data(), rec.data(), W, H, Dp, L, seg_len, nseg, per_xcd, sink.data())
    if (pair) { if (vert) RUN(true, true, true); else RUN(false, true, true); }
    else if (vert) { if (divide) RUN(true, true, false); else RUN(true, false, false); }
    else { if (divide) RUN(false, true, false); else RUN(false, false, false); }
#undef RUN
    for (size_t p = 0; p < P; p++)
        for (int d = 0; d < D; d++) dst_hwd[p * D + d] = bvol[p * Dp + d];
    return 0;
}
