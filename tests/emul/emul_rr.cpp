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
struct uint4 { uint32_t x, y, z, w; };
#include "../../adcensus_amd/csrc/adc_device_fn.h"
#include "../../adcensus_amd/csrc/k_aggregate_rr.h"
#include "../../adcensus_amd/csrc/k_aggregate_rr2.h"

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

// ------------------------------------------------------------------------------------------------------------------
// third generation (k_aggregate_rr2.h): two disparities per lane, ring slots = VGPR pairs, packed adds; optionally the
// fused matching cost (COSTIN) from packed pixel records built exactly as k_cost_records builds them (k_cost.hip)
template <bool VERT, bool DIVIDE, bool COSTIN>
static void run_grid2(const float* src, float* dst, const uint2* rec, int W, int H, int Dp, int L, int chunk_len, int nwaves, int per_xcd,
                      const AggCostIn& ci)
{
    const int armmax[2] = {L, L};
    for (int b = 0; b < per_xcd * 8; b++)
        for (int lane = 0; lane < 64; lane++) {
            rr_emul.block = b;
            rr_emul.lane = lane;
            for (int i = 0; i < 256; i++) rr_emul.vgpr[i] = NAN;
            agg_rr2_body<VERT, DIVIDE, COSTIN>(src, dst, rec, W, H, Dp, L, chunk_len, nwaves, per_xcd, armmax, -1, 0x7fffffff, ci);
        }
}

// costin != 0: src_hwd is ignored, the pass computes the cost from the images / census words (first pass: H, non-dividing)
extern "C" int emul_rr2_pass(const float* src_hwd, float* dst_hwd, const uint8_t* arms, const uint16_t* sup, int W, int H, int D,
                             int vert, int divide, int L, int chunk_len, int costin, const uint8_t* img_l, const uint8_t* img_r,
                             const uint64_t* cen_l, const uint64_t* cen_r, int dmin, int lambda_ad, int lambda_census)
{
    if (2 * L + 1 > RR2_SLOTS || L < 1) return 1;
    if (costin && (vert || divide)) return 1;
    int Dp = D <= 64 ? 64 : (D <= 128 ? 128 : (D <= 256 ? 256 : (D <= 512 ? 512 : 1024))); // the product's padded range
    if (Dp < 128) Dp = 128; // (the product runs this body only when Dp % 128 == 0; a 64-wide range is padded here)
    const size_t P = (size_t)W * H;
    std::vector<float> a(P * Dp, 0.0f), bvol(P * Dp, -12345.0f);
    if (!costin)
        for (size_t p = 0; p < P; p++)
            for (int d = 0; d < D; d++) a[p * Dp + d] = src_hwd[p * D + d];
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
    AggCostIn ci = {};
    std::vector<uint4> rrec, lrec;
    if (costin) {
        for (int k = 0; k < 768; k++) rr2_emul_cost.lut[k] = 0.0f;
        for (int k = 0; k <= 765; k++) rr2_emul_cost.lut[k] = (1.0f - expf(-((float)k / 3.0f) / (float)lambda_ad)) + 1.0f; // capi.hip upload_tables
        for (int hm = 0; hm < 64; hm++) rr2_emul_cost.lut[768 + hm] = expf(-(float)hm / (float)lambda_census);
        const int padl = (dmin + Dp - 1 > 0 ? dmin + Dp - 1 : 0) + 1; // capi.hip alloc_all
        const int pitch = padl + W + (dmin < 0 ? -dmin : 0) + 1;
        rrec.assign((size_t)H * pitch, uint4{0xFFFFFFFFu, 0u, 0u, 0u});
        lrec.resize(P);
        auto pack = [](const uint8_t* px) { return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16); };
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t p = (size_t)y * W + x;
                rrec[(size_t)y * pitch + padl + x] = uint4{pack(img_r + 3 * p), (uint32_t)cen_r[p], (uint32_t)(cen_r[p] >> 32), 0u};
                lrec[p] = uint4{pack(img_l + 3 * p), (uint32_t)cen_l[p], (uint32_t)(cen_l[p] >> 32), 0u};
            }
        ci.rrec = rrec.data(); ci.lrec = lrec.data(); ci.lut_ad = nullptr; ci.lut_census = nullptr;
        ci.rpitch = pitch; ci.padl = padl; ci.dmin = dmin; ci.D = D;
    }
    // launch geometry as launch_pass computes it: a wave = chunk_len outputs of the line-major index space
    const long long nlines = (long long)(vert ? W : H) * (Dp / 128);
    const long long total = nlines * N;
    if (chunk_len < 1) chunk_len = N;
    const int nwaves = (int)((total + chunk_len - 1) / chunk_len);
    const int per_xcd = (nwaves + 7) / 8;
#define RUN2(V, DV, CI) run_grid2<V, DV, CI>(a.data(), bvol.data(), rec.data(), W, H, Dp, L, chunk_len, nwaves, per_xcd, ci)
    if (costin) RUN2(false, false, true);
    else if (vert) { if (divide) RUN2(true, true, false); else RUN2(true, false, false); }
    else { if (divide) RUN2(false, true, false); else RUN2(false, false, false); }
#undef RUN2
    for (size_t p = 0; p < P; p++) {
        for (int d = 0; d < D; d++) dst_hwd[p * D + d] = bvol[p * Dp + d];
        if (costin)
            for (int d = D; d < Dp; d++)
                if (bvol[p * Dp + d] != 0.0f) return 2; // padding disparities are written as 0.0f
    }
    return 0;
}

// Whole-wave model of the two lane windows of the fused cost (k_aggregate_rr2.h RR2_WIN_STEP): A' = B shifted up one lane
// with the new column entering at lane 0, B' = A.  Must reproduce, for every lane and step, the columns x - d0 and
// x - d0 - 1 the per-lane emulation reads directly.  cols: column ids of a padded row (index i -> column id), start = index of
// lane 0's even-disparity column of the entry BEFORE the first step.  Returns the number of mismatches.
extern "C" int emul_rr2_window_identity(int steps, int start)
{
    int bad = 0;
    long A[64], B[64];
    for (int l = 0; l < 64; l++) { A[l] = (long)start - 2 * l; B[l] = (long)start - 2 * l - 1; }
    for (int s = 1; s <= steps; s++) {
        long nA[64];
        for (int l = 63; l > 0; l--) nA[l] = B[l - 1]; // DPP wave_shr:1 of B
        nA[0] = (long)start + s;                        // the new column (wave-uniform) enters at lane 0
        for (int l = 0; l < 64; l++) { B[l] = A[l]; A[l] = nA[l]; }
        for (int l = 0; l < 64; l++) bad += (A[l] != (long)start + s - 2 * l) + (B[l] != (long)start + s - 2 * l - 1);
    }
    return bad;
}

// The launch gate shared by every marching aggregation body (agg_gate_skip, k_aggregate_rr.h): does the launch with the
// given code / packed depths skip its work for an image with these arm maxima?  (CPU tier: exactly one of the two plans of a
// two-plan run works, whatever the image; the per-direction codes of the debug surface; the verify code.)
extern "C" int emul_agg_gate_skip(int armmax_h, int armmax_v, int small_variant, int small_L, int vert)
{
    const int armmax[4] = {armmax_h, armmax_v, 0, 0};
    return agg_gate_skip(armmax, small_variant, small_L, vert != 0) ? 1 : 0;
}
