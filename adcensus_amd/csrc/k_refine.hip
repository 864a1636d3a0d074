// k_refine.hip -- K7..K11 multi-step refinement.
//
// Replaces MultiStepRefiner::{OutlierDetection, IterativeRegionVoting, ProperInterpolation,
// DepthDiscontinuityAdjustment, EdgeDetect} (multistep_refiner.cpp:90-371) and the in-place 3x3
// adcensus_util::MedianFilter (adcensus_util.cpp:55-81, called with in == out at
// multistep_refiner.cpp:86).  The reference's raster-order, in-place (Gauss-Seidel) semantics are
// reproduced EXACTLY by order-aware parallel formulations:
//   K7  LR check      : two phases (invalid mask from the original maps; classification reads the
//                       mask for columns to the left of the pixel)
//   K8  region voting : per pass, fixed-point iteration of the triangular system "value(p) =
//                       vote(p | values of eligible pixels that precede p in raster order)"
//                       with dirty-tile re-evaluation; converges to the sequential result
//   K9  interpolation : Jacobi within a list => plain parallel launch per list
//   K10 discontinuity : Sobel mask + one thread per row (in-row sequential dependency)
//   K11 median        : wavefront t = x + 2y; bands of 64 rows, one wave per band, register-resident, band-to-band hand-off
//                       rows (k_median_banded); single-workgroup wavefront kernel as the time-out fallback
#include "adc_internal.h"
#include <mutex>
#include "adc_device_fn.h"

#include <vector>
#include <stdio.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------------- K7 LR check
__device__ __forceinline__ bool lr_invalid(const float* __restrict__ dl, const float* __restrict__ dr, int W, int x, int y,
                                           float thres, int& col_right, float& disp_r)
{
    const float d = dl[(size_t)y * W + x];
    col_right = -1;
    disp_r = 0.f;
    if (d == ADC_INVALID_FLOAT) return true;
    const long cr = lroundf((float)x - d); // multistep_refiner.cpp:114
    if (cr < 0 || cr >= W) return true;
    col_right = (int)cr;
    disp_r = dr[(size_t)y * W + cr];
    return fabsf(d - disp_r) > thres;
}

__global__ __launch_bounds__(256) void k_lr_phase1(const float* __restrict__ dl, const float* __restrict__ dr,
                                                   uint8_t* __restrict__ inv, int W, int H, float thres)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    int cr;
    float dr_v;
    inv[(size_t)y * W + x] = lr_invalid(dl, dr, W, x, y, thres, cr, dr_v) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_lr_phase2(const float* __restrict__ dl, const float* __restrict__ dr,
                                                   const uint8_t* __restrict__ inv, float* __restrict__ out,
                                                   uint8_t* __restrict__ label, int W, int H, float thres)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const float d = dl[p];
    int cr;
    float disp_r;
    const bool bad = lr_invalid(dl, dr, W, x, y, thres, cr, disp_r);
    uint8_t lab = ADC_LABEL_VALID;
    if (bad) {
        lab = ADC_LABEL_MISMATCH;
        if (d != ADC_INVALID_FLOAT && cr >= 0) {
            const long col_rl = lroundf((float)cr + disp_r); // multistep_refiner.cpp:127
            if (col_rl > 0 && col_rl < W) {
                // in-place raster scan: a pixel to the LEFT that failed the check already holds +inf
                const float disp_l = (col_rl < x && inv[(size_t)y * W + col_rl]) ? ADC_INVALID_FLOAT : dl[(size_t)y * W + col_rl];
                if (disp_l > d) lab = ADC_LABEL_OCCLUSION;
            }
        }
    }
    label[p] = lab;
    out[p] = bad ? ADC_INVALID_FLOAT : d;
}

hipError_t adc_launch_lrcheck(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_lr_phase1, grid, block, 0, h->stream, h->disp_l, h->disp_r, h->elig, p.W, p.H, p.opt.lrcheck_thres);
    hipLaunchKernelGGL(k_lr_phase2, grid, block, 0, h->stream, h->disp_l, h->disp_r, h->elig, h->disp_tmp, h->label, p.W, p.H,
                       p.opt.lrcheck_thres);
    float* t = h->disp_l;
    h->disp_l = h->disp_tmp;
    h->disp_tmp = t;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ K8 region voting: k_voting.hip

// Target list of one interpolation pass: the pixels of the list (`which`) that are still invalid, compacted in chunks of
// 2048 pixels (ONE list-length atomic per workgroup: same-address atomics retire at roughly 8 ns each, so a
// per-256-pixel atomic -- 8100 of them at 1080p -- alone cost ~65 us).
#define ITP_LIST_PPT 8
__global__ __launch_bounds__(256) void k_interp_list(const uint8_t* __restrict__ label, const float* __restrict__ disp,
                                                     int32_t* __restrict__ list, int32_t* __restrict__ counters, int which, int P)
{
    __shared__ int wcnt[ITP_LIST_PPT][4];
    __shared__ int base;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long m[ITP_LIST_PPT];
    bool listed[ITP_LIST_PPT];
#pragma unroll
    for (int k = 0; k < ITP_LIST_PPT; k++) {
        const int p = (blockIdx.x * ITP_LIST_PPT + k) * 256 + threadIdx.x;
        listed[k] = p < P && label[p] == which && disp[p] == ADC_INVALID_FLOAT;
        m[k] = __ballot(listed[k]);
        if (lane == 0) wcnt[k][wave] = __popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < ITP_LIST_PPT; k++) tot += wcnt[k][0] + wcnt[k][1] + wcnt[k][2] + wcnt[k][3];
        base = tot ? atomicAdd(&counters[0], tot) : 0;
    }
    __syncthreads();
    int off = base;
#pragma unroll
    for (int k = 0; k < ITP_LIST_PPT; k++) {
        const int p = (blockIdx.x * ITP_LIST_PPT + k) * 256 + threadIdx.x;
        int mine = off;
        for (int w = 0; w < wave; w++) mine += wcnt[k][w];
        if (listed[k]) list[mine + __popcll(m[k] & ((1ull << lane) - 1ull))] = p;
        off += wcnt[k][0] + wcnt[k][1] + wcnt[k][2] + wcnt[k][3];
    }
}

// Both target lists in ONE pass (round 6): list A = the invalid pixels of the mismatch list, list B = those of the occlusion list.  The
// first list's fills cannot change the second list's targets (a fill only touches list-A pixels), so both are known up front.
__global__ __launch_bounds__(256) void k_interp_list2(const uint8_t* __restrict__ label, const float* __restrict__ disp,
                                                      int32_t* __restrict__ list_a, int32_t* __restrict__ list_b, int32_t* __restrict__ counters, int P)
{
    __shared__ int wcnt[2][ITP_LIST_PPT][4];
    __shared__ int base[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long m[2][ITP_LIST_PPT];
#pragma unroll
    for (int k = 0; k < ITP_LIST_PPT; k++) {
        const int p = (blockIdx.x * ITP_LIST_PPT + k) * 256 + threadIdx.x;
        const bool inv = p < P && disp[p] == ADC_INVALID_FLOAT;
        const int lab = p < P ? (int)label[p] : 0;
        m[0][k] = __ballot(inv && lab == ADC_LABEL_MISMATCH);
        m[1][k] = __ballot(inv && lab == ADC_LABEL_OCCLUSION);
        if (lane == 0) { wcnt[0][k][wave] = __popcll(m[0][k]); wcnt[1][k][wave] = __popcll(m[1][k]); }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < ITP_LIST_PPT; k++) tot += wcnt[threadIdx.x][k][0] + wcnt[threadIdx.x][k][1] + wcnt[threadIdx.x][k][2] + wcnt[threadIdx.x][k][3];
        base[threadIdx.x] = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; l++) {
        int32_t* list = l == 0 ? list_a : list_b;
        int off = base[l];
#pragma unroll
        for (int k = 0; k < ITP_LIST_PPT; k++) {
            const int p = (blockIdx.x * ITP_LIST_PPT + k) * 256 + threadIdx.x;
            int mine = off;
            for (int w = 0; w < wave; w++) mine += wcnt[l][k][w];
            if ((m[l][k] >> lane) & 1ull) list[mine + __popcll(m[l][k] & ((1ull << lane) - 1ull))] = p;
            off += wcnt[l][k][0] + wcnt[l][k][1] + wcnt[l][k][2] + wcnt[l][k][3];
        }
    }
}

// --------------------------------------------------------------------------- K9 proper interpolation
// The target pixels of the list are compacted first (k_interp_list), then 16 lanes
// work on one pixel, one ray each (4 pixels per wave); the 16 first-hits are combined with a 16-lane
// butterfly: mismatch -> lexicographic min of (L1 colour distance, ray index) == "first minimum" of the
// sequential scan over s; occlusion -> smallest disparity.
__global__ __launch_bounds__(256) void k_interpolate_rays(const int32_t* __restrict__ list, const int32_t* __restrict__ counters,
                                                          const float* __restrict__ din, float* __restrict__ dout,
                                                          const uint8_t* __restrict__ img_l, const double* __restrict__ sincos,
                                                          int W, int H, int which, int max_search)
{
    const int n = counters[0];
    const int lane = threadIdx.x & 63;
    const int s = lane & 15;                       // ray index
    const int grp = (blockIdx.x * 256 + threadIdx.x) >> 4; // pixel slot
    const int ngrp = (gridDim.x * 256) >> 4;
    const double sina = sincos[2 * s], cosa = sincos[2 * s + 1];
    const bool mismatch = which == ADC_LABEL_MISMATCH;
    for (int e = grp; e < ((n + 3) & ~3); e += ngrp) { // whole waves iterate together (4 pixels per wave)
        const bool live = e < n;
        const int p = live ? list[e] : 0;
        const int y = p / W, x = p - y * W;
        float hit = ADC_INVALID_FLOAT; // first valid disparity along this ray
        int dist = 0x3fffffff;
        if (live) {
            for (int m = 1; m < max_search; m++) {
                const int yy = (int)lround((double)y + (double)m * sina); // multistep_refiner.cpp:259-260
                const int xx = (int)lround((double)x + (double)m * cosa);
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) break;
                const float d = din[(size_t)yy * W + xx];
                if (d != ADC_INVALID_FLOAT) {
                    hit = d;
                    if (mismatch) dist = adc_color_dist_l1(img_l + (size_t)p * 3, img_l + ((size_t)yy * W + xx) * 3);
                    break;
                }
            }
        }
        // combine the 16 rays of this pixel
        float best;
        bool any;
        if (mismatch) { // colour-nearest, first minimum in ray order (multistep_refiner.cpp:276-289; min_dist starts at 9999)
            int key = (hit != ADC_INVALID_FLOAT && dist < 9999) ? dist * 16 + s : 0x7fffffff;
            float val = hit;
#pragma unroll
            for (int msk = 8; msk >= 1; msk >>= 1) {
                const int ok = __shfl_xor(key, msk, 64);
                const float ov = __shfl_xor(val, msk, 64);
                if (ok < key) { key = ok; val = ov; }
            }
            // the reference keeps d = 0.0f when every collected distance is >= 9999 (impossible: max 765)
            unsigned long long hb = __ballot(hit != ADC_INVALID_FLOAT);
            any = ((hb >> (lane & 48)) & 0xffffull) != 0;
            best = key != 0x7fffffff ? val : 0.0f;
        } else { // smallest disparity (multistep_refiner.cpp:290-296)
            float val = hit;
#pragma unroll
            for (int msk = 8; msk >= 1; msk >>= 1) {
                const float ov = __shfl_xor(val, msk, 64);
                val = ov < val ? ov : val;
            }
            any = val != ADC_INVALID_FLOAT;
            best = val;
        }
        if (live && s == 0) dout[e] = any ? best : 0.0f; // fill value of list entry e; no ray hit: value-initialised fill (multistep_refiner.cpp:246,270-272)
    }
}

// Table variant (default): the reference's step m of ray s lands on (lround(y + m*sin), lround(x + m*cos))
// (multistep_refiner.cpp:259-260).  For integer y that equals y + lround(m*sin) unless m*sin sits within
// rounding distance of a .5 tie, which the host excludes when it builds the table (capi.hip: every entry is
// checked to be > 1e-9 away from a tie; otherwise the f64 kernel above is used).  Table layout [m][16] of
// packed (dy << 16 | dx & 0xffff): one 64-byte segment per step for the 16 rays of a pixel.  The walk issues
// 4 steps' loads at once (the loads past the hit are clamped in-image and ignored).
__global__ void k_pack_bgr(const uint8_t* __restrict__ img, uint32_t* __restrict__ out, int P); // k_arms.hip
__device__ __forceinline__ int packed_l1(uint32_t a, uint32_t b) // adc_color_dist_l1 on packed B | G<<8 | R<<16 (top byte 0 in both): ONE v_sad_u8
{
    return (int)__builtin_amdgcn_sad_u8(a, b, 0u);
}

// 16 lanes per target pixel (one ray each), 4 list-consecutive (mostly x-adjacent) pixels per wave.  Lane layout
// lane = ray*4 + pixel: the texture-address unit merges adjacent lanes that hit the same line, so the 4 pixels'
// steps of one ray cost one line, not four (the kernel is gather-rate bound: ~78% of a noise pair's pixels are
// targets, with ~4.5 steps per ray).  Offsets of steps 1..4 live in registers, the next list entries are
// prefetched, 4 steps' disparities are fetched per round trip (loads past the hit are clamped to the pixel itself
// and ignored), the colour of the hit is fetched once at the end.
// Empty-space skipping.  Most of the walk's steps are spent where there is next to nothing to find: in the band of columns
// x < D of the left view 99.7 % of the pixels are invalid, and 70 % of all ray steps of a noise pair are taken inside it
// (~90 per ray against ~8 elsewhere).  cdist[cell] = a LOWER BOUND of the Chebyshev distance, in cells of 2x2 pixels, from
// the cell to the nearest cell that holds a valid pixel (0 = the cell itself, search window +-ADC_ITP_CAP cells, ADC_ITP_CAP + 1 =
// "further").  A ray standing in a cell with cdist = c >= 2 cannot meet a valid pixel during its next (c - 1) * 2 - 1 steps
// -- a step moves at most one pixel per axis (+1 for the rounding) -- so they are skipped.  Exact: only pixels proven
// invalid (or outside the image, where the ray ends anyway) are passed over.
#define ITP_CELL ADC_ITP_CELL
// cell / row / distance maps (one byte per cell each), then -- 16-byte aligned -- the padded code map of the walk
// (per pass: cell / row / distance maps; behind both sets -- 16-byte aligned -- the two padded code maps of the walk)
static size_t itp_cell_stride(int W, int H) { return 3 * (size_t)((W + ITP_CELL - 1) / ITP_CELL) * ((H + ITP_CELL - 1) / ITP_CELL); }
static size_t itp_code_offset(int W, int H) { return (2 * itp_cell_stride(W, H) + 15) & ~(size_t)15; }
static size_t itp_code_stride(int W, int H, int ms) { return ((size_t)adc_itp_code_pitch(W, ms) * adc_itp_code_rows(H, ms) + 64 + 15) & ~(size_t)15; }
size_t adc_itp_cell_bytes(int W, int H, int ms) { return itp_code_offset(W, H) + 2 * itp_code_stride(W, H, ms) + 64; }
// Validity of both interpolation passes in one set of launches (round 6): the second pass sees the first pass's fills, and EVERY target
// of the first pass is filled (a pixel no ray finds a value for gets 0.0f: multistep_refiner.cpp:246,270-272) -- so the second pass's
// validity = valid pixels + invalid pixels of the mismatch list, known before the first pass runs.  Map 1 of every kernel (blockIdx.y)
// is the second pass's.
__device__ __forceinline__ bool itp_valid(const float* __restrict__ disp, const uint8_t* __restrict__ label, size_t p, int pass2)
{
    return disp[p] != ADC_INVALID_FLOAT || (pass2 && label[p] == ADC_LABEL_MISMATCH);
}
__global__ __launch_bounds__(256) void k_itp_cells(const float* __restrict__ disp, const uint8_t* __restrict__ label, uint8_t* __restrict__ cell, int W, int H, int cw, int ch, size_t map_stride)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cw * ch) return;
    const int cy = c / cw, cx = c - cy * cw;
    bool any = false;
#pragma unroll
    for (int r = 0; r < ITP_CELL; r++)
#pragma unroll
        for (int q = 0; q < ITP_CELL; q++) {
            const int y = cy * ITP_CELL + r, x = cx * ITP_CELL + q;
            if (y < H && x < W) any = any || itp_valid(disp, label, (size_t)y * W + x, (int)blockIdx.y);
        }
    cell[blockIdx.y * map_stride + c] = any ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_itp_rowdist(const uint8_t* __restrict__ cell, uint8_t* __restrict__ rowd, int cw, int ch, size_t map_stride)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cw * ch) return;
    rowd[blockIdx.y * map_stride + c] = (uint8_t)adc_itp_rowdist(cell + blockIdx.y * map_stride, cw, c % cw, c / cw);
}
__global__ __launch_bounds__(256) void k_itp_coldist(const uint8_t* __restrict__ rowd, uint8_t* __restrict__ cdist, int cw, int ch, size_t map_stride)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cw * ch) return;
    cdist[blockIdx.y * map_stride + c] = (uint8_t)adc_itp_coldist(rowd + blockIdx.y * map_stride, cw, ch, c % cw, c / cw);
}

// The code map of the walk (adc_device_fn.h): four pixels of an image row per thread, one 32-bit store; the padding around the image
// (ADC_ITP_OUTSIDE) is written once, when the object is created.  `code` points at the map's row 0, `gx` = columns of padding on the left.
__global__ __launch_bounds__(256) void k_itp_code(const float* __restrict__ disp, const uint8_t* __restrict__ label, const uint8_t* __restrict__ cdist,
                                                  uint8_t* __restrict__ code, int W, int H, int cw, int pitch, int gx, size_t cell_stride, size_t code_stride)
{
    cdist += blockIdx.y * cell_stride;
    code += blockIdx.y * code_stride;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int q4 = (W + 3) >> 2; // threads per row; thread (y, k) covers the padded columns (gx & ~3) + 4k .. + 3
    if (i >= q4 * H + H) return;
    const int y = i / (q4 + 1), k = i - y * (q4 + 1);
    const int c0 = (gx & ~3) + 4 * k;
    uint32_t v = 0u;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int x = c0 + b - gx;
        uint32_t c = ADC_ITP_OUTSIDE;
        if (x >= 0 && x < W) c = itp_valid(disp, label, (size_t)y * W + x, (int)blockIdx.y) ? (uint32_t)ADC_ITP_VALID : (uint32_t)adc_itp_skip(cdist[(y / ITP_CELL) * cw + (x / ITP_CELL)]);
        v |= c << (8 * b);
    }
    *reinterpret_cast<uint32_t*>(code + (size_t)y * pitch + c0) = v; // (pitch and c0 are multiples of 4; c0 + 3 <= gx + W + 6 < pitch: adc_itp_code_pitch)
}

// Round 6: the walk on the CODE MAP (adc_device_fn.h).  SQ counters of the round-5 kernel said 11 400 vector-ALU instructions per wave
// against 1 000 vector loads: with 8 waves per SIMD that is 0.17 of its 0.28 ms in instruction issue alone -- table row, two sign
// extensions, four bounds tests, a multiply-add, 64-bit addresses and the hit bookkeeping per step, ~25 instructions.  Now a step is
// one LDS read of the linear offset (immediate offsets from one address per trip), one add, one byte gather from the padded map (no
// bounds tests: outside the image the map says so; no separate cell-distance gather: the code of an invalid pixel IS its skip) and the
// end test; the value and the colour of the hit are fetched once behind the walk.  (A first attempt of this round replaced the map
// gathers by bit tests on 8x8 validity tiles -- fewer gathers, MORE instructions: 0.28 -> 0.44 ms, profiles/r6_ab_k9_code_map.txt.)
// One trip of N steps of a ray: m = its next step, ho = the linear offset of its hit (ITP_NO_HIT: none so far).  Rays that have ended take no part in the
// gathers (execution mask): the address unit's time goes with the lanes it serves -- measured, 0.235 -> 0.14 ms per list on the noise pair.
template <int N>
__device__ __forceinline__ void itp_trip(const int32_t* __restrict__ lt, const uint8_t* __restrict__ code, uint32_t pb, int s, int max_search,
                                         bool& walking, int& m, int& ho)
{
    const bool w0 = walking && m < max_search;
    uint32_t c[N];
    int o[N];
#pragma unroll
    for (int j = 0; j < N; j++) { c[j] = ADC_ITP_OUTSIDE; o[j] = 0; }
    if (w0) {
        const int32_t* row = lt + (m * 16 + s);
#pragma unroll
        for (int j = 0; j < N; j++) { o[j] = row[j * 16]; c[j] = code[pb + (uint32_t)o[j]]; }
    }
    bool act = w0;
    const int left = max_search - m; // steps left in the search range (> 0 for a walking ray)
#pragma unroll
    for (int j = 0; j < N; j++) {
        const bool end = c[j] >= ADC_ITP_OUTSIDE || j >= left; // a hit, the image border or the end of the range
        ho = (act && end && c[j] == ADC_ITP_VALID && j < left) ? o[j] : ho; // (the hit's linear offset in the padded map)
        act = act && !end;
    }
    walking = act;
    m += act ? N + (int)c[N - 1] : 0; // (the code of the trip's last position: an invalid pixel inside the image = its skip)
}
// Trip lengths (same box, interleaved, refine stage of the noise pair, profiles/r6_ab_k9_code_map.txt): first trip / following trips
// 2 / 4: 0.727 ms, 4 / 4: 0.732, 3 / 6: 0.732, 3 / 3: 0.742, 4 / 8: 0.747, 2 / 8: 0.751, 6 / 8: 0.760 (round 5's kernel: 0.99) -- at 22 % valid
// pixels half of the rays end within two steps, and what a longer trip fetches behind the hit is wasted address-unit time.
#define ITP_NO_HIT 0x7fffffff
#ifndef ITP_NS1
#define ITP_NS1 2 // steps of a ray's first trip ...
#endif
#ifndef ITP_NS2
#define ITP_NS2 4 // ... and of the following ones (<= ADC_ITP_NS: the table's padding)
#endif
#ifndef ITP_NS3
#define ITP_NS3 0 // > 0: steps of the trips from the third one on (2 / 4 / 6 and 2 / 4 / 8 measured: 0.538 - 0.547 ms against 0.540 - 0.542: no difference)
#endif
static_assert(ITP_NS1 <= ADC_ITP_NS && ITP_NS2 <= ADC_ITP_NS && ITP_NS3 <= ADC_ITP_NS, "trip lengths");
// K targets per lane and iteration (ITP_K).  After the code map the walk waits for memory three quarters of its time (SQ counters:
// wait_any 0.73, vector ALU 0.10 busy) at the occupancy limit of 8 waves per SIMD, so K = 2 .. 4 independent targets per lane were
// tried to overlap their round trips: SLOWER (refine stage of the noise pair 0.531 / 0.557 / 0.582 / 0.657 ms for K = 1 / 2 / 3 / 4,
// profiles/r6_ab_k9_code_map.txt) -- the trips of a wave then run until the longest of K x 64 rays has ended, and what the walk waits
// for is the address unit, not latency.  K = 1.
#ifndef ITP_K
#define ITP_K 1
#endif
template <int N, int K>
__device__ __forceinline__ void itp_trips(const int32_t* __restrict__ lt, const uint8_t* __restrict__ code, const uint32_t (&pb)[K], int s, int max_search,
                                          bool (&walking)[K], int (&m)[K], int (&ho)[K])
{
    uint32_t c[K][N];
    int o[K][N];
    bool w0[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        w0[k] = walking[k] && m[k] < max_search;
#pragma unroll
        for (int j = 0; j < N; j++) { c[k][j] = ADC_ITP_OUTSIDE; o[k][j] = 0; }
    }
#pragma unroll
    for (int k = 0; k < K; k++)
        if (w0[k]) { // (rays that have ended take no part in the gathers)
            const int32_t* row = lt + (m[k] * 16 + s);
#pragma unroll
            for (int j = 0; j < N; j++) { o[k][j] = row[j * 16]; c[k][j] = code[pb[k] + (uint32_t)o[k][j]]; }
        }
#pragma unroll
    for (int k = 0; k < K; k++) {
        bool act = w0[k];
        const int left = max_search - m[k]; // steps left in the search range (> 0 for a walking ray)
#pragma unroll
        for (int j = 0; j < N; j++) {
            const bool end = c[k][j] >= ADC_ITP_OUTSIDE || j >= left; // a hit, the image border or the end of the range
            ho[k] = (act && end && c[k][j] == ADC_ITP_VALID && j < left) ? o[k][j] : ho[k]; // (the hit's linear offset in the padded map)
            act = act && !end;
        }
        walking[k] = act;
        m[k] += act ? N + (int)c[k][N - 1] : 0; // (the code of the trip's last position: an invalid pixel inside the image = its skip)
    }
}
template <bool TAB_LDS>
__global__ __launch_bounds__(256) void k_interpolate_tab(const int32_t* __restrict__ list, const int32_t* __restrict__ counters,
                                                         float* dmap /* read at the hits, written at the targets: IN PLACE (see the launch) */,
                                                         const uint32_t* __restrict__ bgr, const int32_t* __restrict__ lin,
                                                         int W, int which, int max_search, const uint8_t* __restrict__ code, int pitch, int gx, float rcp_pitch)
{
    constexpr int K = ITP_K;
    extern __shared__ int32_t lin_lds[]; // [max_search + ADC_ITP_LPAD][16]
    if (TAB_LDS) {
        for (int i = threadIdx.x; i < (max_search + ADC_ITP_LPAD) * 16; i += 256) lin_lds[i] = lin[i];
        __syncthreads();
    }
    const int32_t* __restrict__ lt = TAB_LDS ? lin_lds : lin;
    const int n = counters[0];
    const int lane = threadIdx.x & 63;
    const int s = lane >> 2;                                   // ray index
    const int slot = ((blockIdx.x * 256 + threadIdx.x) >> 6) * 4 + (lane & 3); // pixel slot
    const int nslot = (gridDim.x * 256) >> 4;
    const bool mismatch = which == ADC_LABEL_MISMATCH;
    const int nr = (n + 3) & ~3; // whole waves iterate together (4 pixels per wave and target set)
    int pn[K];
#pragma unroll
    for (int k = 0; k < K; k++) pn[k] = slot + k * nslot < n ? list[slot + k * nslot] : 0;
    for (int e0 = slot; e0 < nr; e0 += K * nslot) {
        bool live[K], walking[K];
        int p[K], m[K], ho[K];
        uint32_t pb[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int e = e0 + k * nslot;
            live[k] = e < n;
            p[k] = pn[k];
            pn[k] = e + K * nslot < n ? list[e + K * nslot] : 0;
            const int y = p[k] / W, x = p[k] - y * W;
            pb[k] = (uint32_t)(y * pitch + x + gx); // the target in the padded map
            ho[k] = ITP_NO_HIT;                      // linear offset of the hit in the padded map
            walking[k] = live[k];
        }
        // the ray's own step counter: next step to evaluate; the target's own code = the steps proven empty around it (a listed
        // pixel is invalid and inside the image)
#pragma unroll
        for (int k = 0; k < K; k++) m[k] = 1 + (int)code[pb[k]];
        itp_trips<ITP_NS1, K>(lt, code, pb, s, max_search, walking, m, ho);
        while (true) {
            bool more = false;
#pragma unroll
            for (int k = 0; k < K; k++) more = more || (walking[k] && m[k] < max_search);
            if (!__any(more)) break;
            itp_trips<ITP_NS2, K>(lt, code, pb, s, max_search, walking, m, ho);
        }
        // the hit: ho = dy * pitch + dx with 0 <= dy and |dx| < gx <= pitch / 2, so dy = (ho + gx) / pitch -- by a float reciprocal
        // and one correction step (ho + gx < 2^24: exact in float; the quotient is off by at most one) -- and its value from the map
        int hitq[K];
        float hv[K];
        uint32_t c_own[K], c_hit[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool found = ho[k] != ITP_NO_HIT;
            const int a = found ? ho[k] + gx : 0;
            int dy = (int)((float)a * rcp_pitch);
            const int r = a - dy * pitch;
            dy += (r >= pitch ? 1 : 0) - (r < 0 ? 1 : 0);
            hitq[k] = found ? p[k] + dy * W + (ho[k] - dy * pitch) : p[k];
            hv[k] = dmap[hitq[k]];
            if (mismatch) { c_own[k] = bgr[p[k]]; c_hit[k] = bgr[hitq[k]]; }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            const float hit = ho[k] != ITP_NO_HIT ? hv[k] : ADC_INVALID_FLOAT; // first valid disparity along this ray
            // combine the 16 rays of this pixel (lanes with equal lane&3)
            float best;
            bool any;
            if (mismatch) { // colour-nearest, first minimum in ray order (multistep_refiner.cpp:276-289; min_dist starts at 9999)
                const int dist = packed_l1(c_own[k], c_hit[k]); // <= 765 < 9999
                int key = hit != ADC_INVALID_FLOAT ? dist * 16 + s : 0x7fffffff;
                float val = hit;
#pragma unroll
                for (int msk = 32; msk >= 4; msk >>= 1) {
                    const int ok = __shfl_xor(key, msk, 64);
                    const float ov = __shfl_xor(val, msk, 64);
                    if (ok < key) { key = ok; val = ov; }
                }
                any = key != 0x7fffffff;
                best = any ? val : 0.0f;
            } else { // smallest disparity (multistep_refiner.cpp:290-296)
                float val = hit;
#pragma unroll
                for (int msk = 32; msk >= 4; msk >>= 1) {
                    const float ov = __shfl_xor(val, msk, 64);
                    val = ov < val ? ov : val;
                }
                any = val != ADC_INVALID_FLOAT;
                best = val;
            }
            if (live[k] && s == 0) dmap[p[k]] = any ? best : 0.0f; // the fill, in place; no ray hit: value-initialised fill (multistep_refiner.cpp:246,270-272)
        }
    }
}

// deferred write-back of a list (multistep_refiner.cpp:298-303): disp[list[e]] = fill[e]
__global__ __launch_bounds__(256) void k_interp_scatter(const int32_t* __restrict__ list, const int32_t* __restrict__ counters,
                                                        const float* __restrict__ fill, float* __restrict__ disp)
{
    const int n = counters[0];
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) disp[list[e]] = fill[e];
}

hipError_t adc_launch_interpolation(adc_handle* h)
{
    const AdcParams& p = h->p;
    const int P = p.W * p.H;
    const int dmaxa = p.dmax < 0 ? -p.dmax : p.dmax, dmina = p.dmin < 0 ? -p.dmin : p.dmin;
    const int max_search = dmaxa > dmina ? dmaxa : dmina; // multistep_refiner.cpp:236
    if (h->ray_tab && max_search == h->ray_tab_rows && max_search == h->itp_ms && (size_t)h->itp_ms * (size_t)h->itp_pitch < ((size_t)1 << 24) && // (linear offsets exact in float: see the kernel)
        (size_t)h->itp_pitch * (size_t)adc_itp_code_rows(p.H, h->itp_ms) < ((size_t)1 << 31)) { // (32-bit positions in the padded map)
        // Round 6: ONE list pass, ONE set of map launches for both passes, fills written IN PLACE.  The reference computes the fill
        // values of a whole list from the UNCHANGED map and writes them back afterwards (fill_disps, multistep_refiner.cpp:246-303); a
        // walk here never reads the map at a pixel its own pass writes: it tests the CODE map (a snapshot) and fetches map values only at
        // hits, which are valid pixels of that snapshot -- targets are invalid in it.  The second pass's snapshot counts the first
        // pass's targets as valid (all of them get filled), and its walk runs behind the first one: it reads their fills.
        hipError_t e;
        if ((e = hipMemsetAsync(h->interp_counters, 0, 8 * sizeof(int32_t), h->stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_interp_list2, dim3((P + 256 * ITP_LIST_PPT - 1) / (256 * ITP_LIST_PPT)), dim3(256), 0, h->stream, h->label, h->disp_l,
                           h->interp_list, h->interp_list + P, h->interp_counters, P);
        if (!h->bgrx_valid) hipLaunchKernelGGL(k_pack_bgr, dim3((P + 255) / 256), dim3(256), 0, h->stream, h->img_l, h->bgrx_l, P);
        const int cw = (p.W + ITP_CELL - 1) / ITP_CELL, ch = (p.H + ITP_CELL - 1) / ITP_CELL, nc = cw * ch;
        const size_t cs = itp_cell_stride(p.W, p.H), ks = itp_code_stride(p.W, p.H, h->itp_ms);
        uint8_t* code = h->itp_cells + itp_code_offset(p.W, p.H);
        hipLaunchKernelGGL(k_itp_cells, dim3((nc + 255) / 256, 2), dim3(256), 0, h->stream, h->disp_l, h->label, h->itp_cells, p.W, p.H, cw, ch, cs);
        hipLaunchKernelGGL(k_itp_rowdist, dim3((nc + 255) / 256, 2), dim3(256), 0, h->stream, h->itp_cells, h->itp_cells + nc, cw, ch, cs);
        hipLaunchKernelGGL(k_itp_coldist, dim3((nc + 255) / 256, 2), dim3(256), 0, h->stream, h->itp_cells + nc, h->itp_cells + 2 * nc, cw, ch, cs);
        hipLaunchKernelGGL(k_itp_code, dim3((((p.W + 3) / 4 + 1) * p.H + 255) / 256, 2), dim3(256), 0, h->stream, h->disp_l, h->label, h->itp_cells + 2 * (size_t)nc,
                           code, p.W, p.H, cw, h->itp_pitch, h->itp_ms, cs, ks);
        const size_t lin_bytes = (size_t)(max_search + ADC_ITP_LPAD) * 16 * sizeof(int32_t);
        const bool tab_lds = lin_bytes <= 40 * 1024; // (ranges up to 576; larger ones read the table from global memory)
        for (int k = 0; k < 2; k++) {
            const int which = k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION;
#define INTERP_TAB(L_)                                                                                                 \
    hipLaunchKernelGGL((k_interpolate_tab<L_>), dim3(2048), dim3(256), L_ ? lin_bytes : 0, h->stream, h->interp_list + (size_t)k * P, h->interp_counters + k, \
                       h->disp_l, h->bgrx_l, h->ray_lin, p.W, which, max_search, code + k * ks, h->itp_pitch, h->itp_ms, 1.0f / (float)h->itp_pitch)
            if (tab_lds) INTERP_TAB(true);
            else INTERP_TAB(false);
#undef INTERP_TAB
        }
        return hipGetLastError();
    }
    for (int k = 0; k < 2; k++) { // (a ray step too close to a rounding tie for the integer tables: the f64 walk, list by list)
        const int which = k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION;
        hipError_t e;
        if ((e = hipMemsetAsync(h->interp_counters, 0, 8 * sizeof(int32_t), h->stream)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_interp_list, dim3((P + 256 * ITP_LIST_PPT - 1) / (256 * ITP_LIST_PPT)), dim3(256), 0, h->stream, h->label, h->disp_l,
                           h->interp_list, h->interp_counters, which, P);
        // fill[e] per list entry (disp_tmp serves as the array), then a scatter into the map in place: no copy of the map, no buffer swap
        hipLaunchKernelGGL(k_interpolate_rays, dim3(2048), dim3(256), 0, h->stream, h->interp_list, h->interp_counters, h->disp_l,
                           h->disp_tmp, h->img_l, h->ray_sincos, p.W, p.H, which, max_search);
        hipLaunchKernelGGL(k_interp_scatter, dim3(1024), dim3(256), 0, h->stream, h->interp_list, h->interp_counters, h->disp_tmp, h->disp_l);
    }
    return hipGetLastError();
}

// -------------------------------------------------------------- K10 discontinuity adjustment (off by default)
__global__ __launch_bounds__(256) void k_edge_detect(const float* __restrict__ dp, uint8_t* __restrict__ edge, int W, int H,
                                                     float threshold)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    uint8_t e = 0;
    if (x >= 1 && x < W - 1 && y >= 1 && y < H - 1) {
#define DP(yy, xx) dp[(size_t)(yy)*W + (xx)]
        const float gx = (-DP(y - 1, x - 1) + DP(y - 1, x + 1)) + (-2 * DP(y, x - 1) + 2 * DP(y, x + 1)) +
                         (-DP(y + 1, x - 1) + DP(y + 1, x + 1));
        const float gy = (-DP(y - 1, x - 1) - 2 * DP(y - 1, x) - DP(y - 1, x + 1)) +
                         (DP(y + 1, x - 1) + 2 * DP(y + 1, x) + DP(y + 1, x + 1));
#undef DP
        if (fabsf(gx) + fabsf(gy) > threshold) e = 1;
    }
    edge[(size_t)y * W + x] = e;
}

// One thread per row: the fix-up is sequential along x (multistep_refiner.cpp:322-350).  The cost index
// is lround(d) WITHOUT "- min_disparity" exactly like the reference (:329-331,:340): with min_disparity != 0 that
// index leaves the pixel's own cost row and lands in a neighbouring pixel's (flat [H][W][D] addressing), so the
// reference's flat element index is mapped back to (pixel, d) of the padded layout.  Indices outside the volume
// (undefined behaviour in the reference; cannot occur on edge pixels, which have 1 <= y <= H-2) are clamped.
__device__ __forceinline__ float dda_cost(const float* __restrict__ vol, long long flat, long long total, int D, int Dp)
{
    flat = flat < 0 ? 0 : (flat >= total ? total - 1 : flat);
    const long long pix = flat / D;
    return vol[pix * Dp + (flat - pix * D)];
}
__global__ void k_discontinuity_rows(float* __restrict__ disp, const uint8_t* __restrict__ edge, const float* __restrict__ vol,
                                     int W, int H, int D, int Dp)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= H) return;
    float* row = disp + (size_t)y * W;
    const long long total = (long long)W * H * D;
    for (int x = 1; x < W - 1; x++) {
        if (edge[(size_t)y * W + x] != 1) continue;
        const float d = row[x];
        if (d == ADC_INVALID_FLOAT) continue;
        const long di = lroundf(d);
        const long long base = ((long long)y * W + x) * D; // cost_ptr of the reference (:330)
        float c0 = dda_cost(vol, base + di, total, D, Dp);
        for (int k = 0; k < 2; k++) {
            const int x2 = k == 0 ? x - 1 : x + 1;
            const float d2 = row[x2];
            if (d2 == ADC_INVALID_FLOAT) continue;
            const long d2i = lroundf(d2);
            const float c = dda_cost(vol, base + (k == 0 ? -(long long)D : (long long)D) + d2i, total, D, Dp);
            if (c < c0) { row[x] = d2; c0 = c; }
        }
    }
}

hipError_t adc_launch_discontinuity(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_edge_detect, grid, block, 0, h->stream, h->disp_l, h->edge, p.W, p.H, 5.0f);
    hipLaunchKernelGGL(k_discontinuity_rows, dim3((p.H + 63) / 64), dim3(64), 0, h->stream, h->disp_l, h->edge, h->vol_a, p.W,
                       p.H, p.D, p.Dp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- K11 median
// In-place raster 3x3 median == recursive filter: the window of (x,y) holds already-filtered values
// at (x-1..x+1, y-1) and (x-1, y).  All pixels with equal t = x + 2y are independent; ONE workgroup
// walks t = 0 .. W-1+2(H-1) with one barrier per level.  Thread i owns RPT consecutive rows.
//   * filtered values travel through a 4-deep LDS ring per row (ring[y][x&3]); a row reads 1 new
//     filtered value of the row above per level and shifts the other two in registers;
//   * unfiltered values (rows y and y+1, never modified: the output goes to a second buffer) slide
//     through registers; the 2 new ones per level are prefetched MED_K levels ahead, so no global
//     load sits on the per-level critical path.
#define MED_K 4

template <int RPT>
__global__ __launch_bounds__(1024) void k_median_wavefront(const float* __restrict__ in, float* __restrict__ out, int W, int H)
{
    extern __shared__ __attribute__((aligned(16))) float mring[]; // [H][4]
    const int tid = threadIdx.x;
    const int nsteps = W + 2 * (H - 1);
    auto clampc = [&](int c) __attribute__((always_inline)) { return c < 0 ? 0 : (c >= W ? W - 1 : c); };

    // per owned row r: y = tid*RPT + r.  Column of row y at level t: x = t - 2y.
    float A0[RPT], A1[RPT];          // in[y][x], in[y][x+1]
    float Bm[RPT], B0[RPT], B1[RPT]; // in[y+1][x-1], [x], [x+1]
    float Fm[RPT], F0[RPT];          // filtered out[y-1][x-1], out[y-1][x]
    float Pv[RPT];                   // filtered out[y][x-1]
    float fa[RPT][MED_K], fb[RPT][MED_K]; // prefetched columns x+2 .. x+1+MED_K of rows y, y+1
    const float* rowA[RPT];
    const float* rowB[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const int y = tid * RPT + r;
        const int ya = y < H ? y : H - 1, yb = y + 1 < H ? y + 1 : H - 1;
        rowA[r] = in + (size_t)ya * W;
        rowB[r] = in + (size_t)yb * W;
        const int x0 = -2 * y; // column at level 0
        A0[r] = rowA[r][clampc(x0)];
        A1[r] = rowA[r][clampc(x0 + 1)];
        Bm[r] = rowB[r][clampc(x0 - 1)];
        B0[r] = rowB[r][clampc(x0)];
        B1[r] = rowB[r][clampc(x0 + 1)];
        Fm[r] = F0[r] = Pv[r] = 0.0f; // columns < 0: never used (validity is decided by index)
#pragma unroll
        for (int k = 0; k < MED_K; k++) {
            fa[r][k] = rowA[r][clampc(x0 + 2 + k)];
            fb[r][k] = rowB[r][clampc(x0 + 2 + k)];
        }
    }

    for (int t0 = 0; t0 < nsteps; t0 += MED_K) {
#pragma unroll
        for (int k = 0; k < MED_K; k++) {
            const int t = t0 + k;
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int y = tid * RPT + r;
                const int x = t - 2 * y;
                // next level's new unfiltered column (x+2) was prefetched MED_K levels ago; refill the slot
                const float na = fa[r][k], nb = fb[r][k];
                fa[r][k] = rowA[r][clampc(x + 2 + MED_K)];
                fb[r][k] = rowB[r][clampc(x + 2 + MED_K)];
                const bool active = (t < nsteps) && (y < H) && (x >= 0) && (x < W);
                // newest filtered value of the row above: out[y-1][x+1] was produced at level t-1
                float F1 = 0.0f;
                if (y > 0 && y < H && x + 1 >= 0 && x + 1 < W) F1 = mring[(y - 1) * 4 + ((x + 1) & 3)];
                if (active) {
                    const bool up = y > 0, dn = y + 1 < H, lf = x > 0, rt = x + 1 < W;
                    float v[9];
                    int n = 1;
                    v[0] = (up && lf) ? Fm[r] : ADC_INVALID_FLOAT; n += (up && lf);
                    v[1] = up ? F0[r] : ADC_INVALID_FLOAT;          n += up;
                    v[2] = (up && rt) ? F1 : ADC_INVALID_FLOAT;     n += (up && rt);
                    v[3] = lf ? Pv[r] : ADC_INVALID_FLOAT;          n += lf;
                    v[4] = A0[r];
                    v[5] = rt ? A1[r] : ADC_INVALID_FLOAT;          n += rt;
                    v[6] = (dn && lf) ? Bm[r] : ADC_INVALID_FLOAT;  n += (dn && lf);
                    v[7] = dn ? B0[r] : ADC_INVALID_FLOAT;          n += dn;
                    v[8] = (dn && rt) ? B1[r] : ADC_INVALID_FLOAT;  n += (dn && rt);
                    adc_sort9(v); // +inf padding sorts to the end
                    const int sel = n / 2; // wnd_data[size/2], adcensus_util.cpp:77
                    float res = v[0];
#pragma unroll
                    for (int i = 1; i < 9; i++) res = (i == sel) ? v[i] : res;
                    out[(size_t)y * W + x] = res;
                    mring[y * 4 + (x & 3)] = res;
                    Pv[r] = res;
                }
                // slide the register windows one column to the right
                Fm[r] = F0[r]; F0[r] = F1;
                A0[r] = A1[r]; A1[r] = na;
                Bm[r] = B0[r]; B0[r] = B1[r]; B1[r] = nb;
            }
            __syncthreads();
        }
    }
}

// ---- multi-workgroup variant: bands of 64 rows, ONE WAVE (one lane per row) per band.
// Inside a band everything stays in registers: the newest filtered value of the row above (out[y-1][x+1],
// produced by lane-1 one level earlier) arrives with a DPP wave_shr:1, the unfiltered row below (in[y+1][x+1])
// is the value lane+1 prefetched for its own window (DPP wave_shl:1), so there is no LDS and no barrier, and a
// level costs ~45 VALU instructions (rank selection by adc_median9 with +-inf padding instead of a 25-exchange
// sort).  Requires W >= 2 and H >= 2.
//
// Band b's first row needs the filtered last row of band b-1.  Device-scope (sc1) stores are expensive here -- the
// bands sit on different XCDs, so such a store goes through to memory; 64 of them per level cost 3x the whole
// filter -- hence the map itself is written with plain stores (made visible by the end of the kernel) and only the
// band's LAST ROW is additionally written, 16 levels at a time, to a small hand-off buffer indexed by LEVEL
// (hand[band][t] = result of the last row at level t) with four 16-byte sc1 stores per block.  The downstream band
// reads hand[band-1][t-1] with sc1 loads one block ahead; the rows are reset to a sentinel before the launch and a band
// re-reads a block in which it still finds the sentinel (MEDB_RECHECK): the data is its own flag.  Dependencies only
// point upstream, all bands are co-resident (<= 256 single-wave workgroups), the re-reads are bounded (error word +
// bail out, reported by adc_wait, which then runs the single-workgroup kernel).
#define MEDB_ROWS 64
#define MEDB_K 16 // levels per block (the asm take-over statements are written for 16)
#define MEDB_HPAD 4 // hand[band][MEDB_HPAD + t]


// Chains of the speculative form (spec = R > 0): EVERY band b >= 1 takes its row above from a private chain of min(b, R) copies of the
// bands above it (round 6: also the bands 1 .. R, whose chains start with an exact copy of band 0 -- chained to the real bands above
// them they had to start with those bands' windows and run to the end of band R's, up to 830 levels where every other wave runs 574).
// Copies of all chains, in the order of their target band: medb_chain_off(b) = links in front of band b's chain.
__host__ __device__ __forceinline__ int medb_chain_depth(int b, int spec) { return b < spec ? b : spec; }
__host__ __device__ __forceinline__ int medb_chain_off(int b, int spec) // sum of min(t, spec) over t = 1 .. b - 1
{
    const int t = b - 1;
    return t <= spec ? t * (t + 1) / 2 : spec * (spec + 1) / 2 + (t - spec) * spec;
}
typedef float medb_v4f __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float medb_dpp(float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xf, 0xf, false));
}
// ... with the value a lane WITHOUT a source lane keeps (lane 0 of wave_shr:1, lane 63 of wave_shl:1)
template <int CTRL> __device__ __forceinline__ float medb_dpp_old(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}

typedef float medb_v2f __attribute__((ext_vector_type(2)));
// PAIRS (W even): a row's column has the parity of the level (x = t - 2y), so all lanes step from an even to an odd column
// together: the window prefetch and the map stores move TWO columns per instruction (8-byte, aligned stores).  With one
// lane per row every vector-memory instruction of this kernel touches 64 different cache lines; SQ counters of the
// one-column form: 35 % of the wave's cycles in s_waitcnt although the data is prefetched a block ahead, and removing a
// third of its vector instructions changed nothing -- it is bound by the rate at which the address coalescer takes those
// 64-line instructions.  Half as many of them: see profiles/README.md.  Odd widths run the one-column form.
//
// SPECULATIVE BANDS (round 4, spec = R > 0).  In the chained form band b trails band b-1, which trails band b-2, ...: 17 bands
// at 1080p, every one of them running all W + 2H levels -- 0.68 ms for an 8 MB map.  But the recursive filter forgets: bands
// that start from the UNFILTERED row above them (instead of the filtered one) produce, R bands further down, exactly the rows the
// true filter produces (measured on the CPU with the oracle's maps, tools/median_spec_bands.py, profiles/r4_median_spec_bands.txt:
// with a run-in of 128 rows no seam of any bench pair differs; with 64 rows 1 of 16 seams of ONE of the four 1080p noise pairs
// tried, by two pixels).  So the real band b > R takes its hand-off not from the real band b-1 but from a private CHAIN OF COPIES
// of the bands b-R .. b-1: the copy of band b-R takes the raw row above it, each further copy the hand-off of the copy before,
// all of them write nothing to the map (workgroups nbands ..: chain of target b = workgroups nbands + medb_chain_off(b) + j).
// The chains of the bands 1 .. R are shorter and start with an exact copy of band 0, which has no row above.  Every dependency chain is R + 1 waves long instead of 17,
// and a wave only runs the levels at which its rows (or the hand-off its successor re-checks) are active: ~2200 instead of 4078.
// k_median_seg_check then compares what the last copy of each chain handed over with what the real band b-1 wrote into the map -- bit
// for bit, every column; a difference raises the error word (2) and adc_wait redoes the filter in the chained form.
//
// SPECULATIVE COLUMN SEGMENTS (round 6, nseg > 1; needs spec > 0 and an even width).  With the chains every wave still runs W + ~200
// levels -- 0.27 ms at 1080p on 45 of the chip's 1024 SIMDs.  The filter forgets sideways as it forgets downwards
// (tools/median_spec_segments.py: with 128 columns of warm-up no seam of any bench pair differs), so every link is cut into nseg
// segments [xs, xe) (boundaries multiples of 16), one wave each: workgroup = link * nseg + segment.  The waves of a chain that serve
// segment s > 0 run ONE window of levels [ts, te):  ts = xs - warm + 2 * (first row of the chain's real band) -- the real band's top
// row starts `warm` columns in front of the segment, every row below it two columns further left --, te = when the real band's last row
// has passed column xe.  Below ts a lane passes the RAW value through as its result (whole blocks: ts is a multiple of 16): the warm-up
// starts from unfiltered pixels exactly like a chain starts from the raw row above it; a link 64 rows further up simply stands 128
// columns further right at the same level, so all links and segments run side by side and the hand-off protocol between a link and
// its upstream link (same segment) is the one of the whole-row form.  A real link writes the columns [xs, xe) of its rows and puts the
// last column pair of its warm-up (xs - 2, xs - 1) into its seam slot instead of the store sink.  k_median_seg_check then compares, bit
// for bit: the hand-off every real band b >= 1 consumed over the columns xs - 1 .. xe with the map row above it, and the seam column
// xs - 1 of every real segment s >= 1 with the map (where segment s - 1 wrote it) -- all equal => the map is the in-place filter
// (induction over bands and segments; CPU emulation: emul_median_spec_segments); a difference raises the error word like a failing
// row seam does, and adc_wait redoes the filter in the chained whole-row form.
template <bool PAIRS>
__global__ __launch_bounds__(MEDB_ROWS) void k_median_banded(const float* __restrict__ in, float* __restrict__ out, int W, int H,
                                                             int* progress, int* error_word, float* hand, int hpitch, int nbands, int spec,
                                                             float* sinks, int nseg, int warm, float* seam, int seg_shift)
{
    const int tid = threadIdx.x;
    if constexpr (!PAIRS) nseg = 1; // (the one-column form runs whole rows: everything below folds to the whole-row code)
    const int link = (int)blockIdx.x / nseg, seg = (int)blockIdx.x - link * nseg;
    const bool is_spec = link >= nbands; // a copy (writes no map): link ck of the chain of target band ct
    int ck = 0, ct = 0;
    if (is_spec) { // (the chains of the bands 1 .. spec are shorter: walk them; behind them every chain has spec links)
        const int c = link - nbands, tri = spec * (spec + 1) / 2;
        if (c < tri) { ct = 1; while (medb_chain_off(ct + 1, spec) <= c) ct++; ck = c - medb_chain_off(ct, spec); }
        else { ct = spec + 1 + (c - tri) / spec; ck = (c - tri) % spec; }
    }
    const int band = is_spec ? ct - medb_chain_depth(ct, spec) + ck : link; // the rows this wave filters
    const int myslot = (int)blockIdx.x;                          // hand-off row it publishes its last row into
    const bool raw_above = is_spec && ck == 0;                    // first link of a chain: the raw row above as its row above
    // hand-off row this wave reads: the link before it / for a real band the last link of its chain, or the real band above (same segment)
    const int upslot = (is_spec ? link - 1 : (spec ? nbands + medb_chain_off(band, spec) + medb_chain_depth(band, spec) - 1 : band - 1)) * nseg + seg;
    // the segment's columns, and the real band at the end of this wave's chain
    const int xs = adc_med_seg_x(W, nseg, seg, seg_shift), xe = adc_med_seg_x(W, nseg, seg + 1, seg_shift);
    const int chain_first = (is_spec ? ct : band) * MEDB_ROWS;
    const int chain_lastband = is_spec ? ct : band;
    const int chain_ylast = adc_imin((chain_lastband + 1) * MEDB_ROWS, H) - 1;
    const int ts = seg > 0 ? adc_imax(0, xs - warm + 2 * chain_first) : 0; // first level that is filtered (a multiple of 16)
    const int y = band * MEDB_ROWS + tid;
    const int nsteps = W + 2 * (H - 1);
    const int yfirst = band * MEDB_ROWS, ylast = adc_imin(yfirst + MEDB_ROWS, H) - 1; // rows of this band (wave-uniform)
    const bool row_ok = y < H;
    const bool up = y > 0, dn = y + 1 < H;
    const bool first_row = tid == 0 && band > 0;                  // reads the upstream band's last row
    const bool own_b = tid == MEDB_ROWS - 1;                      // no lane below: loads row y+1 itself
    const bool last_row = row_ok && (own_b || y == H - 1) && dn;  // feeds the downstream band
    auto clampc = [&](int c) __attribute__((always_inline)) { return c < 0 ? 0 : (c >= W ? W - 1 : c); };
    const int ya = row_ok ? y : H - 1, yb = y + 1 < H ? y + 1 : H - 1;
    const float* rowA = in + (size_t)ya * W;
    // one auxiliary load stream serves two lanes: lane 0 of a band > 0 reads the upstream hand-off row (by level), the
    // band's last lane reads the unfiltered row below (column x+1); every other lane reads dummy elements of `in`,
    // so the loads are unconditional yet cost a single cache line per instruction
    // (a speculative copy reads the RAW row above, indexed by level like a hand-off row: rowX[t] = in[y-1][t - 2y + 1])
    const float* rowX = first_row ? (raw_above ? in + (size_t)(y - 1) * W - 2 * y + 1 : hand + (size_t)upslot * hpitch + MEDB_HPAD - 1)
                                  : in + (own_b ? (size_t)yb * W : 0);
    // levels this wave runs: everything (chained form), or from TWO blocks before its first row becomes active -- the first
    // block is taken over in front of the loop, without a re-check, so it must not hold a level whose hand-off value is used:
    // the first one used is level 2 * yfirst - 1 (the value the first row needs as "above" on column 0), which sits in the
    // second block -- to the last level the band BELOW re-checks hand-off values for (its first row stands on its last column
    // at level W + 2 * ylast + 1 and takes over whole blocks)
    // (segments: not before two blocks in front of the window -- one is taken over without a re-check, one passes raw values through so
    // that the filter finds raw neighbours in its registers -- and, unless the segment ends at the image's right border, to the end of the
    // chain's window: the link below needs this one's last row that far)
    // (A head start of 32 / 48 / 96 levels per link for the links further up in a chain -- so that a reader never finds a hand-off block
    // unwritten and polls -- was measured: no difference, profiles/r6_ab_median_segments.txt.)
    const int tb_rows = spec ? adc_imax(0, 2 * yfirst - 2 * MEDB_K) & ~(MEDB_K - 1) : 0;
    const int tb = seg > 0 ? adc_imax(tb_rows, ts - 2 * MEDB_K) : tb_rows;
    const int te = (nseg > 1 && xe < W) ? adc_imin(nsteps, xe + 2 * chain_ylast + 3 * MEDB_K)
                                        : (spec ? adc_imin(nsteps, W + 2 * ylast + 3 * MEDB_K) : nsteps);
    const float PINF = ADC_INVALID_FLOAT, NINF = -ADC_INVALID_FLOAT;
    int x = tb - 2 * y; // column at the first level
    float A0 = rowA[clampc(x)], A1 = rowA[clampc(x + 1)], A2 = rowA[clampc(x + 2)];
    const float* rowBfull = in + (size_t)yb * W;
    float Bm = rowBfull[clampc(x - 1)], B0 = rowBfull[clampc(x)];
    float Fm = 0.f, F0 = 0.f, Pv = 0.f;
    // Prefetch, one block of MEDB_K levels ahead, with the loads issued from inline asm (the compiler's own model of
    // loop-carried outstanding loads forces near-complete drains of the memory counter, i.e. of the stores in
    // flight).  Every block issues EXACTLY, in this order: NL loads, NS map stores (inactive lanes store to a sink
    // word), 4 hand-off stores, 1 progress store (one column per instruction: NL = 32, NS = 16; PAIRS: 16 and 8);
    // vector-memory operations complete in issue order, so counted waits are exact:
    //   vmcnt(NS + 5) before a block         -> the loads issued one block earlier have landed;
    //   vmcnt(NL + NS + 5) before publishing -> the hand-off stores of the PREVIOUS block have completed (only the previous
    //                                  progress store, this block's loads and stores may be outstanding): "levels
    //                                  completed" is published one block late, so no store drain sits on the
    //                                  critical path.
    float ca[MEDB_K], cx[MEDB_K], hr[MEDB_K];
    float ra[PAIRS ? 1 : MEDB_K], rx[PAIRS ? 1 : MEDB_K];
    medb_v2f ra2a[PAIRS ? MEDB_K / 2 : 1], rx2a[PAIRS ? MEDB_K / 2 : 1];
    medb_v2f ca2[PAIRS ? MEDB_K / 2 : 1], cx2[PAIRS ? MEDB_K / 2 : 1];
// take over the prefetched values: wait + moves in ONE statement, nothing can be hoisted above the wait
#define MEDB_TAKE8(MOV, DST, SRC, O, WAIT)                                                                              \
    asm volatile(WAIT MOV " %0, %8\n\t" MOV " %1, %9\n\t" MOV " %2, %10\n\t" MOV " %3, %11\n\t"                          \
                      MOV " %4, %12\n\t" MOV " %5, %13\n\t" MOV " %6, %14\n\t" MOV " %7, %15"                            \
                 : "=&v"(DST[O]), "=&v"(DST[O + 1]), "=&v"(DST[O + 2]), "=&v"(DST[O + 3]), "=&v"(DST[O + 4]),           \
                   "=&v"(DST[O + 5]), "=&v"(DST[O + 6]), "=&v"(DST[O + 7])                                              \
                 : "v"(SRC[O]), "v"(SRC[O + 1]), "v"(SRC[O + 2]), "v"(SRC[O + 3]), "v"(SRC[O + 4]), "v"(SRC[O + 5]),    \
                   "v"(SRC[O + 6]), "v"(SRC[O + 7])                                                                     \
                 : "memory")
#define MEDB_TAKE(SET, WAIT1, WAIT2)                                                                                    \
    do {                                                                                                                \
        if constexpr (PAIRS) {                                                                                          \
            MEDB_TAKE8("v_mov_b64", ca2, ra2##SET, 0, WAIT2);                                                           \
            MEDB_TAKE8("v_mov_b64", cx2, rx2##SET, 0, "");                                                              \
            _Pragma("unroll") for (int q_ = 0; q_ < MEDB_K / 2; q_++) {                                                 \
                ca[2 * q_] = ca2[q_].x; ca[2 * q_ + 1] = ca2[q_].y;                                                     \
                cx[2 * q_] = cx2[q_].x; cx[2 * q_ + 1] = cx2[q_].y;                                                     \
            }                                                                                                           \
        } else {                                                                                                        \
            MEDB_TAKE8("v_mov_b32", ca, ra, 0, WAIT1);                                                                  \
            MEDB_TAKE8("v_mov_b32", ca, ra, 8, "");                                                                     \
            MEDB_TAKE8("v_mov_b32", cx, rx, 0, "");                                                                     \
            MEDB_TAKE8("v_mov_b32", cx, rx, 8, "");                                                                     \
        }                                                                                                               \
    } while (0)
// The loads of the next block of MEDB_K levels: running per-lane pointers + immediate offsets, NO column clamps (a
// clamp + 64-bit address per load was a fifth of the kernel's instructions).  Row y reads in[y][x+3 ..] with x = t - 2y:
// for x < 0 that is data of earlier rows (the lane is idle then), beyond the row end the next row's head -- never
// consumed (border substitutions / idle lanes), always inside the allocation: the lowest address is in + y*(W-2) + 3,
// the highest in + W*H + 2*MEDB_K + 2 (the disparity maps are allocated with that much slack, capi.hip).  Lanes
// without a row (y >= H) and the auxiliary stream of lanes that need none re-read in[0..15] (stride 0).
#define MEDB_LD1(K)                                                                                                     \
    asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(ra[PAIRS ? 0 : (K)]) : "v"(pA), "n"(4 * (K)) : "memory"); \
    asm volatile("global_load_dword %0, %1, off offset:%2 sc1" : "=v"(rx[PAIRS ? 0 : (K)]) : "v"(pX), "n"(4 * (K)) : "memory");
#define MEDB_LD2(SET, Q)                                                                                                \
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(ra2##SET[PAIRS ? (Q) : 0]) : "v"(pA), "n"(8 * (Q)) : "memory"); \
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2 sc1" : "=v"(rx2##SET[PAIRS ? (Q) : 0]) : "v"(pX), "n"(8 * (Q)) : "memory");
#define MEDB_ISSUE(SET)                                                                                                 \
    do {                                                                                                                \
        if constexpr (PAIRS) {                                                                                          \
            MEDB_LD2(SET, 0) MEDB_LD2(SET, 1) MEDB_LD2(SET, 2) MEDB_LD2(SET, 3) MEDB_LD2(SET, 4) MEDB_LD2(SET, 5)       \
            MEDB_LD2(SET, 6) MEDB_LD2(SET, 7)                                                                           \
        } else {                                                                                                        \
            MEDB_LD1(0) MEDB_LD1(1) MEDB_LD1(2) MEDB_LD1(3) MEDB_LD1(4) MEDB_LD1(5) MEDB_LD1(6) MEDB_LD1(7)             \
            MEDB_LD1(8) MEDB_LD1(9) MEDB_LD1(10) MEDB_LD1(11) MEDB_LD1(12) MEDB_LD1(13) MEDB_LD1(14) MEDB_LD1(15)       \
        }                                                                                                               \
        pA += strideA;                                                                                                  \
        pX += strideX;                                                                                                  \
    } while (0)
    static_assert(MEDB_K == 16, "MEDB_ISSUE is written for 16 levels per block");
    const int strideA = row_ok ? MEDB_K : 0, strideX = (first_row || (own_b && row_ok)) ? MEDB_K : 0;
    const float* pA = row_ok ? rowA + (x + 3) : in;
    // auxiliary stream: lane 0 of a band > 0 -> hand-off row of the upstream band by level (rowX[t + k] = its result of level
    // t + k - 1; block 0 is idle there), the band's last lane -> unfiltered row below at column x + 1 + k
    const float* pX = first_row ? rowX + tb : ((own_b && row_ok) ? rowX + (x + 1) : in);
    // block 0: a band's first row is idle there (x < 0), whatever it reads from the hand-off row is never used
    MEDB_ISSUE(a);
    MEDB_TAKE(a, "s_waitcnt vmcnt(0)\n\t", "s_waitcnt vmcnt(0)\n\t");

    // the compiler-issued window loads above are consumed here, so that no wait for them ends up inside the loop
    asm volatile("" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(Bm), "+v"(B0));
    // Store targets of lanes that have nothing to store (every block keeps its fixed number of vector-memory operations): ONE
    // 16-byte slot PER LANE (sinks[workgroup][lane][4]) -- the 64 stores of an instruction then fall into one contiguous KiB.
    // (Round 4: with a shared 8-byte sink word the speculative copies, whose 64 lanes ALL store there at every level, ran three
    // times slower than a real band: 64 same-address stores per instruction serialise.)
    float* const sinkf = sinks + ((size_t)blockIdx.x * MEDB_ROWS + tid) * 4;
    float* const sink4 = sinkf;
    float* const hrow = hand + (size_t)myslot * hpitch + MEDB_HPAD;
    float* const orow = out + (size_t)(row_ok ? y : 0) * W;
    const bool st_ok = row_ok && !is_spec; // (a speculative copy stores to the sink)
    float* const seamp = seam + ((size_t)(band * nseg + seg) * MEDB_ROWS + tid) * 2; // column pair (xs - 2, xs - 1) of a real segment's warm-up
    const bool seam_ok = st_ok && seg > 0;

    // One block of MEDB_K levels (SI = register set the prefetch of this iteration goes into, ST = set taken over at its end).
    // band > 0 stays behind the upstream band: the block uses hand-off values of levels < t0+MEDB_K-1 and prefetches those of
    // the block(s) ahead.  The progress word is normally read one block ahead of its use (scalar load past the scalar cache,
    // tracked by lgkmcnt: nothing on the critical path); only when that stale value is not enough the band polls, and then
    // it waits for two extra blocks of slack so that the following stale reads succeed.
    // Levels at which some row of the band stands on its first or last column (row y is at x = 0 on level 2y and at x = W-1
    // on level W-1+2y) need the left / right border substitutions: wave-uniform 16-bit mask, one scalar bit test per level;
    // every other level runs the short form of the body.  The newest filtered value of the row above (out[y-1][x+1],
    // produced by lane-1 at the previous level) comes by DPP wave_shr:1; a band's first lane has no lane above and takes the
    // upstream band's hand-off value instead -- the DPP's `old` operand, which lanes without a source keep.  Same for the
    // unfiltered row below (lane+1's prefetch by wave_shl:1; the band's last lane loads that row itself).
    // PAIRS: levels 2q, 2q+1 = columns x (even), x+1 of every row: one aligned 8-byte store (W even: a row is active on both
    // levels or on neither).  The hand-off stores go through to memory (sc1); the s_nop covers the ">64-bit store data, then
    // VALU write of those VGPRs" hazard the assembler cannot see inside an asm statement.
// Hand-off without progress words: the hand-off rows are filled with a sentinel (all ones: a NaN no disparity map contains)
// before the launch, the upstream band overwrites them level by level (16-byte sc1 stores), and a band simply looks at the
// values it has just taken over: a sentinel among the 16 values of the next block means "not written yet" -> read the block
// again (bounded).  The data is its own flag: no publication one block late, no polling protocol, a band trails its upstream
// band by the store latency + the prefetch distance only.  T1 = first level of the block just taken over; values of levels
// before the band's first row becomes active are never consumed and are not waited for.
#define MEDB_RECHECK(T1)                                                                                                \
    do {                                                                                                                \
        if (band > 0 && !raw_above && (T1) + MEDB_K > 2 * yfirst - 2 && (T1) <= W + 2 * yfirst && (seg == 0 || (T1) >= ts - MEDB_K)) { /* (only while the first row still needs them) */ \
            int spins = 0;                                                                                              \
            while (true) {                                                                                              \
                uint32_t mx = 0u;                                                                                       \
                _Pragma("unroll") for (int q_ = 0; q_ < MEDB_K; q_++) mx = mx > __float_as_uint(cx[q_]) ? mx : __float_as_uint(cx[q_]); \
                if (__ballot(first_row && mx == 0xFFFFFFFFu) == 0ull) break;                                            \
                if (++spins > (1 << 20)) { atomicExch(error_word, 1); return; }                                         \
                __builtin_amdgcn_s_sleep(1);                                                                            \
                const float* pr_ = pX - strideX; /* the block just taken over (pX already stands on the next one) */     \
                _Pragma("unroll") for (int q_ = 0; q_ < MEDB_K; q_++) /* device-scope loads: past the caches */           \
                    cx[q_] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(pr_) + q_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
// per block: [loads for the next block][levels + map stores][4 hand-off stores][take + sentinel check] -- vector-memory operations
// per block: one column per instruction 32 + 16 + 4, PAIRS 16 + 8 + 4; wait before the take-over: the stores of this block
// (20 / 12) may stay outstanding.  (Hand-off stores issued every 4 levels instead of at the end of the block were measured:
// 1.10 ms instead of 0.70 at 1080p -- a downstream band then finds blocks half written and pays a re-read per block.)
#define MEDB_BLOCK()                                                                                                    \
    do {                                                                                                                \
        MEDB_ISSUE(a);                                                                                                  \
        uint32_t bmask;                                                                                                 \
        {                                                                                                               \
            const int lo_l = 2 * yfirst - t0, hi_l = 2 * ylast - t0;                                                    \
            const int lo_r = lo_l + (W - 1), hi_r = hi_l + (W - 1);                                                     \
            const uint32_t ml = (hi_l < 0 || lo_l > 15) ? 0u : ((((2u << adc_imin(hi_l, 15)) - 1u) & ~((1u << adc_imax(lo_l, 0)) - 1u)) & 0x5555u); \
            const uint32_t mr = (hi_r < 0 || lo_r > 15) ? 0u : ((((2u << adc_imin(hi_r, 15)) - 1u) & ~((1u << adc_imax(lo_r, 0)) - 1u)) & (((W - 1) & 1) ? 0xAAAAu : 0x5555u)); \
            bmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ml | mr));                                           \
        }                                                                                                               \
        float res_even = 0.f;                                                                                           \
        const bool pass = PAIRS && t0 < ts; /* (uniform; whole blocks) in front of the segment's window: raw values pass through; the one-column form runs whole rows */ \
_Pragma("unroll")                                                                                                       \
        for (int k = 0; k < MEDB_K; k++) {                                                                              \
            const bool active = row_ok && (x >= 0) && (x < W);                                                          \
            const float F1 = medb_dpp_old<0x138>(cx[k], Pv);                                                            \
            const float B1 = medb_dpp_old<0x130>(cx[k], ca[k]);                                                         \
            float res;                                                                                                  \
            if (pass) {                                                                                                 \
                res = A0;                                                                                               \
            } else if (bmask & (1u << k)) {                                                                             \
                const bool lf = x > 0, rt = x + 1 < W;                                                                  \
                const float v0 = (up && lf) ? Fm : PINF, v1 = up ? F0 : NINF, v2 = (up && rt) ? F1 : PINF;              \
                const float v3 = lf ? Pv : NINF, v5 = rt ? A1 : NINF;                                                   \
                const float v6 = (dn && lf) ? Bm : PINF, v7 = dn ? B0 : NINF, v8 = (dn && rt) ? B1 : PINF;              \
                res = adc_median9(v0, v1, v6, v5, v7, v8, v2, v3, A0);                                                  \
            } else {                                                                                                    \
                const float v0 = up ? Fm : PINF, v1 = up ? F0 : NINF, v2 = up ? F1 : PINF;                              \
                const float v6 = dn ? Bm : PINF, v7 = dn ? B0 : NINF, v8 = dn ? B1 : PINF;                              \
                res = adc_median9(v0, v1, v6, A1, v7, v8, v2, Pv, A0);                                                  \
            }                                                                                                           \
            if constexpr (PAIRS) {                                                                                      \
                if ((k & 1) == 0) res_even = res;                                                                       \
                else {                                                                                                  \
                    const medb_v2f pr = {res_even, res};                                                                \
                    float* dst = (st_ok && x > xs && x < xe) ? orow + (x - 1) : ((seam_ok && x == xs - 1) ? seamp : sinkf); \
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(pr) : "memory");                    \
                }                                                                                                       \
            } else {                                                                                                    \
                float* dst = (active && !is_spec) ? orow + x : sinkf;                                                   \
                asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(res) : "memory");                         \
            }                                                                                                           \
            hr[k] = res;                                                                                                \
            Pv = active ? res : Pv;                                                                                     \
            Fm = F0; F0 = F1;                                                                                           \
            A0 = A1; A1 = A2; A2 = ca[k];                                                                               \
            Bm = B0; B0 = B1;                                                                                           \
            x++;                                                                                                        \
        }                                                                                                               \
        {                                                                                                               \
            float* hp = last_row ? hrow + t0 : sink4;                                                                   \
            const int hs = last_row ? 4 : 0;                                                                            \
_Pragma("unroll")                                                                                                       \
            for (int j = 0; j < 4; j++) {                                                                               \
                const medb_v4f hv = {hr[4 * j], hr[4 * j + 1], hr[4 * j + 2], hr[4 * j + 3]};                           \
                float* hq = hp + hs * j;                                                                                \
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(hq), "v"(hv) : "memory");          \
            }                                                                                                           \
        }                                                                                                               \
        t0 += MEDB_K;                                                                                                   \
        done = t0 >= te;                                                                                                \
        if (!done) {                                                                                                    \
            MEDB_TAKE(a, "s_waitcnt vmcnt(20)\n\t", "s_waitcnt vmcnt(12)\n\t");                                         \
            MEDB_RECHECK(t0);                                                                                           \
        }                                                                                                               \
    } while (0)
    {
        int t0 = tb;
        bool done = false;
        while (!done) MEDB_BLOCK();
    }
#undef MEDB_BLOCK
#undef MEDB_RECHECK
    // (the prefetch of the block past the end is never taken: its registers must stay reserved until it has landed)
    if constexpr (PAIRS) {
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(ra2a[0]), "v"(ra2a[1]), "v"(ra2a[2]), "v"(ra2a[3]), "v"(ra2a[4]), "v"(ra2a[5]), "v"(ra2a[6]),
                     "v"(ra2a[7]) : "memory");
        asm volatile("" ::"v"(rx2a[0]), "v"(rx2a[1]), "v"(rx2a[2]), "v"(rx2a[3]), "v"(rx2a[4]), "v"(rx2a[5]), "v"(rx2a[6]), "v"(rx2a[7]) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(ra[0]), "v"(ra[1]), "v"(ra[2]), "v"(ra[3]), "v"(ra[4]), "v"(ra[5]), "v"(ra[6]),
                     "v"(ra[7]), "v"(ra[8 % (PAIRS ? 1 : MEDB_K)]), "v"(ra[9 % (PAIRS ? 1 : MEDB_K)]), "v"(ra[10 % (PAIRS ? 1 : MEDB_K)]),
                     "v"(ra[11 % (PAIRS ? 1 : MEDB_K)]), "v"(ra[12 % (PAIRS ? 1 : MEDB_K)]), "v"(ra[13 % (PAIRS ? 1 : MEDB_K)]),
                     "v"(ra[14 % (PAIRS ? 1 : MEDB_K)]), "v"(ra[15 % (PAIRS ? 1 : MEDB_K)]) : "memory");
        asm volatile("" ::"v"(rx[0]), "v"(rx[1 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[2 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[3 % (PAIRS ? 1 : MEDB_K)]),
                     "v"(rx[4 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[5 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[6 % (PAIRS ? 1 : MEDB_K)]),
                     "v"(rx[7 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[8 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[9 % (PAIRS ? 1 : MEDB_K)]),
                     "v"(rx[10 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[11 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[12 % (PAIRS ? 1 : MEDB_K)]),
                     "v"(rx[13 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[14 % (PAIRS ? 1 : MEDB_K)]), "v"(rx[15 % (PAIRS ? 1 : MEDB_K)]) : "memory");
    }
#undef MEDB_ISSUE
#undef MEDB_LD1
#undef MEDB_LD2
#undef MEDB_TAKE
#undef MEDB_TAKE8
}

// Seam check of the speculative forms (one launch, blockIdx.y = real band b):
//   row seams     the hand-off row band b >= 1 consumed -- the last copy of its chain (b > spec) or the real band above (b <= spec), segment
//                 s -- must equal, at every column xs - 1 .. xe of the segment, the map row above the band (what the real band b - 1 wrote);
//                 level at which the row 64 b - 1 stands on column c: t = c + 2 (64 b - 1)
//   column seams  (nseg > 1) the column xs - 1 a real segment s >= 1 reached in its warm-up must equal the map (segment s - 1 wrote it)
// Whole rows (nseg = 1): the first is round 4's check of the speculative bands (bands 1 .. spec compare a row with itself).
__global__ __launch_bounds__(256) void k_median_seg_check(const float* __restrict__ hand, int hpitch, int nbands, int spec, int nseg, int W, int H,
                                                          const float* __restrict__ out, const float* __restrict__ seam, int* error_word, int seg_shift)
{
    const int b = (int)blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int yf = b * MEDB_ROWS;
    bool bad = false;
    if (i < W) {
        if (b >= 1) {
            const int c = i, t = c + 2 * (yf - 1);
            const int up = spec ? nbands + medb_chain_off(b, spec) + medb_chain_depth(b, spec) - 1 : b - 1;
            const uint32_t want = reinterpret_cast<const uint32_t*>(out)[(size_t)(yf - 1) * W + c];
            for (int sgm = 0; sgm < nseg; sgm++) {
                const int xs = adc_med_seg_x(W, nseg, sgm, seg_shift), xe = adc_med_seg_x(W, nseg, sgm + 1, seg_shift);
                if (c >= xs - 1 && c <= xe) bad = bad || reinterpret_cast<const uint32_t*>(hand)[(size_t)(up * nseg + sgm) * hpitch + MEDB_HPAD + t] != want;
            }
        }
    } else {
        const int j = i - ((W + 255) & ~255); // column seams: (segment, row of the band)
        const int sgm = j / MEDB_ROWS, r = j - sgm * MEDB_ROWS;
        if (j >= 0 && sgm >= 1 && sgm < nseg && yf + r < H) {
            const int xs = adc_med_seg_x(W, nseg, sgm, seg_shift);
            bad = reinterpret_cast<const uint32_t*>(seam)[((size_t)(b * nseg + sgm) * MEDB_ROWS + r) * 2 + 1] !=
                  reinterpret_cast<const uint32_t*>(out)[(size_t)(yf + r) * W + xs - 1];
        }
    }
    if (bad) atomicMax(error_word, 2);
}

static hipError_t launch_median_wavefront(adc_handle* h, const float* in, float* out)
{
    const AdcParams& p = h->p;
    {   // per-DEVICE function attribute (see adc_launch_aggregate)
        static std::mutex attr_mu;
        static bool attr_set[64] = {false};
        std::lock_guard<std::mutex> lk(attr_mu);
        const int dv = (h->device >= 0 && h->device < 64) ? h->device : 0;
        if (!attr_set[dv]) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_median_wavefront<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_median_wavefront<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_median_wavefront<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_median_wavefront<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set[dv] = true;
        }
    }
    const size_t lds = (size_t)p.H * 4 * sizeof(float);
    if (p.H > 8192) return hipErrorInvalidValue; // LDS ring of H*16 B and <= 8 rows per thread
    const int rpt = (p.H + 1023) / 1024;
    if (rpt <= 1) hipLaunchKernelGGL(k_median_wavefront<1>, dim3(1), dim3(1024), lds, h->stream, in, out, p.W, p.H);
    else if (rpt <= 2) hipLaunchKernelGGL(k_median_wavefront<2>, dim3(1), dim3(1024), lds, h->stream, in, out, p.W, p.H);
    else if (rpt <= 4) hipLaunchKernelGGL(k_median_wavefront<4>, dim3(1), dim3(1024), lds, h->stream, in, out, p.W, p.H);
    else hipLaunchKernelGGL(k_median_wavefront<8>, dim3(1), dim3(1024), lds, h->stream, in, out, p.W, p.H);
    return hipGetLastError();
}

// Segments per band link of the speculative form: a segment should be well above its 128 columns of warm-up, and the chip takes
// (bands + copies) x segments single-wave workgroups side by side.  ADC_MEDIAN_SEG overrides (1 = whole rows), ADC_MEDIAN_WARM the warm-up.
#define MEDB_MAX_SEG ADC_MEDB_MAX_SEG
static int median_warm()
{
    static const int v = [] { const char* e = getenv("ADC_MEDIAN_WARM"); const int w = e ? atoi(e) : 128; return adc_imax(0, adc_imin(w, 1024)) & ~15; }();
    return v;
}
// (segment 0 is narrower by `shift` = 128 columns per run-in band minus the warm-up, adc_device_fn.h: adc_med_seg_x; ADC_MEDIAN_SHIFT=0: equal widths)
static int median_seg_shift(int spec)
{
    static const int env = [] { const char* e = getenv("ADC_MEDIAN_SHIFT"); return e ? atoi(e) : -1; }();
    return env >= 0 ? (env & ~15) : adc_imax(0, 2 * MEDB_ROWS * spec - median_warm());
}
static int median_segments(int W, int spec)
{
    static const int env = [] { const char* e = getenv("ADC_MEDIAN_SEG"); return e ? atoi(e) : 0; }();
    if (!spec || (W & 1)) return 1; // (the chained form and odd widths run whole rows)
    // (segments of ~170 columns, 10 at most: 1080p 8 / 10 / 12 segments -> refine 0.518 / 0.510 / 0.513 ms, KITTI size 5 / 8 / 10 -> 0.267 / 0.256 / 0.256)
    int n = env > 0 ? env : adc_imin((W + median_seg_shift(spec)) / 170, 10); // 1080p: 10 segments, KITTI size: 8
    n = adc_imax(1, adc_imin(n, MEDB_MAX_SEG));
    while (n > 1) { // (every segment at least 64 columns wide)
        bool ok = true;
        for (int s = 0; s < n; s++) ok = ok && adc_med_seg_x(W, n, s + 1, median_seg_shift(spec)) - adc_med_seg_x(W, n, s, median_seg_shift(spec)) >= 64;
        if (ok) break;
        n--;
    }
    return n;
}
size_t adc_median_hand_rows(int H) { return (size_t)(5 * ((H + 63) / 64) + 1) * MEDB_MAX_SEG; } // bands + chains of <= 4 copies per band, per segment

// in -> out with the banded kernel; spec > 0: speculative bands (+ column segments) + seam check.  The error word (0 ok, 1 hand-off
// time-out, 2 speculative seam differs) goes to pin_flags[0], looked at by adc_wait.
static hipError_t launch_median_banded(adc_handle* h, const float* in, float* out, int spec)
{
    const AdcParams& p = h->p;
    // The kernel prefetches past the end of the last row without clamping (and a speculative copy reads a few elements behind
    // the row above its band): only the two maps that capi.hip allocates with 1 KiB of slack may be passed.  (It also relies
    // on 0xFFFFFFFF never being a disparity value -- that NaN is the hand-off rows' "not written yet" mark; a debug upload of
    // such a value makes the bands re-read until the bounded spin gives up and adc_wait runs the single-workgroup kernel.)
    if ((in != h->disp_l && in != h->disp_tmp) || (out != h->disp_l && out != h->disp_tmp) || in == out) return hipErrorInvalidValue;
    const int nbands = (p.H + MEDB_ROWS - 1) / MEDB_ROWS;
    const int ncopies = spec ? medb_chain_off(nbands, spec) : 0; // a chain of min(b, spec) copies per target band b = 1 .. nbands - 1
    const int nseg = h->med_seg_off > 0 ? 1 : median_segments(p.W, spec), nlinks = nbands + ncopies;
    if ((size_t)nlinks * nseg + 1 > adc_median_hand_rows(p.H)) return hipErrorInvalidValue;
    // error word + store sinks live in vote_counters[160..]: prog[260] error word, prog[262..267] sinks of idle lanes;
    // the hand-off rows are reset to the "not written yet" sentinel (all ones) before every launch
    int* prog = h->vote_counters + 160;
    hipMemsetAsync(prog + 256, 0, 16 * sizeof(int32_t), h->stream);
    hipMemsetAsync(h->med_hand, 0xFF, ((size_t)nlinks * nseg + 1) * h->med_hpitch * sizeof(float), h->stream);
    float* seam = h->med_sink + (size_t)adc_median_hand_rows(p.H) * 64 * 4; // [bands][segments][64 rows][2] behind the sinks
    // pairs of columns per instruction when the width is even (ADC_MEDIAN_PAIRS=0: always one column per instruction)
    static const bool pairs_env = [] { const char* e = getenv("ADC_MEDIAN_PAIRS"); return e ? atoi(e) != 0 : true; }();
    const bool pairs = pairs_env && (p.W & 1) == 0;
    const int nseg_run = pairs ? nseg : 1;
    if (pairs)
        hipLaunchKernelGGL(k_median_banded<true>, dim3(nlinks * nseg_run), dim3(MEDB_ROWS), 0, h->stream, in, out, p.W, p.H, prog, prog + 260,
                           h->med_hand, h->med_hpitch, nbands, spec, h->med_sink, nseg_run, median_warm(), seam, median_seg_shift(spec));
    else
        hipLaunchKernelGGL(k_median_banded<false>, dim3(nlinks), dim3(MEDB_ROWS), 0, h->stream, in, out, p.W, p.H, prog, prog + 260,
                           h->med_hand, h->med_hpitch, nbands, spec, h->med_sink, 1, 0, seam, 0);
    if (spec)
        hipLaunchKernelGGL(k_median_seg_check, dim3((p.W + 255) / 256 + (MEDB_ROWS * nseg_run + 255) / 256, nbands), dim3(256), 0, h->stream, h->med_hand,
                           h->med_hpitch, nbands, spec, nseg_run, p.W, p.H, out, seam, prog + 260, median_seg_shift(spec));
    h->med_spec_last = spec;
    h->med_seg_last = spec ? nseg_run : 1;
    if (h->pin_flags) hipMemcpyAsync(h->pin_flags, prog + 260, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream); // checked by adc_wait
    return hipGetLastError();
}

hipError_t adc_launch_median(adc_handle* h)
{
    const AdcParams& p = h->p;
    static const bool banded = [] { const char* e = getenv("ADC_MEDIAN_BANDED"); return e ? atoi(e) != 0 : true; }();
    const int nbands = (p.H + MEDB_ROWS - 1) / MEDB_ROWS;
    if (banded && nbands > 1 && nbands <= 256 && p.W >= 2 && p.H >= 2 && h->med_hand) {
        // speculative bands (ADC_MEDIAN_SPEC=0: chained form; also for a while after a seam of this handle has failed)
        // (ADC_MEDIAN_SPEC = run-in in bands, default 2 = 128 rows, 0 = chained form)
        static const int spec_env = [] { const char* e = getenv("ADC_MEDIAN_SPEC"); const int v = e ? atoi(e) : 2; return v < 0 ? 0 : (v > 4 ? 4 : v); }();
        // (round 6: every band has its own chain, and the chains of the bands 1 .. spec are exact copies rooted in band 0 -- images of
        // 2 .. spec + 1 bands run the speculative form too, with nothing speculative about their rows: 1920 x 192 354 -> ~110 us)
        const int spec = h->med_spec_off == 0 ? spec_env : 0;
        launch_median_banded(h, h->disp_l, h->disp_tmp, spec);
        float* t = h->disp_l;
        h->disp_l = h->disp_tmp;
        h->disp_tmp = t;
        return hipGetLastError();
    }
    hipError_t e = launch_median_wavefront(h, h->disp_l, h->disp_tmp);
    if (e != hipSuccess) return e;
    float* t = h->disp_l;
    h->disp_l = h->disp_tmp;
    h->disp_tmp = t;
    return hipGetLastError();
}

// The banded filter reported a hand-off time-out (a band gave up waiting for its upstream band; cannot happen while all
// bands are co-resident, but the spin is bounded on principle): redo the filter with the single-workgroup wavefront
// kernel, which has no inter-workgroup dependency.  The unfiltered input is still intact in disp_tmp (the filter writes
// to the other buffer).  Called by adc_wait after the stream has drained.
hipError_t adc_median_fallback(adc_handle* h)
{
    hipError_t e;
    if (h->med_spec_last && h->pin_flags && (h->pin_flags[0] == 2 || h->force_median_fallback == 2) && h->force_median_fallback != 1) {
        // a speculative seam differed.  With column segments on, the speculative bands alone (whole rows: round 4's form, which has run
        // on every pair since) come first -- segments stay off for the next 64 Matches of the handle; if that fails too, or no segments
        // were on, the chained form of the banded kernel, which needs no assumption (and may itself report a time-out, then the
        // single-workgroup kernel below runs), and whole chained bands for the next 64 Matches.
        h->med_spec_fails++;
        if (h->med_seg_last > 1 && h->force_median_fallback != 2) {
            h->med_seg_off = 64;
            e = launch_median_banded(h, h->disp_tmp, h->disp_l, h->med_spec_last); // (med_seg_off: whole rows)
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) return e;
            if (h->pin_flags[0] == 0) return hipSuccess;
        }
        h->med_spec_off = 64;
        e = launch_median_banded(h, h->disp_tmp, h->disp_l, 0);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return e;
        if (h->pin_flags[0] == 0) return hipSuccess;
    }
    e = launch_median_wavefront(h, h->disp_tmp, h->disp_l);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(h->stream);
}
