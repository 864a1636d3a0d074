// adc_device_fn.h -- per-element arithmetic shared by the HIP kernels (and, for unit checks of the
// order-aware parallel formulations, by a g++-compiled CPU emulation under tests/emul/).
//
// Everything here is exact integer / IEEE-754 arithmetic that must reproduce the reference CPU
// program bit for bit.  Rules: no FMA contraction (files are compiled with
// -ffp-contract=off and carry the pragma below), no fast-math, no approximate reciprocals,
// transcendental values only through host-built tables.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ADC_HD __host__ __device__ __forceinline__
#else
#define ADC_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define ADC_LARGE_FLOAT 99999.0f        // Large_Float,   adcensus_types.h:35
#define ADC_INVALID_FLOAT __builtin_inff() // Invalid_Float, adcensus_types.h:33

#define ADC_LABEL_VALID 0
#define ADC_LABEL_MISMATCH 1
#define ADC_LABEL_OCCLUSION 2

ADC_HD int adc_iabs(int a) { return a < 0 ? -a : a; }
ADC_HD int adc_imax(int a, int b) { return a > b ? a : b; }
ADC_HD int adc_imin(int a, int b) { return a < b ? a : b; }

// gray = uint8(r*0.299 + g*0.587 + b*0.114), operands in double, left to right, truncation
// (cost_computor.cpp:66-69).  Unfused: 3 mul + 2 add.
ADC_HD uint8_t adc_gray(uint8_t b, uint8_t g, uint8_t r)
{
    const double t0 = (double)r * 0.299;
    const double t1 = (double)g * 0.587;
    const double t2 = (double)b * 0.114;
    const double s = (t0 + t1) + t2;
    return (uint8_t)s;
}

// max-channel colour distance (cross_aggregator.h:78-80, scanline_optimizer.h:66-68). p -> B,G,R.
ADC_HD int adc_color_dist_max(const uint8_t* a, const uint8_t* b)
{
    return adc_imax(adc_iabs((int)a[2] - (int)b[2]), adc_imax(adc_iabs((int)a[1] - (int)b[1]), adc_iabs((int)a[0] - (int)b[0])));
}
ADC_HD int adc_color_dist_max_u32(uint32_t a, uint32_t b) // packed 0x00RRGGBB
{
    const int d0 = adc_iabs((int)(a & 255u) - (int)(b & 255u));
    const int d1 = adc_iabs((int)((a >> 8) & 255u) - (int)((b >> 8) & 255u));
    const int d2 = adc_iabs((int)((a >> 16) & 255u) - (int)((b >> 16) & 255u));
    return adc_imax(d2, adc_imax(d1, d0));
}
// L1 colour distance used by the interpolation (multistep_refiner.cpp:281).
ADC_HD int adc_color_dist_l1(const uint8_t* a, const uint8_t* b)
{
    return adc_iabs((int)a[2] - (int)b[2]) + adc_iabs((int)a[1] - (int)b[1]) + adc_iabs((int)a[0] - (int)b[0]);
}

// ---- scanline optimiser: which right-image colour difference a disparity sees ----
// The reference initialises d2 = d1 once per pixel and overwrites it only while 0 < xr < W-1
// (scanline_optimizer.cpp:116-126, :225-235), xr = x - d - dmin descending in d, so out-of-interval
// disparities inherit the last in-interval value ("sticky").  Closed form:
//   xr in (0, W-1)                       -> own column xr
//   xr <= 0 and the interval was entered -> column 1      (entered <=> x - dmin >= 1 and W >= 3)
//   otherwise                            -> d1 (returns -1)
// Returns the column of the right-image difference map to use, or -1 for "use d1".
ADC_HD int adc_so_d2_column(int x, int dmin, int d, int W)
{
    const int xr = x - d - dmin;
    if (xr > 0 && xr < W - 1) return xr;
    if (xr <= 0 && (x - dmin) >= 1 && W >= 3) return 1; // interval was entered at d' = x-dmin-1 < d
    return -1;
}

// penalty class: 0 -> (p1,p2), 1 -> (p1/4,p2/4), 2 -> (p1/10,p2/10)  (scanline_optimizer.cpp:129-141)
ADC_HD int adc_so_penalty_class(int d1, int d2, int tso) { return (d1 >= tso ? 1 : 0) + (d2 >= tso ? 1 : 0); }

// ---- scanline penalty classes derived per lane (k_scanline.hip; CPU emulation: tests/emul/emul.cpp) ----
// LDS byte offsets (class * 8) of the (P1,P2) pairs of this lane's VPL disparities.
//   rb       VPL consecutive bytes of the right-image step map starting at column max(xr_last, 1) (+1 on R->L)
//   c1byte   left-image step d1 of this pixel
//   xr_last  x - dmin - (d0 + VPL-1): right-image column of the lane's LAST disparity (the smallest column)
// (rb = the fetched bytes as little-endian dwords: byte j = (rb[j >> 2] >> (8 * (j & 3))) & 0xff)
template <int VPL>
ADC_HD void adc_so_class_offsets(const uint32_t* rb, int c1byte, int xr_last, int W, int tso, bool row_ok, int* off)
{
    const int c1 = c1byte >= tso ? 8 : 0;
    const int a0 = xr_last > 1 ? xr_last : 1;
    for (int k = 0; k < VPL; k++) {
        const int xr = xr_last + (VPL - 1 - k);
        const int j = (xr > 1 ? xr : 1) - a0; // 0 .. VPL-1: which of the fetched bytes is column max(xr, 1)
        // (VPL <= 4: a single word -- no runtime index, the array must stay in registers: a select tree over the VPL / 4 words)
        uint32_t word = rb[0];
        for (int q = 1; q < (VPL + 3) / 4; q++) word = (j >> 2) == q ? rb[q] : word;
        const int byte = (int)((word >> (8 * (j & 3))) & 0xffu);
        const int c2 = byte >= tso ? 8 : 0;
        const bool use_r = row_ok && xr < W - 1;
        off[k] = c1 + (use_r ? c2 : c1);
    }
}

// Interior form: when every lane of the wave has xr_last >= 1 and no right-image column reaches W-1, the general
// rule above reduces to "byte j = VPL-1-k, always use the right image":
//   x >= dmin + Dpad (Dpad = the disparity range rounded up to VPL)  =>  xr_last >= 1 for every lane that owns a real
//                      disparity, hence a0 = xr_last, j = VPL-1-k, and row_ok holds (x - dmin >= Dpad >= 1, W >= 3);
//   x - dmin < W - 1   =>  xr = x - d <= x - dmin < W - 1 for every real disparity, hence use_r.
// (Lanes that own padding disparities only may differ from the general rule; their results are never used.)
ADC_HD bool adc_so_interior(int x, int W, int dmin, int Dpad)
{
    return W >= 3 && x >= dmin + Dpad && x - dmin < W - 1;
}
template <int VPL>
ADC_HD void adc_so_class_offsets_interior(const uint32_t* rb, int c1byte, int tso, int* off)
{
    const int c1 = c1byte >= tso ? 8 : 0; // wave-uniform
    for (int k = 0; k < VPL; k++) {
        const int j = VPL - 1 - k;
        const int byte = (int)((rb[j >> 2] >> (8 * (j & 3))) & 0xffu);
        off[k] = byte >= tso ? c1 + 8 : c1;
    }
}

// ---- scanline prefetch geometry (k_scanline.hip; checked exhaustively on the CPU: tests/test_emul.py) ----
// Byte offset into the right-image colour-step map of the VPL bytes the path element at coordinate m needs (m = x on a row
// path, y on a column path; `path` = the row / column; cl_last = dmin + the lane's LAST disparity index, so xr = x - cl_last is
// the right-image column of that disparity -- the smallest column of the lane).  Clamped to the columns 1 .. W-1 (the class
// rule never uses the others, see adc_so_class_offsets); R->L row paths read the step to the right neighbour (+1).
ADC_HD int adc_so_rmap_offset(int W, bool vert, int dir, int path, int m, int cl_last)
{
    const int x = vert ? path : m, y = vert ? m : path;
    const int sy = vert ? (dir > 0 ? y : y + 1) : y;
    int xr = x - cl_last;
    xr = xr > W - 1 ? W - 1 : xr;
    return sy * W + (xr > 1 ? xr : 1) + ((!vert && dir < 0) ? 1 : 0);
}
// A chunk of the steady state = path elements i .. i+PF-1, which prefetches elements i+PF .. i+2*PF-1.  It may take the short
// form (interior class rule without its per-step test, running rmap offset without clamps) when
//   * every element it steps on or prefetches, and the d1 word groups behind them, lies inside the path (i + 2*PF + 4 <= plen),
//   * for every such element x >= dmin + Dp (=> x - cl_last >= 1 for every lane: no lower clamp; interior rule, Dp = 64*VPL =
//     the padded range, which equals the range because the short form also requires D == Dp) and x - dmin < W - 1 (=> interior
//     rule; xr <= x - dmin < W - 1: no upper clamp).
ADC_HD bool adc_so_chunk_interior(int i, int PF, int plen, int dir, bool vert, int path, int W, int dmin, int Dp)
{
    if (W < 3 || i + 2 * PF + 4 > plen) return false;
    const int ea = i, eb = i + 2 * PF - 1; // path elements the chunk steps on or prefetches
    const int ma = dir > 0 ? ea : plen - 1 - ea, mb = dir > 0 ? eb : plen - 1 - eb;
    const int xlo = vert ? path : (ma < mb ? ma : mb), xhi = vert ? path : (ma < mb ? mb : ma);
    return xlo >= dmin + Dp && xhi - dmin < W - 1;
}

// ---- scanline row passes cut into VERIFIED SEGMENTS (k_scanline.hip, round 4) ----
// L = (C + min(...)) / 2 halves every perturbation per step, so a pass that starts in the middle of a path from the raw costs
// (like the reference's first pixel) runs into the full pass's state bit for bit after a few dozen steps (measured: <= 42 on
// the row passes, tools/so_merge_length.py).  A path of plen elements is therefore cut into nseg segments, one wave each.
// Segment s > 0 owns the outputs [a, b) and starts at element e0 = a - warm - 1 as a FIRST element, runs `warm` steps whose
// outputs go to its seam slot (the last one stays there: its state at element a - 1), then stores from a on.  Its seam slot is
// compared bit for bit with what segment s - 1 stored at element a - 1 (k_so_seam_check): equal => every output of the
// segment is the full pass's; different => the Match is redone without segments (adc_wait).
// The kernel's prefetch groups need e0 % 4 == 0 (the d1 words hold path elements 1+4g .. 4+4g) and warm % 16 == 0 (the switch
// from the seam slot to the volume happens between chunks of 16 steps): a = 1 (mod 4).  All segments take (nearly) the same
// number of steps T = ceil((plen + (nseg - 1) * (warm + 1)) / nseg).
#define ADC_SO_WARM 64
#define ADC_SO_MAX_SEG 8
ADC_HD int adc_so_seg_start(int plen, int nseg, int warm, int s)
{
    if (s <= 0) return 0;
    if (s >= nseg) return plen;
    const int T = (plen + (nseg - 1) * (warm + 1) + nseg - 1) / nseg;
    return ((T + (s - 1) * (T - warm - 1)) & ~3) + 1;
}
// can a path of plen elements be cut into nseg segments?  (every segment keeps >= 2 chunks of real outputs behind its warm-up)
ADC_HD bool adc_so_seg_ok(int plen, int nseg, int warm)
{
    if (nseg < 2 || nseg > ADC_SO_MAX_SEG || warm < 16 || (warm & 15)) return false;
    for (int s = 1; s <= nseg; s++)
        if (adc_so_seg_start(plen, nseg, warm, s) - adc_so_seg_start(plen, nseg, warm, s - 1) < 32 + (s == 1 ? warm + 1 : 0)) return false;
    return true;
}

// ---- interpolation: empty-space skipping of the ray walk (k_refine.hip; CPU emulation: tests/emul/emul.cpp) ----
// cdist[cell] = a LOWER BOUND of the Chebyshev distance, in cells of ADC_ITP_CELL x ADC_ITP_CELL pixels, from the cell to
// the nearest cell that holds a valid pixel (0 = the cell itself; search window +-ADC_ITP_CAP cells, ADC_ITP_CAP + 1 =
// "further").  A ray standing in a cell with cdist = c >= 2 cannot meet a valid pixel during its next
// (c - 1) * ADC_ITP_CELL - 1 steps: a step moves at most one pixel per axis (+1 for the rounding of lround(m * sin)).
#define ADC_ITP_CELL 2
#define ADC_ITP_CAP 16
ADC_HD int adc_itp_skip(int c) { return c >= 2 ? (c - 1) * ADC_ITP_CELL - 1 : 0; }
// distance (cells) to the nearest non-empty cell of the same cell row, within the window
ADC_HD int adc_itp_rowdist(const uint8_t* cell, int cw, int cx, int cy)
{
    int best = ADC_ITP_CAP + 1;
    for (int dx = -ADC_ITP_CAP; dx <= ADC_ITP_CAP; dx++) {
        const int x = cx + dx;
        if (x >= 0 && x < cw && cell[cy * cw + x]) best = adc_imin(best, adc_iabs(dx));
    }
    return best;
}
// Chebyshev combination over the cell rows of the window: min over dy of max(rowdist(cx, cy + dy), |dy|)
ADC_HD int adc_itp_coldist(const uint8_t* rowd, int cw, int ch, int cx, int cy)
{
    int best = ADC_ITP_CAP + 1;
    for (int dy = -ADC_ITP_CAP; dy <= ADC_ITP_CAP; dy++) {
        const int y = cy + dy;
        if (y >= 0 && y < ch) best = adc_imin(best, adc_imax((int)rowd[y * cw + cx], adc_iabs(dy)));
    }
    return best;
}

// The ray walk's CODE MAP (round 6): one byte per pixel of the image padded by the search range `ms` on the left, on the right and
// below (the rays point into the half plane dy >= 0: angles 0 .. 168.75 degrees, multistep_refiner.cpp:252-257; |dx|, dy < ms), so
// that a ray position is ONE add away from the target's padded index and needs no bounds test:
//   ADC_ITP_VALID    the pixel holds a valid disparity: the ray ends with a hit
//   ADC_ITP_OUTSIDE  outside the image: the ray ends without one (multistep_refiner.cpp:261-263)
//   0 .. 253         invalid pixel inside the image: adc_itp_skip(cell distance) = the steps that may be skipped behind it
// Linear ray offsets lin[m][s] = dy * pitch + dx for 1 <= m < ms, 0 for row 0 and the ADC_ITP_LPAD rows behind the range (a trip of
// ADC_ITP_NS steps that starts below ms, plus the largest skip, stays inside the table).
#define ADC_ITP_VALID 255
#define ADC_ITP_OUTSIDE 254
#ifndef ADC_ITP_NS
#define ADC_ITP_NS 8
#endif
#define ADC_ITP_LPAD 64
ADC_HD int adc_itp_code_pitch(int W, int ms) { return (W + 2 * ms + 8 + 3) & ~3; } // (+8: k_itp_code stores whole dwords around the image's columns)
ADC_HD int adc_itp_code_rows(int H, int ms) { return H + ms; }
static_assert((ADC_ITP_CAP + 1 - 1) * ADC_ITP_CELL - 1 < ADC_ITP_OUTSIDE && ADC_ITP_NS + (ADC_ITP_CAP * ADC_ITP_CELL - 1) + ADC_ITP_NS <= ADC_ITP_LPAD,
              "skip codes stay below the end codes; the table padding covers a trip behind the largest skip");

// ---- K11 median, column segments (k_refine.hip; CPU emulation: tests/emul/emul.cpp) ----
// First column of segment s of nseg (multiples of 16; s <= 0 -> 0, s >= nseg -> W).  Segment 0 starts at the true left border: its
// chain links stand 128 columns further right per link and cannot start before their rows do, so it runs `lead` = 128 * run-in bands
// levels where the other segments run `warm` levels of warm-up -- it is made narrower by the difference (shift) so that all waves run
// about the same number of levels.
ADC_HD int adc_med_seg_x(int W, int nseg, int s, int shift)
{
    if (s <= 0 || nseg <= 1) return s <= 0 ? 0 : W;
    if (s >= nseg) return W;
    const long long x = ((long long)(W + shift) * s / nseg - shift) & ~15LL;
    return (int)(x < 16 ? 16 : (x > W ? W : x));
}

// ---- WTA sub-pixel (ADCensusStereo.cpp:227-240) ----
ADC_HD float adc_subpixel(int best, float c1, float c2, float cmin)
{
    const float denom = c1 + c2 - 2 * cmin;
    if (denom != 0.0f) return (float)best + (c1 - c2) / (denom * 2.0f);
    return (float)best;
}

// ---- right-view WTA marching along a row (k_wta.hip: k_wta_right_march; ADCensusStereo.cpp:245-310) ----
// A workgroup marches along a row of the volume in steps of 64 pixel vectors, which it keeps in an LDS ring of ADC_WTAM_RING
// vectors; the right pixel xr needs the vectors xr + dmin .. xr + dmin + D - 1, so the group of 64 right pixels that starts at
// vector g * 64 is complete when step g + lag has been written.
#define ADC_WTAM_RING 256
ADC_HD int adc_wtam_lag(int D) { return (D + 62) / 64; }
// The launch: one unit per workgroup, one workgroup per CU (the ring takes most of the LDS).  The first rows_full rows (a multiple
// of the CU count) are whole-row units; the remaining H - rows_full rows would leave most CUs idle in the last round, so they are
// cut into nseg segments of segw pixels (a multiple of 64) each -- a segment re-reads D - 1 vectors of its neighbour.
struct AdcWtamPlan { int rows_full, nseg, segw, units; };
ADC_HD AdcWtamPlan adc_wtam_plan(int W, int H, int D, int ncu, int force_nseg = 0) // (force_nseg: tests)
{
    AdcWtamPlan pl;
    ncu = adc_imax(1, ncu);
    pl.rows_full = (H / ncu) * ncu;
    const int rem = H - pl.rows_full;
    pl.nseg = 1;
    pl.segw = ((W + 63) / 64) * 64;
    if (rem > 0) {
        long best = -1;
        for (int n = force_nseg > 0 ? force_nseg : 1; n <= (force_nseg > 0 ? force_nseg : 8); n++) {
            const int sw = (((W + n - 1) / n + 63) / 64) * 64; // pixels per segment
            const int ns = (W + sw - 1) / sw;                  // segments that hold pixels
            const long rounds = ((long)rem * ns + ncu - 1) / ncu;
            const long cost = rounds * (sw + D + 127);         // vectors a workgroup walks per unit + the fill of its pipeline
            if (best < 0 || cost < best) { best = cost; pl.nseg = ns; pl.segw = sw; }
        }
    }
    pl.units = pl.rows_full + rem * pl.nseg;
    return pl;
}
struct AdcWtamUnit { int y, x0, x1; };
ADC_HD AdcWtamUnit adc_wtam_unit(int u, int W, int rows_full, int nseg, int segw)
{
    AdcWtamUnit un;
    if (u < rows_full) { un.y = u; un.x0 = 0; un.x1 = W; return un; }
    const int r = u - rows_full, s = r % nseg;
    un.y = rows_full + r / nseg;
    un.x0 = s * segw;
    un.x1 = adc_imin(W, un.x0 + segw);
    return un;
}

// ---- region voting decision (multistep_refiner.cpp:199-214) ----
// returns the filled disparity or +inf
ADC_HD float adc_vote_decide(int best_bin, int max_ht, int count, int dmin, int irv_ts, float irv_th)
{
    if (max_ht > 0 && count > irv_ts && (float)max_ht * 1.0f / (float)count > irv_th) return (float)(best_bin + dmin);
    return ADC_INVALID_FLOAT;
}

// ---- sorting network for the 3x3 median (adcensus_util.cpp:55-81): full sort of 9, select [n/2] ----
ADC_HD void adc_cswap(float& a, float& b)
{
    // no NaNs ever reach the filter (+inf is the only special value), so min/max are exact selections
#if defined(__HIP_DEVICE_COMPILE__)
    const float lo = __builtin_fminf(a, b);
    const float hi = __builtin_fmaxf(a, b);
#else
    const float lo = b < a ? b : a;
    const float hi = b < a ? a : b;
#endif
    a = lo;
    b = hi;
}
// ---- rank selection without a full sort (used by the banded median kernel) ----
// 3x3 window positions: 0 1 2 / 3 4 5 / 6 7 8.  The reference sorts the n in-image values and takes
// wnd[n/2] (adcensus_util.cpp:64-77).  With W,H >= 2 only n = 9, 6 (edge) and 4 (corner) occur, and
// wnd[n/2] equals the MEDIAN of nine values when the missing positions are padded with 'a' values of -inf
// and 'b' values of +inf where a = 4 - n/2: edge a=1,b=2, corner a=2,b=3.  Rule that yields exactly these
// counts: a missing side-centre (1,3,5,7) is padded with -inf, a missing corner (0,2,6,8) with +inf.
ADC_HD float adc_min3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fminf(__builtin_fminf(a, b), c);
#else
    const float m = b < a ? b : a;
    return c < m ? c : m;
#endif
}
ADC_HD float adc_max3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
#else
    const float m = b < a ? a : b;
    return c < m ? m : c;
#endif
}
ADC_HD float adc_med3(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(a, b, c);
#else
    const float lo = b < a ? b : a, hi = b < a ? a : b;
    const float m = c < hi ? c : hi; // min(hi, c)
    return m < lo ? lo : m;          // max(lo, min(hi, c))
#endif
}
// Median of nine = med3(max of the triple minima, median of the triple medians, min of the triple maxima)
// for ANY partition into three triples (13 min/max/med3 operations; all monotone, checked over all 512
// 0/1 inputs in tests/test_emul.py).  The last triple holds the values that arrive latest in the kernel.
ADC_HD float adc_median9(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float v8)
{
    const float lo1 = adc_min3(v0, v1, v2), me1 = adc_med3(v0, v1, v2), hi1 = adc_max3(v0, v1, v2);
    const float lo2 = adc_min3(v3, v4, v5), me2 = adc_med3(v3, v4, v5), hi2 = adc_max3(v3, v4, v5);
    const float lo3 = adc_min3(v6, v7, v8), me3 = adc_med3(v6, v7, v8), hi3 = adc_max3(v6, v7, v8);
    return adc_med3(adc_max3(lo1, lo2, lo3), adc_med3(me1, me2, me3), adc_min3(hi1, hi2, hi3));
}

// Sorts v[0..8] ascending (25 compare-exchanges, optimal-size network for n=9).
ADC_HD void adc_sort9(float* v)
{
    adc_cswap(v[0], v[3]); adc_cswap(v[1], v[7]); adc_cswap(v[2], v[5]); adc_cswap(v[4], v[8]);
    adc_cswap(v[0], v[7]); adc_cswap(v[2], v[4]); adc_cswap(v[3], v[8]); adc_cswap(v[5], v[6]);
    adc_cswap(v[0], v[2]); adc_cswap(v[1], v[3]); adc_cswap(v[4], v[5]); adc_cswap(v[7], v[8]);
    adc_cswap(v[1], v[4]); adc_cswap(v[3], v[6]); adc_cswap(v[5], v[7]);
    adc_cswap(v[0], v[1]); adc_cswap(v[2], v[4]); adc_cswap(v[3], v[5]); adc_cswap(v[6], v[8]);
    adc_cswap(v[2], v[3]); adc_cswap(v[4], v[5]); adc_cswap(v[6], v[7]);
    adc_cswap(v[1], v[2]); adc_cswap(v[3], v[4]); adc_cswap(v[5], v[6]);
}
