// k_wta.hip -- K6 winner-takes-all + parabola sub-pixel, left and right view.
//
// Replaces ADCensusStereo::ComputeDisparity / ComputeDisparityRight (ADCensusStereo.cpp:188-310).
// Left:  first minimum over d (strict '>' => lowest d wins ties); best at either end of the range
//        => +inf; else best + (c1-c2)/(2*(c1+c2-2*cmin)).
// Right: candidate cost(x+d, y, d) if 0 <= x+d < W else Large_Float (can enter the parabola as a
//        neighbour); at the range ends the INTEGER best is stored (no invalidation).
// One wave per pixel, lanes = disparities; lexicographic (cost, d) wave arg-min.
#include "adc_internal.h"
#include "adc_device_fn.h"

__device__ __forceinline__ void wave_argmin(float& c, int& d)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float oc = __shfl_xor(c, m, 64);
        const int od = __shfl_xor(d, m, 64);
        const bool take = (oc < c) || (oc == c && od < d);
        c = take ? oc : c;
        d = take ? od : d;
    }
}

#define WTA_PPW_MAX 8 // pixels per wave: their loads are all issued before the first reduction

template <int VPL, bool RIGHT>
__global__ __launch_bounds__(256) void k_wta(const float* __restrict__ vol, float* __restrict__ disp, int W, int H, int dmin,
                                             int D)
{
    constexpr int Dp = 64 * VPL;
    constexpr int WTA_PPW = VPL <= 4 ? WTA_PPW_MAX : (VPL == 8 ? 4 : (VPL == 16 ? 2 : 1)); // (register budget: WTA_PPW * VPL costs per lane)
    const int lane = threadIdx.x & 63;
    const long long P = (long long)W * H;
    const long long pix0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * WTA_PPW;
    if (pix0 >= P) return;
    const int dmax = dmin + D;

    float c[WTA_PPW][VPL];
#pragma unroll
    for (int i = 0; i < WTA_PPW; i++) {
        const long long pix = pix0 + i < P ? pix0 + i : P - 1; // clamped: loads stay unconditional
        const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const int di = lane * VPL + k;
            float v = ADC_LARGE_FLOAT;
            if (!RIGHT) {
                v = vol[(size_t)pix * Dp + di]; // padding lanes (di >= D) are masked below
            } else {
                const int col = x + di + dmin; // cost(xr, yr, d) = cost(xr + d, yl, d)
                const int cc = col < 0 ? 0 : (col >= W ? W - 1 : col);
                v = vol[((size_t)y * W + cc) * Dp + (di < Dp ? di : 0)];
                if (col < 0 || col >= W) v = ADC_LARGE_FLOAT; // ADCensusStereo.cpp:281-283
            }
            c[i][k] = di < D ? v : ADC_LARGE_FLOAT;
        }
    }
#pragma unroll
    for (int i = 0; i < WTA_PPW; i++) {
        if (pix0 + i >= P) break;
        // the scan updates only on min_cost > cost, min_cost starting at Large_Float; lowest d wins ties
        float bc = ADC_LARGE_FLOAT;
        int bd = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < VPL; k++)
            if (c[i][k] < bc) { bc = c[i][k]; bd = lane * VPL + k + dmin; }
        wave_argmin(bc, bd);
        const int best = bd == 0x7fffffff ? 0 : bd; // nothing below Large_Float: best_disparity keeps its initial 0
        float out;
        const bool edge = (best == dmin) || (best == dmax - 1);
        if (edge) {
            out = RIGHT ? (float)best : ADC_INVALID_FLOAT;
        } else if (best - 1 - dmin < 0 || best + 1 - dmin >= D) {
            out = (float)best; // reference indexes out of bounds here (only when best stayed 0 and dmin != 0)
        } else {
            // neighbours c1 = cost_local[best-1], c2 = cost_local[best+1]: fetch from the owning lanes
            const int i1 = best - 1 - dmin, i2 = best + 1 - dmin;
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int k = 0; k < VPL; k++) {
                const float a = __shfl(c[i][k], i1 / VPL, 64);
                const float b = __shfl(c[i][k], i2 / VPL, 64);
                if ((i1 % VPL) == k) c1 = a;
                if ((i2 % VPL) == k) c2 = b;
            }
            out = adc_subpixel(best, c1, c2, bc);
        }
        if (lane == 0) disp[pix0 + i] = out;
    }
}

// ------------------------------------------------------------------------------ right view, diagonal bands
// The right-view candidates of pixel xr are cost(xr + d, y, d), d = dmin .. dmax-1: a DIAGONAL of the row slab.
// Gathered per pixel (kernel above, lanes = disparities) that is 128 different cache lines per pixel and the
// texture-address unit becomes the bound (0.8 ms, 1.3 TB/s).  Here a wave owns 64 CONSECUTIVE right pixels
// (lane l = pixel base + l) and walks the pixel columns c = base + dmin + t, t = 0 .. 63 + D - 1: at step t lane l
// looks at disparity index t - l of column c, i.e. the 64 lanes read 64 consecutive floats of ONE pixel's cost vector
// (256 contiguous bytes per load) and every volume element is read exactly once by exactly one lane.  Each lane
// runs the reference's own sequential scan (ADCensusStereo.cpp:276-300): strict '<' update in increasing d (lowest d
// wins ties), remembering the costs just before and just after the running minimum for the parabola.  No LDS, no
// cross-lane traffic, a coalesced 256-byte store at the end.  Works for any disparity range.
template <int UNROLL>
__global__ __launch_bounds__(256) void k_wta_right_band(const float* __restrict__ vol, float* __restrict__ disp, int W, int H,
                                                        int dmin, int D, int Dp)
{
    const int lane = threadIdx.x & 63;
    const int groups = (W + 63) >> 6;
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (gw >= groups * H) return;
    const int y = gw / groups, base = (gw - y * groups) << 6;
    const float* row = vol + (size_t)y * W * Dp;
    const int nsteps = 63 + D;
    float minc = ADC_LARGE_FLOAT, prev = 0.0f, c1 = 0.0f, c2 = 0.0f;
    int best = 0; // best_disparity keeps its initial 0 when nothing is below Large_Float
    bool capture = false;
    for (int t0 = 0; t0 < nsteps; t0 += UNROLL) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { // all loads of the block first (addresses do not depend on the scan state)
            const int t = t0 + u;
            const int c = base + dmin + t;
            const int cc = c < 0 ? 0 : (c >= W ? W - 1 : c);
            int di = t - lane;
            di = di < 0 ? 0 : (di >= D ? D - 1 : di); // clamped: loads stay unconditional and in bounds
            v[u] = row[(size_t)cc * Dp + di];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int t = t0 + u;
            const int c = base + dmin + t;
            const int di = t - lane;
            const bool act = di >= 0 && di < D && t < nsteps;                 // this lane has a candidate at this step
            const float cost = (c >= 0 && c < W) ? v[u] : ADC_LARGE_FLOAT;    // ADCensusStereo.cpp:281-283
            if (act) {
                if (capture) { c2 = cost; capture = false; }                  // cost_local[best + 1]
                if (cost < minc) { minc = cost; best = di + dmin; c1 = prev; capture = true; } // c1 = cost_local[best - 1]
                prev = cost;
            }
        }
    }
    const int x = base + lane;
    if (x >= W) return;
    float out;
    if (best == dmin || best == dmin + D - 1) out = (float)best; // right view keeps the integer at the range ends (:296-300)
    else if (best - 1 - dmin < 0 || best + 1 - dmin >= D) out = (float)best;
    else out = adc_subpixel(best, c1, c2, minc);
    disp[(size_t)y * W + x] = out;
}

hipError_t adc_launch_wta_left(adc_handle* h)
{
    const AdcParams& p = h->p;
    const long long P = (long long)p.W * p.H;
    const int wta_ppw = p.VPL <= 4 ? WTA_PPW_MAX : (p.VPL == 8 ? 4 : (p.VPL == 16 ? 2 : 1)); // == k_wta's WTA_PPW
    const unsigned blocks = (unsigned)((P + 4 * wta_ppw - 1) / (4 * wta_ppw));
#define LAUNCHL(V) hipLaunchKernelGGL((k_wta<V, false>), dim3(blocks), dim3(256), 0, h->heavy, h->vol_a, h->disp_l, p.W, p.H, p.dmin, p.D)
    if (p.VPL == 1) LAUNCHL(1);
    else if (p.VPL == 2) LAUNCHL(2);
    else if (p.VPL == 4) LAUNCHL(4);
    else if (p.VPL == 8) LAUNCHL(8);
    else if (p.VPL == 16) LAUNCHL(16);
    else LAUNCHL(32);
#undef LAUNCHL
    return hipGetLastError();
}

hipError_t adc_launch_wta(adc_handle* h)
{
    const AdcParams& p = h->p;
    const long long P = (long long)p.W * p.H;
    const int wta_ppw = p.VPL <= 4 ? WTA_PPW_MAX : (p.VPL == 8 ? 4 : (p.VPL == 16 ? 2 : 1)); // == k_wta's WTA_PPW
    const unsigned blocks = (unsigned)((P + 4 * wta_ppw - 1) / (4 * wta_ppw));
    static const bool band = [] { const char* e = getenv("ADC_WTA_BAND"); return e ? atoi(e) != 0 : true; }();
    const bool left = !h->wta_left_done; // the last scanline pass of the pipeline already produced the left view
    h->wta_left_done = 0;
#define LAUNCH(V)                                                                                                       \
    do {                                                                                                                \
        if (left) hipLaunchKernelGGL((k_wta<V, false>), dim3(blocks), dim3(256), 0, h->heavy, h->vol_a, h->disp_l, p.W, p.H, p.dmin, p.D); \
        if (band)                                                                                                       \
            hipLaunchKernelGGL((k_wta_right_band<8>), dim3((unsigned)((((p.W + 63) / 64) * p.H + 3) / 4)), dim3(256), 0, h->heavy, \
                               h->vol_a, h->disp_r, p.W, p.H, p.dmin, p.D, p.Dp);                                        \
        else                                                                                                            \
            hipLaunchKernelGGL((k_wta<V, true>), dim3(blocks), dim3(256), 0, h->heavy, h->vol_a, h->disp_r, p.W, p.H, p.dmin, p.D); \
    } while (0)
    if (p.VPL == 1) LAUNCH(1);
    else if (p.VPL == 2) LAUNCH(2);
    else if (p.VPL == 4) LAUNCH(4);
    else if (p.VPL == 8) LAUNCH(8);
    else if (p.VPL == 16) LAUNCH(16);
    else LAUNCH(32);
#undef LAUNCH
    return hipGetLastError();
}
