// k_wta.hip -- K6 winner-takes-all + parabola sub-pixel, left and right view.
//
// Replaces ADCensusStereo::ComputeDisparity / ComputeDisparityRight (ADCensusStereo.cpp:188-310).
// Left:  first minimum over d (strict '>' => lowest d wins ties); best at either end of the range
//        => +inf; else best + (c1-c2)/(2*(c1+c2-2*cmin)).
// Right: candidate cost(x+d, y, d) if 0 <= x+d < W else Large_Float (can enter the parabola as a
//        neighbour); at the range ends the INTEGER best is stored (no invalidation).
// One wave per pixel, lanes = disparities; lexicographic (cost, d) wave arg-min.
#include "adc_internal.h"
#include <mutex>
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

// (left view only since round 6: the per-pixel gather form of the right view -- 128 cache lines per pixel -- was kept behind a switch
// no geometry selected; the right view runs k_wta_right_march or k_wta_right_band)
template <int VPL>
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
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const int di = lane * VPL + k;
            const float v = vol[(size_t)pix * Dp + di]; // padding lanes (di >= D) are masked below
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
            out = ADC_INVALID_FLOAT; // ADCensusStereo.cpp:228-231
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

// ------------------------------------------------------------------------------ right view, marching along a row
// The band kernel above reads every 512-byte pixel vector as two unaligned 256-byte pieces from two different waves (0.29 ms at
// 1080p = 3.6 TB/s).  Here a workgroup of 8 waves marches along a row and reads WHOLE pixel vectors, 64 per step, in full lines --
// every element of the volume once (segments: adc_wtam_plan) -- into an LDS ring of 256 vectors with a pitch of Dp + 1 floats:
// lane d writes element d of a vector (consecutive banks), and the scan of right pixel xr walks the diagonal (xr + d, d), which
// for the 64 pixels of a group (lane = pixel) is 64 addresses at a stride of Dp + 1 floats: consecutive banks again.  The loads
// of steps s + 1 and s + 2 are in flight (registers) while step s is written and group s - lag is scanned: wave q scans the
// disparities [q * Dp / 8, (q + 1) * Dp / 8) of the group's 64 pixels with the reference's strict '<' (lowest d wins inside a
// part), one wave combines the 8 parts in increasing d (strict '<' again: the first minimum overall), fetches the two
// neighbours of the winner from the ring and stores 64 disparities.  Step s + 1 goes into the ring slot that group s - lag no
// longer needs (4 slots of 64 vectors, lag <= 2), so the combining wave can still read while the others write.  D <= 128.
#ifndef WTAM_WAVES
#define WTAM_WAVES 8 // waves per workgroup (8 or 16)
#endif
template <int VPL>
__global__ __launch_bounds__(64 * WTAM_WAVES) void k_wta_right_march(const float* __restrict__ vol, float* __restrict__ disp, int W, int H, int dmin, int D,
                                                         int rows_full, int nseg, int segw)
{
    constexpr int Dp = 64 * VPL, P = Dp + 1, NW = WTAM_WAVES, VPW = 64 / NW, Dq = Dp / NW;
    extern __shared__ float wring[]; // [ADC_WTAM_RING][P], then the parts' results: float pmin[NW][64], int pbest[NW][64]
    float (*pmin)[64] = reinterpret_cast<float (*)[64]>(wring + ADC_WTAM_RING * P);
    int (*pbest)[64] = reinterpret_cast<int (*)[64]>(wring + ADC_WTAM_RING * P + NW * 64);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const AdcWtamUnit un = adc_wtam_unit((int)blockIdx.x, W, rows_full, nseg, segw);
    if (un.x0 >= un.x1) return;
    const int G = (un.x1 - un.x0 + 63) >> 6, lag = adc_wtam_lag(D), last = G - 1 + lag;
    const float* row = vol + (size_t)un.y * W * Dp;
    const int di0 = wave * Dq;
#ifndef WTAM_DEPTH
#define WTAM_DEPTH 2 // steps whose loads are in flight while a step is taken over (2 or 3)
#endif
    float bufA[VPW][VPL], bufB[VPW][VPL], cur[VPW][VPL];
#if WTAM_DEPTH == 3
    float bufC[VPW][VPL];
#endif
    const uint32_t lane4 = (uint32_t)lane * 4u;
    // vector i of the unit is the image column un.x0 + dmin + i; columns outside the image count as Large_Float (:281-283) --
    // their loads are clamped (and so are the loads of the steps behind the last one)
// The loads are issued from inline asm and taken over behind ONE hand-counted wait per step (the compiler's own vmcnt
    // bookkeeping drains the queue at the loop's back edge: built first, it waited for the loads it had just issued): when step s is
    // taken over, the 8 * VPL loads of step s + 1 are the only younger ones of the wave that may still be outstanding (the store of a
    // combining wave can only make the wait stronger); proven on the generated code by tools/check_async_loads.py.
#define WTAM_ISSUE(buf, s)                                                                                            \
    _Pragma("unroll") for (int j = 0; j < VPW; j++) {                                                                 \
        const int c = un.x0 + dmin + (s) * 64 + wave * VPW + j;                                                       \
        const float* vp = row + (size_t)(c < 0 ? 0 : (c >= W ? W - 1 : c)) * Dp;                                      \
        asm volatile("global_load_dword %0, %1, %2" ADC_VOL_NT_STR : "=v"(buf[j][0]) : "v"(lane4), "s"(vp) : "memory");              \
        if constexpr (VPL == 2) asm volatile("global_load_dword %0, %1, %2 offset:256" ADC_VOL_NT_STR : "=v"(buf[j][1]) : "v"(lane4), "s"(vp) : "memory"); \
    }
#define WTAM_TAKE(buf)                                                                                                \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WTAM_DEPTH - 1) * VPW * VPL) : "memory");                               \
    _Pragma("unroll") for (int j = 0; j < VPW; j++) {                                                                 \
        asm volatile("v_mov_b32 %0, %1" : "=v"(cur[j][0]) : "v"(buf[j][0]));                                          \
        if constexpr (VPL == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(cur[j][1]) : "v"(buf[j][1]));                  \
    }
#define WTAM_STEP(buf, s)                                                                                             \
    {                                                                                                                 \
        WTAM_TAKE(buf)                                                                                                \
        _Pragma("unroll") for (int j = 0; j < VPW; j++) {                                                             \
            const int i = (s) * 64 + wave * VPW + j, c = un.x0 + dmin + i;                                            \
            const bool in_img = c >= 0 && c < W;                                                                      \
            _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                           \
                wring[(i & (ADC_WTAM_RING - 1)) * P + lane + 64 * k] = in_img ? cur[j][k] : ADC_LARGE_FLOAT;          \
        }                                                                                                             \
        WTAM_ISSUE(buf, (s) + WTAM_DEPTH)                                                                             \
        __syncthreads();                                                                                              \
        const int g = (s) - lag, b = g * 64 + lane;                                                                   \
        if (g >= 0 && g < G) {                                                                                        \
            float mc = ADC_LARGE_FLOAT;                                                                               \
            int mb = -1;                                                                                              \
            _Pragma("unroll") for (int t = 0; t < Dq; t++) { /* (all reads of the part first; padding disparities never win) */ \
                const int di = di0 + t;                                                                               \
                const float v = wring[((b + di) & (ADC_WTAM_RING - 1)) * P + di];                                     \
                const float cost = di < D ? v : ADC_LARGE_FLOAT;                                                      \
                if (cost < mc) { mc = cost; mb = di; }                                                                \
            }                                                                                                         \
            pmin[wave][lane] = mc;                                                                                    \
            pbest[wave][lane] = mb;                                                                                   \
        }                                                                                                             \
        __syncthreads();                                                                                              \
        if (g >= 0 && g < G && wave == (g & (NW - 1))) {                                                              \
            float minc = ADC_LARGE_FLOAT;                                                                             \
            int bi = -1;                                                                                              \
            _Pragma("unroll") for (int q = 0; q < NW; q++) {                                                          \
                const float m = pmin[q][lane];                                                                        \
                const int mbq = pbest[q][lane];                                                                       \
                if (m < minc) { minc = m; bi = mbq; }                                                                 \
            }                                                                                                         \
            const int x = un.x0 + b;                                                                                  \
            if (x < un.x1) {                                                                                          \
                const int best = bi < 0 ? 0 : bi + dmin; /* best_disparity keeps its initial 0 when nothing is below Large_Float */ \
                const int i1 = best - 1 - dmin, i2 = best + 1 - dmin;                                                 \
                float out = (float)best; /* the integer at the range ends (:296-300) */                               \
                if (best != dmin && best != dmin + D - 1 && i1 >= 0 && i2 < D) {                                      \
                    const float c1 = wring[((b + i1) & (ADC_WTAM_RING - 1)) * P + i1];                                \
                    const float c2 = wring[((b + i2) & (ADC_WTAM_RING - 1)) * P + i2];                                \
                    out = adc_subpixel(best, c1, c2, minc);                                                           \
                }                                                                                                     \
                disp[(size_t)un.y * W + x] = out;                                                                     \
            }                                                                                                         \
        }                                                                                                             \
    }
    WTAM_ISSUE(bufA, 0)
    WTAM_ISSUE(bufB, 1)
#if WTAM_DEPTH == 3
    WTAM_ISSUE(bufC, 2)
#endif
#pragma clang loop unroll(disable)
    for (int s = 0; s <= last; s += WTAM_DEPTH) { // (the last round may add empty steps: their groups lie behind the unit, g >= G)
        WTAM_STEP(bufA, s)
        WTAM_STEP(bufB, s + 1)
#if WTAM_DEPTH == 3
        WTAM_STEP(bufC, s + 2)
#endif
    }
#undef WTAM_STEP
#undef WTAM_TAKE
#undef WTAM_ISSUE
}

hipError_t adc_launch_wta_left(adc_handle* h)
{
    const AdcParams& p = h->p;
    const long long P = (long long)p.W * p.H;
    const int wta_ppw = p.VPL <= 4 ? WTA_PPW_MAX : (p.VPL == 8 ? 4 : (p.VPL == 16 ? 2 : 1)); // == k_wta's WTA_PPW
    const unsigned blocks = (unsigned)((P + 4 * wta_ppw - 1) / (4 * wta_ppw));
#define LAUNCHL(V) hipLaunchKernelGGL((k_wta<V>), dim3(blocks), dim3(256), 0, h->heavy, h->vol_a, h->disp_l, p.W, p.H, p.dmin, p.D)
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
    // marching form of the right view (D <= 128): ADC_WTA_MARCH=0 selects the band kernel, ADC_WTA_NCU overrides the number of
    // workgroups the plan assumes to run at a time, ADC_WTA_NSEG the segments per row of the remainder rows (tests: segments on
    // small images)
    static const bool march_on = [] { const char* e = getenv("ADC_WTA_MARCH"); return e ? atoi(e) != 0 : true; }();
    static const int ncu_env = [] { const char* e = getenv("ADC_WTA_NCU"); return e ? atoi(e) : 0; }();
    static const int nseg_env = [] { const char* e = getenv("ADC_WTA_NSEG"); return e ? atoi(e) : 0; }();
    const bool left = !h->wta_left_done; // the last scanline pass of the pipeline already produced the left view
    h->wta_left_done = 0;
    if (march_on && p.VPL <= 2) {
        static std::mutex attr_mu; // (per-DEVICE function attribute and CU count: see adc_launch_aggregate)
        static bool attr_set[64] = {false};
        static int ncu_dev[64];
        const int dv = (h->device >= 0 && h->device < 64) ? h->device : 0;
        {
            std::lock_guard<std::mutex> lk(attr_mu);
            if (!attr_set[dv]) {
                hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wta_right_march<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (ea == hipSuccess)
                    ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wta_right_march<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (ea != hipSuccess) return ea;
                int n = 0;
                if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dv) != hipSuccess || n <= 0) n = 256;
                ncu_dev[dv] = n;
                attr_set[dv] = true;
            }
        }
        const AdcWtamPlan pl = adc_wtam_plan(p.W, p.H, p.D, ncu_env > 0 ? ncu_env : ncu_dev[dv], nseg_env);
        const size_t lds = ((size_t)ADC_WTAM_RING * (p.Dp + 1) + 2 * WTAM_WAVES * 64) * sizeof(float); // ring + the parts' results
        if (left) {
            const int wl = p.VPL <= 4 ? WTA_PPW_MAX : 1;
            const unsigned bl = (unsigned)((P + 4 * wl - 1) / (4 * wl));
            if (p.VPL == 1) hipLaunchKernelGGL((k_wta<1>), dim3(bl), dim3(256), 0, h->heavy, h->vol_a, h->disp_l, p.W, p.H, p.dmin, p.D);
            else hipLaunchKernelGGL((k_wta<2>), dim3(bl), dim3(256), 0, h->heavy, h->vol_a, h->disp_l, p.W, p.H, p.dmin, p.D);
        }
        if (p.VPL == 1)
            hipLaunchKernelGGL((k_wta_right_march<1>), dim3((unsigned)pl.units), dim3(64 * WTAM_WAVES), lds, h->heavy, h->vol_a, h->disp_r, p.W, p.H, p.dmin, p.D,
                               pl.rows_full, pl.nseg, pl.segw);
        else
            hipLaunchKernelGGL((k_wta_right_march<2>), dim3((unsigned)pl.units), dim3(64 * WTAM_WAVES), lds, h->heavy, h->vol_a, h->disp_r, p.W, p.H, p.dmin, p.D,
                               pl.rows_full, pl.nseg, pl.segw);
        return hipGetLastError();
    }
#define LAUNCH(V)                                                                                                       \
    do {                                                                                                                \
        if (left) hipLaunchKernelGGL((k_wta<V>), dim3(blocks), dim3(256), 0, h->heavy, h->vol_a, h->disp_l, p.W, p.H, p.dmin, p.D); \
        hipLaunchKernelGGL((k_wta_right_band<8>), dim3((unsigned)((((p.W + 63) / 64) * p.H + 3) / 4)), dim3(256), 0, h->heavy,     \
                           h->vol_a, h->disp_r, p.W, p.H, p.dmin, p.D, p.Dp);                                            \
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
