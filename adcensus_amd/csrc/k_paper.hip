// k_paper.hip -- OPT-IN "paper" modes: features of the AD-Census paper that the reference declares or
// stores but does not implement.  They are NOT the reference's behaviour (adc_set_paper_modes, default 0 = off) and have
// their own checker -- the test suite's plain-C restatement of the same definitions -- against which tests/test_gpu_paper.py compares them
// bit for bit.  Functional kernels, not tuned: one thread per volume element.
//   ADC_PAPER_CENSUS5X5    5x5 census window (adcensus_types.h:39-42 declares CensusSize::Census5x5): in k_cost.hip
//   ADC_PAPER_SO_SUM       the four scanline paths are computed independently from the aggregated volume and averaged
//                          (paper eq. 10) instead of chained (scanline_optimizer.cpp:54-60): k_scanline.hip + k_vol_accumulate
//   ADC_PAPER_RIGHT_ARMS   the support region of (p, d) is limited by BOTH images: arm = min(left arm at p, right arm at
//                          (x - d, y)) (the reference stores img_right_ for this, cross_aggregator.h:91, and never reads
//                          it); the divisor is the number of contributing cost values
#include "adc_internal.h"
#include "adc_device_fn.h"

__global__ __launch_bounds__(256) void k_vol_accumulate(float* __restrict__ acc, const float* __restrict__ src, size_t n, int first, int last)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = first ? src[i] : acc[i] + src[i]; // sum order: L->R, R->L, T->B, B->T
        if (last) v = v * 0.25f;
        acc[i] = v;
    }
}

hipError_t adc_paper_accumulate(adc_handle* h, float* acc, const float* src, int first, int last)
{
    const size_t n = (size_t)h->p.W * h->p.H * h->p.Dp;
    hipLaunchKernelGGL(k_vol_accumulate, dim3(4096), dim3(256), 0, h->heavy, acc, src, n, first, last);
    return hipGetLastError();
}

__device__ __forceinline__ uchar4 paper_arms_at(const uchar4* __restrict__ al, const uchar4* __restrict__ ar, int W, int x, int y, int d)
{
    uchar4 a = al[(size_t)y * W + x];
    const int xr = x - d;
    if (xr >= 0 && xr < W) {
        const uchar4 b = ar[(size_t)y * W + xr];
        a.x = a.x < b.x ? a.x : b.x; a.y = a.y < b.y ? a.y : b.y;
        a.z = a.z < b.z ? a.z : b.z; a.w = a.w < b.w ? a.w : b.w;
    }
    return a;
}

// one pass over the padded volume: VERT = sum along the column, DIVIDE = second pass of an iteration (divide by the number of
// contributing cost values = sum over this pass's span of the other direction's span lengths at the same disparity)
template <bool VERT, bool DIVIDE>
__global__ __launch_bounds__(256) void k_agg_rarms(const float* __restrict__ src, float* __restrict__ dst, const uchar4* __restrict__ al,
                                                   const uchar4* __restrict__ ar, int W, int H, int Dp, int D, int dmin)
{
    const size_t total = (size_t)W * H * Dp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / Dp;
        const int di = (int)(i % Dp);
        if (di >= D) { dst[i] = 0.0f; continue; } // padding disparities
        const int y = (int)(pix / W), x = (int)(pix - (size_t)y * W), d = di + dmin;
        const uchar4 a = paper_arms_at(al, ar, W, x, y, d);
        const int lo = VERT ? a.z : a.x, hi = VERT ? a.w : a.y;
        float acc = 0.0f;
        int cnt = 0;
        for (int t = -lo; t <= hi; t++) { // ordered sum from 0.0f, t = -arm .. +arm
            const int xx = VERT ? x : x + t, yy = VERT ? y + t : y;
            acc += src[((size_t)yy * W + xx) * Dp + di];
            if (DIVIDE) {
                const uchar4 b = paper_arms_at(al, ar, W, xx, yy, d);
                cnt += VERT ? (int)b.x + (int)b.y + 1 : (int)b.z + (int)b.w + 1;
            }
        }
        dst[i] = DIVIDE ? acc / (float)cnt : acc;
    }
}

// arms of the RIGHT image (same rule as the left ones, k_arms.hip) -- called by adc_launch_arms when the mode is on
hipError_t adc_paper_aggregate(adc_handle* h, int iterations)
{
    const AdcParams& p = h->p;
    const uchar4* al = reinterpret_cast<const uchar4*>(h->arms);
    const uchar4* ar = reinterpret_cast<const uchar4*>(h->arms_r);
    float* cur = h->vol_a;
    float* oth = h->vol_b;
    bool hf = true; // cross_aggregator.cpp:100: horizontal first, alternating
    const dim3 grid(256 * 16), block(256);
    for (int k = 0; k < iterations; k++) {
        if (hf) {
            hipLaunchKernelGGL((k_agg_rarms<false, false>), grid, block, 0, h->heavy, cur, oth, al, ar, p.W, p.H, p.Dp, p.D, p.dmin);
            hipLaunchKernelGGL((k_agg_rarms<true, true>), grid, block, 0, h->heavy, oth, cur, al, ar, p.W, p.H, p.Dp, p.D, p.dmin);
        } else {
            hipLaunchKernelGGL((k_agg_rarms<true, false>), grid, block, 0, h->heavy, cur, oth, al, ar, p.W, p.H, p.Dp, p.D, p.dmin);
            hipLaunchKernelGGL((k_agg_rarms<false, true>), grid, block, 0, h->heavy, oth, cur, al, ar, p.W, p.H, p.Dp, p.D, p.dmin);
        }
        hf = !hf;
    }
    h->agg_first_fused = 0;
    h->agg_launches = 0;
    h->agg_passes = 0;
    h->agg_kernel = "k_agg_rarms (paper mode: right-image arms, one thread per volume element)";
    return hipGetLastError();
}
