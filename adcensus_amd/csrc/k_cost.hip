// k_cost.hip -- K1 gray + 9x7 census, K2 AD-census cost volume.
//
// Replaces CostComputor::{ComputeGray, CensusTransform, ComputeCost} (cost_computor.cpp:58-121)
// and adcensus_util::{census_transform_9x7, Hamming64} (adcensus_util.cpp:10-53).
//
// K2 is a pure streaming write of the volume (algorithmic bytes: V = 4*W*H*Dp written once,
// inputs 2*(3+8)*W*H): lanes = disparities, one wave-store of Dp floats per pixel (256/512/1024 B
// contiguous), the right image row segment and its census strings staged in LDS, the two
// exponentials replaced by host-built tables (A[766], C[64]) so the result is bit-identical to
// glibc expf on the host.
#include "adc_internal.h"
#include "adc_device_fn.h"

// ------------------------------------------------------------------------------------------- K1
#define CT_W 32
#define CT_H 8
#define CT_LW (CT_W + 6)
#define CT_LH (CT_H + 8)

__global__ __launch_bounds__(CT_W* CT_H) void k_gray_census(const uint8_t* __restrict__ img_l,
                                                             const uint8_t* __restrict__ img_r,
                                                             uint8_t* __restrict__ gray_l, uint8_t* __restrict__ gray_r,
                                                             uint64_t* __restrict__ census_l,
                                                             uint64_t* __restrict__ census_r, int W, int H, int census5)
{
    __shared__ uint8_t tile[CT_LH][CT_LW + 2];
    const uint8_t* img = blockIdx.z == 0 ? img_l : img_r;
    uint8_t* gray = blockIdx.z == 0 ? gray_l : gray_r;
    uint64_t* census = blockIdx.z == 0 ? census_l : census_r;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    const int tid = threadIdx.y * CT_W + threadIdx.x;
    for (int i = tid; i < CT_LH * CT_LW; i += CT_W * CT_H) {
        const int ly = i / CT_LW, lx = i % CT_LW;
        const int gx = x0 + lx - 3, gy = y0 + ly - 4;
        uint8_t g = 0;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            const uint8_t* p = img + ((size_t)gy * W + gx) * 3;
            g = adc_gray(p[0], p[1], p[2]);
        }
        tile[ly][lx] = g;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int lx = threadIdx.x + 3, ly = threadIdx.y + 4;
    const uint8_t c = tile[ly][lx];
    gray[(size_t)y * W + x] = c;
    uint64_t v = 0;
    // interior only, whole transform skipped for tiny images (adcensus_util.cpp:12,17-18); others stay 0
    if (census5) { // opt-in paper mode (k_paper.hip): 5x5 window, same conventions -- 25 bits, MSB first, interior only
        if (W > 5 && H > 5 && y >= 2 && y < H - 2 && x >= 2 && x < W - 2) {
#pragma unroll
            for (int r = -2; r <= 2; r++)
#pragma unroll
                for (int cc = -2; cc <= 2; cc++) {
                    v <<= 1;
                    v += (tile[ly + r][lx + cc] < c) ? 1u : 0u;
                }
        }
    } else if (W > 9 && H > 7 && y >= 4 && y < H - 4 && x >= 3 && x < W - 3) {
#pragma unroll
        for (int r = -4; r <= 4; r++)
#pragma unroll
            for (int cc = -3; cc <= 3; cc++) {
                v <<= 1;
                v += (tile[ly + r][lx + cc] < c) ? 1u : 0u;
            }
    }
    census[(size_t)y * W + x] = v;
}

hipError_t adc_launch_gray_census(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + CT_W - 1) / CT_W, (p.H + CT_H - 1) / CT_H, 2), block(CT_W, CT_H, 1);
    hipLaunchKernelGGL(k_gray_census, grid, block, 0, h->heavy, h->img_l, h->img_r, h->gray_l, h->gray_r, h->census_l,
                       h->census_r, p.W, p.H, (h->paper & ADC_PAPER_CENSUS5X5) ? 1 : 0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------- K2
#define COST_TX 32           // pixels of one row per block
#define COST_MAXSEG (COST_TX + ADC_MAX_DISP_RANGE)

template <int VPL>
__global__ __launch_bounds__(256) void k_cost(const uint8_t* __restrict__ img_l, const uint8_t* __restrict__ img_r,
                                              const uint64_t* __restrict__ census_l,
                                              const uint64_t* __restrict__ census_r, const float* __restrict__ lut_ad,
                                              const float* __restrict__ lut_census, float* __restrict__ vol, int W,
                                              int H, int dmin, int D)
{
    constexpr int Dp = 64 * VPL;
    __shared__ float sA[768];
    __shared__ float sC[64];
    __shared__ uint32_t sBgr[COST_MAXSEG];
    __shared__ uint64_t sCen[COST_MAXSEG];

    const int y = blockIdx.y;
    const int x0 = blockIdx.x * COST_TX;
    const int tid = threadIdx.x;
    for (int i = tid; i < 766; i += 256) sA[i] = lut_ad[i];
    if (tid < 64) sC[tid] = lut_census[tid];
    // right-image columns needed: xr = x - d, x in [x0, x0+TX), d in [dmin, dmin+D)
    const int xr_lo = x0 - (dmin + D - 1);
    const int nseg = COST_TX + D - 1;
    for (int i = tid; i < nseg; i += 256) {
        const int xr = xr_lo + i;
        uint32_t c = 0xFFFFFFFFu; // out-of-image marker (cost_computor.cpp:101-104)
        uint64_t cs = 0;
        if (xr >= 0 && xr < W) {
            const uint8_t* pr = img_r + ((size_t)y * W + xr) * 3;
            c = (uint32_t)pr[0] | ((uint32_t)pr[1] << 8) | ((uint32_t)pr[2] << 16);
            cs = census_r[(size_t)y * W + xr];
        }
        sBgr[i] = c;
        sCen[i] = cs;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    for (int px = wave; px < COST_TX; px += 4) {
        const int x = x0 + px;
        if (x >= W) break;
        const uint8_t* pl = img_l + ((size_t)y * W + x) * 3;
        const int bl = pl[0], gl = pl[1], rl = pl[2];
        const uint64_t cl = census_l[(size_t)y * W + x];
        float out[VPL];
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            const int di = lane * VPL + k;
            float c = 0.0f; // padding
            if (di < D) {
                const int xr = x - (di + dmin);
                const uint32_t pr = sBgr[xr - xr_lo];
                if (pr == 0xFFFFFFFFu) {
                    c = 1.0f;
                } else {
                    const int ad = adc_iabs(bl - (int)(pr & 255u)) + adc_iabs(gl - (int)((pr >> 8) & 255u)) +
                                   adc_iabs(rl - (int)((pr >> 16) & 255u));
                    const int hm = __popcll(cl ^ sCen[xr - xr_lo]);
                    c = sA[ad] - sC[hm]; // == ((1 - ea) + 1) - ec, cost_computor.cpp:117
                }
            }
            out[k] = c;
        }
        float* dst = vol + ((size_t)y * W + x) * Dp + lane * VPL;
        if constexpr (VPL == 1) dst[0] = out[0];
        else if constexpr (VPL == 2) *reinterpret_cast<float2*>(dst) = make_float2(out[0], out[1]);
        else {
#pragma unroll
            for (int q = 0; q < VPL; q += 4) *reinterpret_cast<float4*>(dst + q) = make_float4(out[q], out[q + 1], out[q + 2], out[q + 3]);
        }
    }
}

hipError_t adc_launch_cost(adc_handle* h, float* vol_out)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + COST_TX - 1) / COST_TX, p.H, 1), block(256, 1, 1);
#define LAUNCH(V)                                                                                                     \
    hipLaunchKernelGGL(k_cost<V>, grid, block, 0, h->heavy, h->img_l, h->img_r, h->census_l, h->census_r, h->lut_ad, \
                       h->lut_census, vol_out, p.W, p.H, p.dmin, p.D)
    if (p.VPL == 1) LAUNCH(1);
    else if (p.VPL == 2) LAUNCH(2);
    else if (p.VPL == 4) LAUNCH(4);
    else if (p.VPL == 8) LAUNCH(8);
    else if (p.VPL == 16) LAUNCH(16);
    else LAUNCH(32);
#undef LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------- packed pixel records (fused cost)
// Inputs of the cost computation that is fused into the first aggregation pass (k_agg_march<.., COSTIN>):
// per pixel {B | G<<8 | R<<16, census lo, census hi, 0}.  The right-image rows are padded on both sides with
// out-of-image markers (bgrx = 0xFFFFFFFF -> cost 1.0, cost_computor.cpp:101-104), so the marching kernel needs
// no bounds logic: row pitch = padl + W + padr.
__global__ __launch_bounds__(256) void k_cost_records(const uint8_t* __restrict__ img_l, const uint8_t* __restrict__ img_r,
                                                      const uint64_t* __restrict__ census_l,
                                                      const uint64_t* __restrict__ census_r, uint4* __restrict__ lrec,
                                                      uint4* __restrict__ rrec, int W, int H, int pitch, int padl)
{
    const int y = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < pitch) {
        const int c = i - padl;
        uint4 r = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
        if (c >= 0 && c < W) {
            const size_t p = (size_t)y * W + c;
            const uint64_t cs = census_r[p];
            r = make_uint4((uint32_t)img_r[3 * p] | ((uint32_t)img_r[3 * p + 1] << 8) | ((uint32_t)img_r[3 * p + 2] << 16),
                           (uint32_t)cs, (uint32_t)(cs >> 32), 0u);
        }
        rrec[(size_t)y * pitch + i] = r;
    }
    if (i < W) {
        const size_t p = (size_t)y * W + i;
        const uint64_t cs = census_l[p];
        lrec[p] = make_uint4((uint32_t)img_l[3 * p] | ((uint32_t)img_l[3 * p + 1] << 8) | ((uint32_t)img_l[3 * p + 2] << 16),
                             (uint32_t)cs, (uint32_t)(cs >> 32), 0u);
    }
}

hipError_t adc_launch_cost_records(adc_handle* h)
{
    const AdcParams& p = h->p;
    hipLaunchKernelGGL(k_cost_records, dim3((h->rrec_pitch + 255) / 256, p.H), dim3(256), 0, h->heavy, h->img_l, h->img_r,
                       h->census_l, h->census_r, reinterpret_cast<uint4*>(h->cost_lrec), reinterpret_cast<uint4*>(h->cost_rrec),
                       p.W, p.H, h->rrec_pitch, h->rrec_padl);
    return hipGetLastError();
}

// ---------------------------------------------------------------------- volume pad / unpad (debug)
__global__ void k_pad_volume(const float* __restrict__ src, float* __restrict__ dst, size_t P, int D, int Dp, int to_padded)
{
    const size_t n = P * (size_t)Dp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i / Dp;
        const int d = (int)(i % Dp);
        if (to_padded) dst[i] = d < D ? src[px * D + d] : 0.0f;
        else if (d < D) dst[px * D + d] = src[i];
    }
}

hipError_t adc_launch_pad_volume(adc_handle* h, const float* src_HWD, float* dst_HWDp)
{
    const AdcParams& p = h->p;
    hipLaunchKernelGGL(k_pad_volume, dim3(2048), dim3(256), 0, h->stream, src_HWD, dst_HWDp, (size_t)p.W * p.H, p.D, p.Dp, 1);
    return hipGetLastError();
}
hipError_t adc_launch_unpad_volume(adc_handle* h, const float* src_HWDp, float* dst_HWD)
{
    const AdcParams& p = h->p;
    hipLaunchKernelGGL(k_pad_volume, dim3(2048), dim3(256), 0, h->stream, src_HWDp, dst_HWD, (size_t)p.W * p.H, p.D, p.Dp, 0);
    return hipGetLastError();
}
