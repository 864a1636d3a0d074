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
//   K11 median        : level-synchronous wavefront t = x + 2y inside one workgroup
#include "adc_internal.h"
#include "adc_device_fn.h"

#include <vector>

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

// ------------------------------------------------------------------------------ K8 region voting
#define IRV_TILE 16

__global__ __launch_bounds__(256) void k_irv_begin(const uint8_t* __restrict__ label, const float* __restrict__ disp,
                                                   uint8_t* __restrict__ elig, int32_t* __restrict__ list,
                                                   int32_t* __restrict__ counters, int which, int P)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const bool e = (label[p] == which) && (disp[p] == ADC_INVALID_FLOAT);
    elig[p] = e ? 1 : 0;
    if (e) list[atomicAdd(&counters[0], 1)] = p;
}

// One wave evaluates the vote of one eligible pixel per loop trip.
__global__ __launch_bounds__(256) void k_irv_round(const int32_t* __restrict__ list, int n, float* disp,
                                                   const uint8_t* __restrict__ elig, const uchar4* __restrict__ arms,
                                                   const uint8_t* __restrict__ chg_prev, uint8_t* __restrict__ chg_next,
                                                   int32_t* __restrict__ counters, int W, int H, int dmin, int D, int irv_ts,
                                                   float irv_th, int round, int Lmax)
{
    __shared__ int hist_all[4][ADC_MAX_DISP_RANGE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* hist = hist_all[wave];
    const int tiles_x = (W + IRV_TILE - 1) / IRV_TILE;
    const int nwaves = gridDim.x * 4;
    int evals = 0;
    for (int e = blockIdx.x * 4 + wave; e < n; e += nwaves) {
        const int p = list[e];
        const int y = p / W, x = p - y * W;
        if (round > 0) {
            // re-evaluate only if a pixel of the dependency box (rows y-L..y, cols x-L..x+L) changed last round
            const int tx0 = adc_imax(0, x - Lmax) / IRV_TILE, tx1 = adc_imin(W - 1, x + Lmax) / IRV_TILE;
            const int ty0 = adc_imax(0, y - Lmax) / IRV_TILE, ty1 = y / IRV_TILE;
            const int ntx = tx1 - tx0 + 1, nt = ntx * (ty1 - ty0 + 1);
            bool dirty = false;
            for (int i = lane; i < nt; i += 64) dirty |= chg_prev[(ty0 + i / ntx) * tiles_x + tx0 + i % ntx] != 0;
            if (__ballot(dirty) == 0ull) continue;
        }
        for (int b = lane; b < D; b += 64) hist[b] = 0;
        const uchar4 arm = arms[p];
        const int sub = lane >> 4, sl = lane & 15;
        for (int t0 = -(int)arm.z; t0 <= (int)arm.w; t0 += 4) {
            const int t = t0 + sub;
            if (t <= (int)arm.w) {
                const int yt = y + t;
                const uchar4 arm2 = arms[yt * W + x];
                for (int s = -(int)arm2.x + sl; s <= (int)arm2.y; s += 16) {
                    const int q = yt * W + x + s;
                    float v = disp[q];
                    // eligible pixels of this pass: visible only if they precede p in raster order
                    // (already processed by the sequential scan), otherwise still invalid
                    if (elig[q] && q >= p) v = ADC_INVALID_FLOAT;
                    if (v != ADC_INVALID_FLOAT) {
                        const int b = (int)lroundf(v) - dmin; // multistep_refiner.cpp:193-196
                        if (b >= 0 && b < D) atomicAdd(&hist[b], 1);
                    }
                }
            }
        }
        // first maximum (lowest bin on ties) and total count (multistep_refiner.cpp:199-209)
        int bh = 0, bb = 0x7fffffff, cnt = 0;
        for (int b = lane; b < D; b += 64) {
            const int hv = hist[b];
            cnt += hv;
            if (hv > bh) { bh = hv; bb = b; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int oh = __shfl_xor(bh, m, 64), ob = __shfl_xor(bb, m, 64);
            cnt += __shfl_xor(cnt, m, 64);
            const bool take = (oh > bh) || (oh == bh && ob < bb);
            bh = take ? oh : bh;
            bb = take ? ob : bb;
        }
        const float nv = adc_vote_decide(bb, bh, cnt, dmin, irv_ts, irv_th);
        evals++;
        if (lane == 0) {
            const float cur = disp[p];
            if (__float_as_uint(cur) != __float_as_uint(nv)) {
                disp[p] = nv;
                chg_next[(y / IRV_TILE) * tiles_x + x / IRV_TILE] = 1;
                counters[1] = 1;
            }
        }
    }
    if (lane == 0 && evals) atomicAdd(&counters[2], evals);
}

hipError_t adc_run_region_voting(adc_handle* h)
{
    const AdcParams& p = h->p;
    const int P = p.W * p.H;
    const int tiles = ((p.W + IRV_TILE - 1) / IRV_TILE) * ((p.H + IRV_TILE - 1) / IRV_TILE);
    const int Lmax = adc_imax(0, adc_imin(p.opt.cross_L1, 255));
    h->vote_rounds = 0;
    h->vote_evals = 0;
    hipError_t e = hipSuccess;
    int32_t host_cnt[4];
    for (int it = 0; it < 5; it++) {         // multistep_refiner.cpp:167
        for (int k = 0; k < 2; k++) {        // mismatches, then occlusions (:170-171)
            if ((e = hipMemsetAsync(h->vote_counters, 0, 4 * sizeof(int32_t), h->stream)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_irv_begin, dim3((P + 255) / 256), dim3(256), 0, h->stream, h->label, h->disp_l, h->elig,
                               h->vote_list, h->vote_counters, k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION, P);
            if ((e = hipMemcpyAsync(host_cnt, h->vote_counters, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
            const int n = host_cnt[0];
            if (n == 0) continue;
            const unsigned blocks = (unsigned)adc_imin((n + 3) / 4, 256 * 8);
            for (int round = 0;; round++) {
                hipMemsetAsync(h->chg_b, 0, tiles, h->stream);
                hipMemsetAsync(h->vote_counters + 1, 0, sizeof(int32_t), h->stream);
                hipLaunchKernelGGL(k_irv_round, dim3(blocks), dim3(256), 0, h->stream, h->vote_list, n, h->disp_l, h->elig,
                                   reinterpret_cast<const uchar4*>(h->arms), h->chg_a, h->chg_b, h->vote_counters, p.W, p.H,
                                   p.dmin, p.D, p.opt.irv_ts, p.opt.irv_th, round, Lmax);
                if ((e = hipMemcpyAsync(host_cnt, h->vote_counters, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return e;
                if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return e;
                h->vote_rounds++;
                h->vote_evals += host_cnt[2];
                hipMemsetAsync(h->vote_counters + 2, 0, sizeof(int32_t), h->stream);
                uint8_t* t = h->chg_a;
                h->chg_a = h->chg_b;
                h->chg_b = t;
                if (host_cnt[1] == 0) break; // a full round without any change: fixed point == sequential result
                if (round > n + 8) return hipErrorUnknown; // cannot happen (triangular system converges in <= n rounds)
            }
        }
    }
    return hipGetLastError();
}

// --------------------------------------------------------------------------- K9 proper interpolation
__global__ __launch_bounds__(256) void k_interpolate(const float* __restrict__ din, float* __restrict__ dout,
                                                     const uint8_t* __restrict__ label, const uint8_t* __restrict__ img_l,
                                                     const double* __restrict__ sincos, int W, int H, int which,
                                                     int max_search)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const float d0 = din[p];
    if (!(label[p] == which && d0 == ADC_INVALID_FLOAT)) {
        dout[p] = d0;
        return;
    }
    const uint8_t* c0 = img_l + p * 3;
    const bool mismatch = which == ADC_LABEL_MISMATCH;
    int min_dist = 9999;
    float best = mismatch ? 0.0f : ADC_LARGE_FLOAT;
    bool any = false;
    for (int s = 0; s < 16; s++) {
        const double sina = sincos[2 * s], cosa = sincos[2 * s + 1];
        for (int m = 1; m < max_search; m++) {
            const int yy = (int)lround((double)y + (double)m * sina); // multistep_refiner.cpp:259-260
            const int xx = (int)lround((double)x + (double)m * cosa);
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) break;
            const float d = din[(size_t)yy * W + xx];
            if (d != ADC_INVALID_FLOAT) {
                any = true;
                if (mismatch) { // colour-nearest, first minimum (multistep_refiner.cpp:276-289)
                    const int dist = adc_color_dist_l1(c0, img_l + ((size_t)yy * W + xx) * 3);
                    if (min_dist > dist) { min_dist = dist; best = d; }
                } else { // smallest disparity (multistep_refiner.cpp:290-296)
                    best = d < best ? d : best;
                }
                break;
            }
        }
    }
    dout[p] = any ? best : 0.0f; // no ray hit: value-initialised fill (multistep_refiner.cpp:246,270-272)
}

hipError_t adc_launch_interpolation(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    const int dmaxa = p.dmax < 0 ? -p.dmax : p.dmax, dmina = p.dmin < 0 ? -p.dmin : p.dmin;
    const int max_search = dmaxa > dmina ? dmaxa : dmina; // multistep_refiner.cpp:236
    for (int k = 0; k < 2; k++) {
        hipLaunchKernelGGL(k_interpolate, grid, block, 0, h->stream, h->disp_l, h->disp_tmp, h->label, h->img_l, h->ray_sincos,
                           p.W, p.H, k == 0 ? ADC_LABEL_MISMATCH : ADC_LABEL_OCCLUSION, max_search);
        float* t = h->disp_l;
        h->disp_l = h->disp_tmp;
        h->disp_tmp = t;
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
// is lround(d) WITHOUT "- min_disparity" exactly like the reference (:329-331,:340).
__global__ void k_discontinuity_rows(float* __restrict__ disp, const uint8_t* __restrict__ edge, const float* __restrict__ vol,
                                     int W, int H, int Dp)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= H) return;
    float* row = disp + (size_t)y * W;
    for (int x = 1; x < W - 1; x++) {
        if (edge[(size_t)y * W + x] != 1) continue;
        const float d = row[x];
        if (d == ADC_INVALID_FLOAT) continue;
        const long di = lroundf(d);
        const float* cp = vol + ((size_t)y * W + x) * Dp;
        float c0 = cp[di];
        for (int k = 0; k < 2; k++) {
            const int x2 = k == 0 ? x - 1 : x + 1;
            const float d2 = row[x2];
            if (d2 == ADC_INVALID_FLOAT) continue;
            const long d2i = lroundf(d2);
            const float c = k == 0 ? cp[-Dp + d2i] : cp[Dp + d2i];
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
                       p.H, p.Dp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- K11 median
// In-place raster 3x3 median == recursive filter: the window of (x,y) holds already-filtered values
// at (x-1..x+1, y-1) and (x-1, y).  All pixels with equal t = x + 2y are independent; one workgroup
// walks t = 0 .. W-1+2(H-1) with one barrier per level.  Filtered values travel through a 4-deep LDS
// ring per row (ring[y][x&3]); unfiltered values are read from the (unmodified) input map.
__global__ __launch_bounds__(1024) void k_median_wavefront(const float* __restrict__ in, float* __restrict__ out, int W, int H)
{
    extern __shared__ __attribute__((aligned(16))) float mring[]; // [H][4]
    const int tid = threadIdx.x;
    const int nsteps = W + 2 * (H - 1);
    for (int t = 0; t < nsteps; t++) {
        for (int y = tid; y < H; y += 1024) {
            const int x = t - 2 * y;
            if (x < 0 || x >= W) continue;
            float v[9];
            int n = 0;
#pragma unroll
            for (int r = -1; r <= 1; r++)
#pragma unroll
                for (int c = -1; c <= 1; c++) {
                    const int row = y + r, col = x + c;
                    float val = ADC_INVALID_FLOAT; // padding sorts to the end
                    if (row >= 0 && row < H && col >= 0 && col < W) {
                        n++;
                        const bool filtered = (r < 0) || (r == 0 && c < 0);
                        val = filtered ? mring[row * 4 + (col & 3)] : in[(size_t)row * W + col];
                    }
                    v[(r + 1) * 3 + (c + 1)] = val;
                }
            adc_sort9(v);
            const int sel = n / 2; // wnd_data[size/2], adcensus_util.cpp:77
            float res = v[0];
#pragma unroll
            for (int i = 1; i < 9; i++) res = (i == sel) ? v[i] : res;
            out[(size_t)y * W + x] = res;
            mring[y * 4 + (x & 3)] = res;
        }
        __syncthreads();
    }
}

hipError_t adc_launch_median(adc_handle* h)
{
    const AdcParams& p = h->p;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_median_wavefront), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const size_t lds = (size_t)p.H * 4 * sizeof(float);
    hipLaunchKernelGGL(k_median_wavefront, dim3(1), dim3(1024), lds, h->stream, h->disp_l, h->disp_tmp, p.W, p.H);
    float* t = h->disp_l;
    h->disp_l = h->disp_tmp;
    h->disp_tmp = t;
    return hipGetLastError();
}
