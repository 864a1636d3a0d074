// k_arms.hip -- K3 cross arms, support-pixel counts and the colour-difference maps used by the
// scanline penalties.
//
// Replaces CrossAggregator::{BuildArms, FindHorizontalArm, FindVerticalArm, ComputeSupPixelCount}
// (cross_aggregator.cpp:76-86,135-325).  Integer-only, O(W*H*34); the left image (3*W*H bytes) is
// L2-resident, so one thread per pixel reading global memory directly is sufficient (this stage is
// <1% of the pipeline traffic).
#include "adc_internal.h"
#include "adc_device_fn.h"

// Left image packed B | G<<8 | R<<16 per pixel: the arm search reads ONE dword per visited pixel instead of three bytes
// (it is a chain of dependent L2 hits: 213 -> ~... us on the structured 1080p pair), and the interpolation gathers reuse it.
__global__ __launch_bounds__(256) void k_pack_bgr(const uint8_t* __restrict__ img, uint32_t* __restrict__ out, int P)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P) out[p] = (uint32_t)img[3 * (size_t)p] | ((uint32_t)img[3 * (size_t)p + 1] << 8) | ((uint32_t)img[3 * (size_t)p + 2] << 16);
}

// One arm from (x,y) along (dx,dy): rules of cross_aggregator.cpp:151-198.
// (The colour distance max(|dB|, |dG|, |dR|) on packed pixels: each channel masked out once per pixel, then ONE v_sad_u8 per channel and
// a v_max3 -- the search is bound by its vector-ALU instructions (2 273 per wave on the structured 1080p pair, SQ counters), and the
// byte-extract / subtract / abs form cost three times as many per distance.  Same integers: adc_color_dist_max_u32 is the definition.)
struct ArmPix { uint32_t b, g, r; };
__device__ __forceinline__ ArmPix arm_pix(uint32_t c) { return ArmPix{c & 0x000000FFu, c & 0x0000FF00u, c & 0x00FF0000u}; }
__device__ __forceinline__ int arm_dist(const ArmPix& a, const ArmPix& b)
{
    const uint32_t d0 = __builtin_amdgcn_sad_u8(a.b, b.b, 0u), d1 = __builtin_amdgcn_sad_u8(a.g, b.g, 0u), d2 = __builtin_amdgcn_sad_u8(a.r, b.r, 0u);
    return (int)(d2 > d1 ? (d2 > d0 ? d2 : d0) : (d1 > d0 ? d1 : d0));
}
// The four arms of a pixel in ONE branch-free loop (round 6, second session): the rules of cross_aggregator.cpp:151-198
// per step n = 0, 1, ... of an arm -- the pixel must lie inside the image, d1 = dist(pixel, anchor) < t1, from the second step on d2 =
// dist(pixel, previous pixel) < t1, and beyond L2 steps d1 < t2 -- as predicates: `room` (steps that stay inside the image) folds the
// bounds test into the step limit, a lane whose arm has ended stays where it is (its loads hit the same line), and the wave leaves the
// loop when no arm of its 64 pixels is alive.  The loop-with-four-breaks form spent more scalar instructions on execution masks (2 725
// per wave) than vector instructions on pixels (2 273), once per direction; here the four directions' loads are in flight together.
__device__ __forceinline__ uchar4 arm_lengths(const uint32_t* __restrict__ img, int W, int H, int x, int y, int L1, int L2, int t1, int t2)
{
    const int p = y * W + x;
    const int lim = adc_imin(L1, 255); // MAX_ARM_LENGTH, cross_aggregator.h:22
    const int step[4] = {-1, +1, -W, +W};                       // left, right, top, bottom
    const int nmax[4] = {adc_imin(lim, x), adc_imin(lim, W - 1 - x), adc_imin(lim, y), adc_imin(lim, H - 1 - y)};
    const ArmPix c0 = arm_pix(img[p]);
    ArmPix cl[4] = {c0, c0, c0, c0};
    int q[4] = {p, p, p, p}, len[4] = {0, 0, 0, 0};
    bool alive[4];
#pragma unroll
    for (int k = 0; k < 4; k++) alive[k] = nmax[k] > 0;
    for (int n = 0; __any(alive[0] || alive[1] || alive[2] || alive[3]); n++) {
        ArmPix c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            q[k] += alive[k] ? step[k] : 0;
            c[k] = arm_pix(img[q[k]]);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int d1 = arm_dist(c[k], c0), d2 = arm_dist(c[k], cl[k]);
            const bool ok = d1 < t1 && (n == 0 || d2 < t1) && !(n + 1 > L2 && d1 >= t2);
            alive[k] = alive[k] && ok;
            len[k] += alive[k] ? 1 : 0;
            cl[k] = c[k];
            alive[k] = alive[k] && n + 1 < nmax[k];
        }
    }
    return make_uchar4((unsigned char)len[0], (unsigned char)len[1], (unsigned char)len[2], (unsigned char)len[3]);
}

__global__ __launch_bounds__(256) void k_build_arms(const uint32_t* __restrict__ img_l, uchar4* __restrict__ arms, int W,
                                                    int H, int L1, int L2, int t1, int t2, int* __restrict__ armmax)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    __shared__ int smax[2];
    if (threadIdx.x < 2) smax[threadIdx.x] = 0;
    __syncthreads();
    uchar4 a = make_uchar4(0, 0, 0, 0);
    if (x < W && y < H) {
    a = arm_lengths(img_l, W, H, x, y, L1, L2, t1, t2); // {left, right, top, bottom}
    arms[(size_t)y * W + x] = a;
    }
    // image-wide maximum arm per direction (lets the aggregation pick its window depth from the data)
    atomicMax(&smax[0], adc_imax((int)a.x, (int)a.y));
    atomicMax(&smax[1], adc_imax((int)a.z, (int)a.w));
    __syncthreads();
    // monotone maximum: skip the (serialising, same-address) atomic when the value already there is large enough;
    // a stale read can only be too small, which just means one redundant atomic
    if (threadIdx.x < 2 && smax[threadIdx.x] > __hip_atomic_load(&armmax[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&armmax[threadIdx.x], smax[threadIdx.x]);
}

// Support counts (cross_aggregator.cpp:271-325):
//   id 0 (horizontal first): cnt = sum_{t=-top..bottom} (left+right+1)(x, y+t)
//   id 1 (vertical first)  : cnt = sum_{t=-left..right} (top+bottom+1)(x+t, y)
// The walk over the pixel's region rows also yields the widest H arms of those rows -- the bounding box of the cross region, which
// the region-voting chain wants per work-list entry (k_voting.hip: read box of a vote, change tiles to watch): bbox[p] = {0, 0, widest
// left arm, widest right arm} (round 6: a kernel of its own, 48 us at 1080p, before).
__global__ __launch_bounds__(256) void k_sup_counts(const uchar4* __restrict__ arms, uint16_t* __restrict__ sup_h,
                                                    uint16_t* __restrict__ sup_v, int W, int H, uchar4* __restrict__ bbox)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uchar4 a = arms[(size_t)y * W + x];
    int ch = 0, cv = 0, mla = 0, mra = 0;
    for (int t = -(int)a.z; t <= (int)a.w; t++) {
        const uchar4 q = arms[(size_t)(y + t) * W + x];
        ch += (int)q.x + (int)q.y + 1;
        mla = adc_imax(mla, (int)q.x);
        mra = adc_imax(mra, (int)q.y);
    }
    bbox[(size_t)y * W + x] = make_uchar4(0, 0, (unsigned char)mla, (unsigned char)mra);
    for (int t = -(int)a.x; t <= (int)a.y; t++) {
        const uchar4 q = arms[(size_t)y * W + x + t];
        cv += (int)q.z + (int)q.w + 1;
    }
    sup_h[(size_t)y * W + x] = (uint16_t)ch;
    sup_v[(size_t)y * W + x] = (uint16_t)cv;
}

// Colour-difference maps: dh[y][x] = ColorDist((x,y),(x-1,y)) (0 at x=0), dv[y][x] = ColorDist((x,y),(x,y-1))
// (0 at y=0), for both images.  They are what ScanlineOptimizer evaluates on the fly as d1 / d2
// (scanline_optimizer.cpp:114-126, :223-235); forward passes read [p], backward passes read [p+1].
__global__ __launch_bounds__(256) void k_color_diffs(const uint8_t* __restrict__ img_l, const uint8_t* __restrict__ img_r,
                                                     uint8_t* __restrict__ lh, uint8_t* __restrict__ lv,
                                                     uint8_t* __restrict__ rh, uint8_t* __restrict__ rv, int W, int H)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const uint8_t* img = blockIdx.z == 0 ? img_l : img_r;
    uint8_t* dh = blockIdx.z == 0 ? lh : rh;
    uint8_t* dv = blockIdx.z == 0 ? lv : rv;
    const uint8_t* p = img + ((size_t)y * W + x) * 3;
    dh[(size_t)y * W + x] = x > 0 ? (uint8_t)adc_color_dist_max(p, p - 3) : 0;
    dv[(size_t)y * W + x] = y > 0 ? (uint8_t)adc_color_dist_max(p, p - (size_t)W * 3) : 0;
}

// Packed per-pixel records for the aggregation passes: {arm_lo, arm_hi, divisor(u16)}.
//   rec_h[y][x]  = {left, right, sup_v}   (H passes march along x; the dividing H pass is the 2nd pass of a
//                                          vertical-first iteration -> vec_sup_count_[1])
//   rec_v[x][y]  = {top, bottom, sup_h}   (V passes march along y: stored TRANSPOSED so a line is contiguous)
// rec2_* (register-ring kernels, k_aggregate_rr.h): 8 bytes {lob | span << 8 | divisor << 16, RN(1 / divisor)} with
// lob = arm_lo + L + 1 (the BIASED arm: first ring slot of the span = write slot - lob, L = the arm limit = ring
// half-depth), span = arm_lo + arm_hi + 1 (both <= 255 for L <= 84; the register rings need L <= 35) and the
// correctly rounded reciprocal (IEEE division here: no fast-math) that Markstein's division sequence starts from.
__global__ __launch_bounds__(256) void k_make_records(const uchar4* __restrict__ arms, const uint16_t* __restrict__ sup_h,
                                                      const uint16_t* __restrict__ sup_v, uint32_t* __restrict__ rec_h,
                                                      uint32_t* __restrict__ rec_v, uint2* __restrict__ rec2_h,
                                                      uint2* __restrict__ rec2_v, int W, int H, int L)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const uchar4 a = arms[p];
    const uint32_t ch = sup_h[p], cv = sup_v[p];
    rec_h[p] = (uint32_t)a.x | ((uint32_t)a.y << 8) | (cv << 16);
    rec_v[(size_t)x * H + y] = (uint32_t)a.z | ((uint32_t)a.w << 8) | (ch << 16);
    if (rec2_h) {
        const uint32_t nh = (uint32_t)a.x + a.y + 1u, nv = (uint32_t)a.z + a.w + 1u;
        const uint32_t bias = (uint32_t)L + 1u;
        rec2_h[p] = make_uint2((((uint32_t)a.x + bias) & 255u) | ((nh & 255u) << 8) | (cv << 16), __float_as_uint(1.0f / (float)cv));
        rec2_v[(size_t)x * H + y] = make_uint2((((uint32_t)a.z + bias) & 255u) | ((nv & 255u) << 8) | (ch << 16), __float_as_uint(1.0f / (float)ch));
    }
}

hipError_t adc_launch_records(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_make_records, grid, block, 0, h->heavy, reinterpret_cast<const uchar4*>(h->arms), h->sup_h, h->sup_v,
                       h->rec_h, h->rec_v, reinterpret_cast<uint2*>(h->rec2_h), reinterpret_cast<uint2*>(h->rec2_v), p.W, p.H,
                       adc_imax(0, adc_imin(p.opt.cross_L1, 255)));
    return hipGetLastError();
}

// The part of the arms stage that needs only the LEFT image (the aggregation arms come from the left image alone,
// cross_aggregator.cpp:76-86): packed pixels, arms + image-wide maxima, support counts.  The pipeline runs it FIRST, while the right
// image of a host caller is still on the bus (capi.hip: run_heavy).
hipError_t adc_launch_arms_left(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    hipMemsetAsync(h->armmax, 0, 4 * sizeof(int), h->heavy); // [0],[1] maxima, [3] "assumed ring depth too small" flag (k_aggregate.hip)
    hipLaunchKernelGGL(k_pack_bgr, dim3((p.W * p.H + 255) / 256), dim3(256), 0, h->heavy, h->img_l, h->bgrx_l, p.W * p.H);
    h->bgrx_valid = 1;
    hipLaunchKernelGGL(k_build_arms, grid, block, 0, h->heavy, h->bgrx_l, reinterpret_cast<uchar4*>(h->arms), p.W, p.H,
                       p.opt.cross_L1, p.opt.cross_L2, p.opt.cross_t1, p.opt.cross_t2, h->armmax);
    hipLaunchKernelGGL(k_sup_counts, grid, block, 0, h->heavy, reinterpret_cast<const uchar4*>(h->arms), h->sup_h, h->sup_v,
                       p.W, p.H, reinterpret_cast<uchar4*>(h->irv_bbox));
    return hipGetLastError();
}
// Support counts + region boxes alone, from the arms in HBM (debug surface: a stage test writes the arms and runs the region voting
// without the arms stage in front of it)
hipError_t adc_launch_sup_counts(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    hipLaunchKernelGGL(k_sup_counts, grid, block, 0, h->stream, reinterpret_cast<const uchar4*>(h->arms), h->sup_h, h->sup_v,
                       p.W, p.H, reinterpret_cast<uchar4*>(h->irv_bbox));
    return hipGetLastError();
}
// ... and the part that reads BOTH images: the colour-step maps of the scanline penalties (scanline_optimizer.cpp:114-126), and
// the right-image arms of the opt-in paper mode.
hipError_t adc_launch_arms_rest(adc_handle* h)
{
    const AdcParams& p = h->p;
    dim3 grid((p.W + 63) / 64, (p.H + 3) / 4, 1), block(256, 1, 1);
    if ((h->paper & ADC_PAPER_RIGHT_ARMS) && h->arms_r) { // opt-in paper mode (k_paper.hip): the same arms on the RIGHT image
        hipLaunchKernelGGL(k_pack_bgr, dim3((p.W * p.H + 255) / 256), dim3(256), 0, h->heavy, h->img_r, h->bgrx_r, p.W * p.H);
        hipLaunchKernelGGL(k_build_arms, grid, block, 0, h->heavy, h->bgrx_r, reinterpret_cast<uchar4*>(h->arms_r), p.W, p.H,
                           p.opt.cross_L1, p.opt.cross_L2, p.opt.cross_t1, p.opt.cross_t2, h->armmax_r);
    }
    dim3 grid2(grid.x, grid.y, 2);
    hipLaunchKernelGGL(k_color_diffs, grid2, block, 0, h->heavy, h->img_l, h->img_r, h->cdiff_lh, h->cdiff_lv, h->cdiff_rh,
                       h->cdiff_rv, p.W, p.H);
    return hipGetLastError();
}
hipError_t adc_launch_arms(adc_handle* h) // (the whole stage: debug surface)
{
    const hipError_t e = adc_launch_arms_left(h);
    return e != hipSuccess ? e : adc_launch_arms_rest(h);
}
