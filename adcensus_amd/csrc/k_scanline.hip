// k_scanline.hip -- K5 scanline optimisation: 4 CHAINED semi-global DP passes.
//
// Replaces ScanlineOptimizer::{Optimize, ScanlineOptimizeLeftRight, ScanlineOptimizeUpDown}
// (scanline_optimizer.cpp:40-279).  L->R (a->b), R->L (b->a), T->B (a->b), B->T (b->a); the data
// term of a pass is the previous pass's output and the recurrence is
//     L(p,d) = ( C(p,d) + min( L(q,d), L(q,d-1)+P1, L(q,d+1)+P1, min_k L(q,k)+P2 ) ) / 2
// (divide by 2, no subtraction of the path minimum -- SURVEY.md A.5), with colour-adaptive
// P1/P2 chosen per (pixel, disparity) from d1 (left image step) and the "sticky" d2 (right image
// step at xr = x-d-dmin; closed form in adc_device_fn.h).
//
// MI355X mapping: one wave per path (row for L/R, column for U/D), lanes = disparities, lane l
// holds VPL consecutive disparities in registers, so d+-1 neighbours are in-lane except at the
// lane edges (one DPP wave-shift each way) and min_k is one wave reduction.  The pass is a pure
// stream of the volume (read V + write V); the dependent chain per pixel is short (~10 VALU), so
// the cost vectors / penalty bytes of the next SO_PF pixels are kept in flight in registers.
#include "adc_internal.h"
#include "adc_device_fn.h"

#define SO_PF 8

__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float o = __shfl_xor(v, m, 64);
        v = o < v ? o : v;
    }
    return v;
}

// lane l gets lane l-1's value (lane 0: fill)
__device__ __forceinline__ float lane_up(float v, float fill, int lane)
{
    const float o = __shfl_up(v, 1, 64);
    return lane == 0 ? fill : o;
}
// lane l gets lane l+1's value (lane 63: fill)
__device__ __forceinline__ float lane_down(float v, float fill, int lane)
{
    const float o = __shfl_down(v, 1, 64);
    return lane == 63 ? fill : o;
}

template <int VPL>
struct VecT;
template <>
struct VecT<1> { typedef float type; };
template <>
struct VecT<2> { typedef float2 type; };
template <>
struct VecT<4> { typedef float4 type; };

template <int VPL>
__device__ __forceinline__ void vload(const float* p, float* r)
{
    if constexpr (VPL == 1) r[0] = p[0];
    else if constexpr (VPL == 2) { const float2 v = *reinterpret_cast<const float2*>(p); r[0] = v.x; r[1] = v.y; }
    else { const float4 v = *reinterpret_cast<const float4*>(p); r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
}
template <int VPL>
__device__ __forceinline__ void vstore(float* p, const float* r)
{
    if constexpr (VPL == 1) p[0] = r[0];
    else if constexpr (VPL == 2) *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
    else *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
}

// Inputs of one path element: data term c[], d1 (wave-uniform) and the per-lane d2 of each of the VPL
// disparities (raw diff-map byte; whether d1 is used instead is decided at consumption).
template <int VPL>
struct SoElem {
    float c[VPL];
    int d2[VPL];
    int d1;
};

struct SoGeom {
    int W, H, dmin, dir, path, plen, d0;
    int vzero; // per-lane zero the compiler cannot see through (prevents an early readfirstlane + wait on d1)
};

template <bool VERT>
__device__ __forceinline__ void so_coord(const SoGeom& g, int i, int& x, int& y)
{
    const int m = g.dir > 0 ? i : g.plen - 1 - i;
    if (VERT) { x = g.path; y = m; } else { x = m; y = g.path; }
}

// diff maps: forward pass reads [p], backward pass reads [p + one step] (see k_arms.hip)
template <int VPL, bool VERT>
__device__ __forceinline__ SoElem<VPL> so_load(const SoGeom& g, const float* __restrict__ src,
                                                const uint8_t* __restrict__ cd_left, const uint8_t* __restrict__ cd_right,
                                                int i)
{
    constexpr int Dp = 64 * VPL;
    int x, y;
    so_coord<VERT>(g, i, x, y);
    SoElem<VPL> e;
    vload<VPL>(src + ((size_t)y * g.W + x) * Dp + g.d0, e.c);
    const int sx = VERT ? x : (g.dir > 0 ? x : x + 1);
    const int sy = VERT ? (g.dir > 0 ? y : y + 1) : y;
    e.d1 = cd_left[(size_t)sy * g.W + sx + g.vzero]; // vzero: opaque per-lane 0 keeps this a plain VMEM load
    const uint8_t* row = cd_right + (size_t)sy * g.W; // row of the right-image diff map
    const int shift = VERT ? 0 : (g.dir > 0 ? 0 : 1);
#pragma unroll
    for (int k = 0; k < VPL; k++) {
        const int col = adc_so_d2_column(x, g.dmin, g.d0 + k, g.W);
        // unconditional load of the raw byte (a branch, or any ALU on the loaded value here, would make the
        // compiler wait for it at issue time); "use d1 instead" is resolved when the element is consumed
        e.d2[k] = (int)row[(col >= 0 ? col : 0) + shift];
    }
    return e;
}

// VERT=false: path = image row `path`, marching in x.  VERT=true: path = column, marching in y.
// dir=+1 forward, -1 backward.
template <int VPL, bool VERT>
__global__ __launch_bounds__(256) void k_scanline(const float* __restrict__ src, float* __restrict__ dst,
                                                  const uint8_t* __restrict__ cd_left,  // left-image diff map (h or v)
                                                  const uint8_t* __restrict__ cd_right, // right-image diff map (h or v)
                                                  int W, int H, int dmin, int D, int dir, int tso, float P1a, float P1b,
                                                  float P1c, float P2a, float P2b, float P2c)
{
    constexpr int Dp = 64 * VPL;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int npaths = VERT ? W : H;
    SoGeom g;
    g.W = W; g.H = H; g.dmin = dmin; g.dir = dir;
    g.path = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
    if (g.path >= npaths) return;
    g.plen = VERT ? H : W;
    g.d0 = lane * VPL; // first disparity index of this lane
    asm volatile("v_mov_b32 %0, 0" : "=v"(g.vzero));

    float Lp[VPL]; // previous path element's costs; padding lanes (d >= D) hold the sentinel
    float minLp;
    {
        int x, y;
        so_coord<VERT>(g, 0, x, y);
        float c[VPL];
        vload<VPL>(src + ((size_t)y * W + x) * Dp + g.d0, c);
        vstore<VPL>(dst + ((size_t)y * W + x) * Dp + g.d0, c); // first pixel: dst = src (scanline_optimizer.cpp:99,208)
        float lmin = ADC_LARGE_FLOAT; // sentinels take part in the first minimum (scanline_optimizer.cpp:107-110)
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            Lp[k] = (g.d0 + k) < D ? c[k] : ADC_LARGE_FLOAT;
            lmin = Lp[k] < lmin ? Lp[k] : lmin;
        }
        minLp = wave_min_f32(lmin);
    }
    if (g.plen <= 1) return;

// one DP step for path element I with inputs E (a macro keeps every array in registers)
#define SO_STEP(I, E)                                                                                      \
    do {                                                                                                   \
        int sx_, sy_;                                                                                      \
        so_coord<VERT>(g, (I), sx_, sy_);                                                                  \
        const float up_ = lane_up(Lp[VPL - 1], ADC_LARGE_FLOAT, lane); /* L(q, d0-1), sentinel at d=-1 */  \
        const float dn_ = lane_down(Lp[0], ADC_LARGE_FLOAT, lane);    /* L(q, d0+VPL), sentinel at d=D */ \
        float out_[VPL];                                                                                   \
        float omin_ = ADC_LARGE_FLOAT;                                                                     \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            const int dd2_ = adc_so_d2_column(sx_, dmin, g.d0 + k, W) >= 0 ? (E).d2[k] : (E).d1;           \
            const int cls_ = adc_so_penalty_class((E).d1, dd2_, tso);                                      \
            const float P1_ = cls_ == 0 ? P1a : (cls_ == 1 ? P1b : P1c);                                   \
            const float P2_ = cls_ == 0 ? P2a : (cls_ == 1 ? P2b : P2c);                                   \
            const float lm1_ = k == 0 ? up_ : Lp[k == 0 ? 0 : k - 1];                                      \
            const float lp1_ = k == VPL - 1 ? dn_ : Lp[k == VPL - 1 ? k : k + 1];                          \
            const float l1_ = Lp[k];                                                                       \
            const float l2_ = lm1_ + P1_;                                                                  \
            const float l3_ = lp1_ + P1_;                                                                  \
            const float l4_ = minLp + P2_;                                                                 \
            const float m12_ = l2_ < l1_ ? l2_ : l1_;                                                      \
            const float m34_ = l4_ < l3_ ? l4_ : l3_;                                                      \
            const float mm_ = m34_ < m12_ ? m34_ : m12_;                                                   \
            float cs_ = (E).c[k] + mm_;                                                                    \
            cs_ = cs_ / 2; /* scanline_optimizer.cpp:151 */                                                \
            out_[k] = cs_;                                                                                 \
        }                                                                                                  \
        vstore<VPL>(dst + ((size_t)sy_ * W + sx_) * Dp + g.d0, out_);                                      \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            Lp[k] = (g.d0 + k) < D ? out_[k] : ADC_LARGE_FLOAT;                                            \
            omin_ = Lp[k] < omin_ ? Lp[k] : omin_;                                                         \
        }                                                                                                  \
        minLp = wave_min_f32(omin_);                                                                       \
    } while (0)

    // software prefetch ring: the inputs of the next SO_PF path elements stay in flight in registers.
    // Prefetch loads are unconditional (index clamped to the path end).
    SoElem<VPL> pre[SO_PF];
#pragma unroll
    for (int u = 0; u < SO_PF; u++) pre[u] = so_load<VPL, VERT>(g, src, cd_left, cd_right, adc_imin(1 + u, g.plen - 1));

    int i = 1;
    for (; i + SO_PF <= g.plen; i += SO_PF) {
#pragma unroll
        for (int u = 0; u < SO_PF; u++) {
            const SoElem<VPL> cur = pre[u];
            pre[u] = so_load<VPL, VERT>(g, src, cd_left, cd_right, adc_imin(i + u + SO_PF, g.plen - 1));
            __builtin_amdgcn_sched_barrier(0); // keep the refill loads ahead of the dependent chain, in program order
            SO_STEP(i + u, cur);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < SO_PF; u++) {
        if (i + u < g.plen) SO_STEP(i + u, pre[u]);
    }
#undef SO_STEP
}

template <int VPL>
static hipError_t launch_so(adc_handle* h, const float* src, float* dst, bool vert, int dir)
{
    const AdcParams& p = h->p;
    const int npaths = vert ? p.W : p.H;
    const unsigned blocks = (unsigned)((npaths + 3) / 4);
    const uint8_t* cdl = vert ? h->cdiff_lv : h->cdiff_lh;
    const uint8_t* cdr = vert ? h->cdiff_rv : h->cdiff_rh;
    if (vert)
        hipLaunchKernelGGL((k_scanline<VPL, true>), dim3(blocks), dim3(256), 0, h->stream, src, dst, cdl, cdr, p.W, p.H,
                           p.dmin, p.D, dir, p.opt.so_tso, h->so_P1[0], h->so_P1[1], h->so_P1[2], h->so_P2[0], h->so_P2[1],
                           h->so_P2[2]);
    else
        hipLaunchKernelGGL((k_scanline<VPL, false>), dim3(blocks), dim3(256), 0, h->stream, src, dst, cdl, cdr, p.W, p.H,
                           p.dmin, p.D, dir, p.opt.so_tso, h->so_P1[0], h->so_P1[1], h->so_P1[2], h->so_P2[0], h->so_P2[1],
                           h->so_P2[2]);
    return hipGetLastError();
}

template <int VPL>
static hipError_t run_so(adc_handle* h, int passes)
{
    // scanline_optimizer.cpp:54-60 (cost_aggr_ == vol_a, cost_init_ == vol_b)
    hipError_t e = launch_so<VPL>(h, h->vol_a, h->vol_b, false, +1);
    if (e == hipSuccess && passes >= 2) e = launch_so<VPL>(h, h->vol_b, h->vol_a, false, -1);
    if (e == hipSuccess && passes >= 3) e = launch_so<VPL>(h, h->vol_a, h->vol_b, true, +1);
    if (e == hipSuccess && passes >= 4) e = launch_so<VPL>(h, h->vol_b, h->vol_a, true, -1);
    if (e == hipSuccess && (passes == 1 || passes == 3)) // debug: leave the partial result in vol_a
        e = hipMemcpyAsync(h->vol_a, h->vol_b, (size_t)h->p.W * h->p.H * h->p.Dp * sizeof(float), hipMemcpyDeviceToDevice,
                           h->stream);
    return e;
}

hipError_t adc_launch_scanline(adc_handle* h, int passes)
{
    if (passes <= 0 || passes > 4) passes = 4;
    if (h->p.VPL == 1) return run_so<1>(h, passes);
    if (h->p.VPL == 2) return run_so<2>(h, passes);
    return run_so<4>(h, passes);
}
