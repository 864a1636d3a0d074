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

// Cross-lane primitives.  DPP=true: data-parallel-primitive moves (no LDS crossbar round trip):
// wave_shr:1 / wave_shl:1 for the d-1 / d+1 neighbours, row_ror butterflies + 4 readlanes for the
// wave minimum.  DPP=false: ds_bpermute based __shfl (kept as a cross-check, ADC_SO_DPP=0).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}

template <bool DPP>
__device__ __forceinline__ float wave_min_f32(float v)
{
    if constexpr (DPP) {
        // six in-place v_min_f32 with a DPP source (costs are finite, never NaN: an exact selection); lanes without
        // a DPP source / outside the row mask keep their value.  Written as asm because the builtin route costs 4
        // instructions per stage (copy, v_mov_dpp, canonicalise, min); "s_nop 1" = the 2 wait states a DPP read
        // needs after a VALU write of the same register, which the assembler cannot insert inside an asm block.
        asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"  /* 16-lane row minima */
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" /* rows 1,3 += rows 0,2 */
            "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" /* row 3 = all */
            "s_nop 1"
            : "+v"(v));
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); // uniform (SGPR) result
    } else {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float o = __shfl_xor(v, m, 64);
            v = o < v ? o : v;
        }
        return v;
    }
}

// lane l gets lane l-1's value (lane 0: fill)
template <bool DPP>
__device__ __forceinline__ float lane_up(float v, float fill, int lane)
{
    if constexpr (DPP) return dpp_mov<0x138>(fill, v); // wave_shr:1, lane 0 keeps `old` = fill
    const float o = __shfl_up(v, 1, 64);
    return lane == 0 ? fill : o;
}
// lane l gets lane l+1's value (lane 63: fill)
template <bool DPP>
__device__ __forceinline__ float lane_down(float v, float fill, int lane)
{
    if constexpr (DPP) return dpp_mov<0x130>(fill, v); // wave_shl:1, lane 63 keeps `old` = fill
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

// ------------------------------------------------------------------------------ penalty-class maps
// The (P1,P2) class of (pixel, disparity) depends only on the two images: it is precomputed once per
// Match for the four pass types into cls[pass][pixel][lane] (one byte per lane holding the 2-bit classes
// of its VPL disparities), so the DP kernel's per-step inputs are two fully coalesced loads (512 B data
// + 64 B classes per pixel at D=128) and the sticky-d2 logic (adc_so_d2_column) runs in a massively
// parallel kernel instead of on the sequential critical path.
//   pass 0: L->R   d1 = dh_left[y][x]      d2 = dh_right[y][col]
//   pass 1: R->L   d1 = dh_left[y][x+1]    d2 = dh_right[y][col+1]
//   pass 2: T->B   d1 = dv_left[y][x]      d2 = dv_right[y][col]
//   pass 3: B->T   d1 = dv_left[y+1][x]    d2 = dv_right[y+1][col]
// with col = adc_so_d2_column(x, dmin, d, W) (or "use d1" when it returns -1).
// One thread produces the class bytes of 4 consecutive lanes of one pixel for all four passes
// (dword stores: a wave writes 4 pixels x 64 B contiguous per pass).
template <int VPL>
__global__ __launch_bounds__(256) void k_so_classes(const uint8_t* __restrict__ lh, const uint8_t* __restrict__ lv,
                                                    const uint8_t* __restrict__ rh, const uint8_t* __restrict__ rv,
                                                    uint8_t* __restrict__ cls, int W, int H, int dmin, int D, int tso)
{
    const long long P = (long long)W * H;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid >> 4;
    if (pix >= P) return;
    const int quad = (int)(gid & 15); // lanes 4*quad .. 4*quad+3
    const int y = (int)(pix / W), x = (int)(pix - (long long)y * W);
    int col[4 * VPL];
#pragma unroll
    for (int j = 0; j < 4 * VPL; j++) col[j] = adc_so_d2_column(x, dmin, quad * 4 * VPL + j, W);
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const bool vert = pass >= 2, bwd = (pass & 1) != 0;
        const bool has_pred = vert ? (bwd ? y + 1 < H : y > 0) : (bwd ? x + 1 < W : x > 0);
        uint32_t word = 0;
        if (has_pred) {
            const int sx = vert ? x : (bwd ? x + 1 : x);
            const int sy = vert ? (bwd ? y + 1 : y) : y;
            const uint8_t* dl = vert ? lv : lh;
            const uint8_t* dr = (vert ? rv : rh) + (size_t)sy * W;
            const int shift = (!vert && bwd) ? 1 : 0;
            const int d1 = dl[(size_t)sy * W + sx];
            int d2[4 * VPL];
#pragma unroll
            for (int j = 0; j < 4 * VPL; j++) d2[j] = (int)dr[(col[j] >= 0 ? col[j] : 0) + shift];
#pragma unroll
            for (int j = 0; j < 4 * VPL; j++) {
                const int dd2 = col[j] >= 0 ? d2[j] : d1;
                const int l = j / VPL, k = j % VPL; // lane within the quad, disparity within the lane
                // VPL <= 2: classes stored pre-scaled as LDS table offsets (k = 0: class*8 in bits 3-4, k = 1: class*32
                // in bits 5-6, see SO_STEP); VPL = 4: four packed 2-bit fields
                word |= (uint32_t)adc_so_penalty_class(d1, dd2, tso) << (8 * l + 2 * k + (VPL <= 2 ? 3 : 0));
            }
        }
        *reinterpret_cast<uint32_t*>(cls + ((size_t)pass * P + pix) * 64 + quad * 4) = word;
    }
}

hipError_t adc_launch_so_classes(adc_handle* h)
{
    const AdcParams& p = h->p;
    const long long P = (long long)p.W * p.H;
    const unsigned blocks = (unsigned)((P * 16 + 255) / 256);
#define CLS_LAUNCH(V)                                                                                                    \
    hipLaunchKernelGGL(k_so_classes<V>, dim3(blocks), dim3(256), 0, h->heavy, h->cdiff_lh, h->cdiff_lv, h->cdiff_rh, \
                       h->cdiff_rv, h->so_cls, p.W, p.H, p.dmin, p.D, p.opt.so_tso)
    if (p.VPL == 1) CLS_LAUNCH(1);
    else if (p.VPL == 2) CLS_LAUNCH(2);
    else CLS_LAUNCH(4);
#undef CLS_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- DP kernel
#define SO_PF 16

template <int VPL>
struct SoElem {
    float c[VPL]; // data term
    int cls;      // packed 2-bit penalty classes of this lane's VPL disparities
};

struct SoGeom {
    int W, H, dir, path, plen, d0, lane;
};

template <bool VERT>
__device__ __forceinline__ size_t so_pixel(const SoGeom& g, int i)
{
    const int m = g.dir > 0 ? i : g.plen - 1 - i;
    return VERT ? (size_t)m * g.W + g.path : (size_t)g.path * g.W + m;
}

template <int VPL, bool VERT>
__device__ __forceinline__ SoElem<VPL> so_load(const SoGeom& g, const float* __restrict__ src, const uint8_t* __restrict__ cls, int i)
{
    constexpr int Dp = 64 * VPL;
    const size_t pix = so_pixel<VERT>(g, i);
    SoElem<VPL> e;
    vload<VPL>(src + pix * Dp + g.d0, e.c);
    e.cls = cls[pix * 64 + g.lane];
    return e;
}

// VERT=false: path = image row `path`, marching in x.  VERT=true: path = column, marching in y.
// dir=+1 forward, -1 backward.  cls = class map of this pass type.
template <int VPL, bool VERT, bool DPP>
__global__ __launch_bounds__(256) void k_scanline(const float* __restrict__ src, float* __restrict__ dst,
                                                  const uint8_t* __restrict__ cls, int W, int H, int D, int dir, float P1a,
                                                  float P1b, float P1c, float P2a, float P2b, float P2c)
{
    constexpr int Dp = 64 * VPL;
    // (P1,P2) by penalty class, addressed with the pre-scaled class bits of the class byte: table A has 8-byte
    // entries (class*8), table B 32-byte entries (class*32); class 3 == class 2 (both differences above the threshold)
    __shared__ float2 so_tabA[4];
    __shared__ float2 so_tabB[16];
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        const float2 pp = make_float2(c == 0 ? P1a : (c == 1 ? P1b : P1c), c == 0 ? P2a : (c == 1 ? P2b : P2c));
        so_tabA[c] = pp;
        so_tabB[4 * c] = pp;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int npaths = VERT ? W : H;
    SoGeom g;
    g.W = W; g.H = H; g.dir = dir; g.lane = lane;
    g.path = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
    if (g.path >= npaths) return;
    g.plen = VERT ? H : W;
    g.d0 = lane * VPL; // first disparity index of this lane

    float Lp[VPL]; // previous path element's costs; padding lanes (d >= D) hold the sentinel
    float minLp;
    {
        const size_t pix = so_pixel<VERT>(g, 0);
        float c[VPL];
        vload<VPL>(src + pix * Dp + g.d0, c);
        vstore<VPL>(dst + pix * Dp + g.d0, c); // first pixel: dst = src (scanline_optimizer.cpp:99,208)
        float lmin = ADC_LARGE_FLOAT; // sentinels take part in the first minimum (scanline_optimizer.cpp:107-110)
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            Lp[k] = (g.d0 + k) < D ? c[k] : ADC_LARGE_FLOAT;
            lmin = Lp[k] < lmin ? Lp[k] : lmin;
        }
        minLp = wave_min_f32<DPP>(lmin);
    }
    if (g.plen <= 1) return;

// one DP step for path element I with inputs E (a macro keeps every array in registers)
#define SO_STEP(I, E)                                                                                      \
    do {                                                                                                   \
        const float up_ = lane_up<DPP>(Lp[VPL - 1], ADC_LARGE_FLOAT, lane); /* L(q, d0-1), sentinel at d=-1 */ \
        const float dn_ = lane_down<DPP>(Lp[0], ADC_LARGE_FLOAT, lane); /* L(q, d0+VPL), sentinel at d=D */ \
        float out_[VPL];                                                                                   \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            float P1_, P2_;                                                                                \
            if constexpr (VPL <= 2) { /* one LDS read of the (P1,P2) pair at the pre-scaled class offset */ \
                const float2 pp_ = *reinterpret_cast<const float2*>(                                       \
                    reinterpret_cast<const char*>(k == 0 ? so_tabA : so_tabB) + ((E).cls & (k == 0 ? 0x18 : 0x60))); \
                P1_ = pp_.x; P2_ = pp_.y;                                                                  \
            } else {                                                                                       \
                const int cls_ = ((E).cls >> (2 * k)) & 3;                                                 \
                P1_ = cls_ == 0 ? P1a : (cls_ == 1 ? P1b : P1c);                                           \
                P2_ = cls_ == 0 ? P2a : (cls_ == 1 ? P2b : P2c);                                           \
            }                                                                                              \
            const float lm1_ = k == 0 ? up_ : Lp[k == 0 ? 0 : k - 1];                                      \
            const float lp1_ = k == VPL - 1 ? dn_ : Lp[k == VPL - 1 ? k : k + 1];                          \
            const float l1_ = Lp[k];                                                                       \
            const float l2_ = lm1_ + P1_;                                                                  \
            const float l3_ = lp1_ + P1_;                                                                  \
            const float l4_ = minLp + P2_;                                                                 \
            /* min(min(l1,l2),min(l3,l4)): all finite, never NaN -> v_min3/v_min are exact selections */   \
            const float mm_ = __builtin_fminf(__builtin_fminf(__builtin_fminf(l1_, l2_), l3_), l4_);       \
            float cs_ = (E).c[k] + mm_;                                                                    \
            cs_ = cs_ * 0.5f; /* == cs / 2 exactly (scanline_optimizer.cpp:151) */                         \
            out_[k] = cs_;                                                                                 \
        }                                                                                                  \
        SO_STORE(I, out_);                                                                                 \
        float omin_ = ADC_LARGE_FLOAT;                                                                     \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            Lp[k] = (g.d0 + k) < D ? out_[k] : ADC_LARGE_FLOAT;                                            \
            omin_ = __builtin_fminf(Lp[k], omin_);                                                         \
        }                                                                                                  \
        minLp = wave_min_f32<DPP>(omin_);                                                                  \
    } while (0)
// output store of the prefetch path: running pointer (path elements are visited in order)
#define SO_STORE(I, OUT)              \
    do {                              \
        vstore<VPL>(dpn, OUT);        \
        dpn += fstep;                 \
    } while (0)

    if constexpr (VPL <= 2) {
        // Prefetch ring with asm-issued loads and hand-counted vmcnt (same technique and the same reasons as
        // k_agg_march, see k_aggregate.hip): SO_PF path elements (data term + class byte) in flight.
        // VMEM ops per steady-state step in program order: [data load][class load] ... [output store].
        typedef typename VecT<VPL>::type vec_t;
        vec_t pfc[SO_PF];
        int pfk[SO_PF];
        // running pointers: next element to prefetch (data, classes) and next element to store
        const long long pstep = (long long)(VERT ? W : 1) * dir; // pixels per path step
        const long long fstep = pstep * Dp, cstep = pstep * 64;
        const size_t px1 = so_pixel<VERT>(g, 1);
        const float* spn = src + px1 * Dp + g.d0;
        const uint8_t* cpn = cls + px1 * 64 + g.lane;
        float* dpn = dst + px1 * Dp + g.d0;
#define SO_ISSUE(U, I)                                                                                         \
    do {                                                                                                       \
        if constexpr (VPL == 1) asm volatile("global_load_dword %0, %1, off" : "=v"(pfc[U]) : "v"(spn) : "memory"); \
        else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pfc[U]) : "v"(spn) : "memory");              \
        asm volatile("global_load_ubyte %0, %1, off" : "=v"(pfk[U]) : "v"(cpn) : "memory");                    \
        spn += fstep;                                                                                          \
        cpn += cstep;                                                                                          \
    } while (0)
#define SO_TAKE(U, WAITN, E)                                                                                   \
    do {                                                                                                       \
        vec_t tc_;                                                                                             \
        int tk_;                                                                                               \
        if constexpr (VPL == 1)                                                                                \
            asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"                         \
                         : "=&v"(tc_), "=&v"(tk_) : "v"(pfc[U]), "v"(pfk[U]), "n"(WAITN) : "memory");          \
        else                                                                                                   \
            asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b64 %0, %2\n\tv_mov_b32 %1, %3"                         \
                         : "=&v"(tc_), "=&v"(tk_) : "v"(pfc[U]), "v"(pfk[U]), "n"(WAITN) : "memory");          \
        if constexpr (VPL == 1) (E).c[0] = tc_;                                                                \
        else { (E).c[0] = tc_.x; (E).c[VPL - 1] = tc_.y; }                                                     \
        (E).cls = tk_;                                                                                         \
    } while (0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // start the manual bookkeeping from an empty queue
        int i = 1;
        if (i + 2 * SO_PF <= g.plen) {
#pragma unroll
            for (int u = 0; u < SO_PF; u++) SO_ISSUE(u, i + u);
            // first iteration: younger ops = 2 per not-yet-consumed prologue slot + 3 per finished step
#define SO_FIRST(U)                                     \
    do {                                                \
        SoElem<VPL> cur_;                               \
        SO_TAKE(U, 2 * (SO_PF - 1 - (U)) + 3 * (U), cur_); \
        SO_ISSUE(U, i + (U) + SO_PF);                   \
        SO_STEP(i + (U), cur_);                         \
    } while (0)
            static_assert(SO_PF == 16, "peeled first iteration written for SO_PF == 16");
            SO_FIRST(0); SO_FIRST(1); SO_FIRST(2); SO_FIRST(3); SO_FIRST(4); SO_FIRST(5); SO_FIRST(6); SO_FIRST(7);
            SO_FIRST(8); SO_FIRST(9); SO_FIRST(10); SO_FIRST(11); SO_FIRST(12); SO_FIRST(13); SO_FIRST(14); SO_FIRST(15);
#undef SO_FIRST
            i += SO_PF;
            for (; i + 2 * SO_PF <= g.plen; i += SO_PF) {
#pragma unroll
                for (int u = 0; u < SO_PF; u++) {
                    SoElem<VPL> cur;
                    SO_TAKE(u, 3 * (SO_PF - 1), cur); // steady state: 3 ops per younger step (+ own store): one stricter
                    SO_ISSUE(u, i + u + SO_PF);
                    SO_STEP(i + u, cur);
                }
            }
            // drain: the SO_PF elements still in flight are elements i .. i+SO_PF-1
#pragma unroll
            for (int u = 0; u < SO_PF; u++) {
                SoElem<VPL> cur;
                SO_TAKE(u, 0, cur);
                SO_STEP(i + u, cur);
            }
            i += SO_PF;
        }
        for (; i < g.plen; i++) { // tail (< 2*SO_PF elements): compiler-scheduled loads
            const SoElem<VPL> cur = so_load<VPL, VERT>(g, src, cls, i);
            SO_STEP(i, cur);
        }
#undef SO_ISSUE
#undef SO_TAKE
#undef SO_STORE
#define SO_STORE(I, OUT) vstore<VPL>(dst + so_pixel<VERT>(g, (I)) * Dp + g.d0, OUT)
    } else {
        // VPL == 4 (disparity range > 128): compiler-scheduled prefetch ring
        SoElem<VPL> pre[SO_PF];
#pragma unroll
        for (int u = 0; u < SO_PF; u++) pre[u] = so_load<VPL, VERT>(g, src, cls, adc_imin(1 + u, g.plen - 1));
        int i = 1;
        for (; i + SO_PF <= g.plen; i += SO_PF) {
#pragma unroll
            for (int u = 0; u < SO_PF; u++) {
                const SoElem<VPL> cur = pre[u];
                pre[u] = so_load<VPL, VERT>(g, src, cls, adc_imin(i + u + SO_PF, g.plen - 1));
                __builtin_amdgcn_sched_barrier(0);
                SO_STEP(i + u, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < SO_PF; u++) {
            if (i + u < g.plen) SO_STEP(i + u, pre[u]);
        }
    }
#undef SO_STEP
#undef SO_STORE
}

static bool so_use_dpp()
{
    static const bool v = [] { const char* e = getenv("ADC_SO_DPP"); return e ? atoi(e) != 0 : true; }();
    return v;
}

template <int VPL>
static hipError_t launch_so(adc_handle* h, const float* src, float* dst, bool vert, int dir)
{
    const AdcParams& p = h->p;
    const int npaths = vert ? p.W : p.H;
    const unsigned blocks = (unsigned)((npaths + 3) / 4);
    const int pass = (vert ? 2 : 0) + (dir > 0 ? 0 : 1);
    const uint8_t* cls = h->so_cls + (size_t)pass * p.W * p.H * 64;
#define SO_LAUNCH(VERT_, DPP_)                                                                                         \
    hipLaunchKernelGGL((k_scanline<VPL, VERT_, DPP_>), dim3(blocks), dim3(256), 0, h->heavy, src, dst, cls, p.W, p.H, \
                       p.D, dir, h->so_P1[0], h->so_P1[1], h->so_P1[2], h->so_P2[0], h->so_P2[1], h->so_P2[2])
    const bool dpp = so_use_dpp();
    if (vert) { if (dpp) SO_LAUNCH(true, true); else SO_LAUNCH(true, false); }
    else { if (dpp) SO_LAUNCH(false, true); else SO_LAUNCH(false, false); }
#undef SO_LAUNCH
    return hipGetLastError();
}

template <int VPL>
static hipError_t run_so(adc_handle* h, int passes)
{
    // scanline_optimizer.cpp:54-60 (cost_aggr_ == vol_a, cost_init_ == vol_b)
    hipError_t e = adc_launch_so_classes(h);
    if (e == hipSuccess) e = launch_so<VPL>(h, h->vol_a, h->vol_b, false, +1);
    if (e == hipSuccess && passes >= 2) e = launch_so<VPL>(h, h->vol_b, h->vol_a, false, -1);
    if (e == hipSuccess && passes >= 3) e = launch_so<VPL>(h, h->vol_a, h->vol_b, true, +1);
    if (e == hipSuccess && passes >= 4) e = launch_so<VPL>(h, h->vol_b, h->vol_a, true, -1);
    if (e == hipSuccess && (passes == 1 || passes == 3)) // debug: leave the partial result in vol_a
        e = hipMemcpyAsync(h->vol_a, h->vol_b, (size_t)h->p.W * h->p.H * h->p.Dp * sizeof(float), hipMemcpyDeviceToDevice,
                           h->heavy);
    return e;
}

hipError_t adc_launch_scanline(adc_handle* h, int passes)
{
    if (passes <= 0 || passes > 4) passes = 4;
    if (h->p.VPL == 1) return run_so<1>(h, passes);
    if (h->p.VPL == 2) return run_so<2>(h, passes);
    return run_so<4>(h, passes);
}
