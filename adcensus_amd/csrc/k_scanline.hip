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
    else {
#pragma unroll
        for (int q = 0; q < VPL; q += 4) {
            const float4 v = *reinterpret_cast<const float4*>(p + q);
            r[q] = v.x; r[q + 1] = v.y; r[q + 2] = v.z; r[q + 3] = v.w;
        }
    }
}
template <int VPL>
__device__ __forceinline__ void vstore(float* p, const float* r)
{
    // (volume stores: streamed, ADC_VOL_STORE in adc_internal.h)
    typedef float so_f2 __attribute__((ext_vector_type(2)));
    typedef float so_f4 __attribute__((ext_vector_type(4)));
    if constexpr (VPL == 1) ADC_VOL_STORE(p, r[0]);
    else if constexpr (VPL == 2) { so_f2 v_ = {r[0], r[1]}; ADC_VOL_STORE(reinterpret_cast<so_f2*>(p), v_); }
    else {
#pragma unroll
        for (int q = 0; q < VPL; q += 4) { so_f4 v_ = {r[q], r[q + 1], r[q + 2], r[q + 3]}; ADC_VOL_STORE(reinterpret_cast<so_f4*>(p + q), v_); }
    }
}

// ------------------------------------------------------------------------------ penalty classes
// The (P1,P2) class of (pixel, disparity) is (d1 >= tso) + (d2' >= tso) (scanline_optimizer.cpp:129-141):
//   d1  = colour step of the LEFT image along the path at this pixel (one value per pixel),
//   d2' = colour step of the RIGHT image at column col = adc_so_d2_column(x, dmin, d, W), or d1 again when that
//         returns -1 (the "sticky d2" quirk, closed form in adc_device_fn.h):
//         use the right image  <=>  W >= 3  and  x - dmin >= 1  and  xr < W-1,   col = max(xr, 1),  xr = x - d - dmin.
//   pass 0: L->R   d1 = dh_left[y][x]      d2 = dh_right[y][col]
//   pass 1: R->L   d1 = dh_left[y][x+1]    d2 = dh_right[y][col+1]
//   pass 2: T->B   d1 = dv_left[y][x]      d2 = dv_right[y][col]
//   pass 3: B->T   d1 = dv_left[y+1][x]    d2 = dv_right[y+1][col]
// The DP passes are memory-bound with ~1 wave per SIMD, i.e. their ALUs idle, so the classes are derived INSIDE
// the DP kernel (no class-map pass over P*D elements, no class stream): a lane's VPL disparities need VPL
// consecutive bytes of the right-image step map (columns max(xr_last,1) .., one ubyte/ushort/dword gather per
// step from a 2 MB, L2-resident map; the clamp max(.,1) maps onto "read the same byte twice"), and d1 comes from
// a tiny PATH-ORDERED copy of the left-image step map: c1w[pass][path][g] = the four d1 bytes of path elements
// 1+4g .. 4+4g (element i sits at coordinate i on a forward path, plen-1-i on a backward one), one uniform
// dword per four steps.
struct SoC1Layout {
    long long off[4]; // dword offset of each pass
    int ngr[4];       // groups (of 4 path elements) per path
    long long total;  // dwords
};
static SoC1Layout so_c1_layout(int W, int H)
{
    SoC1Layout L;
    const int gh = (W - 1 + 3) / 4 > 0 ? (W - 1 + 3) / 4 : 1, gv = (H - 1 + 3) / 4 > 0 ? (H - 1 + 3) / 4 : 1;
    const long long sh = (long long)H * gh, sv = (long long)W * gv;
    L.ngr[0] = L.ngr[1] = gh;
    L.ngr[2] = L.ngr[3] = gv;
    L.off[0] = 0; L.off[1] = sh; L.off[2] = 2 * sh; L.off[3] = 2 * sh + sv;
    L.total = 2 * sh + 2 * sv;
    return L;
}
size_t adc_so_cls_bytes(int W, int H) { return (size_t)so_c1_layout(W, H).total * 4 + 64; }

__global__ __launch_bounds__(256) void k_so_c1(const uint8_t* __restrict__ lh, const uint8_t* __restrict__ lv,
                                               uint32_t* __restrict__ c1w, int W, int H, long long off1, long long off2,
                                               long long off3, int ngr_h, int ngr_v)
{
    const int pass = blockIdx.z;
    const bool vert = pass >= 2, bwd = (pass & 1) != 0;
    const int plen = vert ? H : W, npaths = vert ? W : H, ngr = vert ? ngr_v : ngr_h;
    // row paths: a thread = four consecutive bytes of a row, neighbouring threads = neighbouring groups (loads and stores coalesced).
    // Column paths (round 6): neighbouring LANES = neighbouring columns, a wave = one group of four rows -- the byte loads of a wave
    // fall into one line per row; with the row-path mapping every lane read its own line (64 lines per load: 43 us of this kernel's
    // 43 us at 1080p; the dword stores are the scattered side now: one per path and group, absorbed by the L2).
    // (one block index per pass: row passes = (row, 256 groups), column passes = (64 columns, 4 groups))
    const int nbx = vert ? (W + 63) / 64 : (ngr + 255) / 256;
    const int bx = (int)blockIdx.x % nbx, by = (int)blockIdx.x / nbx;
    const int path = vert ? bx * 64 + (int)(threadIdx.x & 63) : by;
    const int g = vert ? by * 4 + (int)(threadIdx.x >> 6) : bx * 256 + (int)threadIdx.x;
    if (path >= npaths || g >= ngr) return;
    const uint8_t* dl = vert ? lv : lh;
    uint32_t word = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int i0 = 1 + 4 * g + b;
        const int i = i0 < plen ? i0 : plen - 1; // clamped: loads stay unconditional
        const int m = bwd ? plen - 1 - i : i;
        const int x = vert ? path : m, y = vert ? m : path;
        const int sx = vert ? x : (bwd ? x + 1 : x);
        const int sy = vert ? (bwd ? y + 1 : y) : y;
        word |= (uint32_t)dl[(size_t)sy * W + sx] << (8 * b);
    }
    const long long off = pass == 0 ? 0 : (pass == 1 ? off1 : (pass == 2 ? off2 : off3));
    c1w[off + (long long)path * ngr + g] = word;
}

hipError_t adc_launch_so_classes(adc_handle* h, hipStream_t stream)
{
    const AdcParams& p = h->p;
    const SoC1Layout L = so_c1_layout(p.W, p.H);
    // blocks per pass: row passes rows x ceil(groups / 256), column passes ceil(columns / 64) x ceil(groups / 4)
    const long long nb_row = (long long)p.H * ((L.ngr[0] + 255) / 256), nb_col = (long long)((p.W + 63) / 64) * ((L.ngr[2] + 3) / 4);
    const long long nb = nb_row > nb_col ? nb_row : nb_col;
    if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_so_c1, dim3((unsigned)nb, 1, 4), dim3(256), 0, stream, h->cdiff_lh, h->cdiff_lv,
                       reinterpret_cast<uint32_t*>(h->so_cls), p.W, p.H, L.off[1], L.off[2], L.off[3], L.ngr[0], L.ngr[2]);
    return hipGetLastError();
}

// (the class derivation itself, adc_so_class_offsets, lives in adc_device_fn.h: it is shared with the CPU emulation)

// ------------------------------------------------------------------------------------- DP kernel
#define SO_PF 16

template <int VPL>
struct SoElem {
    float c[VPL];                // data term
    uint32_t rb[(VPL + 3) / 4];  // right-image step bytes, little-endian dwords (see adc_so_class_offsets)
    int c1;                      // left-image step byte d1
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

// byte offset into the right-image step map of the VPL bytes path element i needs (this lane)
template <int VPL, bool VERT>
__device__ __forceinline__ int so_rmap_offset(const SoGeom& g, int i, int cl_last, int dmin_unused)
{
    const int m = g.dir > 0 ? i : g.plen - 1 - i;
    const int x = VERT ? g.path : m, y = VERT ? m : g.path;
    const int sy = VERT ? (g.dir > 0 ? y : y + 1) : y;
    int xr = x - cl_last; // (negative min_disparity: up to x + |dmin| -- columns >= W-1 are never used by the class rule)
    xr = xr > g.W - 1 ? g.W - 1 : xr;
    return sy * g.W + (xr > 1 ? xr : 1) + ((!VERT && g.dir < 0) ? 1 : 0);
}
// same, from the coordinate m of the path element (x on a row path, y on a column path)
template <int VPL, bool VERT>
__device__ __forceinline__ int so_rmap_offset_m(const SoGeom& g, int m, int cl_last)
{
    const int x = VERT ? g.path : m, y = VERT ? m : g.path;
    const int sy = VERT ? (g.dir > 0 ? y : y + 1) : y;
    int xr = x - cl_last;
    xr = xr > g.W - 1 ? g.W - 1 : xr;
    return sy * g.W + (xr > 1 ? xr : 1) + ((!VERT && g.dir < 0) ? 1 : 0);
}
template <int VPL>
__device__ __forceinline__ uint32_t so_rmap_load(const uint8_t* __restrict__ rmap, int off)
{
    if constexpr (VPL == 1) return rmap[off];
    else if constexpr (VPL == 2) { uint16_t v; __builtin_memcpy(&v, rmap + off, 2); return v; }
    else { uint32_t v; __builtin_memcpy(&v, rmap + off, 4); return v; }
}
template <int VPL>
__device__ __forceinline__ void so_rmap_load_words(const uint8_t* __restrict__ rmap, int off, uint32_t* w)
{
    if constexpr (VPL <= 4) w[0] = so_rmap_load<VPL>(rmap, off);
    else {
#pragma unroll
        for (int q = 0; q < VPL / 4; q++) __builtin_memcpy(&w[q], rmap + off + 4 * q, 4);
    }
}

// VERT=false: path = image row `path`, marching in x.  VERT=true: path = column, marching in y.
// dir=+1 forward, -1 backward.  c1w = path-ordered d1 words of this pass type, rmap = right-image step map of the
// pass direction (horizontal / vertical), at least VPL bytes of slack behind its last element.
// WTA=true (last pass of the production pipeline only): the left-view winner-takes-all + sub-pixel step
// (ADCensusStereo::ComputeDisparity, ADCensusStereo.cpp:188-245; same rules as k_wta<VPL,false>) is evaluated on the
// final costs while they are still in registers and written to `disp`: the path minimum the recurrence needs anyway IS
// the winning cost, the winner is the lowest set bit of a ballot, its two neighbours come with two readlanes -- the
// separate pass that re-reads the whole volume disappears.
#ifndef SO_INTERIOR
#define SO_INTERIOR 1 // A/B switch: 0 = always the general class rule
#endif
#ifndef SO_FAST
#define SO_FAST 1 // A/B switch: 0 = never the short form of a whole interior chunk
#endif
// Prefetch slots of the VPL <= 2 kernels: REGISTERS THE COMPILER CANNOT ALLOCATE (it is limited to v0 .. v[SO_V0-1] by
// amdgpu_num_vgpr, like the ring of k_aggregate_rr2.h), named directly in the asm statements:
//   data slot U   v[SO_V0 + 2U : SO_V0 + 2U + 1]   (VPL = 1 uses the low register)
//   rmap slot U   v[SO_V0 + 32 + U]
//   d1 word G     v[SO_V0 + 48 + G]
// A load is in flight for 16 steps; in compiler-allocated registers nothing stops the register allocator from copying a slot
// while its load has not landed (it did, as soon as the steady state had two forms: tools/check_async_loads.py).
#define SO_V0 96
#define SO_RC(U) (SO_V0 + 2 * (U))
#define SO_RR(U) (SO_V0 + 32 + (U))
#define SO_RW(G) (SO_V0 + 48 + (G))
#define SO_CLOBBERS "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147"
// SEG (row passes, VPL <= 2): the path is cut into nseg verified segments (adc_device_fn.h: adc_so_seg_start), wave = (segment,
// path); `seam` = this pass's seam slots [path][nseg - 1][Dp].
// AGG (round 5, the L->R row pass of short-arm images): the pass ALSO computes the last aggregation pass (horizontal, dividing:
// cross_aggregator.cpp:327-394 with the arms of `aggrec`, k_arms.hip's rec_h = {arm_lo, arm_hi, divisor}) on its input -- `src`
// is the volume BEFORE that pass; the aggregated volume is never written or read back (one launch and 2 V of traffic less).
// The data stream runs SO_AGG_LA = 4 elements ahead of the recurrence; the last 9 raw pixel vectors live in a register shift
// ring (position k = raw element e + 4 - k); element e's value is the sequential f32 sum from 0.0f over t = -arm_lo .. +arm_hi
// (ring positions 4 + arm_lo down to 4 - arm_hi; the masked-out positions add +0.0f to a non-negative sum: exact) divided by
// the count, like the separate pass.  Arms up to 4 only (the host launches this form when the assumed depth allows it, and the
// other horizontal passes of the Match verify the assumption on the device).
#define SO_AGG_LA 4
template <int VPL, bool VERT, bool DPP, bool WTA, bool PIN, bool SEG = false, bool AGG = false>
__device__ __forceinline__ void so_body(const float* __restrict__ src, float* __restrict__ dst,
                                        const uint32_t* __restrict__ c1w, int ngr,
                                        const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso,
                                        int dir, float P1a, float P1b, float P1c, float P2a, float P2b, float P2c,
                                        float* __restrict__ disp, int allow_fast, int nseg = 1, int warm = 0, float* __restrict__ seam = nullptr,
                                        const uint32_t* __restrict__ aggrec = nullptr)
{
    static_assert(!AGG || (SEG && !PIN && !VERT && !WTA && VPL == 2), "fused aggregation: the L->R row pass, two disparities per lane, compiler-allocated slots");
    static_assert(!WTA || DPP, "the fused winner-takes-all relies on the uniform (SGPR) path minimum of the DPP reduction");
    static_assert(!SEG || (!WTA && !VERT && VPL <= 2), "verified segments: row passes of the asm-prefetch kernels");
    constexpr int Dp = 64 * VPL;
    // (P1,P2) by penalty class, 8-byte entries addressed with class*8 (selecting them with v_cndmask instead was
    // measured: no faster)
    __shared__ float2 so_tab[4];
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        so_tab[c] = make_float2(c == 0 ? P1a : (c == 1 ? P1b : P1c), c == 0 ? P2a : (c == 1 ? P2b : P2c));
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int npaths = VERT ? W : H;
    SoGeom g;
    g.W = W; g.H = H; g.dir = dir; g.lane = lane;
    g.path = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (int)(blockDim.x >> 6) + wave);
    int seg = 0;
    if constexpr (SEG) { // waves 0 .. npaths-1 run segment 0, the next npaths segment 1, ...
        seg = g.path / npaths;
        g.path -= seg * npaths;
        if (seg >= nseg) return;
    }
    if (g.path >= npaths) return;
    g.plen = VERT ? H : W;
    // the elements this wave visits: [e0, e1); a segment behind the first starts warm + 1 elements before its first output
    int e0 = 0, e1 = g.plen, efirst = 0;
    if constexpr (SEG) {
        e1 = adc_so_seg_start(g.plen, nseg, warm, seg + 1);
        if (seg > 0) {
            efirst = adc_so_seg_start(g.plen, nseg, warm, seg);
            e0 = efirst - warm - 1;
        }
    }
    const int plen_v = e1 - e0; // path length as this wave sees it
    g.d0 = lane * VPL; // first disparity index of this lane
    const uint32_t* c1p = c1w + (size_t)g.path * ngr; // this path's d1 words (uniform)
    const int cl_last = g.d0 + VPL - 1 + dmin;         // xr of the lane's last disparity = x - cl_last

    // ---- fused aggregation (AGG): the register ring of raw pixel vectors and the arm records of this row
    float agr[AGG ? 2 * SO_AGG_LA + 1 : 1][VPL];
    const uint32_t* recp = AGG ? aggrec + (size_t)g.path * W : nullptr; // (dir > 0: path element e is column e)
#define SO_AGG_PUSH(C)                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int t_ = 2 * SO_AGG_LA; t_ > 0; t_--)                                           \
            _Pragma("unroll") for (int k_ = 0; k_ < VPL; k_++) agr[t_][k_] = agr[t_ - 1][k_];                  \
        _Pragma("unroll") for (int k_ = 0; k_ < VPL; k_++) agr[0][k_] = (C)[k_];                               \
    } while (0)
#define SO_AGG_EVAL(EIDX, C)                                                                                   \
    do {                                                                                                       \
        const uint32_t rec_ = recp[(EIDX)];                                                                    \
        const int top_ = SO_AGG_LA + (int)(rec_ & 255u), bot_ = SO_AGG_LA - (int)((rec_ >> 8) & 255u);         \
        const float cnt_ = (float)(rec_ >> 16);                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < VPL; k_++) {                                                   \
            float acc_ = 0.0f;                                                                                 \
            _Pragma("unroll") for (int t_ = 2 * SO_AGG_LA; t_ >= 0; t_--) acc_ += (t_ <= top_ && t_ >= bot_) ? agr[t_][k_] : 0.0f; \
            (C)[k_] = acc_ / cnt_; /* cross_aggregator.cpp:389 (x / 1 == x) */                                  \
        }                                                                                                      \
    } while (0)
    float Lp[VPL]; // previous path element's costs; padding lanes (d >= D) hold the sentinel
    float minLp;
    {
        const size_t pix = so_pixel<VERT>(g, e0);
        float c[VPL];
        if constexpr (AGG) {
#pragma unroll
            for (int t_ = 0; t_ <= 2 * SO_AGG_LA; t_++) { // ring position t_ = raw element e0 + 4 - t_ (clamped: the arms never reach outside the row)
                const int er_ = adc_imax(0, adc_imin(g.plen - 1, e0 + SO_AGG_LA - t_));
                vload<VPL>(src + so_pixel<VERT>(g, er_) * Dp + g.d0, agr[t_]);
            }
            SO_AGG_EVAL(e0, c);
        } else
        vload<VPL>(src + pix * Dp + g.d0, c);
        if (!SEG || seg == 0) vstore<VPL>(dst + pix * Dp + g.d0, c); // first pixel: dst = src (scanline_optimizer.cpp:99,208)
        float lmin = ADC_LARGE_FLOAT; // sentinels take part in the first minimum (scanline_optimizer.cpp:107-110)
#pragma unroll
        for (int k = 0; k < VPL; k++) {
            Lp[k] = (g.d0 + k) < D ? c[k] : ADC_LARGE_FLOAT;
            lmin = Lp[k] < lmin ? Lp[k] : lmin;
        }
        minLp = wave_min_f32<DPP>(lmin);
    }
    // d-1 / d+1 neighbours of the lane's first / last disparity in the CURRENT costs (sentinels beyond the range):
    // needed by the next DP step and by the fused winner-takes-all, so they are computed once, right after Lp
    float upN = lane_up<DPP>(Lp[VPL - 1], ADC_LARGE_FLOAT, lane);
    float dnN = lane_down<DPP>(Lp[0], ADC_LARGE_FLOAT, lane);
// Left-view winner of path element I, whose final costs are in Lp / minLp (pads hold Large_Float).  Only the cheap,
// mostly scalar part runs per step: winner index (lowest set bit of a ballot against the path minimum -- lowest d
// wins ties, like the reference's strict '>' scan), its two neighbour costs (readlane), and the minimum; they are
// parked in lane (I & 63) of four accumulator registers.  Every 64 elements (and at the end of the path) SO_WTA_FLUSH
// evaluates the edge rules and the sub-pixel parabola for 64 pixels at once, one pixel per lane.
    int wtaB = 0;
    float wtaC1 = 0.f, wtaC2 = 0.f, wtaM = 0.f;
#define SO_WTA(I)                                                                                          \
    if constexpr (WTA) {                                                                                   \
        int bi_ = 0x7fffffff;                                                                              \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            const unsigned long long m_ = __ballot(Lp[k] == minLp);                                        \
            const int c_ = m_ ? (int)__builtin_ctzll(m_) * VPL + k : 0x7fffffff;                           \
            bi_ = c_ < bi_ ? c_ : bi_;                                                                     \
        }                                                                                                  \
        /* the reference scan only updates on cost < min, starting at Large_Float: nothing below -> best stays 0 */ \
        const int best_ = minLp < ADC_LARGE_FLOAT ? bi_ + dmin : 0;                                        \
        /* its two neighbour costs: every lane already holds the d-1 / d+1 neighbours of its own disparities */ \
        const int idx_ = (best_ - dmin) & (64 * VPL - 1);                                                  \
        const int kw_ = idx_ % VPL, lw_ = idx_ / VPL;                                                      \
        float s1_ = upN, s2_ = VPL == 1 ? dnN : Lp[VPL == 1 ? 0 : 1];                                      \
        _Pragma("unroll") for (int k = 1; k < VPL; k++)                                                    \
        {                                                                                                  \
            s1_ = kw_ == k ? Lp[k - 1] : s1_;                                                              \
            s2_ = kw_ == k ? (k == VPL - 1 ? dnN : Lp[k == VPL - 1 ? k : k + 1]) : s2_;                    \
        }                                                                                                  \
        const int c1_ = __builtin_amdgcn_readlane(__float_as_int(s1_), lw_);                               \
        const int c2_ = __builtin_amdgcn_readlane(__float_as_int(s2_), lw_);                               \
        const int slot_ = (I)&63;                                                                          \
        const bool mine_ = lane == slot_; /* park the four scalars in lane (I & 63) */                       \
        wtaB = mine_ ? best_ : wtaB;                                                                       \
        wtaC1 = mine_ ? __int_as_float(c1_) : wtaC1;                                                       \
        wtaC2 = mine_ ? __int_as_float(c2_) : wtaC2;                                                       \
        wtaM = mine_ ? minLp : wtaM;                                                                       \
        if (slot_ == 63 || (I) == g.plen - 1) {                                                            \
            const int e_ = ((I) & ~63) + lane; /* path element parked in this lane */                      \
            if (e_ <= (I)) {                                                                               \
                float o_;                                                                                  \
                if (wtaB == dmin || wtaB == dmin + D - 1) o_ = ADC_INVALID_FLOAT;                          \
                else if (wtaB - 1 - dmin < 0 || wtaB + 1 - dmin >= D) o_ = (float)wtaB;                    \
                else o_ = adc_subpixel(wtaB, wtaC1, wtaC2, wtaM);                                          \
                disp[so_pixel<VERT>(g, e_)] = o_;                                                          \
            }                                                                                              \
        }                                                                                                  \
    }
    SO_WTA(0);
    if (plen_v <= 1) return;

    int mcur = dir > 0 ? e0 + 1 : g.plen - 2 - e0; // coordinate of path element e0 + 1, advanced by every SO_STEP
    const int dpad = (D + VPL - 1) / VPL * VPL;
// one DP step for path element I with inputs E (a macro keeps every array in registers)
#define SO_STEP(I, E)                                                                                      \
    do {                                                                                                   \
        const int x_ = VERT ? g.path : mcur; /* mcur = coordinate of path element I (running counter) */   \
        int off_[VPL];                                                                                     \
        if (SO_INTERIOR && adc_so_interior(x_, W, dmin, dpad)) /* wave-uniform: 2 instead of ~10 VALU per class */ \
            adc_so_class_offsets_interior<VPL>((E).rb, (E).c1, tso, off_);                                 \
        else                                                                                               \
            adc_so_class_offsets<VPL>((E).rb, (E).c1, x_ - cl_last, W, tso, W >= 3 && x_ - dmin >= 1, off_);   \
        const float up_ = upN; /* L(q, d0-1), sentinel at d=-1 */                                          \
        const float dn_ = dnN; /* L(q, d0+VPL), sentinel at d=D */                                         \
        float out_[VPL];                                                                                   \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            const float2 pp_ = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(so_tab) + off_[k]); \
            const float P1_ = pp_.x, P2_ = pp_.y;                                                          \
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
        upN = lane_up<DPP>(Lp[VPL - 1], ADC_LARGE_FLOAT, lane);                                            \
        dnN = lane_down<DPP>(Lp[0], ADC_LARGE_FLOAT, lane);                                                \
        SO_WTA(I);                                                                                         \
        mcur += dir;                                                                                       \
    } while (0)
// output store of the prefetch path: running pointer (path elements are visited in order)
#define SO_STORE(I, OUT)              \
    do {                              \
        vstore<VPL>(dpn, OUT);        \
        dpn += dstep;                 \
    } while (0)
// The same step for a chunk of the path that is interior as a whole (see the steady-state loop below): interior class
// rule without its per-step test, no padding lanes (D == 64 * VPL), the four-way minimum as v_min3 + v_min without the
// canonicalising copies the compiler puts in front of fminf (costs are finite, never NaN: exact selections either way),
// and the lane-edge sentinels of the d-1 / d+1 neighbours simply stay where they are (a DPP move leaves lanes without a
// source untouched: lane 0 of upN / lane 63 of dnN have held Large_Float since the first element).
#define SO_STEP_F(I, E)                                                                                    \
    do {                                                                                                   \
        int off_[VPL];                                                                                     \
        adc_so_class_offsets_interior<VPL>((E).rb, (E).c1, tso, off_);                                     \
        float out_[VPL];                                                                                   \
        _Pragma("unroll") for (int k = 0; k < VPL; k++)                                                    \
        {                                                                                                  \
            const float2 pp_ = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(so_tab) + off_[k]); \
            const float lm1_ = k == 0 ? upN : Lp[k == 0 ? 0 : k - 1];                                      \
            const float lp1_ = k == VPL - 1 ? dnN : Lp[k == VPL - 1 ? k : k + 1];                          \
            const float l2_ = lm1_ + pp_.x;                                                                \
            const float l3_ = lp1_ + pp_.x;                                                                \
            const float l4_ = minLp + pp_.y;                                                               \
            float t_, mm_;                                                                                 \
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t_) : "v"(Lp[k]), "v"(l2_), "v"(l3_));                  \
            asm("v_min_f32 %0, %1, %2" : "=v"(mm_) : "v"(t_), "v"(l4_));                                   \
            out_[k] = ((E).c[k] + mm_) * 0.5f;                                                             \
        }                                                                                                  \
        SO_STORE(I, out_);                                                                                 \
        float omin_ = out_[0];                                                                             \
        _Pragma("unroll") for (int k = 0; k < VPL; k++) Lp[k] = out_[k];                                   \
        if constexpr (VPL == 2) asm("v_min_f32 %0, %1, %2" : "=v"(omin_) : "v"(out_[0]), "v"(out_[1]));    \
        minLp = wave_min_f32<DPP>(omin_);                                                                  \
        if constexpr (DPP) {                                                                               \
            upN = dpp_mov<0x138>(upN, Lp[VPL - 1]);                                                        \
            dnN = dpp_mov<0x130>(dnN, Lp[0]);                                                              \
        } else {                                                                                           \
            upN = lane_up<DPP>(Lp[VPL - 1], ADC_LARGE_FLOAT, lane);                                        \
            dnN = lane_down<DPP>(Lp[0], ADC_LARGE_FLOAT, lane);                                            \
        }                                                                                                  \
        SO_WTA(I);                                                                                         \
    } while (0)

    if constexpr (VPL <= 2) {
        // Prefetch ring with asm-issued loads and hand-counted vmcnt (same technique and the same reasons as
        // k_agg_march, see k_aggregate.hip): SO_PF = 16 path elements = 4 groups in flight.  Per group of 4 steps the
        // vector-memory operations are, in program order:
        //   [data 0][rmap 0][d1 word][store 0] [data 1][rmap 1][store 1] [data 2][rmap 2][store 2] [data 3][rmap 3][store 3]
        // (13 per group, 52 of the 63 operations the memory counter can track), which gives exact wait counts --
        // operations younger than everything the step takes (its data, its rmap bytes and, on the first step of a
        // group, the group's d1 word):
        //   steady state    : 49 on the first step of a group, 50 on the others
        //   first iteration : 33 + 4*group on the first step, 34 + 4*group + step otherwise (the prologue issued no stores)
        //   final chunk     : 33 - 9*group - 2*step, a lower bound that ignores the stores (always safe)
        // Loads past the end of the path are clamped to its last element / group (harmless duplicates).
        typedef typename VecT<VPL>::type vec_t;
        constexpr int PF = 16, NG = PF / 4;
        // PIN: slots in the reserved registers (see SO_V0; the empty statement makes the kernel's register count include
        // them).  !PIN: slots in compiler-allocated registers -- sound for the single-form loop of these kernels, which is
        // what the generated-code check (tools/check_async_loads.py) verifies on every build.
        if constexpr (PIN) asm volatile("" ::: SO_CLOBBERS);
        vec_t pfc[PIN ? 1 : PF];
        uint32_t pfr[PIN ? 1 : PF];
        uint32_t pfw[PIN ? 1 : NG];
        const long long pstep = (long long)(VERT ? W : 1) * dir; // pixels per path step
        const long long fstep = pstep * Dp;
        const size_t px1 = so_pixel<VERT>(g, e0 + 1);
        // (AGG: the data stream runs SO_AGG_LA elements ahead of the recurrence, clamped at the end of the ROW)
        const float* spn = src + (AGG ? so_pixel<VERT>(g, adc_imin(e0 + 1 + SO_AGG_LA, g.plen - 1)) : px1) * Dp + g.d0; // next element to prefetch
        const uint32_t* cwn = c1p + (e0 >> 2);     // next d1 word to prefetch (e0 is a multiple of 4)
        float* dpn = dst + px1 * Dp + g.d0;        // next element to store
        long long dstep = fstep;                   // ... and how far the store pointer moves per step
        if constexpr (SEG) {
            // warm-up steps store to the segment's seam slot (every step keeps its store: the wait counts below assume it);
            // the last of them leaves the state at element efirst - 1 there
            if (seg > 0) {
                dpn = seam + ((size_t)g.path * (nseg - 1) + (seg - 1)) * Dp + g.d0;
                dstep = 0;
            }
        }
        const int last = e1 - 1;
        int ii = e0 + 1, gi = e0 >> 2; // element / group the prefetch stands on
        int mpf = mcur; // coordinate of element ii
#define SO_ISSUE_D(U)                                                                                          \
    do {                                                                                                       \
        const int ro_ = so_rmap_offset_m<VPL, VERT>(g, mpf, cl_last);                                          \
        if constexpr (PIN) {                                                                                   \
            if constexpr (VPL == 1) {                                                                          \
                asm volatile("global_load_dword v[%1], %0, off" ADC_VOL_NT_STR ::"v"(spn), "n"(SO_RC(U)) : "memory");         \
                asm volatile("global_load_ubyte v[%2], %0, %1" ::"v"(ro_), "s"(rmap), "n"(SO_RR(U)) : "memory"); \
            } else {                                                                                           \
                asm volatile("global_load_dwordx2 v[%1:%2], %0, off" ADC_VOL_NT_STR ::"v"(spn), "n"(SO_RC(U)), "n"(SO_RC(U) + 1) : "memory"); \
                asm volatile("global_load_ushort v[%2], %0, %1" ::"v"(ro_), "s"(rmap), "n"(SO_RR(U)) : "memory"); \
            }                                                                                                  \
        } else {                                                                                               \
            if constexpr (VPL == 1) {                                                                          \
                asm volatile("global_load_dword %0, %1, off" ADC_VOL_NT_STR : "=v"(pfc[PIN ? 0 : (U)]) : "v"(spn) : "memory"); \
                asm volatile("global_load_ubyte %0, %1, %2" : "=v"(pfr[PIN ? 0 : (U)]) : "v"(ro_), "s"(rmap) : "memory"); \
            } else {                                                                                           \
                asm volatile("global_load_dwordx2 %0, %1, off" ADC_VOL_NT_STR : "=v"(pfc[PIN ? 0 : (U)]) : "v"(spn) : "memory"); \
                asm volatile("global_load_ushort %0, %1, %2" : "=v"(pfr[PIN ? 0 : (U)]) : "v"(ro_), "s"(rmap) : "memory"); \
            }                                                                                                  \
        }                                                                                                      \
        spn += (ii < last && (!AGG || ii + SO_AGG_LA < g.plen - 1)) ? fstep : 0;                               \
        mpf += ii < last ? dir : 0;                                                                            \
        ii += ii < last ? 1 : 0;                                                                               \
    } while (0)
#define SO_ISSUE_C(G)                                                                                          \
    do {                                                                                                       \
        if constexpr (PIN) asm volatile("global_load_dword v[%1], %0, off" ::"v"(cwn), "n"(SO_RW(G)) : "memory"); \
        else asm volatile("global_load_dword %0, %1, off" : "=v"(pfw[PIN ? 0 : (G)]) : "v"(cwn) : "memory");    \
        cwn += gi + 1 < ngr ? 1 : 0;                                                                           \
        gi++;                                                                                                  \
    } while (0)
// clamp-free forms for a chunk whose prefetches all stay inside the path (running rmap offset `rof`)
#define SO_ISSUE_DF(U)                                                                                         \
    do {                                                                                                       \
        if constexpr (VPL == 1) {                                                                              \
            asm volatile("global_load_dword v[%1], %0, off" ADC_VOL_NT_STR ::"v"(spn), "n"(SO_RC(U)) : "memory");               \
            asm volatile("global_load_ubyte v[%2], %0, %1" ::"v"(rof), "s"(rmap), "n"(SO_RR(U)) : "memory");     \
        } else {                                                                                               \
            asm volatile("global_load_dwordx2 v[%1:%2], %0, off" ADC_VOL_NT_STR ::"v"(spn), "n"(SO_RC(U)), "n"(SO_RC(U) + 1) : "memory"); \
            asm volatile("global_load_ushort v[%2], %0, %1" ::"v"(rof), "s"(rmap), "n"(SO_RR(U)) : "memory");    \
        }                                                                                                      \
        spn += fstep;                                                                                          \
        rof += rstep;                                                                                          \
    } while (0)
#define SO_ISSUE_CF(G)                                                                                         \
    do {                                                                                                       \
        asm volatile("global_load_dword v[%1], %0, off" ::"v"(cwn), "n"(SO_RW(G)) : "memory");                   \
        cwn += 1;                                                                                              \
        gi++;                                                                                                  \
    } while (0)
// take element U (and, on the first step of a group, the group's d1 word) after waiting for <= WAITN younger ops
#define SO_TAKE(U, WAITN, E)                                                                                   \
    do {                                                                                                       \
        vec_t tc_;                                                                                             \
        uint32_t tr_;                                                                                          \
        if constexpr (PIN) {                                                                                   \
            if constexpr (((U)&3) == 0) {                                                                          \
                if constexpr (VPL == 1)                                                                            \
                    asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v[%4]\n\tv_mov_b32 %1, v[%5]\n\tv_mov_b32 %2, v[%6]" \
                                 : "=&v"(tc_), "=&v"(tr_), "=&v"(cw)                                               \
                                 : "n"(WAITN), "n"(SO_RC(U)), "n"(SO_RR(U)), "n"(SO_RW((U) >> 2)) : "memory");     \
                else                                                                                               \
                    asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b64 %0, v[%4:%5]\n\tv_mov_b32 %1, v[%6]\n\tv_mov_b32 %2, v[%7]" \
                                 : "=&v"(tc_), "=&v"(tr_), "=&v"(cw)                                               \
                                 : "n"(WAITN), "n"(SO_RC(U)), "n"(SO_RC(U) + 1), "n"(SO_RR(U)), "n"(SO_RW((U) >> 2)) : "memory"); \
            } else {                                                                                               \
                if constexpr (VPL == 1)                                                                            \
                    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, v[%3]\n\tv_mov_b32 %1, v[%4]"                 \
                                 : "=&v"(tc_), "=&v"(tr_) : "n"(WAITN), "n"(SO_RC(U)), "n"(SO_RR(U)) : "memory");  \
                else                                                                                               \
                    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b64 %0, v[%3:%4]\n\tv_mov_b32 %1, v[%5]"            \
                                 : "=&v"(tc_), "=&v"(tr_) : "n"(WAITN), "n"(SO_RC(U)), "n"(SO_RC(U) + 1), "n"(SO_RR(U)) : "memory"); \
            }                                                                                                      \
        } else {                                                                                               \
            if constexpr (((U)&3) == 0) {                                                                          \
                if constexpr (VPL == 1)                                                                            \
                    asm volatile("s_waitcnt vmcnt(%6)\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" \
                                 : "=&v"(tc_), "=&v"(tr_), "=&v"(cw)                                               \
                                 : "v"(pfc[PIN ? 0 : (U)]), "v"(pfr[PIN ? 0 : (U)]), "v"(pfw[PIN ? 0 : ((U) >> 2)]), "n"(WAITN) : "memory"); \
                else                                                                                               \
                    asm volatile("s_waitcnt vmcnt(%6)\n\tv_mov_b64 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" \
                                 : "=&v"(tc_), "=&v"(tr_), "=&v"(cw)                                               \
                                 : "v"(pfc[PIN ? 0 : (U)]), "v"(pfr[PIN ? 0 : (U)]), "v"(pfw[PIN ? 0 : ((U) >> 2)]), "n"(WAITN) : "memory"); \
            } else {                                                                                               \
                if constexpr (VPL == 1)                                                                            \
                    asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"                     \
                                 : "=&v"(tc_), "=&v"(tr_) : "v"(pfc[PIN ? 0 : (U)]), "v"(pfr[PIN ? 0 : (U)]), "n"(WAITN) : "memory"); \
                else                                                                                               \
                    asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b64 %0, %2\n\tv_mov_b32 %1, %3"                     \
                                 : "=&v"(tc_), "=&v"(tr_) : "v"(pfc[PIN ? 0 : (U)]), "v"(pfr[PIN ? 0 : (U)]), "n"(WAITN) : "memory"); \
            }                                                                                                      \
        }                                                                                                      \
        if constexpr (VPL == 1) (E).c[0] = tc_;                                                                \
        else { (E).c[0] = tc_.x; (E).c[VPL - 1] = tc_.y; }                                                     \
        (E).rb[0] = tr_;                                                                                       \
        (E).c1 = (int)((cw >> (8 * ((U)&3))) & 0xffu);                                                         \
        if constexpr (AGG) { /* (E).c holds the RAW element e + SO_AGG_LA: push it, aggregate element e = e0 + i + U */ \
            SO_AGG_PUSH((E).c);                                                                                \
            SO_AGG_EVAL(e0 + i + (U), (E).c);                                                                  \
        }                                                                                                      \
    } while (0)
// one full step of the pipelined loop: take, re-issue the slot for the element PF ahead, DP step
#define SO_PIPE(U, WAITN)                                                                                      \
    do {                                                                                                       \
        SoElem<VPL> cur_;                                                                                      \
        SO_TAKE(U, WAITN, cur_);                                                                               \
        SO_ISSUE_D(U);                                                                                         \
        if constexpr (((U)&3) == 0) SO_ISSUE_C((U) >> 2);                                                      \
        SO_STEP(i + (U), cur_);                                                                                \
    } while (0)
#define SO_PIPE_F(U, WAITN)                                                                                    \
    do {                                                                                                       \
        SoElem<VPL> cur_;                                                                                      \
        SO_TAKE(U, WAITN, cur_);                                                                               \
        SO_ISSUE_DF(U);                                                                                        \
        if constexpr (((U)&3) == 0) SO_ISSUE_CF((U) >> 2);                                                     \
        SO_STEP_F(i + (U), cur_);                                                                              \
    } while (0)
#define SO_FAST4(G) SO_PIPE_F(4 * (G), 49); SO_PIPE_F(4 * (G) + 1, 50); SO_PIPE_F(4 * (G) + 2, 50); SO_PIPE_F(4 * (G) + 3, 50)
#define SO_FIRST4(G)                                                                                           \
    SO_PIPE(4 * (G), 33 + 4 * (G)); SO_PIPE(4 * (G) + 1, 35 + 4 * (G)); SO_PIPE(4 * (G) + 2, 36 + 4 * (G));     \
    SO_PIPE(4 * (G) + 3, 37 + 4 * (G))
#define SO_STEADY4(G) SO_PIPE(4 * (G), 49); SO_PIPE(4 * (G) + 1, 50); SO_PIPE(4 * (G) + 2, 50); SO_PIPE(4 * (G) + 3, 50)
#define SO_LAST(U)                                                                                             \
    if (i + (U) < plen_v) {                                                                                    \
        SoElem<VPL> cur_;                                                                                      \
        SO_TAKE(U, 33 - 9 * ((U) >> 2) - 2 * ((U)&3), cur_);                                                   \
        SO_STEP(i + (U), cur_);                                                                                \
    }
#define SO_LAST4(G) SO_LAST(4 * (G)) SO_LAST(4 * (G) + 1) SO_LAST(4 * (G) + 2) SO_LAST(4 * (G) + 3)
        static_assert(PF == 16, "wait counts and the unrolled groups are written for 16 elements / 4 groups in flight");
        uint32_t cw = 0; // d1 word of the current group
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // start the manual bookkeeping from an empty queue
#pragma unroll
        for (int G = 0; G < NG; G++) { // prologue, same order as the steady state without the stores
            SO_ISSUE_D(4 * G);
            SO_ISSUE_C(G);
            SO_ISSUE_D(4 * G + 1);
            SO_ISSUE_D(4 * G + 2);
            SO_ISSUE_D(4 * G + 3);
        }
        int i = 1; // (relative to e0)
// a segment's warm-up ends between two chunks: from element efirst on the outputs go to the volume
#define SO_SEG_SWITCH()                                                                                        \
    do {                                                                                                       \
        if constexpr (SEG) {                                                                                   \
            if (seg > 0 && i == warm + 1) {                                                                    \
                dpn = dst + so_pixel<VERT>(g, efirst) * Dp + g.d0;                                             \
                dstep = fstep;                                                                                 \
            }                                                                                                  \
        }                                                                                                      \
    } while (0)
        if (i + PF <= plen_v) {
            SO_FIRST4(0); SO_FIRST4(1); SO_FIRST4(2); SO_FIRST4(3);
            i += PF;
            // Steady state.  A chunk of PF steps takes the short form when it is interior as a whole: no padding lanes,
            // every element of the chunk AND every element it prefetches (the PF behind it, their d1 groups included)
            // inside the path and inside the columns where the class rule is the interior one and the rmap gather needs
            // no clamp (x - (dmin + Dp - 1) >= 1, x - dmin < W - 1).  Same vector-memory operations in the same order as
            // the general form, so the two can alternate under the same wait counts; at 1080p 109 of the 120 chunks of a
            // row qualify.  The general form keeps its per-step tests and clamps for the rest.
            const int rstep = (VERT ? W : 1) * dir;
            for (; i + PF <= plen_v; i += PF) {
                SO_SEG_SWITCH();
                if constexpr (PIN) { // (two forms of the loop body need the pinned slots, see SO_V0)
                    // (adc_device_fn.h restates this predicate for the CPU checks, with e0 = 0: adc_so_chunk_interior)
                    bool fast = SO_INTERIOR && SO_FAST && allow_fast && D == Dp && W >= 3 && i + 2 * PF + 4 <= plen_v;
                    if (fast) {
                        const int ea = e0 + i, eb = e0 + i + 2 * PF - 1; // path elements the chunk steps on or prefetches
                        const int ma = dir > 0 ? ea : g.plen - 1 - ea, mb = dir > 0 ? eb : g.plen - 1 - eb;
                        const int xlo = VERT ? g.path : (ma < mb ? ma : mb), xhi = VERT ? g.path : (ma < mb ? mb : ma);
                        fast = xlo >= dmin + Dp && xhi - dmin < W - 1;
                    }
                    if (fast) {
                        int rof = so_rmap_offset_m<VPL, VERT>(g, mpf, cl_last); // exact: no clamp is active in this chunk
                        SO_FAST4(0); SO_FAST4(1); SO_FAST4(2); SO_FAST4(3);
                        mcur += PF * dir;
                        mpf += PF * dir;
                        ii += PF;
                    } else {
                        SO_STEADY4(0); SO_STEADY4(1); SO_STEADY4(2); SO_STEADY4(3);
                    }
                } else {
                    SO_STEADY4(0); SO_STEADY4(1); SO_STEADY4(2); SO_STEADY4(3);
                }
            }
        }
        // final chunk: fewer than PF elements left, all of them in flight
        SO_SEG_SWITCH();
        SO_LAST4(0) SO_LAST4(1) SO_LAST4(2) SO_LAST4(3)
#undef SO_SEG_SWITCH
        // Slots past the end of the path were loaded (clamped) but never taken.  PIN: harmless, their registers are
        // reserved.  !PIN: keep their destination registers alive until those loads have landed, or the compiler may reuse
        // them for the values of the steps above and a late-landing load overwrites them.
        if constexpr (PIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(pfc[PIN ? 0 : 0]), "v"(pfc[PIN ? 0 : 1]), "v"(pfc[PIN ? 0 : 2]), "v"(pfc[PIN ? 0 : 3]),
                         "v"(pfc[PIN ? 0 : 4]), "v"(pfc[PIN ? 0 : 5]), "v"(pfc[PIN ? 0 : 6]), "v"(pfc[PIN ? 0 : 7]), "v"(pfc[PIN ? 0 : 8]),
                         "v"(pfc[PIN ? 0 : 9]), "v"(pfc[PIN ? 0 : 10]), "v"(pfc[PIN ? 0 : 11]), "v"(pfc[PIN ? 0 : 12]),
                         "v"(pfc[PIN ? 0 : 13]), "v"(pfc[PIN ? 0 : 14]), "v"(pfc[PIN ? 0 : 15]) : "memory");
            asm volatile("" ::"v"(pfr[PIN ? 0 : 0]), "v"(pfr[PIN ? 0 : 1]), "v"(pfr[PIN ? 0 : 2]), "v"(pfr[PIN ? 0 : 3]), "v"(pfr[PIN ? 0 : 4]),
                         "v"(pfr[PIN ? 0 : 5]), "v"(pfr[PIN ? 0 : 6]), "v"(pfr[PIN ? 0 : 7]), "v"(pfr[PIN ? 0 : 8]), "v"(pfr[PIN ? 0 : 9]),
                         "v"(pfr[PIN ? 0 : 10]), "v"(pfr[PIN ? 0 : 11]), "v"(pfr[PIN ? 0 : 12]), "v"(pfr[PIN ? 0 : 13]),
                         "v"(pfr[PIN ? 0 : 14]), "v"(pfr[PIN ? 0 : 15]), "v"(pfw[PIN ? 0 : 0]), "v"(pfw[PIN ? 0 : 1]),
                         "v"(pfw[PIN ? 0 : 2]), "v"(pfw[PIN ? 0 : 3]) : "memory");
        }
#undef SO_ISSUE_D
#undef SO_ISSUE_C
#undef SO_ISSUE_DF
#undef SO_ISSUE_CF
#undef SO_PIPE_F
#undef SO_FAST4
#undef SO_TAKE
#undef SO_PIPE
#undef SO_FIRST4
#undef SO_STEADY4
#undef SO_LAST
#undef SO_LAST4
#undef SO_STORE
#define SO_STORE(I, OUT) vstore<VPL>(dst + so_pixel<VERT>(g, (I)) * Dp + g.d0, OUT)
    } else {
        // VPL == 4 (disparity range > 128): compiler-scheduled prefetch ring
        auto so_load = [&](int i) __attribute__((always_inline)) {
            SoElem<VPL> e;
            vload<VPL>(src + so_pixel<VERT>(g, i) * Dp + g.d0, e.c);
            so_rmap_load_words<VPL>(rmap, so_rmap_offset<VPL, VERT>(g, i, cl_last, 0), e.rb);
            e.c1 = (int)((c1p[(i - 1) >> 2] >> (8 * ((i - 1) & 3))) & 0xffu);
            return e;
        };
        constexpr int SO_PFV = VPL <= 4 ? SO_PF : (VPL == 8 ? 8 : (VPL == 16 ? 4 : 2)); // register budget: SO_PFV * (VPL + 2) per lane
        SoElem<VPL> pre[SO_PFV];
#pragma unroll
        for (int u = 0; u < SO_PFV; u++) pre[u] = so_load(adc_imin(1 + u, g.plen - 1));
        int i = 1;
        for (; i + SO_PFV <= g.plen; i += SO_PFV) {
#pragma unroll
            for (int u = 0; u < SO_PFV; u++) {
                const SoElem<VPL> cur = pre[u];
                pre[u] = so_load(adc_imin(i + u + SO_PFV, g.plen - 1));
                __builtin_amdgcn_sched_barrier(0);
                SO_STEP(i + u, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < SO_PFV; u++) {
            if (i + u < g.plen) SO_STEP(i + u, pre[u]);
        }
    }
#undef SO_STEP
#undef SO_STEP_F
#undef SO_STORE
#undef SO_WTA
#undef SO_AGG_PUSH
#undef SO_AGG_EVAL
}

// VPL <= 2 (disparity ranges up to 128), asm prefetch.  Two kernels: k_scanline keeps the slots in compiler-allocated
// registers and has ONE form of the steady state (the form every pass with >= 1024 paths runs: such a pass is bound by its
// memory streams; same-box A/B at 1080p: scanline stage 1.934 ms against 1.963 with the pinned kernel, 1.984 / 2.029 on the
// structured pair); k_scanline_pin keeps them in the reserved registers and adds the short form of whole interior chunks,
// for passes that are lone-wave chains (< 1024 paths; KITTI-size row passes 287 -> 194 us).
template <int VPL, bool VERT, bool DPP, bool WTA>
__global__ __launch_bounds__(256) void k_scanline(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ c1w, int ngr,
    const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso, int dir, float P1a, float P1b, float P1c,
    float P2a, float P2b, float P2c, float* __restrict__ disp)
{
    static_assert(VPL <= 2, "wider lanes use k_scanline_wide");
    so_body<VPL, VERT, DPP, WTA, false>(src, dst, c1w, ngr, rmap, W, H, D, dmin, tso, dir, P1a, P1b, P1c, P2a, P2b, P2c, disp, 0);
}
template <int VPL, bool VERT, bool DPP, bool WTA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(SO_V0))) void k_scanline_pin(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ c1w, int ngr,
    const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso, int dir, float P1a, float P1b, float P1c,
    float P2a, float P2b, float P2c, float* __restrict__ disp, int allow_fast)
{
    static_assert(VPL <= 2, "wider lanes use k_scanline_wide");
    so_body<VPL, VERT, DPP, WTA, true>(src, dst, c1w, ngr, rmap, W, H, D, dmin, tso, dir, P1a, P1b, P1c, P2a, P2b, P2c, disp, allow_fast);
}
// Row passes cut into verified segments (round 4): the same two families, wave = (segment, row)
template <int VPL, bool DPP>
__global__ __launch_bounds__(256) void k_scanline_seg(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ c1w, int ngr,
    const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso, int dir, float P1a, float P1b, float P1c,
    float P2a, float P2b, float P2c, int nseg, int warm, float* __restrict__ seam)
{
    static_assert(VPL <= 2, "asm-prefetch kernels only");
    so_body<VPL, false, DPP, false, false, true>(src, dst, c1w, ngr, rmap, W, H, D, dmin, tso, dir, P1a, P1b, P1c, P2a, P2b, P2c, nullptr, 0,
                                                 nseg, warm, seam);
}
// ... and the L->R pass that also computes the last aggregation pass on its input (AGG, see so_body)
template <int VPL, bool DPP>
__global__ __launch_bounds__(256) void k_scanline_seg_agg(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ c1w, int ngr,
    const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso, int dir, float P1a, float P1b, float P1c,
    float P2a, float P2b, float P2c, int nseg, int warm, float* __restrict__ seam, const uint32_t* __restrict__ aggrec)
{
    so_body<VPL, false, DPP, false, false, true, true>(src, dst, c1w, ngr, rmap, W, H, D, dmin, tso, dir, P1a, P1b, P1c, P2a, P2b, P2c, nullptr, 0,
                                                       nseg, warm, seam, aggrec);
}
template <int VPL, bool DPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(SO_V0))) void k_scanline_pin_seg(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ c1w, int ngr,
    const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso, int dir, float P1a, float P1b, float P1c,
    float P2a, float P2b, float P2c, int allow_fast, int nseg, int warm, float* __restrict__ seam)
{
    static_assert(VPL <= 2, "asm-prefetch kernels only");
    so_body<VPL, false, DPP, false, true, true>(src, dst, c1w, ngr, rmap, W, H, D, dmin, tso, dir, P1a, P1b, P1c, P2a, P2b, P2c, nullptr,
                                                allow_fast, nseg, warm, seam);
}
// Seam check of the two row passes of a Match (both outputs are still intact behind the second pass: L->R wrote dst0, R->L
// wrote dst1): one wave per (pass, row, seam); the segment's state at its last warm-up element must equal what its predecessor
// stored there, bit for bit, in every disparity of the range.  A difference is counted; adc_wait then redoes the Match
// without segments.
__global__ __launch_bounds__(256) void k_so_seam_check(const float* __restrict__ dst0, const float* __restrict__ dst1,
                                                       const float* __restrict__ seam, int W, int H, int Dp, int D, int nseg, int warm,
                                                       int* __restrict__ fails)
{
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), per_pass = H * (nseg - 1);
    if (gw >= 2 * per_pass) return;
    const int pass = gw / per_pass, r = gw - pass * per_pass, path = r / (nseg - 1), s = 1 + r % (nseg - 1);
    const int e = adc_so_seg_start(W, nseg, warm, s) - 1; // last warm-up element of segment s = last output of segment s - 1
    const int x = pass == 0 ? e : W - 1 - e;
    const uint32_t* a = reinterpret_cast<const uint32_t*>(pass == 0 ? dst0 : dst1) + ((size_t)path * W + x) * Dp;
    const uint32_t* b = reinterpret_cast<const uint32_t*>(seam) + ((size_t)pass * per_pass + r) * Dp;
    bool bad = false;
    for (int d = lane; d < D; d += 64) bad |= a[d] != b[d];
    if (__ballot(bad) != 0ull && lane == 0) atomicAdd(fails, 1);
}
// VPL >= 4: compiler-scheduled prefetch, no reserved registers
template <int VPL, bool VERT, bool DPP, bool WTA>
__global__ __launch_bounds__(256) void k_scanline_wide(
    const float* __restrict__ src, float* __restrict__ dst, const uint32_t* __restrict__ c1w, int ngr,
    const uint8_t* __restrict__ rmap, int W, int H, int D, int dmin, int tso, int dir, float P1a, float P1b, float P1c,
    float P2a, float P2b, float P2c, float* __restrict__ disp)
{
    static_assert(VPL >= 4, "VPL <= 2 uses k_scanline / k_scanline_pin");
    so_body<VPL, VERT, DPP, WTA, false>(src, dst, c1w, ngr, rmap, W, H, D, dmin, tso, dir, P1a, P1b, P1c, P2a, P2b, P2c, disp, 0);
}

static bool so_use_dpp()
{
    static const bool v = [] { const char* e = getenv("ADC_SO_DPP"); return e ? atoi(e) != 0 : true; }();
    return v;
}

// Number of verified segments a row pass of this handle is cut into (1 = whole rows).  Automatic choice: enough chains for
// two to three waves on every SIMD of the chip (1080 rows -> 3 x 1080, 375 rows -> 5 x 375); ADC_SO_SEG = 0 / 1 switches the segments
// off, N >= 2 forces N; ADC_SO_WARM = warm-up steps (multiple of 16; the tests use short ones to provoke seam failures).
int adc_so_segments(const adc_handle* h, int* warm_out)
{
    static const int seg_env = [] { const char* e = getenv("ADC_SO_SEG"); return e ? atoi(e) : -1; }();
    static const int warm_env = [] { const char* e = getenv("ADC_SO_WARM"); return e ? atoi(e) : ADC_SO_WARM; }();
    const AdcParams& p = h->p;
    if (warm_out) *warm_out = warm_env;
    if (p.VPL > 2 || h->so_seg_off || !h->so_seam || seg_env == 0 || seg_env == 1) return 1;
    int n = seg_env >= 2 ? seg_env : (2048 + p.H / 2) / p.H;
    // (round 6, same box, interleaved -- profiles/r6_ab_scanline_segments.txt: at 1080 rows three segments per row instead of two, 3240
    // chains for 1024 SIMDs: scanline stage of the noise pair 1.990 / 1.981 -> 1.937 ms, structured 1.888 / 1.879 -> 1.873; four: 1.965)
    if (seg_env < 2 && n == 2 && 3 * p.H <= 4096) n = 3;
    if (n > ADC_SO_MAX_SEG) n = ADC_SO_MAX_SEG;
    while (n >= 2 && !adc_so_seg_ok(p.W, n, warm_env)) n--;
    return n >= 2 ? n : 1;
}
size_t adc_so_seam_bytes(int W, int H, int Dp) { (void)W; return (size_t)2 * H * (ADC_SO_MAX_SEG - 1) * Dp * sizeof(float); }

// Passes that are lone-wave chains (fewer paths than the 1024 SIMDs of the chip) run k_scanline_pin with the short form
// of whole interior chunks (KITTI-size row passes: 375 paths, scanline stage 0.873 -> 0.675 ms, same-box A/B); where
// every SIMD has a wave or two the pass is bound by its memory streams and runs k_scanline.  ADC_SO_FAST=0 / 1 forces
// k_scanline / k_scanline_pin everywhere.
static bool so_uses_pin(int npaths, int nseg)
{
    static const int so_fast_env = [] { const char* e = getenv("ADC_SO_FAST"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    return so_fast_env >= 0 ? so_fast_env != 0 : npaths * nseg < 1024;
}
// Can the L->R row pass of the next scanline run take over the last aggregation pass (k_scanline_seg_agg)?  Needs the segment
// form of the compiler-allocated family with two disparities per lane.  (adc_launch_aggregate asks before it drops that pass.)
bool adc_so_can_fuse_agg(const adc_handle* h)
{
    static const bool env = [] { const char* e = getenv("ADC_FUSE_AGG_SO"); return e ? atoi(e) != 0 : true; }();
    if (!env || h->p.VPL != 2 || !so_use_dpp() || h->paper) return false;
    const int nseg = adc_so_segments(h, nullptr);
    return nseg > 1 && !so_uses_pin(h->p.H, nseg);
}

template <int VPL>
static hipError_t launch_so(adc_handle* h, const float* src, float* dst, bool vert, int dir, float* disp = nullptr, int nseg = 1, int warm = 0,
                            bool agg = false)
{
    const AdcParams& p = h->p;
    const int npaths = vert ? p.W : p.H;
    // Waves per workgroup (ADC_SO_WPB_ROW / ADC_SO_WPB_COL: 1, 2 or 4).  A 1080p row pass has 1080 paths for 1024 SIMDs:
    // with 4-wave workgroups 14 CUs end up with eight waves and set the kernel time (two waves per SIMD take ~660
    // instead of ~520 cycles per step; 1024 rows: 0.42 ms, 1028 rows: 0.61 ms); single-wave workgroups spread the 56
    // extra waves over 56 CUs (scanline stage 2.15 -> 1.97 ms).  Column passes (1920 paths) measured best with 4.
    static const int wpb_row = [] { const char* e = getenv("ADC_SO_WPB_ROW"); const int v = e ? atoi(e) : 1; return v == 1 || v == 2 ? v : 4; }();
    static const int wpb_col = [] { const char* e = getenv("ADC_SO_WPB_COL"); const int v = e ? atoi(e) : 4; return v == 1 || v == 2 ? v : 4; }();
    const int wpb = vert ? wpb_col : wpb_row;
    const unsigned blocks = (unsigned)((npaths * nseg + wpb - 1) / wpb);
    const int pass = (vert ? 2 : 0) + (dir > 0 ? 0 : 1);
    const SoC1Layout L = so_c1_layout(p.W, p.H);
    const uint32_t* c1w = reinterpret_cast<const uint32_t*>(h->so_cls) + L.off[pass];
    const uint8_t* rmap = vert ? h->cdiff_rv : h->cdiff_rh;
#define SO_ARGS src, dst, c1w, L.ngr[pass], rmap, p.W, p.H, p.D, p.dmin, p.opt.so_tso, dir, h->so_P1[0], h->so_P1[1], h->so_P1[2], \
                h->so_P2[0], h->so_P2[1], h->so_P2[2], disp
#define SO_LAUNCH(VERT_, DPP_, WTA_)                                                                                   \
    do {                                                                                                               \
        if constexpr (VPL <= 2) {                                                                                      \
            if (so_pin)                                                                                                \
                hipLaunchKernelGGL((k_scanline_pin<VPL, VERT_, DPP_, WTA_>), dim3(blocks), dim3(64 * wpb), 0, h->heavy, SO_ARGS, 1); \
            else                                                                                                       \
                hipLaunchKernelGGL((k_scanline<VPL, VERT_, DPP_, WTA_>), dim3(blocks), dim3(64 * wpb), 0, h->heavy, SO_ARGS); \
        } else                                                                                                         \
            hipLaunchKernelGGL((k_scanline_wide<VPL, VERT_, DPP_, WTA_>), dim3(blocks), dim3(64 * wpb), 0, h->heavy, SO_ARGS); \
    } while (0)
    const bool dpp = so_use_dpp();
    const bool so_pin = so_uses_pin(npaths, nseg);
    if constexpr (VPL <= 2) {
        if (nseg > 1 && !vert && dpp) { // verified segments: wave = (segment, row); seam slots of this pass
            float* seam = h->so_seam + (size_t)pass * p.H * (nseg - 1) * p.Dp;
            if constexpr (VPL == 2) {
                if (agg && !so_pin && dir > 0) { // the pass also computes the last aggregation pass on its input (so_body, AGG)
                    hipLaunchKernelGGL((k_scanline_seg_agg<VPL, true>), dim3(blocks), dim3(64 * wpb), 0, h->heavy, src, dst, c1w, L.ngr[pass], rmap, p.W,
                                       p.H, p.D, p.dmin, p.opt.so_tso, dir, h->so_P1[0], h->so_P1[1], h->so_P1[2], h->so_P2[0], h->so_P2[1],
                                       h->so_P2[2], nseg, warm, seam, h->rec_h);
                    return hipGetLastError();
                }
            }
            if (agg) return hipErrorInvalidValue; // (adc_so_can_fuse_agg promised this form)
            if (so_pin)
                hipLaunchKernelGGL((k_scanline_pin_seg<VPL, true>), dim3(blocks), dim3(64 * wpb), 0, h->heavy, src, dst, c1w, L.ngr[pass], rmap,
                                   p.W, p.H, p.D, p.dmin, p.opt.so_tso, dir, h->so_P1[0], h->so_P1[1], h->so_P1[2], h->so_P2[0], h->so_P2[1],
                                   h->so_P2[2], 1, nseg, warm, seam);
            else
                hipLaunchKernelGGL((k_scanline_seg<VPL, true>), dim3(blocks), dim3(64 * wpb), 0, h->heavy, src, dst, c1w, L.ngr[pass], rmap, p.W,
                                   p.H, p.D, p.dmin, p.opt.so_tso, dir, h->so_P1[0], h->so_P1[1], h->so_P1[2], h->so_P2[0], h->so_P2[1],
                                   h->so_P2[2], nseg, warm, seam);
            return hipGetLastError();
        }
    }
    if (vert) {
        if (dpp && disp) SO_LAUNCH(true, true, true);
        else if (dpp) SO_LAUNCH(true, true, false);
        else SO_LAUNCH(true, false, false);
    } else {
        if (dpp) SO_LAUNCH(false, true, false);
        else SO_LAUNCH(false, false, false);
    }
#undef SO_LAUNCH
#undef SO_ARGS
    return hipGetLastError();
}

template <int VPL>
static hipError_t run_so(adc_handle* h, int passes)
{
    // scanline_optimizer.cpp:54-60 (cost_aggr_ == vol_a, cost_init_ == vol_b)
    hipError_t e = adc_launch_so_classes(h, h->heavy); // the path-ordered d1 words (tiny)
    if ((h->paper & ADC_PAPER_SO_SUM) && h->vol_c && passes == 4) {
        // opt-in paper mode (k_paper.hip): every path from the SAME aggregated volume, averaged -- vol_a -> vol_b per path,
        // accumulated in vol_c, which then becomes vol_a
        for (int r = 0; r < 4 && e == hipSuccess; r++) {
            e = launch_so<VPL>(h, h->vol_a, h->vol_b, r >= 2, (r & 1) ? -1 : +1);
            if (e == hipSuccess) e = adc_paper_accumulate(h, h->vol_c, h->vol_b, r == 0, r == 3);
        }
        { float* t = h->vol_a; h->vol_a = h->vol_c; h->vol_c = t; }
        h->wta_left_done = 0;
        if (e == hipSuccess && h->fuse_wta) { // (the left view cannot ride on a pass here: computed from the averaged volume)
            e = adc_launch_wta_left(h);
            h->wta_left_done = 1;
        }
        return e;
    }
    // Row passes as verified segments (needs both passes: the seam check runs once, behind the second one, while both outputs
    // are intact).  h->armmax[2] counts the seams that failed; adc_wait looks at it.
    int warm = 0;
    const int nseg = (passes >= 2 && so_use_dpp()) ? adc_so_segments(h, &warm) : 1;
    h->so_nseg_last = nseg;
    if (e == hipSuccess && nseg > 1) e = hipMemsetAsync(h->armmax + 2, 0, sizeof(int), h->heavy);
    bool agg = h->so_agg_fused != 0; // vol_a holds the volume BEFORE the last aggregation pass (adc_launch_aggregate dropped it)
    h->so_agg_fused = 0;
    // (round-5 advisor finding) adc_launch_aggregate asked adc_so_can_fuse_agg at ITS time; should the plan here differ after all
    // (whole rows, the pinned family, a single pass), the dropped pass runs as a launch of its own instead of failing the Match
    if (agg && e == hipSuccess && (nseg <= 1 || passes < 1 || VPL != 2 || so_uses_pin(h->p.H, nseg))) {
        e = adc_launch_aggregate_tail(h);
        agg = false;
    }
    if (e == hipSuccess) e = launch_so<VPL>(h, h->vol_a, h->vol_b, false, +1, nullptr, nseg, warm, agg);
    if (e == hipSuccess && passes >= 2) e = launch_so<VPL>(h, h->vol_b, h->vol_a, false, -1, nullptr, nseg, warm);
    if (e == hipSuccess && nseg > 1) {
        const int waves = 2 * h->p.H * (nseg - 1);
        hipLaunchKernelGGL(k_so_seam_check, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, h->heavy, h->vol_b, h->vol_a, h->so_seam, h->p.W,
                           h->p.H, h->p.Dp, h->p.D, nseg, warm, h->armmax + 2);
        e = hipGetLastError();
    }
    if (e == hipSuccess && passes >= 3) e = launch_so<VPL>(h, h->vol_a, h->vol_b, true, +1);
    if (e == hipSuccess && passes >= 4) {
        // production pipeline: the left-view winner-takes-all rides on the last pass (capi.hip sets fuse_wta)
        float* disp = (h->fuse_wta && so_use_dpp()) ? h->disp_l : nullptr;
        e = launch_so<VPL>(h, h->vol_b, h->vol_a, true, -1, disp);
        h->wta_left_done = disp ? 1 : 0;
    }
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
    if (h->p.VPL == 4) return run_so<4>(h, passes);
    if (h->p.VPL == 8) return run_so<8>(h, passes);
    if (h->p.VPL == 16) return run_so<16>(h, passes);
    return run_so<32>(h, passes);
}
