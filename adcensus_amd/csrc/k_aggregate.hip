// k_aggregate.hip -- K4 cross-based cost aggregation (the roofline kernel of the pipeline).
//
// Replaces CrossAggregator::{Aggregate, AggregateInArms} (cross_aggregator.cpp:89-118,327-394):
// 4 iterations (H-first, V-first, H-first, V-first); per iteration two 1-D passes over the whole
// volume, each output = sequential f32 sum (from 0.0f, in the order t = -arm ... +arm, which is
// observable in the WTA result -- SURVEY.md A.4) of the inputs inside the pixel's own arm; the second
// pass of an iteration divides by the support count.
//
// MI355X mapping ("marching ring"):
//   * layout [y][x][Dp], lanes = disparities: a wave owns 64 consecutive floats (256 B) of every
//     pixel of ONE image line (a row for the H pass, a column for the V pass) and marches along it.
//     Arms and support counts are wave-uniform (scalar loads, no divergence).
//   * every input element is read from HBM exactly once per pass (+ the 34-pixel halo at segment
//     ends): the last 2L+1 line entries live in an LDS ring (256 B x (2L+1) per wave, each lane
//     only ever reads what it wrote itself, so no barriers), the next AGG_PF entries are in flight
//     in registers (software prefetch) -- that is what keeps ~10 MB of loads outstanding chip-wide.
//   * algorithmic bytes per pass: read V + write V (+4 B/pixel arms, +2 B/pixel counts), V = 4*W*H*Dp.
#include "adc_internal.h"
#include "adc_device_fn.h"

#define AGG_PF 16  // loads in flight per wave (x 256 B)

// ------------------------------------------------------------------------------- direct (fallback)
// One thread per volume element reading its arm span straight from global memory.  Used when the
// LDS ring does not fit (cross_L1 > 79) and as an A/B cross-check of the marching kernel in tests.
template <bool VERT, bool DIVIDE>
__global__ __launch_bounds__(256) void k_agg_direct(const float* __restrict__ src, float* __restrict__ dst,
                                                    const uchar4* __restrict__ arms, const uint16_t* __restrict__ sup,
                                                    int W, int H, int Dp)
{
    const size_t total = (size_t)W * H * Dp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i / Dp;
        const int d = (int)(i % Dp);
        const uchar4 a = arms[pix];
        const int lo = VERT ? a.z : a.x, hi = VERT ? a.w : a.y;
        const size_t stride = VERT ? (size_t)W * Dp : (size_t)Dp;
        float acc = 0.0f;
        for (int t = -lo; t <= hi; t++) acc += src[(size_t)((long long)pix * Dp + (long long)t * (long long)stride) + d];
        if (DIVIDE) acc = acc / (float)sup[pix];
        dst[i] = acc;
    }
}

// ---------------------------------------------------------------------------------- marching ring
template <bool VERT, bool DIVIDE>
__global__ __launch_bounds__(256) void k_agg_march(const float* __restrict__ src, float* __restrict__ dst,
                                                   const uchar4* __restrict__ arms, const uint16_t* __restrict__ sup,
                                                   int W, int H, int Dp, int L, int seg_len, int nseg)
{
    extern __shared__ __attribute__((aligned(16))) float ring_all[];
    const int R = 2 * L + 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float* ring = ring_all + (size_t)wave * R * 64 + lane; // entry s at ring[s*64]

    const int chunks = Dp >> 6;                 // 64-float chunks per pixel
    const int N = VERT ? H : W;                 // length of a line
    const int nlines = (VERT ? W : H) * chunks; // independent lines
    const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
    if (gw >= nlines * nseg) return;
    const int seg = gw / nlines;
    const int line = gw - seg * nlines;
    const int fixed = line / chunks; // x (V pass) or y (H pass)
    const int chunk = line - fixed * chunks;

    const int m0 = seg * seg_len;
    const int m1 = adc_imin(N, m0 + seg_len);
    if (m0 >= m1) return;
    const int lo = adc_imax(0, m0 - L);
    const int hi = adc_imin(N, m1 + L);

    // element (m) of this line: pixel index and float offset
    const long long pix_step = VERT ? (long long)W : 1LL;
    const long long pix0 = VERT ? (long long)fixed : (long long)fixed * W;
    const long long fstep = pix_step * Dp;
    const float* sp = src + pix0 * Dp + chunk * 64 + lane;
    float* dp = dst + pix0 * Dp + chunk * 64 + lane;

    // Software prefetch: the next AGG_PF line entries (and the arms / counts of the outputs they will
    // trigger) are in flight in registers.  All prefetch loads are UNCONDITIONAL (index clamped into
    // the segment) -- a load under a branch makes the compiler drain vmcnt at the join.
    auto pix_of = [&](int m) __attribute__((always_inline)) -> long long { return pix0 + (long long)m * pix_step; };
    float pf[AGG_PF];
    uint32_t pa[AGG_PF]; // arms (uchar4 as u32) of output m = entry - L
    uint32_t ps[AGG_PF]; // support count of that output (DIVIDE passes)
    const uint32_t* arms32 = reinterpret_cast<const uint32_t*>(arms);
#pragma unroll
    for (int u = 0; u < AGG_PF; u++) {
        const int e = adc_imin(lo + u, hi - 1);
        const int mo = adc_imin(adc_imax(e - L, m0), m1 - 1);
        pf[u] = sp[(long long)e * fstep];
        pa[u] = arms32[pix_of(mo)];
        ps[u] = DIVIDE ? (uint32_t)sup[pix_of(mo)] : 1u;
    }

    int slot_w = 0;                 // ring slot of the next entry to be written
    int slot_m = (m0 - lo);         // ring slot of entry m0 (m0 - lo <= L < R)

    auto emit = [&](int m, uint32_t a32, uint32_t cnt) __attribute__((always_inline)) {
        const int a_lo = __builtin_amdgcn_readfirstlane((int)(VERT ? (a32 >> 16) & 255u : a32 & 255u));
        const int a_hi = __builtin_amdgcn_readfirstlane((int)(VERT ? (a32 >> 24) & 255u : (a32 >> 8) & 255u));
        int n = a_lo + a_hi + 1;
        int idx = slot_m - a_lo;
        if (idx < 0) idx += R;
        float acc = 0.0f;
        while (n > 0) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int s = idx + k;
                if (s >= R) s -= R;
                v[k] = ring[s * 64];
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < n) acc += v[k]; // sequential order t = -arm .. +arm
            idx += 8;
            if (idx >= R) idx -= R;
            n -= 8;
        }
        if (DIVIDE) acc = acc / (float)cnt; // cross_aggregator.cpp:389
        dp[(long long)m * fstep] = acc;
        slot_m++;
        if (slot_m == R) slot_m = 0;
    };

    int j = lo;
    for (; j + AGG_PF <= hi; j += AGG_PF) {
#pragma unroll
        for (int u = 0; u < AGG_PF; u++) {
            const int jj = j + u;
            const float v = pf[u];
            const uint32_t a32 = pa[u], cnt = ps[u];
            {
                const int e = adc_imin(jj + AGG_PF, hi - 1);
                const int mo = adc_imin(adc_imax(e - L, m0), m1 - 1);
                pf[u] = sp[(long long)e * fstep];
                pa[u] = arms32[pix_of(mo)];
                if (DIVIDE) ps[u] = (uint32_t)sup[pix_of(mo)];
            }
            ring[slot_w * 64] = v;
            slot_w++;
            if (slot_w == R) slot_w = 0;
            const int m = jj - L;
            if (m >= m0 && m < m1) emit(m, a32, cnt);
        }
    }
    // remainder (< AGG_PF entries): their values / arms are already in the registers
#pragma unroll
    for (int u = 0; u < AGG_PF; u++) {
        const int jj = j + u;
        if (jj < hi) {
            ring[slot_w * 64] = pf[u];
            slot_w++;
            if (slot_w == R) slot_w = 0;
            const int m = jj - L;
            if (m >= m0 && m < m1) emit(m, pa[u], ps[u]);
        }
    }
    // outputs whose +L look-ahead ends beyond the last loaded entry (image end): arms loaded directly
    for (int m = adc_imax(m0, hi - L); m < m1; m++)
        emit(m, arms32[pix_of(m)], DIVIDE ? (uint32_t)sup[pix_of(m)] : 1u);
}

static int env_int(const char* name, int dflt)
{
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

template <bool VERT, bool DIVIDE>
static hipError_t launch_pass(adc_handle* h, const float* src, float* dst, const uint16_t* sup, bool direct)
{
    const AdcParams& p = h->p;
    const int L = adc_imax(0, adc_imin(p.opt.cross_L1, 255));
    const size_t lds = (size_t)4 * (2 * L + 1) * 64 * sizeof(float);
    if (direct || lds > 150 * 1024) {
        hipLaunchKernelGGL((k_agg_direct<VERT, DIVIDE>), dim3(256 * 16), dim3(256), 0, h->stream, src, dst,
                           reinterpret_cast<const uchar4*>(h->arms), sup, p.W, p.H, p.Dp);
        return hipGetLastError();
    }
    const int N = VERT ? p.H : p.W;
    int nseg = env_int(VERT ? "ADC_AGG_VSEG" : "ADC_AGG_HSEG", VERT ? 2 : 4);
    if (nseg < 1) nseg = 1;
    int seg_len = (N + nseg - 1) / nseg;
    if (seg_len < 1) seg_len = 1;
    nseg = (N + seg_len - 1) / seg_len;
    const long long nlines = (long long)(VERT ? p.W : p.H) * (p.Dp / 64);
    const long long waves = nlines * nseg;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL((k_agg_march<VERT, DIVIDE>), dim3(blocks), dim3(256), lds, h->stream, src, dst,
                       reinterpret_cast<const uchar4*>(h->arms), sup, p.W, p.H, p.Dp, L, seg_len, nseg);
    return hipGetLastError();
}

// vol_a -> (H,V | V,H alternating) -> vol_a.  Every iteration is two launches: a -> b -> a.
hipError_t adc_launch_aggregate(adc_handle* h, int iterations)
{
    static const bool direct = env_int("ADC_AGG_DIRECT", 0) != 0;
    static bool attr_set = false;
    if (!attr_set) {
        // allow > 64 KiB dynamic LDS for the ring
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_agg_march<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipError_t e = hipSuccess;
    bool horizontal_first = true; // cross_aggregator.cpp:100
    int launch = 0;
    for (int k = 0; k < iterations && e == hipSuccess; k++) {
        if (h->profiling && launch < 8) hipEventRecord(h->ev_agg[launch], h->stream);
        if (horizontal_first) {
            e = launch_pass<false, false>(h, h->vol_a, h->vol_b, nullptr, direct);
            launch++;
            if (h->profiling && launch < 8) hipEventRecord(h->ev_agg[launch], h->stream);
            if (e == hipSuccess) e = launch_pass<true, true>(h, h->vol_b, h->vol_a, h->sup_h, direct);
        } else {
            e = launch_pass<true, false>(h, h->vol_a, h->vol_b, nullptr, direct);
            launch++;
            if (h->profiling && launch < 8) hipEventRecord(h->ev_agg[launch], h->stream);
            if (e == hipSuccess) e = launch_pass<false, true>(h, h->vol_b, h->vol_a, h->sup_v, direct);
        }
        launch++;
        horizontal_first = !horizontal_first;
    }
    if (h->profiling) hipEventRecord(h->ev_agg[launch < 8 ? launch : 8], h->stream);
    h->agg_launches = launch < 8 ? launch : 8;
    return e;
}
