// capi.hip -- the C ABI of include/adcensus_c_api.h: object lifetime, the Match pipeline
// (ADCensusStereo.cpp:69-132 stage order) and the test-only per-stage debug surface.
// Host code only; kernels live in the k_*.hip files.
#include "adc_internal.h"
#include "adc_device_fn.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <new>
#include <mutex>
#include <atomic>

static thread_local std::string g_last_error;

#ifdef ADC_FAULT_INJECTION // (test builds only, adc_internal.h)
static std::atomic<long> g_fi_calls{0};
static std::atomic<long> g_fi_fail_at{[] { const char* e = getenv("ADC_TEST_FAIL_AT"); return e ? atol(e) : 0L; }()};
extern "C" int adc_test_fault_now(void) { const long n = ++g_fi_calls; return n == g_fi_fail_at.load(); }
extern "C" void adc_test_fail_at(long n) { g_fi_calls = 0; g_fi_fail_at = n; } // the n-th call from now on fails (0: none)
extern "C" long adc_test_hip_calls(void) { return g_fi_calls.load(); }
#endif

// Host ranges the CALLER has page-locked for the library (adc_host_register): images / maps inside such a range are
// transferred by DMA straight from / to the caller's memory, without the pinned staging copies.  Opt-in on purpose: a
// registration must not outlive the allocation (a freed and re-used address range would DMA into stale pages), which only
// the caller can guarantee.
struct HostRange { const char* base; size_t bytes; };
static std::mutex g_host_mu;
static std::vector<HostRange> g_host_ranges;
static bool host_registered(const void* p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_host_mu);
    const char* c = static_cast<const char*>(p);
    for (const HostRange& r : g_host_ranges)
        if (c >= r.base && c + bytes <= r.base + r.bytes) return true;
    return false;
}

static void set_error(const char* what, hipError_t e)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
}
#define HIP_OK(call)                          \
    do {                                      \
        hipError_t e__ = ADC_HIP(call);       \
        if (e__ != hipSuccess) {              \
            set_error(#call, e__);            \
            return e__;                       \
        }                                     \
    } while (0)

// The guide's yardstick (MI355X_MICROARCH.md: "float4 copy"): a grid-stride copy kernel, 16 bytes per lane, four loads in
// flight per lane; best time over a few grid sizes and plain / non-temporal accesses (tools/ubench/copy_ceiling.hip sweeps more
// shapes on the same box: profiles/r4_ubench_copy_ceiling.txt).  bench.py reports the better of this and hipMemcpyAsync.
typedef float adc_vf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy_yardstick(const adc_vf4* __restrict__ src, adc_vf4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        adc_vf4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
extern "C" {

void adc_option_default(adc_option* o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->min_disparity = 0;  o->max_disparity = 64; // adcensus_types.h:67-74
    o->lambda_ad = 10;     o->lambda_census = 30;
    o->cross_L1 = 34;      o->cross_L2 = 17;
    o->cross_t1 = 20;      o->cross_t2 = 6;
    o->so_p1 = 1.0f;       o->so_p2 = 3.0f;      o->so_tso = 15;
    o->irv_ts = 20;        o->irv_th = 0.4f;
    o->lrcheck_thres = 1.0f;
    o->do_lr_check = 1;    o->do_filling = 1;    o->do_discontinuity_adjustment = 0;
}

int adc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
// (a library whose asm-prefetch / register-ring kernels were NOT re-verified on the generated code -- build() with
// ADC_BUILD_SKIP_CODEGEN_CHECK=1 -- says so: the hand-counted s_waitcnt values are only proven for a checked build)
#ifdef ADC_CODEGEN_UNCHECKED
const char* adc_version(void) { return "adcensus-mi355x 0.2 (gfx950) [generated-code checks SKIPPED]"; }
#else
const char* adc_version(void) { return "adcensus-mi355x 0.2 (gfx950)"; }
#endif
const char* adc_last_error(void) { return g_last_error.c_str(); }

static hipError_t alloc_all(adc_handle* h)
{
    const AdcParams& p = h->p;
    const size_t P = (size_t)p.W * p.H;
    const size_t VB = P * p.Dp * sizeof(float);
    HIP_OK(hipMalloc(&h->img_l_own, P * 3));
    HIP_OK(hipMalloc(&h->img_r_own, P * 3));
    h->img_l = h->img_l_own;
    h->img_r = h->img_r_own;
    HIP_OK(hipMalloc(&h->gray_l, P));
    HIP_OK(hipMalloc(&h->bgrx_l, P * 4));
    // sized from the PADDED range: the fused-cost pass also marches over padding chunks / lanes (d >= D), whose
    // right-image column x - d lies up to dmin + Dp - 1 columns to the left of the row
    h->rrec_padl = (p.dmin + p.Dp - 1 > 0 ? p.dmin + p.Dp - 1 : 0) + 1;
    h->rrec_pitch = h->rrec_padl + p.W + (p.dmin < 0 ? -p.dmin : 0) + 1;
    HIP_OK(hipMalloc(&h->cost_rrec, (size_t)p.H * h->rrec_pitch * 16));
    HIP_OK(hipMalloc(&h->cost_lrec, P * 16));
    h->med_hpitch = ((p.W + 2 * p.H + 64 + 15) / 16) * 16;
    HIP_OK(hipMalloc(&h->med_hand, adc_median_hand_rows(p.H) * h->med_hpitch * sizeof(float))); // (bands + chains of <= 4 speculative copies per band, per column segment)
    HIP_OK(hipMalloc(&h->med_sink, adc_median_hand_rows(p.H) * 64 * 16 + (size_t)((p.H + 63) / 64) * ADC_MEDB_MAX_SEG * 64 * 8 + 64)); // store sinks, then the segments' seam columns
    HIP_OK(hipMalloc(&h->gray_r, P));
    HIP_OK(hipMalloc(&h->census_l, P * 8));
    HIP_OK(hipMalloc(&h->census_r, P * 8));
    HIP_OK(hipMalloc(&h->arms, P * 4));
    HIP_OK(hipMalloc(&h->sup_h, P * 2));
    HIP_OK(hipMalloc(&h->sup_v, P * 2));
    HIP_OK(hipMalloc(&h->armmax, 4 * sizeof(int)));
    HIP_OK(hipMalloc(&h->rec_h, P * 4));
    HIP_OK(hipMalloc(&h->rec_v, P * 4));
    HIP_OK(hipMalloc(&h->rec2_h, P * 8));
    HIP_OK(hipMalloc(&h->rec2_v, P * 8));
    HIP_OK(hipMalloc(&h->agg_sink, 1024 * 64 * sizeof(float)));
    // + slack: the scanline kernels fetch up to VPL (<= 32) bytes starting at a column <= W-1 (+1 on R->L passes)
    HIP_OK(hipMalloc(&h->cdiff_lh, P + 64));
    HIP_OK(hipMalloc(&h->cdiff_lv, P + 64));
    HIP_OK(hipMalloc(&h->cdiff_rh, P + 64));
    HIP_OK(hipMalloc(&h->cdiff_rv, P + 64));
    HIP_OK(hipMalloc(&h->so_cls, adc_so_cls_bytes(p.W, p.H)));
    if (p.VPL <= 2) HIP_OK(hipMalloc(&h->so_seam, adc_so_seam_bytes(p.W, p.H, p.Dp))); // (verified segments of the scanline row passes)
    HIP_OK(hipMalloc(&h->vol_a, VB));
    HIP_OK(hipMalloc(&h->vol_b, VB));
    HIP_OK(hipMalloc(&h->lut_ad, 768 * sizeof(float)));
    HIP_OK(hipMalloc(&h->lut_census, 64 * sizeof(float)));
    HIP_OK(hipMalloc(&h->ray_sincos, 32 * sizeof(double)));
    // (+ 1 KiB: the banded median prefetches a few columns past the last row's end without clamping, k_refine.hip)
    HIP_OK(hipMalloc(&h->disp_l, P * 4 + 1024));
    HIP_OK(hipMalloc(&h->disp_r, P * 4));
    HIP_OK(hipMalloc(&h->disp_tmp, P * 4 + 1024));
    HIP_OK(hipMemset(h->disp_l + P, 0, 1024));
    HIP_OK(hipMemset(h->disp_tmp + P, 0, 1024));
    HIP_OK(hipMalloc(&h->label, P));
    HIP_OK(hipMalloc(&h->elig, P + 64)); // (LR check: invalid mask, 1 byte per pixel; then the voting chain's bitmap of listed pixels, whole 64-bit words)
    HIP_OK(hipMalloc(&h->irv_bbox, P * 4));
    h->irv_grid = adc_irv_grid(P);
    h->irv_xcd_mode = adc_irv_probe_xcd_mode(h->device);
    HIP_OK(hipMalloc(&h->vote_list, adc_irv_list_entries(p.W, p.H, p.D, h->irv_grid) * 16)); // int4 per entry, one segment per workgroup (irv_plan.h)
    HIP_OK(hipMemset(h->vote_list, 0xFF, adc_irv_list_entries(p.W, p.H, p.D, h->irv_grid) * 16)); // every slot = IRV_LIST_END
    HIP_OK(hipMalloc(&h->vote_evals_arr, adc_irv_waves(h->irv_grid) * sizeof(int32_t)));
    HIP_OK(hipMemset(h->vote_evals_arr, 0, adc_irv_waves(h->irv_grid) * sizeof(int32_t)));
    HIP_OK(hipMalloc(&h->interp_list, 2 * P * 4)); // both target lists of the interpolation, P entries each
    HIP_OK(hipMalloc(&h->interp_counters, 64 * sizeof(int32_t)));
    {
        const int da = p.dmax < 0 ? -p.dmax : p.dmax, di = p.dmin < 0 ? -p.dmin : p.dmin;
        h->itp_ms = da > di ? da : di; // multistep_refiner.cpp:236
        h->itp_pitch = adc_itp_code_pitch(p.W, h->itp_ms);
        HIP_OK(hipMalloc(&h->itp_cells, adc_itp_cell_bytes(p.W, p.H, h->itp_ms)));
        // (everything outside the image never changes: the per-Match kernel only rewrites the image's own columns and rows)
        HIP_OK(hipMemset(h->itp_cells, ADC_ITP_OUTSIDE, adc_itp_cell_bytes(p.W, p.H, h->itp_ms)));
    }
    h->st16_pitch = (p.W + 7) & ~7;
    HIP_OK(hipMalloc(&h->st16, ((size_t)h->st16_pitch * p.H + 64) * sizeof(uint16_t))); // (an uncached allocation -- visible across XCDs inside a kernel -- measured equal)
    HIP_OK(hipMemset(h->st16, 0xFF, ((size_t)h->st16_pitch * p.H + 64) * sizeof(uint16_t))); // padding columns: invalid bin
    HIP_OK(hipMalloc(&h->disp_vote, P * 4));
    HIP_OK(hipMalloc(&h->vote_counters, 512 * sizeof(int32_t)));
    // voting chain budget of the FIRST Match of a handle (later ones adapt: kernels used + 40 % + 2): a natural 1080p image needs
    // ~50-75 kernels (round 5: all passes iterate at once; ~350 before); kernels past the end of the chain are no-ops of ~4 us, an
    // exhausted budget costs a synchronous continuation
    h->irv_budget = 256;
    // change-tile map of the voting rounds: one BYTE per 8x8 tile, rows padded to a multiple of 4 (+16: a 16-byte load
    // may start at the last dword of a row)
    h->chg_pitch = (((p.W + 7) / 8 + 3) & ~3) + 16;
    const size_t tiles = (size_t)h->chg_pitch * ((p.H + 7) / 8) + 64;
    HIP_OK(hipMalloc(&h->chg_a, 2 * tiles)); // two planes (round parity)
    HIP_OK(hipMalloc(&h->irv_cold, 64));
    HIP_OK(hipMalloc(&h->irv_px, adc_irv_px_words(p.W, p.H) * sizeof(uint32_t)));
    HIP_OK(hipMemset(h->irv_px, 0, adc_irv_px_words(p.W, p.H) * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&h->edge, P));
    HIP_OK(hipHostMalloc(&h->pin_in, P * 6, hipHostMallocDefault));
    HIP_OK(hipHostMalloc(&h->pin_out, P * 4, hipHostMallocDefault));
    HIP_OK(hipHostMalloc(&h->pin_flags, 64 * sizeof(int32_t), hipHostMallocDefault));
    memset(h->pin_flags, 0, 64 * sizeof(int32_t)); // [0] median error, [4..7] armmax + violation flag, [16..23] voting state, [32..63] staging of the voting chain's cold block
    HIP_OK(hipMemset(h->label, 0, P));
    HIP_OK(hipMemset(h->chg_a, 0, 2 * tiles));
    HIP_OK(hipMemset(h->vol_a, 0, VB));
    HIP_OK(hipMemset(h->vol_b, 0, VB));
    return hipSuccess;
}

// Host-built tables: evaluated with the host's libm so the GPU result is
// bit-identical to what the CPU reference computes with the same libm.
static hipError_t upload_tables(adc_handle* h)
{
    const adc_option& o = h->p.opt;
    float A[768], C[64];
    memset(A, 0, sizeof(A));
    for (int k = 0; k <= 765; k++) {
        const float cost_ad = (float)k / 3.0f;                     // cost_computor.cpp:110
        const float ea = expf(-cost_ad / (float)o.lambda_ad);      // :117
        A[k] = (1.0f - ea) + 1.0f;
    }
    for (int hm = 0; hm < 64; hm++) C[hm] = expf(-(float)hm / (float)o.lambda_census);
    HIP_OK(hipMemcpy(h->lut_ad, A, sizeof(A), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(h->lut_census, C, sizeof(C), hipMemcpyHostToDevice));
    // 16 ray angles: ang = 0.0 (double); ang += pi/16 with float pi, float divide (multistep_refiner.cpp:234,254-268)
    double sc[32];
    const float pi = 3.1415926f;
    double ang = 0.0;
    for (int s = 0; s < 16; s++) {
        sc[2 * s] = sin(ang);
        sc[2 * s + 1] = cos(ang);
        ang += pi / 16;
    }
    HIP_OK(hipMemcpy(h->ray_sincos, sc, sizeof(sc), hipMemcpyHostToDevice));
    {   // integer ray offsets: lround(y + m*sin) == y + lround(m*sin) for every integer 0 <= y < 2^20 as long as m*sin is
        // not within 1e-9 of a .5 tie (the addition's rounding error is < 2^-32); one unsafe entry disables the table
        const int da = o.max_disparity < 0 ? -o.max_disparity : o.max_disparity, di = o.min_disparity < 0 ? -o.min_disparity : o.min_disparity;
        const int ms = da > di ? da : di; // multistep_refiner.cpp:236
        h->ray_tab = nullptr;
        h->ray_lin = nullptr;
        h->ray_tab_rows = 0;
        if (ms >= 1 && ms <= 30000 && h->p.W < (1 << 20) && h->p.H < (1 << 20)) {
            // packed offsets [ms][16], then the linear offsets into the padded code map [ms + ADC_ITP_LPAD][16] (adc_device_fn.h)
            std::vector<int32_t> tab((size_t)ms * 16 + (size_t)(ms + ADC_ITP_LPAD) * 16, 0);
            int32_t* lin = tab.data() + (size_t)ms * 16;
            bool safe = true;
            for (int m = 1; m < ms && safe; m++)
                for (int s = 0; s < 16; s++) {
                    const double fy = (double)m * sc[2 * s], fx = (double)m * sc[2 * s + 1];
                    const double ry = fabs(fabs(fy - floor(fy)) - 0.5), rx = fabs(fabs(fx - floor(fx)) - 0.5);
                    if (ry < 1e-9 || rx < 1e-9) { safe = false; break; }
                    const long dy = lround(fy), dx = lround(fx);
                    tab[(size_t)m * 16 + s] = (int32_t)(((uint32_t)(dy & 0xffff) << 16) | (uint32_t)(dx & 0xffff));
                    lin[(size_t)m * 16 + s] = (int32_t)(dy * h->itp_pitch + dx);
                    if (dy < 0 || dy >= ms || dx <= -ms || dx >= ms) safe = false; // (the padding of the code map assumes it; always true)
                }
            if (safe) {
                HIP_OK(hipMalloc(&h->ray_tab, tab.size() * sizeof(int32_t)));
                HIP_OK(hipMemcpy(h->ray_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
                h->ray_tab_rows = ms;
                h->ray_lin = h->ray_tab + (size_t)ms * 16;
            }
        }
    }
    // penalty classes (scanline_optimizer.cpp:129-141): f32 divides on the host
    h->so_P1[0] = o.so_p1;      h->so_P2[0] = o.so_p2;
    h->so_P1[1] = o.so_p1 / 4;  h->so_P2[1] = o.so_p2 / 4;
    h->so_P1[2] = o.so_p1 / 10; h->so_P2[2] = o.so_p2 / 10;
    return hipSuccess;
}

// One "bandwidth lane" stream per device, shared by every object on it (never destroyed).  ADC_SHARED_HEAVY=0
// makes every object use its own stream for everything.
static hipStream_t shared_heavy_stream(int dev, hipStream_t own)
{
    static const bool shared = [] { const char* e = getenv("ADC_SHARED_HEAVY"); return e ? atoi(e) != 0 : false; }();
    if (!shared) return own;
    static std::mutex mu;
    static hipStream_t lanes[64] = {nullptr};
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64) return own;
    if (!lanes[dev] && hipStreamCreateWithFlags(&lanes[dev], hipStreamNonBlocking) != hipSuccess) return nullptr;
    return lanes[dev];
}

adc_handle* adc_create(int32_t width, int32_t height, const adc_option* opt, int device)
{
    g_last_error.clear();
    if (!opt) { g_last_error = "adc_create: null option"; return nullptr; }
    if (width <= 0 || height <= 0) { g_last_error = "adc_create: width/height <= 0"; return nullptr; }            // ADCensusStereo.cpp:31-33
    const long long range = (long long)opt->max_disparity - (long long)opt->min_disparity;
    if (range <= 0) { g_last_error = "adc_create: disparity range <= 0"; return nullptr; }                         // :38-40
    if (range > ADC_MAX_DISP_RANGE) { g_last_error = "adc_create: disparity range > ADC_MAX_DISP_RANGE"; return nullptr; }
    if ((long long)width * height > (1LL << 30)) { g_last_error = "adc_create: image too large"; return nullptr; }
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { g_last_error = "adc_create: hipSetDevice failed"; return nullptr; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { g_last_error = "adc_create: no HIP device (the HIP path is mandatory, there is no CPU fallback)"; return nullptr; }

    adc_handle* h = new (std::nothrow) adc_handle();
    if (!h) return nullptr;
    memset(h, 0, sizeof(*h));
    h->device = dev;
    h->p.W = width; h->p.H = height;
    h->p.dmin = opt->min_disparity; h->p.dmax = opt->max_disparity; h->p.D = (int)range;
    h->p.VPL = range <= 64 ? 1 : (range <= 128 ? 2 : (range <= 256 ? 4 : (range <= 512 ? 8 : (range <= 1024 ? 16 : 32))));
    h->p.Dp = 64 * h->p.VPL;
    h->p.opt = *opt;
    bool ok = ADC_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) == hipSuccess;
    h->own_stream = ok;
    if (ok) ok = (h->heavy = shared_heavy_stream(dev, h->stream)) != nullptr;
    if (ok) ok = ADC_HIP(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming)) == hipSuccess;
    if (ok) ok = ADC_HIP(hipEventCreateWithFlags(&h->ev_heavy_done, hipEventDisableTiming)) == hipSuccess;
    for (int i = 0; ok && i <= ADC_STAGE_COUNT; i++) ok = ADC_HIP(hipEventCreate(&h->ev[i])) == hipSuccess;
    for (int i = 0; ok && i < 9; i++) ok = ADC_HIP(hipEventCreate(&h->ev_agg[i])) == hipSuccess;
    if (ok) ok = alloc_all(h) == hipSuccess;
    if (ok) ok = upload_tables(h) == hipSuccess;
    if (!ok) {
        if (g_last_error.empty()) g_last_error = "adc_create: HIP resource creation failed";
        std::string keep = g_last_error;
        adc_destroy(h);
        g_last_error = keep;
        return nullptr;
    }
    return h;
}

void adc_destroy(adc_handle* h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->heavy) hipStreamSynchronize(h->heavy);
    void* bufs[] = {h->img_l_own, h->img_r_own, h->gray_l, h->gray_r, h->census_l, h->census_r, h->arms, h->sup_h, h->sup_v,
                    h->armmax, h->rec_h, h->rec_v, h->rec2_h, h->rec2_v, h->agg_sink, h->so_cls, h->so_seam, h->cdiff_lh, h->cdiff_lv, h->cdiff_rh, h->cdiff_rv, h->vol_a, h->vol_b, h->lut_ad, h->lut_census,
                    h->ray_sincos, h->ray_tab, h->bgrx_l, h->cost_rrec, h->cost_lrec, h->med_hand, h->med_sink, h->disp_l, h->disp_r, h->disp_tmp, h->label, h->elig, h->irv_bbox, h->vote_list, h->vote_evals_arr, h->interp_list, h->interp_counters, h->itp_cells, h->st16, h->disp_vote, h->vote_counters,
                    h->chg_a, h->irv_px, h->irv_cold, h->edge, h->arms_r, h->bgrx_r, h->armmax_r, h->vol_c};
    for (void* b : bufs) if (b) hipFree(b);
    if (h->pin_in) hipHostFree(h->pin_in);
    if (h->pin_out) hipHostFree(h->pin_out);
    if (h->pin_flags) hipHostFree(h->pin_flags);
    for (int i = 0; i <= ADC_STAGE_COUNT; i++) if (h->ev[i]) hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 9; i++) if (h->ev_agg[i]) hipEventDestroy(h->ev_agg[i]);
    if (h->ev_in) hipEventDestroy(h->ev_in);
    if (h->ev_heavy_done) hipEventDestroy(h->ev_heavy_done);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
}

// ------------------------------------------------------------------------------ the pipeline
// the stages behind the region voting (redone by adc_wait when the voting chain had to be continued)
static hipError_t run_refine_tail(adc_handle* h)
{
    const adc_option& o = h->p.opt;
    if (o.do_filling && o.do_lr_check) HIP_OK(adc_launch_interpolation(h));
    if (o.do_discontinuity_adjustment) HIP_OK(adc_launch_discontinuity(h));
    HIP_OK(adc_launch_median(h));
    return hipSuccess;
}
static hipError_t run_refine(adc_handle* h)
{
    // MultiStepRefiner::Refine (multistep_refiner.cpp:60-87); do_filling drives both the voting and the
    // interpolation (ADCensusStereo.cpp:182-183).  Without an LR check both lists are empty.
    const adc_option& o = h->p.opt;
    const size_t P = (size_t)h->p.W * h->p.H;
    if (o.do_lr_check) HIP_OK(adc_launch_lrcheck(h));
    else HIP_OK(hipMemsetAsync(h->label, 0, P, h->stream));
    h->irv_pending = 0;
    if (o.do_filling && o.do_lr_check) HIP_OK(adc_run_region_voting(h));
    h->tail_disp_l = h->disp_l;
    h->tail_disp_tmp = h->disp_tmp;
    return run_refine_tail(h);
}

// The streaming phase (cost .. WTA, ~26 passes over the volume) of different objects on one device is
// made mutually exclusive with a host lock: bandwidth-bound kernels of two pairs only fight for HBM/L2,
// while the latency-bound refinement (one-CU median, voting rounds) of one pair overlaps the streaming
// phase of the next.  Measured on MI355X this is SLOWER than free overlap (56 vs 74 pairs/s with two objects),
// so it is opt-in: ADC_HEAVY_EXCLUSIVE=1.
static std::mutex& heavy_lock(int dev)
{
    static std::mutex locks[64];
    return locks[(dev >= 0 && dev < 64) ? dev : 0];
}
static bool heavy_exclusive()
{
    static const bool v = [] { const char* e = getenv("ADC_HEAVY_EXCLUSIVE"); return e ? atoi(e) != 0 : false; }();
    return v;
}

// from_aggregation: a redo by adc_wait that keeps what the stages in front of the aggregation produced (gray / census, the pixel
// records of the fused cost, arms, support counts, aggregation records: none of them is touched by the later stages) and
// restarts at the first aggregation pass -- possible whenever that pass computes the matching cost itself (no input volume).
static hipError_t run_heavy(adc_handle* h, bool from_aggregation = false)
{
    const bool prof = h->profiling == 1; // (level 2: only the marks around the aggregation launches, k_aggregate.hip)
#define MARK(i, s) do { if (prof) HIP_OK(hipEventRecord(h->ev[i], s)); } while (0)
    if (h->heavy != h->stream) {
        HIP_OK(hipEventRecord(h->ev_in, h->stream));
        HIP_OK(hipStreamWaitEvent(h->heavy, h->ev_in, 0));
    }
    MARK(0, h->heavy);
    static const bool fuse_cost = [] { const char* e = getenv("ADC_FUSE_COST"); return e ? atoi(e) != 0 : true; }();
    const bool fuse_cost_now = fuse_cost && !(h->paper & ADC_PAPER_RIGHT_ARMS); // (paper mode: plain kernels on a stored cost volume)
    h->match_pending = 1; // (adc_wait looks at this Match's arm maxima / speculation flags exactly once)
    if (from_aggregation) {
        MARK(1, h->heavy);
        MARK(2, h->heavy);
        HIP_OK(hipMemsetAsync(h->armmax + 2, 0, 2 * sizeof(int), h->heavy)); // failed seams, "assumed ring too shallow" flag
        h->armmax_valid = 3; // the full ring: valid for every image
    } else {
    HIP_OK(adc_launch_gray_census(h));           // ComputeCost, ADCensusStereo.cpp:84
    // ADC_FUSE_COST (default on): the cost volume is never written -- the first aggregation pass computes each cost
    // in registers from packed pixel records (k_agg_march<.., COSTIN>); otherwise K2 writes it and pass 1 reads it back
    if (fuse_cost_now) HIP_OK(adc_launch_cost_records(h));
    else HIP_OK(adc_launch_cost(h, h->vol_a));
    MARK(1, h->heavy);
    HIP_OK(adc_launch_arms(h));                  // CostAggregation, :92
    MARK(2, h->heavy);
    {   // The maximum arm lengths decide the ring depth of the aggregation kernels and whether same-direction pass pairs
        // can share a launch (k_aggregate.hip).  The host does NOT wait for them: it assumes the maxima of the previous
        // Match of this handle (exact ring depth for that image), the small-ring kernels verify the assumption on the
        // device (armmax[3] is raised and the pass skipped when an arm is longer) and adc_wait redoes the Match with the
        // full ring -- which is valid for every image and is what the first Match of a handle uses.
        // ADC_AGG_HOST_ARMS=1: read the two maxima back instead (one early host synchronisation, the round-1 behaviour).
        static const bool host_arms = [] { const char* e = getenv("ADC_AGG_HOST_ARMS"); return e ? atoi(e) != 0 : false; }();
        if (host_arms && h->pin_flags) {
            HIP_OK(hipMemcpyAsync(h->pin_flags + 4, h->armmax, 2 * sizeof(int), hipMemcpyDeviceToHost, h->heavy));
            HIP_OK(hipStreamSynchronize(h->heavy));
            h->armmax_host[0] = h->pin_flags[4];
            h->armmax_host[1] = h->pin_flags[5];
            h->armmax_valid = 1;
        } else {
            h->armmax_valid = h->arm_known ? 2 : 3;
        }
    }
    HIP_OK(adc_launch_records(h));
    } // (!from_aggregation)
    h->fuse_cost = fuse_cost_now ? 1 : 0;
    h->fuse_agg_so = 1; // (the launcher decides: short-arm plan, arms <= 4, segmented row passes)
    {
        const hipError_t e_ = adc_launch_aggregate(h, 4); // aggregator_.Aggregate(4), :164
        h->fuse_cost = 0;
        h->fuse_agg_so = 0;
        h->armmax_valid = 0;
        HIP_OK(e_);
    }
    MARK(3, h->heavy);
    static const bool fuse_wta = [] { const char* e = getenv("ADC_FUSE_WTA"); return e ? atoi(e) != 0 : true; }();
    h->fuse_wta = fuse_wta ? 1 : 0;
    {
        const hipError_t e_ = adc_launch_scanline(h, 4); // ScanlineOptimize, :100 (+ left-view ComputeDisparity, :108)
        h->fuse_wta = 0;
        HIP_OK(e_);
    }
    MARK(4, h->heavy);
    HIP_OK(adc_launch_wta(h));                   // ComputeDisparity + ComputeDisparityRight, :108-109
    MARK(5, h->heavy);
    // maxima + violation flag of this pair, looked at by adc_wait (they seed the next Match's assumption)
    if (h->pin_flags) HIP_OK(hipMemcpyAsync(h->pin_flags + 4, h->armmax, 4 * sizeof(int), hipMemcpyDeviceToHost, h->heavy));
    HIP_OK(hipEventRecord(h->ev_heavy_done, h->heavy));
    if (h->heavy != h->stream) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_heavy_done, 0));
#undef MARK
    return hipSuccess;
}

static hipError_t run_pipeline(adc_handle* h, bool from_aggregation = false)
{
    if (heavy_exclusive()) {
        // the uploads of this pair (already queued on the object's stream) run before the lock is taken
        std::lock_guard<std::mutex> lk(heavy_lock(h->device));
        HIP_OK(run_heavy(h, from_aggregation));
        HIP_OK(hipEventSynchronize(h->ev_heavy_done)); // hold the lane until the streaming phase has drained
    } else {
        HIP_OK(run_heavy(h, from_aggregation));
    }
    HIP_OK(run_refine(h));                       // MultiStepRefine, :117 (object stream)
    if (h->profiling == 1) HIP_OK(hipEventRecord(h->ev[6], h->stream));
    h->timings_pending = h->profiling != 0;
    return hipSuccess;
}

static void collect_timings(adc_handle* h)
{
    if (!h->timings_pending) return;
    h->timings_pending = false;
    for (int i = 0; i < ADC_STAGE_COUNT; i++) {
        float ms = 0.f;
        if (h->profiling != 1 || hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) != hipSuccess) ms = -1.f; // (level 2: no stage marks were recorded)
        h->stage_ms[i] = ms;
    }
    float tot = 0.f;
    // average duration of a REGULAR aggregation pass (read V + write V); a fused first pass (write-only) is left out
    const int first = h->agg_first_fused ? 1 : 0;
    if (h->agg_dual_last) h->agg_pass_ms = 0.f; // (two plans were enqueued: the marks bracket launches that may have been skipped)
    else if (h->agg_launches > first && hipEventElapsedTime(&tot, h->ev_agg[first], h->ev_agg[h->agg_launches]) == hipSuccess)
        h->agg_pass_ms = tot / (float)(h->agg_launches - first);
    if (h->verbose) { // the reference's stage lines (ADCensusStereo.cpp:88-129)
        printf("computing cost! timing :	%lf s\n", (h->stage_ms[0]) / 1000.0);
        printf("cost aggregating! timing :	%lf s\n", (h->stage_ms[1] + h->stage_ms[2]) / 1000.0);
        printf("scanline optimizing! timing :	%lf s\n", h->stage_ms[3] / 1000.0);
        printf("computing disparities! timing :	%lf s\n", h->stage_ms[4] / 1000.0);
        printf("multistep refining! timing :	%lf s\n", h->stage_ms[5] / 1000.0);
        printf("output disparities! timing :	%lf s\n", 0.0);
    }
}

// the final map -> where the caller wants it (pinned staging for host callers, the caller's device buffer otherwise)
static hipError_t enqueue_output(adc_handle* h)
{
    const size_t P = (size_t)h->p.W * h->p.H;
    if (h->async_dst && h->async_dst_direct == 1) return ADC_HIP(hipMemcpyAsync(h->async_dst, h->disp_l, P * 4, hipMemcpyDeviceToHost, h->stream)); // page-locked by the caller
    if (h->async_dst && h->async_dst_direct == 2) return hipSuccess; // (pageable, ADC_HOST_DIRECT: copied by adc_wait after the stream has drained)
    if (h->async_dst) return ADC_HIP(hipMemcpyAsync(h->pin_out, h->disp_l, P * 4, hipMemcpyDeviceToHost, h->stream));
    if (h->device_dst) return ADC_HIP(hipMemcpyAsync(h->device_dst, h->disp_l, P * 4, hipMemcpyDeviceToDevice, h->stream));
    return hipSuccess;
}

// A HIP call of a Match failed half-way (the reference's contract: Match returns false and the object stays usable,
// ADCensusStereo.cpp:71-76).  Whatever was already enqueued is drained, every per-Match flag of the handle goes back to its idle
// value -- a later Match must not find a half-described predecessor: a pending voting chain, a dropped aggregation pass the
// scanline stage never consumed (round-5 advisor finding), speculation flags of launches that never ran -- and the caller's
// buffers are forgotten.  What the handle has LEARNED from earlier pairs (arm maxima, chain budget) stays: it is verified on the
// device for every pair anyway.
static void abort_match(adc_handle* h)
{
    if (h->heavy) hipStreamSynchronize(h->heavy);
    if (h->stream) hipStreamSynchronize(h->stream);
    (void)hipGetLastError();
    h->match_pending = 0;
    h->irv_pending = 0;
    h->so_agg_fused = 0;
    h->fuse_cost = 0; h->fuse_agg_so = 0; h->fuse_wta = 0;
    h->armmax_valid = 0;
    h->agg_gate = 0;
    h->wta_left_done = 0;
    h->timings_pending = false;
    h->force_median_fallback = 0;
    h->async_dst = nullptr;
    h->device_dst = nullptr;
    if (h->pin_flags) { h->pin_flags[0] = 0; h->pin_flags[4] = h->pin_flags[5] = h->pin_flags[6] = h->pin_flags[7] = 0; }
    if (h->img_l != h->img_l_own || h->img_r != h->img_r_own) { h->img_l = h->img_l_own; h->img_r = h->img_r_own; }
    h->bgrx_valid = 0;
}

int adc_match_device(adc_handle* h, const void* d_left, const void* d_right, void* d_disp)
{
    if (!h || !d_left || !d_right || !d_disp) return 1; // ADCensusStereo.cpp:71-76
    hipSetDevice(h->device);
    // the caller's device images are BORROWED until adc_wait returns (like the reference borrows the host pointers for the
    // duration of Match, ADCensusStereo.cpp:78-79): no copy
    h->img_l = const_cast<uint8_t*>(static_cast<const uint8_t*>(d_left));
    h->img_r = const_cast<uint8_t*>(static_cast<const uint8_t*>(d_right));
    if (run_pipeline(h) != hipSuccess) { abort_match(h); return 2; }
    h->device_dst = d_disp;
    h->async_dst = nullptr;
    if (enqueue_output(h) != hipSuccess) { set_error("adc_match_device: output copy", hipGetLastError()); abort_match(h); return 2; }
    return 0;
}

static int match_async_impl(adc_handle* h, const uint8_t* left, const uint8_t* right, float* disp, bool sync_call)
{
    if (!h || !left || !right || !disp) return 1;
    hipSetDevice(h->device);
    const size_t P = (size_t)h->p.W * h->p.H;
    h->img_l = h->img_l_own;
    h->img_r = h->img_r_own;
    // Synchronous adc_match only (the caller cannot touch its buffers before the call returns): hand the pageable pointers to
    // the runtime (its own chunked staging / pin-in-place: measured 145 vs 140 pairs/s at 1080p; ADC_HOST_DIRECT=0 switches it
    // off).  The asynchronous entry points promise "the images may be reused as soon as the call returns", so they always
    // stage through the handle's pinned buffers (complete on return) unless the caller registered its memory.
    static const bool direct_env = [] { const char* e = getenv("ADC_HOST_DIRECT"); return e ? atoi(e) != 0 : true; }();
    const bool direct = direct_env && sync_call;
    // Registered (page-locked) INPUT images are DMA-ed in place only by the synchronous adc_match, which returns after the
    // copy: the asynchronous entry points (adc_match_async, adc_farm_submit) promise that the caller may refill its images as
    // soon as the call returns, so they always stage the inputs (a DMA still in flight would read the refilled pixels).  The
    // OUTPUT map of a registered range is written in place by every entry point (it is the caller's until adc_wait anyway).
    const bool reg_in = sync_call && host_registered(left, P * 3) && host_registered(right, P * 3);
    // (Round 6, measured and NOT adopted: left image first, the kernels that need only the left image -- arms, support counts,
    // aggregation records -- enqueued, the right image on a second stream behind an event.  Pageable buffers: no gain; buffers the
    // caller registered: 192 -> 180 pairs/s -- the cross-stream dependency costs more than the ~0.1 ms of overlap it buys,
    // profiles/r6_ab_upload_overlap.txt.)
    const uint8_t *lsrc = left, *rsrc = right;
    if (!(reg_in || direct)) { // staging: the second image is copied while the first one is on the bus
        memcpy(h->pin_in, left, P * 3);
        lsrc = h->pin_in;
    }
    // (reg_in / direct: DMA from the caller's memory -- page-locked by the caller: asynchronous; pageable: the runtime stages)
    if (ADC_HIP(hipMemcpyAsync(h->img_l, lsrc, P * 3, hipMemcpyHostToDevice, h->stream)) != hipSuccess) {
        set_error("adc_match: upload of the left image", hipGetLastError());
        abort_match(h);
        return 2;
    }
    if (!(reg_in || direct)) {
        memcpy(h->pin_in + P * 3, right, P * 3);
        rsrc = h->pin_in + P * 3;
    }
    if (ADC_HIP(hipMemcpyAsync(h->img_r, rsrc, P * 3, hipMemcpyHostToDevice, h->stream)) != hipSuccess) {
        set_error("adc_match: upload of the right image", hipGetLastError());
        abort_match(h);
        return 2;
    }
    if (run_pipeline(h) != hipSuccess) { abort_match(h); return 2; }
    h->async_dst = disp;
    h->async_dst_direct = host_registered(disp, P * 4) ? 1 : (direct ? 2 : 0);
    h->device_dst = nullptr;
    if (enqueue_output(h) != hipSuccess) { set_error("adc_match: output copy", hipGetLastError()); abort_match(h); return 2; }
    return 0;
}

int adc_match_async(adc_handle* h, const uint8_t* left, const uint8_t* right, float* disp)
{
    return match_async_impl(h, left, right, disp, false);
}

int adc_host_register(void* ptr, size_t bytes)
{
    if (!ptr || !bytes) return 1;
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) { set_error("adc_host_register", hipGetLastError()); return 2; }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_ranges.push_back(HostRange{static_cast<const char*>(ptr), bytes});
    return 0;
}
int adc_host_unregister(void* ptr)
{
    if (!ptr) return 1;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        bool found = false;
        for (size_t i = 0; i < g_host_ranges.size(); i++)
            if (g_host_ranges[i].base == static_cast<const char*>(ptr)) { g_host_ranges.erase(g_host_ranges.begin() + (long)i); found = true; break; }
        if (!found) return 1;
    }
    return hipHostUnregister(ptr) == hipSuccess ? 0 : 2;
}

int adc_wait(adc_handle* h)
{
    if (!h) return 1;
    hipSetDevice(h->device);
    if (ADC_HIP(hipStreamSynchronize(h->stream)) != hipSuccess) { set_error("adc_wait", hipGetLastError()); abort_match(h); return 2; }
    // (1) the aggregation assumed the arm maxima of the previous Match; a longer arm raised the flag and the pass was
    //     skipped: redo with the full ring (valid for every image).
    // (1b) a row of the scanline passes was cut into segments and a segment's warm-up did not reach the state of the full
    //     pass (pin_flags[6] = seams that failed): redo with whole rows, and keep them for the next Matches.
    //     Both redos restart at the aggregation when its first pass computes the matching cost itself (the default): the
    //     pixel records, arms and aggregation records of this pair are still in HBM; otherwise the whole Match runs again
    //     (the inputs are still there).  EVERY redo runs whole scanline rows: its volume differs from the first run's when the
    //     aggregation was skipped, so a seam could fail there that did not fail before (round-4 advisor finding) -- and the
    //     seam count is looked at again behind the redo.
    if (h->pin_flags) {
        for (int attempt = 0; attempt < 2 && (h->pin_flags[7] != 0 || h->pin_flags[6] != 0); attempt++) {
            // (round-5 advisor finding) a too-shallow ring skipped aggregation passes: the row passes and their seam check then ran
            // on a stale volume -- a seam that failed THERE says nothing about this image and must not cost 64 Matches of whole
            // rows (which would also switch the fused tail pass off); the redo re-checks the seams on the real volume
            if (h->pin_flags[7] != 0) { h->arm_redos++; h->arm_known = 0; }
            else if (h->pin_flags[6] != 0) { h->so_seam_redos++; h->so_seg_off = 64; }
            if (h->so_seg_off < 1) h->so_seg_off = 1; // whole rows in every redo
            const bool partial = h->agg_first_fused != 0 && !(h->paper & ADC_PAPER_RIGHT_ARMS);
            if (partial) h->redo_partial++;
            hipError_t e = run_pipeline(h, partial);
            if (e == hipSuccess) e = enqueue_output(h);
            if (e == hipSuccess) e = ADC_HIP(hipStreamSynchronize(h->stream));
            if (e != hipSuccess) { set_error("adc_wait: redo (full aggregation ring / whole scanline rows)", e); abort_match(h); return 2; }
        }
        if (h->pin_flags[7] != 0 || h->pin_flags[6] != 0) { g_last_error = "adc_wait: redo did not clear the speculation flags"; abort_match(h); return 2; }
        h->armmax_host[0] = h->pin_flags[4];
        h->armmax_host[1] = h->pin_flags[5];
        h->arm_known = 1;
        if (h->match_pending && h->so_seg_off > 0) h->so_seg_off--; // (whole rows for a while after a failed seam)
        if (h->match_pending) { // (once per Match: a second adc_wait without a Match in between must not count again)
            // which plan did this image need?  Consecutive Matches that need different plans = a mixed stream: the next 64
            // Matches enqueue both plans and let the device choose (k_aggregate.hip) instead of assuming and redoing
            const int small_L = adc_agg_small_L(h);
            const int plan = (h->pin_flags[4] <= small_L && h->pin_flags[5] <= small_L) ? 1 : 2;
            if (plan == 1) { h->armmax_small[0] = h->pin_flags[4] > 0 ? h->pin_flags[4] : 1; h->armmax_small[1] = h->pin_flags[5] > 0 ? h->pin_flags[5] : 1; }
            if (h->agg_last_plan != 0 && plan != h->agg_last_plan) { h->agg_switches++; h->agg_dual = 64; }
            else if (h->agg_dual > 0) h->agg_dual--;
            h->agg_last_plan = plan;
        }
        h->match_pending = 0;
    }
    // (2) the voting chain ran out of its launch budget before it converged: continue it, redo the stages behind it
    int continued = 0;
    hipError_t e = hipSuccess;
    if (h->irv_pending) {
        // the continuation delivers into the buffer that was disp_l when the chain was enqueued (the stages behind the
        // voting have swapped the roles since)
        float *now_l = h->disp_l, *now_tmp = h->disp_tmp;
        h->disp_l = h->tail_disp_l;
        h->disp_tmp = h->tail_disp_tmp;
        e = adc_voting_finish(h, &continued);
        if (!continued) { h->disp_l = now_l; h->disp_tmp = now_tmp; }
    }
    if (e == hipSuccess && continued) {
        e = run_refine_tail(h);
        if (e == hipSuccess) e = enqueue_output(h);
        if (e == hipSuccess) e = ADC_HIP(hipStreamSynchronize(h->stream));
    }
    if (e != hipSuccess) { set_error("adc_wait: region voting continuation", e); abort_match(h); return 2; }
    // (3) a median band gave up waiting for its upstream band: the map is incomplete -- redo the filter with the
    //     single-workgroup kernel (no inter-workgroup dependency) and deliver that result
    if (h->pin_flags && (h->pin_flags[0] != 0 || h->force_median_fallback)) {
        e = adc_median_fallback(h); // (looks at pin_flags[0]: 2 = a speculative seam differed -> chained form first)
        h->pin_flags[0] = 0;
        if (e == hipSuccess) e = enqueue_output(h);
        if (e == hipSuccess) e = ADC_HIP(hipStreamSynchronize(h->stream));
        h->median_fallbacks++;
        if (e != hipSuccess) { set_error("adc_wait: median fallback", e); abort_match(h); return 2; }
    }
    h->force_median_fallback = 0;
    if (h->med_spec_off > 0 && h->med_spec_last == 0) h->med_spec_off--;
    else if (h->med_seg_off > 0 && h->med_seg_last <= 1) h->med_seg_off--; // (whole rows again because a segment seam had failed)
    if (h->async_dst) {
        if (h->async_dst_direct == 2) {
            if (ADC_HIP(hipMemcpy(h->async_dst, h->disp_l, (size_t)h->p.W * h->p.H * 4, hipMemcpyDeviceToHost)) != hipSuccess) { set_error("adc_wait: copy-out", hipGetLastError()); abort_match(h); return 2; }
        } else if (h->async_dst_direct == 0) {
            memcpy(h->async_dst, h->pin_out, (size_t)h->p.W * h->p.H * 4);
        }
        h->async_dst = nullptr;
    }
    h->device_dst = nullptr;
    // adc_match_device BORROWED the caller's device images until here: nothing of the handle may point at them any
    // more (a later debug stage would otherwise read memory the caller has reused or freed)
    if (h->img_l != h->img_l_own || h->img_r != h->img_r_own) {
        h->img_l = h->img_l_own;
        h->img_r = h->img_r_own;
        h->bgrx_valid = 0;
    }
    collect_timings(h);
    return 0;
}

int adc_match(adc_handle* h, const uint8_t* left, const uint8_t* right, float* disp)
{
    const int rc = match_async_impl(h, left, right, disp, true);
    if (rc != 0) return rc;
    return adc_wait(h);
}

// ------------------------------------------------------------------------------ pair farm
struct adc_farm {
    std::vector<adc_handle*> pipes;
    std::vector<uint64_t> in_flight; // ticket of the pair in flight on each pipeline (0 = idle)
    uint64_t next_ticket = 1;
    int64_t delivered = 0;
};

adc_farm* adc_farm_create(int32_t width, int32_t height, const adc_option* opt, int device, int pipelines)
{
    if (pipelines < 1 || pipelines > 64) { g_last_error = "adc_farm_create: pipelines must be 1..64"; return nullptr; }
    adc_farm* f = new (std::nothrow) adc_farm();
    if (!f) return nullptr;
    for (int i = 0; i < pipelines; i++) {
        adc_handle* h = adc_create(width, height, opt, device);
        if (!h) { adc_farm_destroy(f); return nullptr; }
        f->pipes.push_back(h);
        f->in_flight.push_back(0);
    }
    return f;
}
void adc_farm_destroy(adc_farm* f)
{
    if (!f) return;
    for (size_t i = 0; i < f->pipes.size(); i++) {
        if (f->in_flight[i]) adc_wait(f->pipes[i]);
        adc_destroy(f->pipes[i]);
    }
    delete f;
}
static int farm_collect(adc_farm* f, size_t slot)
{
    if (!f->in_flight[slot]) return 0;
    const int rc = adc_wait(f->pipes[slot]);
    f->in_flight[slot] = 0;
    if (rc == 0) f->delivered++;
    return rc;
}
int adc_farm_submit(adc_farm* f, const uint8_t* left, const uint8_t* right, float* disp, uint64_t* ticket)
{
    if (!f || !left || !right || !disp) return 1;
    const uint64_t t = f->next_ticket;
    const size_t slot = (size_t)((t - 1) % f->pipes.size());
    // the pipeline's previous pair (if any) must be delivered before its staging is reused.  When THAT pair failed, the new
    // pair is still enqueued (the caller gets its ticket) and the failure is reported as ADC_FARM_PREVIOUS_FAILED with the
    // failed ticket in adc_last_error(): the caller can tell which output is invalid
    const uint64_t prev = f->in_flight[slot];
    const int rc_prev = farm_collect(f, slot);
    const std::string prev_error = rc_prev != 0 ? g_last_error : std::string();
    int rc = adc_match_async(f->pipes[slot], left, right, disp);
    if (rc != 0) {
        // nothing was enqueued; when the pipeline's previous pair failed as well, say so first (its output is invalid too)
        if (rc_prev != 0)
            g_last_error = "adc_farm_submit: the pair with ticket " + std::to_string((unsigned long long)prev) + " failed (" + prev_error +
                           ") AND the new pair could not be enqueued (" + g_last_error + ")";
        return rc;
    }
    f->in_flight[slot] = t;
    f->next_ticket++;
    if (ticket) *ticket = t;
    if (rc_prev != 0) {
        g_last_error = "adc_farm_submit: the pair with ticket " + std::to_string((unsigned long long)prev) + " failed (" + prev_error + "); the new pair was enqueued";
        return ADC_FARM_PREVIOUS_FAILED;
    }
    return 0;
}
int adc_farm_wait(adc_farm* f, uint64_t ticket)
{
    if (!f || ticket == 0 || ticket >= f->next_ticket) return 1;
    const size_t slot = (size_t)((ticket - 1) % f->pipes.size());
    if (f->in_flight[slot] && f->in_flight[slot] <= ticket) return farm_collect(f, slot);
    return 0; // already delivered (a later pair of the pipeline is in flight, or the pipeline is idle)
}
int64_t adc_farm_drain(adc_farm* f)
{
    if (!f) return -1;
    // oldest first
    for (size_t k = 0; k < f->pipes.size(); k++) {
        size_t best = f->pipes.size();
        for (size_t i = 0; i < f->pipes.size(); i++)
            if (f->in_flight[i] && (best == f->pipes.size() || f->in_flight[i] < f->in_flight[best])) best = i;
        if (best == f->pipes.size()) break;
        if (farm_collect(f, best) != 0) return -1;
    }
    return f->delivered;
}

// ------------------------------------------------------------------------------ misc plumbing
const char* adc_stage_name(int s)
{
    static const char* names[ADC_STAGE_COUNT] = {"cost", "arms", "aggregate", "scanline", "wta", "refine"};
    return (s >= 0 && s < ADC_STAGE_COUNT) ? names[s] : "";
}
int adc_set_paper_modes(adc_handle* h, uint32_t modes)
{
    if (!h || (modes & ~(ADC_PAPER_CENSUS5X5 | ADC_PAPER_SO_SUM | ADC_PAPER_RIGHT_ARMS))) return 1;
    hipSetDevice(h->device);
    const size_t P = (size_t)h->p.W * h->p.H;
    if ((modes & ADC_PAPER_RIGHT_ARMS) && !(h->arms_r && h->bgrx_r && h->armmax_r)) {
        // all three or none: a partial set left behind by a failed call must not pass for "allocated" in the next one
        if ((!h->arms_r && hipMalloc(&h->arms_r, P * 4) != hipSuccess) || (!h->bgrx_r && hipMalloc(&h->bgrx_r, P * 4) != hipSuccess) ||
            (!h->armmax_r && hipMalloc(&h->armmax_r, 4 * sizeof(int)) != hipSuccess)) {
            if (h->arms_r) hipFree(h->arms_r);
            if (h->bgrx_r) hipFree(h->bgrx_r);
            if (h->armmax_r) hipFree(h->armmax_r);
            h->arms_r = nullptr; h->bgrx_r = nullptr; h->armmax_r = nullptr;
            g_last_error = "adc_set_paper_modes: allocation failed";
            return 2;
        }
    }
    if ((modes & ADC_PAPER_SO_SUM) && !h->vol_c) {
        if (hipMalloc(&h->vol_c, P * h->p.Dp * sizeof(float)) != hipSuccess) { g_last_error = "adc_set_paper_modes: allocation failed"; return 2; }
    }
    h->paper = modes;
    return 0;
}
void adc_set_profiling(adc_handle* h, int on) { if (h) h->profiling = on; }
void adc_set_verbose(adc_handle* h, int on) { if (h) { h->verbose = on; if (on) h->profiling = 1; } }
int adc_get_stage_ms(adc_handle* h, float* ms, int n)
{
    if (!h || !ms) return 1;
    for (int i = 0; i < n && i < ADC_STAGE_COUNT; i++) ms[i] = h->stage_ms[i];
    return 0;
}
int adc_get_aggregate_pass_ms(adc_handle* h, float* avg_ms, int* launches)
{
    if (!h) return 1;
    if (avg_ms) *avg_ms = h->agg_pass_ms;
    if (launches) *launches = h->agg_launches - (h->agg_first_fused ? 1 : 0);
    return 0;
}
int adc_get_aggregate_info(adc_handle* h, float* avg_launch_ms, int* launches, int* passes, int* first_fused)
{
    if (!h) return 1;
    const int ff = h->agg_first_fused ? 1 : 0;
    if (avg_launch_ms) *avg_launch_ms = h->agg_pass_ms;
    if (launches) *launches = h->agg_launches - ff;
    if (passes) *passes = h->agg_passes - ff;
    if (first_fused) *first_fused = ff;
    return 0;
}
const char* adc_get_aggregate_kernel(adc_handle* h) { return (h && h->agg_kernel) ? h->agg_kernel : ""; }
void* adc_get_stream(adc_handle* h) { return h ? (void*)h->stream : nullptr; }
int adc_device_synchronize(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : 1; }
void* adc_device_malloc(size_t bytes) { void* p = nullptr; return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr; }
void adc_device_free(void* p) { if (p) hipFree(p); }
double adc_device_copy_ms(void* dst, const void* src, size_t bytes, int reps)
{
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    double best = -1.0;
    for (int r = 0; r < (reps < 1 ? 1 : reps) + 1; r++) { // first copy = warm-up
        hipEventRecord(e0, 0);
        if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0) != hipSuccess) { best = -1.0; break; }
        hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { best = -1.0; break; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && (best < 0 || ms < best)) best = ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best;
}
double adc_device_copy_kernel_ms(void* dst, const void* src, size_t bytes, int reps)
{
    hipEvent_t e0, e1;
    if ((bytes & 15) || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
    double best = -1.0;
    const size_t n = bytes / 16;
    for (int variant = 0; variant < 6 && best > -2.0; variant++) {
        const unsigned grid = variant < 2 ? 8192u : (variant < 4 ? 32768u : 65536u);
        for (int r = 0; r < (reps < 1 ? 1 : reps) + 1; r++) { // first copy = warm-up
            hipEventRecord(e0, 0);
            if (variant & 1) hipLaunchKernelGGL((k_copy_yardstick<true>), dim3(grid), dim3(256), 0, 0, (const adc_vf4*)src, (adc_vf4*)dst, n);
            else hipLaunchKernelGGL((k_copy_yardstick<false>), dim3(grid), dim3(256), 0, 0, (const adc_vf4*)src, (adc_vf4*)dst, n);
            hipEventRecord(e1, 0);
            if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { best = -3.0; break; }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (r > 0 && (best < 0 || ms < best)) best = ms;
        }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best < 0 ? -1.0 : best;
}
int adc_memcpy_h2d(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; }
int adc_memcpy_d2h(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1; }

// ------------------------------------------------------------------------------ debug surface
struct BufDesc { void* ptr; size_t bytes; bool volume; };
static BufDesc buf_desc(adc_handle* h, int which)
{
    const size_t P = (size_t)h->p.W * h->p.H;
    switch (which) {
    case ADC_BUF_GRAY_LEFT: return {h->gray_l, P, false};
    case ADC_BUF_GRAY_RIGHT: return {h->gray_r, P, false};
    case ADC_BUF_CENSUS_LEFT: return {h->census_l, P * 8, false};
    case ADC_BUF_CENSUS_RIGHT: return {h->census_r, P * 8, false};
    case ADC_BUF_ARMS: return {h->arms, P * 4, false};
    case ADC_BUF_SUPCOUNT_H: return {h->sup_h, P * 2, false};
    case ADC_BUF_SUPCOUNT_V: return {h->sup_v, P * 2, false};
    case ADC_BUF_VOLUME_A: return {h->vol_a, P * h->p.D * 4, true};
    case ADC_BUF_DISP_LEFT: return {h->disp_l, P * 4, false};
    case ADC_BUF_DISP_RIGHT: return {h->disp_r, P * 4, false};
    case ADC_BUF_OUTLIER_LABEL: return {h->label, P, false};
    default: return {nullptr, 0, false};
    }
}

int adc_debug_read(adc_handle* h, int which, void* dst)
{
    if (!h || !dst) return 1;
    hipSetDevice(h->device);
    const BufDesc b = buf_desc(h, which);
    if (!b.ptr) return 1;
    if (b.volume) { // de-pad through vol_b (scratch at stage boundaries)
        if (adc_launch_unpad_volume(h, h->vol_a, h->vol_b) != hipSuccess) return 2;
        if (hipMemcpyAsync(dst, h->vol_b, b.bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 2;
    } else if (hipMemcpyAsync(dst, b.ptr, b.bytes, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 2;
    return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : 2;
}

int adc_debug_write(adc_handle* h, int which, const void* src)
{
    if (!h || !src) return 1;
    hipSetDevice(h->device);
    const BufDesc b = buf_desc(h, which);
    if (!b.ptr) return 1;
    if (b.volume) {
        if (hipMemcpyAsync(h->vol_b, src, b.bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 2;
        if (adc_launch_pad_volume(h, h->vol_b, h->vol_a) != hipSuccess) return 2;
    } else if (hipMemcpyAsync(b.ptr, src, b.bytes, hipMemcpyHostToDevice, h->stream) != hipSuccess) return 2;
    return hipStreamSynchronize(h->stream) == hipSuccess ? 0 : 2;
}

int adc_debug_set_images(adc_handle* h, const uint8_t* left, const uint8_t* right)
{
    if (!h || !left || !right) return 1;
    hipSetDevice(h->device);
    const size_t P = (size_t)h->p.W * h->p.H;
    h->img_l = h->img_l_own;
    h->img_r = h->img_r_own;
    if (hipMemcpy(h->img_l, left, P * 3, hipMemcpyHostToDevice) != hipSuccess) return 2;
    if (hipMemcpy(h->img_r, right, P * 3, hipMemcpyHostToDevice) != hipSuccess) return 2;
    h->bgrx_valid = 0;
    return 0;
}

int adc_debug_run(adc_handle* h, int stage, int arg)
{
    if (!h) return 1;
    hipSetDevice(h->device);
    hipError_t e = hipSuccess;
    switch (stage) {
    case ADC_RUN_GRAY_CENSUS: e = adc_launch_gray_census(h); break;
    case ADC_RUN_COST: e = adc_launch_cost(h, h->vol_a); break;
    case ADC_RUN_ARMS: e = adc_launch_arms(h); break;
    case ADC_RUN_AGGREGATE: // arg = iterations (default 4); arg >= 100: first pass with the fused cost computation
        e = adc_launch_records(h); // (needs ADC_RUN_GRAY_CENSUS before; reads the images instead of ADC_BUF_COST_INIT)
        // arg >= 200: additionally read the maximum arms back (needs ADC_RUN_ARMS before) so that the launcher picks
        // the ring depth on the host and fuses same-direction pass pairs -- the production pipeline's path
        if (e == hipSuccess && arg >= 200) {
            e = hipMemcpy(h->armmax_host, h->armmax, 2 * sizeof(int), hipMemcpyDeviceToHost);
            h->armmax_valid = e == hipSuccess ? 1 : 0;
            arg -= 200;
        }
        if (e == hipSuccess && arg >= 100) {
            if (h->paper & ADC_PAPER_RIGHT_ARMS) e = adc_launch_cost(h, h->vol_a); // (no fused form in this mode: recompute the volume)
            else { e = adc_launch_cost_records(h); h->fuse_cost = 1; }
            arg -= 100;
        }
        if (e == hipSuccess) e = adc_launch_aggregate(h, arg > 0 ? arg : 4);
        h->fuse_cost = 0;
        h->armmax_valid = 0;
        break;
    case ADC_RUN_SCANLINE: // arg = passes (default 4); arg >= 100: the production form of the last pass, which also
                           // writes the left-view disparity map (ADC_BUF_DISP_LEFT) -- ADC_RUN_WTA then only adds the right view
        h->fuse_wta = arg >= 100 ? 1 : 0;
        h->wta_left_done = 0;
        e = adc_launch_scanline(h, arg >= 100 ? arg - 100 : arg);
        h->fuse_wta = 0;
        break;
    case ADC_RUN_WTA: e = adc_launch_wta(h); break;
    case ADC_RUN_LRCHECK: e = adc_launch_lrcheck(h); break;
    case ADC_RUN_REGION_VOTING: // arg > 0: launch budget (kernels) of this run, e.g. 4 to force the continuation path
        if (arg < 0) { h->irv_budget = -arg; return 0; } // test hook: only set the budget of the NEXT Match's chain (continuation inside adc_wait)
        if (arg > 0) h->irv_budget = arg;
        e = adc_launch_sup_counts(h); // (the region boxes of the votes come out of the arms stage; here the arms may have been written by the test)
        if (e == hipSuccess) e = adc_run_region_voting(h);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e == hipSuccess) { int cont = 0; e = adc_voting_finish(h, &cont); }
        break;
    case ADC_RUN_INTERPOLATION: e = adc_launch_interpolation(h); break;
    case ADC_RUN_DISCONTINUITY: e = adc_launch_discontinuity(h); break;
    case ADC_RUN_MEDIAN: // arg 100: test hook -- arm the fallback path of the NEXT adc_wait (as if a band had timed out)
        if (arg == 100) { h->force_median_fallback = 1; return 0; }
        if (arg == 101) { h->force_median_fallback = 2; return 0; } // ... as if a speculative seam had differed (chained form redone)
        e = adc_launch_median(h);
        break;
    default: return 1;
    }
    if (e != hipSuccess) { set_error("adc_debug_run launch", e); return 2; }
    e = hipStreamSynchronize(h->heavy);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { set_error("adc_debug_run sync", e); return 2; }
    if (stage == ADC_RUN_SCANLINE && h->so_nseg_last > 1) {
        // (round-4 advisor finding) the row passes ran as speculative segments; behind a Match adc_wait redoes the stage with whole
        // rows when a seam failed -- here the aggregated volume is gone (the passes ping-pong over it), so a failed seam is an
        // ERROR of the debug call instead of a silently inexact volume
        int fails = 0;
        if (hipMemcpy(&fails, h->armmax + 2, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { set_error("adc_debug_run: seam flag", hipGetLastError()); return 2; }
        if (fails != 0) { g_last_error = "adc_debug_run(ADC_RUN_SCANLINE): a speculative row segment failed its seam check; rerun with ADC_SO_SEG=1"; return 3; }
    }
    if (stage == ADC_RUN_MEDIAN && h->pin_flags && h->pin_flags[0] != 0) { // what adc_wait does behind a Match
        e = adc_median_fallback(h);
        h->pin_flags[0] = 0;
        h->median_fallbacks++;
        if (e != hipSuccess) { set_error("adc_debug_run: median fallback", e); return 2; }
    }
    return 0;
}

int64_t adc_debug_counter(adc_handle* h, int which)
{
    if (!h) return -1;
    switch (which) {
    case 0: return h->median_fallbacks;
    case 1: return h->irv_overflows;
    case 2: return h->arm_redos;
    case 9: return h->agg_switches;   // consecutive Matches that needed different aggregation plans
    case 10: return h->agg_dual_runs; // Matches whose aggregation was enqueued as two plans (the device chose)
    case 11: return h->redo_partial;  // redos that restarted at the aggregation (not the whole Match)
    case 12: return h->agg_dual;      // > 0: the next Match enqueues both plans
    case 13: return h->agg_so_fusions; // Matches whose last aggregation pass ran inside the first scanline pass
    case 14: return h->irv_xcd_mode;   // the voting chain sweeps band -> XCD (the mapping was probed on this device)
    case 15: return h->med_seg_last;   // column segments per band link of the last banded median launch (1: whole rows)
    case 3: return h->irv_budget;
    case 7: return h->med_spec_fails;
    case 8: return h->med_spec_last;
    case 4: return h->so_seam_redos;
    case 5: return h->so_nseg_last; // segments per row of the last scanline run
    case 6: { // seams that failed in the last scanline run (debug surface: nothing redoes it there)
        int v = -1;
        return hipMemcpy(&v, h->armmax + 2, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess ? v : -1;
    }
    default: return -1;
    }
}

int adc_debug_voting_stats(adc_handle* h, int64_t* rounds, int64_t* evaluations)
{
    if (!h) return 1;
    if (rounds) *rounds = h->vote_rounds;
    if (evaluations) *evaluations = h->vote_evals;
    return 0;
}

} // extern "C"
