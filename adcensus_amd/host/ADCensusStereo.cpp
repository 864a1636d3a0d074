// ADCensusStereo.cpp -- C++ facade over the C ABI (include/adcensus_c_api.h).  Host C++ only: every
// device operation happens behind adc_* (HIP, gfx950).  Mirrors the reference's error behaviour
// (ADCensusStereo.cpp:21-144): bool returns, no exceptions.
#include "ADCensusStereo.h"
#include "adcensus_c_api.h"

#include <stdlib.h>

static adc_option to_c(const ADCensusOption& o)
{
    adc_option c;
    adc_option_default(&c);
    c.min_disparity = o.min_disparity;  c.max_disparity = o.max_disparity;
    c.lambda_ad = o.lambda_ad;          c.lambda_census = o.lambda_census;
    c.cross_L1 = o.cross_L1;            c.cross_L2 = o.cross_L2;
    c.cross_t1 = o.cross_t1;            c.cross_t2 = o.cross_t2;
    c.so_p1 = o.so_p1;                  c.so_p2 = o.so_p2;            c.so_tso = o.so_tso;
    c.irv_ts = o.irv_ts;                c.irv_th = o.irv_th;          c.lrcheck_thres = o.lrcheck_thres;
    c.do_lr_check = o.do_lr_check ? 1 : 0;
    c.do_filling = o.do_filling ? 1 : 0;
    c.do_discontinuity_adjustment = o.do_discontinuity_adjustment ? 1 : 0;
    return c;
}

// The reference's Match always prints its six stage-timing lines (ADCensusStereo.cpp:88-129): the look-alike facade does
// too; ADC_VERBOSE=0 in the environment or SetVerbose(false) switches them off (the C ABI underneath, which
// the benchmarks use, prints nothing unless asked).
static bool default_verbose()
{
    const char* e = getenv("ADC_VERBOSE");
    return e ? atoi(e) != 0 : true;
}
ADCensusStereo::ADCensusStereo() : impl_(nullptr), device_(-1), verbose_(default_verbose()), profiling_(false), paper_(0) {}
ADCensusStereo::~ADCensusStereo() { Release(); }

void ADCensusStereo::Release()
{
    if (impl_) adc_destroy(impl_);
    impl_ = nullptr;
}

bool ADCensusStereo::Initialize(const sint32& width, const sint32& height, const ADCensusOption& option)
{
    Release(); // the reference leaks here when called twice without Reset (ADCensusStereo.cpp:43-44); we do not
    const adc_option c = to_c(option);
    impl_ = adc_create(width, height, &c, device_);
    if (!impl_) return false;
    if (profiling_) adc_set_profiling(impl_, 1);
    if (verbose_) adc_set_verbose(impl_, 1);
    if (paper_ && adc_set_paper_modes(impl_, paper_) != 0) return false;
    return true;
}

bool ADCensusStereo::Match(const uint8* img_left, const uint8* img_right, float32* disp_left)
{
    if (!impl_) return false;                                    // ADCensusStereo.cpp:71-73
    if (!img_left || !img_right || !disp_left) return false;     // :74-76
    return adc_match(impl_, img_left, img_right, disp_left) == 0;
}

bool ADCensusStereo::Reset(const uint32& width, const uint32& height, const ADCensusOption& option)
{
    Release();
    return Initialize(static_cast<sint32>(width), static_cast<sint32>(height), option);
}

void ADCensusStereo::SetVerbose(bool on)
{
    verbose_ = on;
    if (impl_) adc_set_verbose(impl_, on ? 1 : 0);
}
void ADCensusStereo::SetProfiling(bool on)
{
    profiling_ = on;
    if (impl_) adc_set_profiling(impl_, on ? 1 : 0);
}
bool ADCensusStereo::StageMilliseconds(float ms[6]) const { return impl_ && adc_get_stage_ms(impl_, ms, 6) == 0; }
bool ADCensusStereo::MatchAsync(const uint8* l, const uint8* r, float32* d)
{
    if (!impl_ || !l || !r || !d) return false;
    return adc_match_async(impl_, l, r, d) == 0;
}
bool ADCensusStereo::Wait() { return impl_ && adc_wait(impl_) == 0; }
bool ADCensusStereo::SetPaperModes(unsigned modes)
{
    paper_ = modes;
    return impl_ ? adc_set_paper_modes(impl_, modes) == 0 : true; // (before Initialize: applied there)
}
const char* ADCensusStereo::LastError() const { return adc_last_error(); }
