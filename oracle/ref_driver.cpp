/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Drives the *unmodified* reference sources (compiled in place from /root/reference/AD-Census by
 * oracle/Makefile, nothing is copied) stage by stage and dumps every intermediate result through
 * the ABI in oracle_abi.h.  The private stage methods of ADCensusStereo (ComputeCost,
 * CostAggregation, ScanlineOptimize, ComputeDisparity, ComputeDisparityRight,
 * ADCensusStereo.h:43-62) and of MultiStepRefiner (multistep_refiner.h:58-77) are reached with
 * the test-only `#define private public` below; class layout is unchanged by it.
 *
 * The canonical recipe is part of the oracle's definition:
 *   g++ -std=c++14 -O2 -ffp-contract=off -include math.h -include stdlib.h -include string.h
 *       -include stdio.h
 * which gives exp(float)->float and abs(float)->float like MSVC's UCRT.  Guarded below.
 */
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <limits>
#include <type_traits>
#include <utility>
#include <vector>

#define private public
#include "ADCensusStereo.h"
#include "adcensus_util.h"
#undef private

#include "oracle_abi.h"

static_assert(std::is_same<decltype(exp(1.0f)), float>::value,
              "oracle recipe broken: exp(float) must resolve to the float overload (use -include math.h)");
static_assert(std::is_same<decltype(abs(1.0f)), float>::value,
              "oracle recipe broken: abs(float) must resolve to the float overload (use -include stdlib.h/math.h)");
#ifdef __FP_FAST_FMA
// -march=native style builds contract a*b+c unless told not to; the Makefile passes -ffp-contract=off.
#endif

namespace {

ADCensusOption to_ref_option(const adc_option* o)
{
    ADCensusOption r;
    r.min_disparity = o->min_disparity;
    r.max_disparity = o->max_disparity;
    r.lambda_ad = o->lambda_ad;
    r.lambda_census = o->lambda_census;
    r.cross_L1 = o->cross_L1;
    r.cross_L2 = o->cross_L2;
    r.cross_t1 = o->cross_t1;
    r.cross_t2 = o->cross_t2;
    r.so_p1 = o->so_p1;
    r.so_p2 = o->so_p2;
    r.so_tso = o->so_tso;
    r.irv_ts = o->irv_ts;
    r.irv_th = o->irv_th;
    r.lrcheck_thres = o->lrcheck_thres;
    r.do_lr_check = o->do_lr_check != 0;
    r.do_filling = o->do_filling != 0;
    r.do_discontinuity_adjustment = o->do_discontinuity_adjustment != 0;
    return r;
}

template <class T>
void dump(T* dst, const T* src, size_t n)
{
    if (dst && src) memcpy(dst, src, n * sizeof(T));
}

} // namespace

extern "C" {

const char* adc_oracle_kind(void) { return "reference"; }
/* compiler + flags this checker was built with (reported next to the CPU baseline by bench.py) */
#ifndef ADC_ORACLE_FLAGS
#define ADC_ORACLE_FLAGS "?"
#endif
const char* adc_oracle_build_info(void) { return "compiler " __VERSION__ ", flags " ADC_ORACLE_FLAGS; }

int adc_oracle_run(int32_t width, int32_t height, const adc_option* opt,
                   const uint8_t* bgr_left, const uint8_t* bgr_right, adc_oracle_dump* out)
{
    ADCensusStereo st;
    const ADCensusOption ro = to_ref_option(opt);
    if (!st.Initialize(width, height, ro)) return 1;
    if (!bgr_left || !bgr_right) return 2;
    adc_oracle_dump none;
    memset(&none, 0, sizeof(none));
    if (!out) out = &none;

    const size_t P = size_t(width) * size_t(height);
    const size_t D = size_t(ro.max_disparity - ro.min_disparity);

    // ADCensusStereo::Match body (ADCensusStereo.cpp:78-125), one stage at a time.
    st.img_left_ = bgr_left;
    st.img_right_ = bgr_right;

    st.ComputeCost(); // :84
    dump(out->gray_left, st.cost_computer_.gray_left_.data(), P);
    dump(out->gray_right, st.cost_computer_.gray_right_.data(), P);
    dump(out->census_left, (const uint64_t*)st.cost_computer_.census_left_.data(), P);
    dump(out->census_right, (const uint64_t*)st.cost_computer_.census_right_.data(), P);
    dump(out->cost_init, st.cost_computer_.get_cost_ptr(), P * D);

    st.CostAggregation(); // :92
    static_assert(sizeof(CrossArm) == 4, "CrossArm is 4 x uint8");
    dump(out->arms, (const uint8_t*)st.aggregator_.get_arms_ptr(), P * 4);
    dump(out->sup_count_h, (const uint16_t*)st.aggregator_.vec_sup_count_[0].data(), P);
    dump(out->sup_count_v, (const uint16_t*)st.aggregator_.vec_sup_count_[1].data(), P);
    dump(out->cost_aggr, st.aggregator_.get_cost_ptr(), P * D);

    st.ScanlineOptimize(); // :100
    dump(out->cost_so, st.aggregator_.get_cost_ptr(), P * D);

    st.ComputeDisparity();      // :108
    st.ComputeDisparityRight(); // :109
    dump(out->disp_left_wta, st.disp_left_, P);
    dump(out->disp_right_wta, st.disp_right_, P);

    // ADCensusStereo::MultiStepRefine (ADCensusStereo.cpp:177-186) + MultiStepRefiner::Refine
    // (multistep_refiner.cpp:60-87), unrolled so that each step can be dumped.
    MultiStepRefiner& rf = st.refiner_;
    rf.SetData(st.img_left_, st.aggregator_.get_cost_ptr(), st.aggregator_.get_arms_ptr(), st.disp_left_, st.disp_right_);
    rf.SetParam(ro.min_disparity, ro.max_disparity, ro.irv_ts, ro.irv_th, ro.lrcheck_thres,
                ro.do_lr_check, ro.do_filling, ro.do_filling, ro.do_discontinuity_adjustment);
    if (rf.do_lr_check_) rf.OutlierDetection();
    if (out->outlier_label) {
        memset(out->outlier_label, 0, P);
        for (auto& p : rf.mismatches_) out->outlier_label[size_t(p.second) * width + p.first] = 1;
        for (auto& p : rf.occlusions_) out->outlier_label[size_t(p.second) * width + p.first] = 2;
    }
    dump(out->disp_after_lr, st.disp_left_, P);
    if (rf.do_region_voting_) rf.IterativeRegionVoting();
    dump(out->disp_after_irv, st.disp_left_, P);
    if (rf.do_interpolating_) rf.ProperInterpolation();
    dump(out->disp_after_interp, st.disp_left_, P);
    if (rf.do_discontinuity_adjustment_) rf.DepthDiscontinuityAdjustment();
    dump(out->disp_after_dda, st.disp_left_, P);
    adcensus_util::MedianFilter(st.disp_left_, st.disp_left_, width, height, 3);
    dump(out->disp_final, st.disp_left_, P);
    return 0;
}

int adc_oracle_match(int32_t width, int32_t height, const adc_option* opt,
                     const uint8_t* bgr_left, const uint8_t* bgr_right, float* disp_left,
                     double* seconds_match)
{
    ADCensusStereo st;
    if (!st.Initialize(width, height, to_ref_option(opt))) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = st.Match(bgr_left, bgr_right, disp_left);
    const auto t1 = std::chrono::steady_clock::now();
    if (seconds_match) *seconds_match = std::chrono::duration<double>(t1 - t0).count();
    return ok ? 0 : 2;
}

void adc_oracle_median3_inplace(float* disp, int32_t width, int32_t height)
{
    adcensus_util::MedianFilter(disp, disp, width, height, 3);
}

} // extern "C"
