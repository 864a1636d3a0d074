/*
 * ADCensusStereo.h -- drop-in C++ facade of the MI355X-native AD-Census matcher.
 *
 * Same public surface as the reference's class (ADCensusStereo.h:14-41):
 *     bool Initialize(const sint32& width, const sint32& height, const ADCensusOption& option);
 *     bool Match(const uint8* img_left, const uint8* img_right, float32* disp_left);
 *     bool Reset(const uint32& width, const uint32& height, const ADCensusOption& option);
 * so the reference's caller (main.cpp:80-118) compiles unchanged.  The private part is a pimpl over
 * the C ABI of include/adcensus_c_api.h (HIP kernels for gfx950); there is no CPU path.
 *
 * Additive members (no reference counterpart): SetDevice, SetVerbose (prints the reference's six stage
 * timing lines, ADCensusStereo.cpp:88-129: ON by default like the reference, ADC_VERBOSE=0 or SetVerbose(false) turns them off), StageMilliseconds, MatchAsync/Wait.
 */
#pragma once

#include "adcensus_types.h"

struct adc_handle;

class ADCensusStereo {
public:
    ADCensusStereo();
    ~ADCensusStereo();
    ADCensusStereo(const ADCensusStereo&) = delete;
    ADCensusStereo& operator=(const ADCensusStereo&) = delete;

    /** Allocates all device buffers once. false: width/height <= 0, empty disparity range
     *  (ADCensusStereo.cpp:31-40), range > ADC_MAX_DISP_RANGE (2047), W*H > 2^30, or a HIP failure. */
    bool Initialize(const sint32& width, const sint32& height, const ADCensusOption& option);

    /** Left-view sub-pixel disparity map of the pair (uint8 [H][W][3] BGR each) into the caller's
     *  float32 [H][W].  false: not initialised or a null pointer (ADCensusStereo.cpp:71-76), HIP failure. */
    bool Match(const uint8* img_left, const uint8* img_right, float32* disp_left);

    /** Release + Initialize (ADCensusStereo.cpp:134-144). */
    bool Reset(const uint32& width, const uint32& height, const ADCensusOption& option);

    // ---- additive API ----
    void SetDevice(int device) { device_ = device; }
    void SetVerbose(bool on);
    /** ms of the 6 stages of the last Match (cost, arms, aggregate, scanline, wta, refine); needs SetProfiling(true). */
    void SetProfiling(bool on);
    bool StageMilliseconds(float ms[6]) const;
    bool MatchAsync(const uint8* img_left, const uint8* img_right, float32* disp_left);
    bool Wait();
    /** Opt-in paper features the reference declares / stores but does not implement (bit 0: 5x5 census, adcensus_types.h:39-42;
     *  bit 1: averaged instead of chained scanline paths; bit 2: right-image arms, cross_aggregator.h:91).  0 (default) = the
     *  reference's behaviour; anything else changes the results by definition.  Call after Initialize. */
    bool SetPaperModes(unsigned modes);
    const char* LastError() const;

private:
    void Release();
    adc_handle* impl_;
    int device_;
    bool verbose_, profiling_;
    unsigned paper_;
};
