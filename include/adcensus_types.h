/*
 * adcensus_types.h -- public types of the MI355X-native AD-Census matcher.
 *
 * Source-compatible with the reference's adcensus_types.h (typedefs :21-30, Invalid_Float :33,
 * Large_Float / Small_Float :35-36, struct ADCensusOption :45-75 -- same fields, order and
 * defaults), so code written against the reference's ADCensusStereo compiles unchanged.
 */
#ifndef ADCENSUS_AMD_TYPES_H_
#define ADCENSUS_AMD_TYPES_H_

#include <cstdint>
#include <limits>

typedef int8_t   sint8;
typedef uint8_t  uint8;
typedef int16_t  sint16;
typedef uint16_t uint16;
typedef int32_t  sint32;
typedef uint32_t uint32;
typedef int64_t  sint64;
typedef uint64_t uint64;
typedef float    float32;
typedef double   float64;

constexpr auto Invalid_Float = std::numeric_limits<float32>::infinity();
constexpr auto Large_Float = 99999.0f;
constexpr auto Small_Float = -99999.0f;

/** AD-Census parameters; every field is live in the HIP implementation. */
struct ADCensusOption {
    sint32  min_disparity;
    sint32  max_disparity;
    sint32  lambda_ad;
    sint32  lambda_census;
    sint32  cross_L1;
    sint32  cross_L2;
    sint32  cross_t1;
    sint32  cross_t2;
    float32 so_p1;
    float32 so_p2;
    sint32  so_tso;
    sint32  irv_ts;
    float32 irv_th;
    float32 lrcheck_thres;
    bool    do_lr_check;
    bool    do_filling;
    bool    do_discontinuity_adjustment;

    ADCensusOption() : min_disparity(0), max_disparity(64), lambda_ad(10), lambda_census(30), cross_L1(34), cross_L2(17),
                       cross_t1(20), cross_t2(6), so_p1(1.0f), so_p2(3.0f), so_tso(15), irv_ts(20), irv_th(0.4f),
                       lrcheck_thres(1.0f), do_lr_check(true), do_filling(true), do_discontinuity_adjustment(false) {}
};

#endif
