/*
 * adcensus_types.h -- public types of the MI355X-native AD-Census matcher.
 *
 * Source-compatible with the reference's adcensus_types.h: every name that header puts into the
 * global namespace exists here with the same meaning -- the using-declarations for vector / pair
 * (:12-13), SAFE_DELETE (:15-17), the integer / float typedefs (:21-30), Invalid_Float (:33),
 * Large_Float / Small_Float (:35-36), enum CensusSize (:39-42), struct ADCensusOption (:45-75, same
 * fields, order and defaults) and struct ADColor (:80-86) -- so code written against the reference's
 * headers (main.cpp, or a caller that also uses these helper names) compiles unchanged.
 */
#ifndef ADCENSUS_AMD_TYPES_H_
#define ADCENSUS_AMD_TYPES_H_

#include <cstdint>
#include <limits>
#include <utility>
#include <vector>
using std::pair;
using std::vector;

#ifndef SAFE_DELETE
/* array delete + reset, as callers of the reference expect it (adcensus_types.h:15-17) */
#define SAFE_DELETE(P) { if (P) delete[] (P); (P) = nullptr; }
#endif

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

/** Census window selector of the reference (adcensus_types.h:39-42).  The reference declares it and never reads it;
 *  the matcher always uses the 9x7 window (adcensus_util.cpp:10-39), and so does this implementation. */
enum CensusSize { Census5x5 = 0, Census9x7 };

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

/** One colour sample; note the constructor's argument order (blue, green, red), adcensus_types.h:80-86. */
struct ADColor {
    uint8 r, g, b;
    ADColor() : r(0), g(0), b(0) {}
    ADColor(uint8 blue, uint8 green, uint8 red) : r(red), g(green), b(blue) {}
};

#endif
