/*
 * oracle_abi.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Common C ABI exported by both CPU oracles:
 *   oracle/_ref/libadcensus_ref.so   = the reference's own six library .cpp files compiled in
 *                                      place from /root/reference (oracle/Makefile, kind
 *                                      "reference"), driven by oracle/ref_driver.cpp;
 *   oracle/_port/libadcensus_port.so = oracle/adcensus_port.c, a plain-C restatement of the
 *                                      same algorithm (kind "port").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these.
 * The product (adcensus_amd/, include/) never links, imports or calls anything in oracle/.
 */
#ifndef ADCENSUS_ORACLE_ABI_H_
#define ADCENSUS_ORACLE_ABI_H_

#include <stdint.h>
#include "../include/adcensus_c_api.h" /* adc_option (plain-C mirror of ADCensusOption) */

#ifdef __cplusplus
extern "C" {
#endif

/* Every pointer may be NULL (= do not dump that stage).  Layouts are the reference's. */
typedef struct adc_oracle_dump {
    uint8_t*  gray_left;        /* u8  [H][W]      cost_computor.cpp:58-73                        */
    uint8_t*  gray_right;
    uint64_t* census_left;      /* u64 [H][W]      adcensus_util.cpp:10-39                        */
    uint64_t* census_right;
    float*    cost_init;        /* f32 [H][W][D]   cost_computor.cpp:82-121                       */
    uint8_t*  arms;             /* u8  [H][W][4]   left,right,top,bottom cross_aggregator.cpp:76-86 */
    uint16_t* sup_count_h;      /* u16 [H][W]      cross_aggregator.cpp:271-325 (id 0)            */
    uint16_t* sup_count_v;      /*                 (id 1)                                          */
    float*    cost_aggr;        /* f32 [H][W][D]   after Aggregate(4), cross_aggregator.cpp:89-118 */
    float*    cost_so;          /* f32 [H][W][D]   after Optimize(),   scanline_optimizer.cpp:40-61 */
    float*    disp_left_wta;    /* f32 [H][W]      ADCensusStereo.cpp:188-243                     */
    float*    disp_right_wta;   /* f32 [H][W]      ADCensusStereo.cpp:245-310                     */
    uint8_t*  outlier_label;    /* u8  [H][W]      0 valid,1 mismatch,2 occlusion multistep_refiner.cpp:90-151 */
    float*    disp_after_lr;    /* f32 [H][W]                                                      */
    float*    disp_after_irv;   /* f32 [H][W]      multistep_refiner.cpp:153-227                  */
    float*    disp_after_interp;/* f32 [H][W]      multistep_refiner.cpp:229-305                  */
    float*    disp_after_dda;   /* f32 [H][W]      multistep_refiner.cpp:307-352 (== interp if off) */
    float*    disp_final;       /* f32 [H][W]      after the in-place 3x3 median, multistep_refiner.cpp:86 */
} adc_oracle_dump;

/* "reference" or "port". */
const char* adc_oracle_kind(void);
const char* adc_oracle_build_info(void);

/* Whole pipeline stage by stage with dumps.  0 = ok, nonzero = Initialize failed. */
int adc_oracle_run(int32_t width, int32_t height, const adc_option* opt,
                   const uint8_t* bgr_left, const uint8_t* bgr_right, adc_oracle_dump* dump);

/* Plain Initialize + Match through the public API only.  Returns 0 ok.
 * seconds_match (nullable) receives the steady_clock time of Match alone. */
int adc_oracle_match(int32_t width, int32_t height, const adc_option* opt,
                     const uint8_t* bgr_left, const uint8_t* bgr_right, float* disp_left,
                     double* seconds_match);

/* Stand-alone stage helpers (inputs supplied by the caller) used by property tests. */
/* in-place-semantics 3x3 median (adcensus_util.cpp:55-81 called with in==out). */
void adc_oracle_median3_inplace(float* disp, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif
