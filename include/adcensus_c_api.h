/*
 * adcensus_c_api.h -- C ABI of the MI355X-native AD-Census stereo matcher.
 *
 * This is the drop-in boundary underneath the C++ facade `ADCensusStereo`
 * (include/ADCensusStereo.h).  Everything crossing it is plain C: pointers, sizes, PODs.
 * No torch / HIP types appear in any signature (a stream is passed as an opaque void*).
 *
 * Each entry point names the reference interface it replaces
 * (paths are relative to the reference checkout, AD-Census/...):
 *
 *   adc_option            <- struct ADCensusOption            adcensus_types.h:45-75
 *   adc_create            <- ADCensusStereo::Initialize        ADCensusStereo.cpp:21-67
 *   adc_match             <- ADCensusStereo::Match             ADCensusStereo.cpp:69-132
 *   adc_destroy           <- ~ADCensusStereo / Release         ADCensusStereo.cpp:15-19,312-316
 *   (Reset == adc_destroy + adc_create                         ADCensusStereo.cpp:134-144)
 *
 * Additive entry points (no reference counterpart; they do not change Match semantics):
 *   adc_match_device      device-resident in/out buffers (bench: inputs already in HBM)
 *   adc_match_async/wait  several objects in flight from one host thread
 *   adc_get_stage_ms      HIP-event stage timers (the reference printf()s stage times,
 *                         ADCensusStereo.cpp:81-129)
 *   adc_debug_*           per-stage entry points used ONLY by the parity tests
 *
 * Image format (cost_computor.cpp:66-68, main.cpp:65-76): tightly packed row-major
 * uint8[H][W][3], channel order B,G,R.  Output: float32[H][W] left-view disparity.
 */
#ifndef ADCENSUS_C_API_H_
#define ADCENSUS_C_API_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Plain-C mirror of ADCensusOption (adcensus_types.h:45-75); same fields, same order,
 * same defaults (adc_option_default).  bools are uint8_t. */
typedef struct adc_option {
    int32_t min_disparity;               /* default 0   */
    int32_t max_disparity;               /* default 64  */
    int32_t lambda_ad;                   /* default 10  */
    int32_t lambda_census;               /* default 30  */
    int32_t cross_L1;                    /* default 34  */
    int32_t cross_L2;                    /* default 17  */
    int32_t cross_t1;                    /* default 20  */
    int32_t cross_t2;                    /* default 6   */
    float   so_p1;                       /* default 1.0 */
    float   so_p2;                       /* default 3.0 */
    int32_t so_tso;                      /* default 15  */
    int32_t irv_ts;                      /* default 20  */
    float   irv_th;                      /* default 0.4 */
    float   lrcheck_thres;               /* default 1.0 */
    uint8_t do_lr_check;                 /* default 1   */
    uint8_t do_filling;                  /* default 1   */
    uint8_t do_discontinuity_adjustment; /* default 0   */
    uint8_t reserved_;
} adc_option;

typedef struct adc_handle adc_handle;

/* Fills *opt with the reference defaults (adcensus_types.h:67-74). */
void adc_option_default(adc_option* opt);

/* Library / device info. Returns number of visible HIP devices (<=0: none / error). */
int adc_device_count(void);
const char* adc_version(void);
/* Last error text of the calling thread ("" if none). */
const char* adc_last_error(void);

/*
 * Initialize.  Returns NULL when the reference's Initialize returns false
 * (width<=0 || height<=0, ADCensusStereo.cpp:31-33; max_disparity-min_disparity<=0, :38-40),
 * on a HIP failure (including out of memory), or when the disparity range exceeds ADC_MAX_DISP_RANGE
 * (2047 = 32 disparities per lane and the 11-bit histogram bins of the voting state map; the reference accepts any positive range
 * its host memory holds -- at 2047 a 1080p cost volume is 17 GB --, larger ones return NULL here).
 * device < 0 means "current device".  All device scratch is allocated here, once.
 */
#define ADC_MAX_DISP_RANGE 2047
adc_handle* adc_create(int32_t width, int32_t height, const adc_option* opt, int device);
void adc_destroy(adc_handle* h);

/*
 * Match (synchronous).  Host pointers.  0 = ok; nonzero = the reference's `false`
 * (any pointer NULL, ADCensusStereo.cpp:74-76) or a HIP error.
 * Does H2D (2 x 3*W*H bytes), the kernels, D2H (4*W*H bytes) on the handle's stream.
 */
int adc_match(adc_handle* h, const uint8_t* bgr_left, const uint8_t* bgr_right, float* disp_left);

/* Same pipeline, device-resident buffers (already in HBM); asynchronous on the handle's stream: the call only ENQUEUES
 * (no host synchronisation anywhere in the pipeline) and returns; call adc_wait() before reading d_disp_left.  The two
 * image buffers are BORROWED until adc_wait returns (not copied: the caller must not overwrite or free them before; after
 * adc_wait the handle holds no pointer to them).  Where the reference would decide something on the
 * host in mid-pipeline, the device decides or verifies: the aggregation uses the ring depth of the previous Match of the
 * handle and checks it on the device; the region voting is a kernel chain driven by a device-side state machine with a
 * launch budget adapted from the previous Match.  adc_wait completes whatever such an assumption left open (redo with
 * the full aggregation ring, continuation of the voting chain) -- slower for that one call, identical results always.
 * To overlap several pairs on one GPU, keep several handles in flight from ONE host thread (bench.py --inflight N,
 * adc_farm_* below). */
int adc_match_device(adc_handle* h, const void* d_bgr_left, const void* d_bgr_right, void* d_disp_left);

/* Host buffers, asynchronous (pinned staging inside the handle); adc_wait() completes it and
 * copies the result to disp_left given here. */
int adc_match_async(adc_handle* h, const uint8_t* bgr_left, const uint8_t* bgr_right, float* disp_left);
int adc_wait(adc_handle* h);

/* Opt-in for host callers that keep their buffers alive: page-lock a host range (hipHostRegister) and tell the library.
 * Images / disparity maps passed to adc_match / adc_match_async / adc_farm_submit that lie inside a registered range are
 * then transferred by DMA straight from / to the caller's memory -- no pinned staging copies (2 x 6.2 MB in, 8.3 MB out
 * per 1080p pair).  INPUT images are read in place only by the synchronous adc_match (which returns after the copy);
 * adc_match_async and adc_farm_submit keep their contract "the images may be reused as soon as the call returns" and still
 * stage them.  A registered OUTPUT map is written in place by every entry point (it belongs to the library until adc_wait /
 * adc_farm_wait has delivered it).  The caller must adc_host_unregister(ptr) BEFORE freeing the memory.  Process-wide, thread-safe.
 * 0 ok, 1 bad argument / unknown pointer, 2 HIP failure. */
int adc_host_register(void* ptr, size_t bytes);
int adc_host_unregister(void* ptr);

/* -------------------------------------------------------------------------------------------
 * Pair farm: a persistent set of `pipelines` matcher objects of one geometry on one device, each
 * with its own stream and pinned staging buffers (the ring), fed round-robin.  adc_farm_submit copies the pair into the
 * next pipeline's pinned slot and enqueues the whole Match asynchronously -- it blocks only when that pipeline is still
 * busy with an earlier pair, and then exactly until that pair is done and delivered.  Results arrive in the caller's
 * `disp_left` buffers in submission order; adc_farm_wait(ticket) / adc_farm_drain complete them.  Match semantics are
 * those of adc_match (same values, same error codes); the caller's image buffers may be reused as soon as submit returns.
 * ------------------------------------------------------------------------------------------- */
typedef struct adc_farm adc_farm;
adc_farm* adc_farm_create(int32_t width, int32_t height, const adc_option* opt, int device, int pipelines);
void adc_farm_destroy(adc_farm* f);
/* Returns 0 and the ticket (1, 2, 3, ...) of the pair; 1 on bad arguments, 2 on a HIP failure (nothing enqueued);
 * ADC_FARM_PREVIOUS_FAILED when the pair that occupied the pipeline before (ticket - pipelines) failed while it was
 * collected: the NEW pair was enqueued all the same and *ticket is valid; adc_last_error() names the failed ticket. */
#define ADC_FARM_PREVIOUS_FAILED 3
int adc_farm_submit(adc_farm* f, const uint8_t* bgr_left, const uint8_t* bgr_right, float* disp_left, uint64_t* ticket);
/* Blocks until the pair with this ticket (and every earlier pair of its pipeline) has been delivered. */
int adc_farm_wait(adc_farm* f, uint64_t ticket);
/* Completes everything submitted so far; returns the number of pairs delivered since creation, negative on failure. */
int64_t adc_farm_drain(adc_farm* f);

/* Stage timers (ms, HIP events on the handle's stream) of the most recent completed match.
 * Enable with adc_set_profiling(h,1).  Order: see adc_stage_name().
 * Level 2 records only the marks around the aggregation launches (adc_aggregate_info: the live duration of the roofline kernel) and
 * no stage marks -- every event record on the stream costs ~6 us of its time, ten of them 1.3 % of a 1080p Match (bench.py times its
 * region at level 2 and measures the stage times on extra Matches behind it). */
enum {
    ADC_STAGE_COST = 0,       /* gray + census + AD-census cost volume   (cost_computor.cpp)      */
    ADC_STAGE_ARMS,           /* cross arms + support counts             (cross_aggregator.cpp:76-86,271-325) */
    ADC_STAGE_AGGREGATE,      /* 4 iterations x (H,V) passes             (cross_aggregator.cpp:89-118) */
    ADC_STAGE_SCANLINE,       /* 4 chained DP passes                     (scanline_optimizer.cpp:40-61) */
    ADC_STAGE_WTA,            /* left + right WTA / sub-pixel            (ADCensusStereo.cpp:188-310) */
    ADC_STAGE_REFINE,         /* LR check, region voting, interpolation, [DDA], median (multistep_refiner.cpp:60-87) */
    ADC_STAGE_COUNT
};
const char* adc_stage_name(int stage);
void adc_set_profiling(adc_handle* h, int on);
int adc_get_stage_ms(adc_handle* h, float* ms, int n);
/* Per-kernel-launch average of the aggregation pass kernel over the last match (ms), and the
 * number of launches it averaged (8 for 4 iterations). */
int adc_get_aggregate_pass_ms(adc_handle* h, float* avg_ms, int* launches);
/* The same average with what it covers: number of REGULAR aggregation launches of the last match (the first pass is
 * left out when it computed the matching cost itself: first_fused = 1, write-only), and the number of algorithmic
 * passes (cross_aggregator.cpp:100-118, two per iteration) those launches covered -- a pass-pair launch covers two. */
int adc_get_aggregate_info(adc_handle* h, float* avg_launch_ms, int* launches, int* passes, int* first_fused);
/* Name of the kernel family the last regular aggregation launch of the handle used (static string, "" before the first Match). */
const char* adc_get_aggregate_kernel(adc_handle* h);

/* OPT-IN paper modes: features of the AD-Census paper the reference declares or stores but does not
 * implement.  Default 0 = exactly the reference.  Any other value changes the results BY DEFINITION (no parity with the
 * reference; checked against the test suite's own plain-C restatement of the same definitions).  Functional, not tuned.
 *   ADC_PAPER_CENSUS5X5   5x5 census window (adcensus_types.h:39-42, CensusSize::Census5x5, declared / unimplemented)
 *   ADC_PAPER_SO_SUM      the four scanline paths computed independently and averaged (paper eq. 10) instead of chained
 *                         (scanline_optimizer.cpp:54-60)
 *   ADC_PAPER_RIGHT_ARMS  support regions limited by the arms of BOTH images (cross_aggregator.h:91 stores img_right_, unused)
 * Call between matches (not while one is in flight).  Returns 0, 1 on bad arguments, 2 on an allocation failure. */
#define ADC_PAPER_CENSUS5X5 1u
#define ADC_PAPER_SO_SUM 2u
#define ADC_PAPER_RIGHT_ARMS 4u
int adc_set_paper_modes(adc_handle* h, uint32_t modes);

/* Print the reference's six timing lines from Match (ADCensusStereo.cpp:88-129); default off. */
void adc_set_verbose(adc_handle* h, int on);

/* Plumbing for callers that own device memory / streams elsewhere (e.g. torch). */
void* adc_get_stream(adc_handle* h);                 /* hipStream_t as void* */
int   adc_device_synchronize(void);
void* adc_device_malloc(size_t bytes);
void  adc_device_free(void* p);
int   adc_memcpy_h2d(void* dst, const void* src, size_t bytes);
int   adc_memcpy_d2h(void* dst, const void* src, size_t bytes);
/* Measured device-to-device copy time of `bytes` bytes (hipMemcpyAsync on the null stream, best of `reps`), in ms;
 * negative on error.  bench.py reports 2*bytes/time next to the 8 TB/s peak: the practical HBM ceiling of this device
 * for a pass that reads one volume and writes another. */
double adc_device_copy_ms(void* dst, const void* src, size_t bytes, int reps);
/* the same with a float4 grid-stride copy KERNEL (the hardware guide's yardstick shape), best of a few grid sizes; -1 on error */
double adc_device_copy_kernel_ms(void* dst, const void* src, size_t bytes, int reps);

/* -------------------------------------------------------------------------------------------
 * Test-only debug surface (parity tests drive single stages with oracle-provided inputs).
 * Volumes cross this boundary in the REFERENCE layout [H][W][D] float32 (D = max-min disparity);
 * the padded internal layout is private.
 * ------------------------------------------------------------------------------------------- */
enum {
    ADC_BUF_GRAY_LEFT = 0,    /* u8  [H][W]                                                    */
    ADC_BUF_GRAY_RIGHT,       /* u8  [H][W]                                                    */
    ADC_BUF_CENSUS_LEFT,      /* u64 [H][W]                                                    */
    ADC_BUF_CENSUS_RIGHT,     /* u64 [H][W]                                                    */
    ADC_BUF_ARMS,             /* u8  [H][W][4] = left,right,top,bottom (CrossArm, cross_aggregator.h:17-20) */
    ADC_BUF_SUPCOUNT_H,       /* u16 [H][W]  horizontal-first support count (vec_sup_count_[0]) */
    ADC_BUF_SUPCOUNT_V,       /* u16 [H][W]  vertical-first support count   (vec_sup_count_[1]) */
    ADC_BUF_VOLUME_A,         /* f32 [H][W][D]  the volume holding the latest stage result      */
    ADC_BUF_DISP_LEFT,        /* f32 [H][W]  current left disparity map                         */
    ADC_BUF_DISP_RIGHT,       /* f32 [H][W]                                                     */
    ADC_BUF_OUTLIER_LABEL,    /* u8  [H][W]  0 valid, 1 mismatch, 2 occlusion                   */
    ADC_BUF_COUNT
};
/* Copies a device buffer to host (de-padding volumes). dst must hold the full buffer. */
int adc_debug_read(adc_handle* h, int which, void* dst);
/* Overwrites a device buffer from host (padding volumes). */
int adc_debug_write(adc_handle* h, int which, const void* src);
/* Uploads the image pair without running anything. */
int adc_debug_set_images(adc_handle* h, const uint8_t* bgr_left, const uint8_t* bgr_right);

enum {
    ADC_RUN_GRAY_CENSUS = 0,  /* images -> gray, census                                        */
    ADC_RUN_COST,             /* images, census -> VOLUME_A                                    */
    ADC_RUN_ARMS,             /* left image -> arms, support counts                            */
    ADC_RUN_AGGREGATE,        /* VOLUME_A, arms, counts -> VOLUME_A (4 iterations)             */
    ADC_RUN_SCANLINE,         /* VOLUME_A, images -> VOLUME_A (4 passes)                       */
    ADC_RUN_WTA,              /* VOLUME_A -> DISP_LEFT, DISP_RIGHT                             */
    ADC_RUN_LRCHECK,          /* DISP_LEFT, DISP_RIGHT -> DISP_LEFT, OUTLIER_LABEL             */
    ADC_RUN_REGION_VOTING,    /* DISP_LEFT, OUTLIER_LABEL, arms -> DISP_LEFT                   */
    ADC_RUN_INTERPOLATION,    /* DISP_LEFT, OUTLIER_LABEL, left image -> DISP_LEFT             */
    ADC_RUN_DISCONTINUITY,    /* DISP_LEFT, VOLUME_A -> DISP_LEFT                              */
    ADC_RUN_MEDIAN,           /* DISP_LEFT -> DISP_LEFT (in-place semantics)                   */
    ADC_RUN_COUNT
};
/* Runs ONE stage on the handle's current device buffers and synchronizes. `arg` is stage
 * specific (ADC_RUN_AGGREGATE: number of iterations, 0 -> 4, +100 fused cost, +200 host-chosen
 * ring / pass pairs; ADC_RUN_SCANLINE: number of chained passes 1..4, 0 -> 4, +100 = the
 * production form whose last pass also writes DISP_LEFT (the fused left-view winner-takes-all);
 * ADC_RUN_MEDIAN: 100 = do not run, arm the fallback path of the next adc_wait (single-workgroup kernel); 101 = the same as if
 * a seam of the speculative bands had differed (the chained form of the banded kernel is redone); else ignored). */
int adc_debug_run(adc_handle* h, int stage, int arg);
/* Test-only event counters of the handle: which = 0 -> number of times adc_wait had to redo the median filter with the
 * single-workgroup kernel (hand-off time-out of the banded kernel); 1 -> continuations of the voting chain (launch budget
 * too small); 2 -> aggregation redos (assumed ring depth too small); 3 -> launch budget (kernels) of the next Match's
 * voting chain; 4 -> Matches redone because a scanline row segment failed its seam check, 5 -> segments per row of the last
 * scanline run, 6 -> seams that failed in it; 7 -> speculative median seams that differed, 8 -> the last median used speculative
 * bands; 9 -> consecutive Matches that needed different aggregation plans (short-arm / long-arm image), 10 -> Matches whose
 * aggregation was enqueued as two plans (the device chose), 11 -> redos that restarted at the aggregation, 12 -> Matches for which
 * both plans will still be enqueued, 13 -> Matches whose last aggregation pass ran inside the first scanline pass (the fused tail of
 * short-arm images, k_scanline_seg_agg), 14 -> the voting chain's band -> XCD sweep is in use (0: the device's workgroup -> XCD
 * mapping is not the assumed round-robin, plain schedule), 15 -> column segments per band link of the last banded median launch (1:
 * whole rows -- odd widths, the chained form, or a segment seam failed within the last 64 Matches).  ADC_RUN_REGION_VOTING of adc_debug_run takes the budget of that run as `arg` (0 = keep;
 * arg < 0: run nothing, set the budget of the NEXT Match's chain to -arg). */
int64_t adc_debug_counter(adc_handle* h, int which);
/* Statistics of the last region-voting run: rounds of the fixed-point iteration (all ten passes of the reference iterate at
 * once since round 5) and total vote evaluations. */
int adc_debug_voting_stats(adc_handle* h, int64_t* rounds, int64_t* evaluations);

#ifdef __cplusplus
}
#endif
#endif /* ADCENSUS_C_API_H_ */
