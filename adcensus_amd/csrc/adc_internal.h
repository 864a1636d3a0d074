// adc_internal.h -- private state of an adc_handle and the kernel-launch entry points.
//
// HBM layout (all buffers allocated once in adc_create, 288 GB HBM3E is plentiful):
//   img_l / img_r      u8  [H][W][3]  BGR, as handed over by the caller
//   gray_l / gray_r    u8  [H][W]
//   census_l/census_r  u64 [H][W]
//   arms               u8x4[H][W]     left,right,top,bottom
//   sup_h / sup_v      u16 [H][W]     support counts of the H-first / V-first iterations
//   cdiff_*            u8  [H][W]     max-channel colour difference to the left / upper neighbour of
//                                     the left (L) and right (R) image (scanline penalties)
//   vol_a / vol_b      f32 [H][W][Dp] cost volumes (ping-pong).  d innermost, Dp = 64*VPL, VPL in
//                                     {1,2,4}: lanes = disparities, lane l holds d = l*VPL .. l*VPL+VPL-1.
//                                     d >= D are padding (never read by a reduction).
//   disp_l / disp_r    f32 [H][W]
//   label              u8  [H][W]     outlier class (0 valid, 1 mismatch, 2 occlusion)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/adcensus_c_api.h"

#define ADC_WAVE 64

// Streaming hint for the volume accesses (every element is read once and written once per pass; 1 GB >> every cache):
// the loads / stores of the marching kernels (K4, K5, K6) are marked non-temporal: same-box A/B at 1080p, K4 launch 0.394 -> 0.376 ms
// (0.462 -> 0.445 in the slower clock state of the same box), scanline stage -1 %, right-view WTA -4 % (profiles/r5_ab_nontemporal.txt);
// -DADC_VOL_NT=0 switches it off (tools/build_variant.sh).
#if !defined(ADC_VOL_NT) || ADC_VOL_NT
#define ADC_VOL_NT_STR " nt"
#define ADC_VOL_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define ADC_VOL_NT_STR ""
#define ADC_VOL_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif


// Fault injection -- TEST BUILDS ONLY.  libadcensus_hip_faultinj.so (csrc/Makefile, target faultinj) is capi.hip compiled with
// -DADC_FAULT_INJECTION=1 and linked with the product's kernel objects; the product library never defines it.  There every HIP
// call of capi.hip's object-lifetime and Match path goes through ADC_HIP(call): the n-th one (ADC_TEST_FAIL_AT=<n> in the
// environment, or adc_test_fail_at(n)) is NOT executed and reports hipErrorOutOfMemory -- the reference's contract for that is
// `return false` (ADCensusStereo.cpp:31-40,71-76; SURVEY.md 8b "HIP failure -> false").  tests/test_gpu_faults.py.
#ifdef ADC_FAULT_INJECTION
extern "C" int adc_test_fault_now(void);
#define ADC_HIP(call) (adc_test_fault_now() ? hipErrorOutOfMemory : (call))
#else
#define ADC_HIP(call) (call)
#endif

struct AdcParams {
    int W, H;
    int dmin, dmax, D; // D = dmax - dmin
    int VPL;           // disparities per lane (1,2,4,8,16)
    int Dp;            // padded disparity count = 64*VPL
    adc_option opt;
};

struct adc_handle {
    AdcParams p;
    int device;
    hipStream_t stream; // per-object stream: uploads, refinement (latency-bound kernels), downloads
    hipStream_t heavy;  // bandwidth lane: cost/arms/aggregate/scanline/WTA kernels.  Shared by all objects of a
                        // device (FIFO), so the streaming kernels of different pairs never fight for HBM while
                        // the latency-bound refinement of one pair overlaps the streaming phase of the next
    hipEvent_t ev_in, ev_heavy_done;
    bool own_stream;

    // images + per-pixel maps
    uint8_t *img_l, *img_r;         // the pair being matched: the handle's own buffers, or the caller's (adc_match_device)
    uint8_t *img_l_own, *img_r_own;
    uint8_t *gray_l, *gray_r;
    uint64_t *census_l, *census_r;
    uint8_t* arms;
    uint16_t *sup_h, *sup_v;
    int* armmax;             // [0] max horizontal arm, [1] max vertical arm of the current left image
    uint32_t *rec_h, *rec_v; // packed {arm_lo, arm_hi, divisor} per pixel, line-major (rec_v transposed)
    uint32_t *rec2_h, *rec2_v; // uint2 {arm_lo | span<<8 | divisor<<16, RN(1/divisor)}: records of the register-ring kernels
    float* agg_sink;           // 256 KiB scratch: store target of the halo steps of a pass pair (k_aggregate_rr.h)
    uint8_t *cdiff_lh, *cdiff_lv, *cdiff_rh, *cdiff_rv;
    uint8_t* so_cls; // path-ordered left-image colour-step words (d1) of the 4 scanline pass types (k_scanline.hip)
    float* so_seam;  // seam slots of the row passes cut into verified segments: [2 passes][H][ADC_SO_MAX_SEG - 1][Dp] (k_scanline.hip)
    int so_seg_off;  // > 0: row passes run whole (a seam failed: the redo and the next so_seg_off Matches of the handle)
    int so_nseg_last; // segments per row of the last scanline run (1 = whole rows)
    int so_seam_redos; // how often adc_wait had to redo a Match because a seam failed
    // volumes
    float *vol_a, *vol_b;
    // host-built tables: same libm as the CPU reference
    float* lut_ad;      // [766] A[k] = (1 - expf(-(k/3.0f)/lambda_ad)) + 1
    float* lut_census;  // [64]  C[h] = expf(-(float)h/lambda_census)
    double* ray_sincos; // [16][2] (sin, cos) of the 16 interpolation angles
    float so_P1[3], so_P2[3]; // penalty classes (p, p/4, p/10)
    // disparity maps and refinement state
    float *disp_l, *disp_r, *disp_tmp;
    uint8_t* label;
    uint8_t* elig;       // scratch: invalid mask of the LR check (byte per pixel), then the voting chain's bitmap of the pixels on its work list
    uint8_t* irv_bbox;   // uchar4 per pixel: widest H arms {left, right} over ALL region rows (k_sup_counts): read box / dirty box of a vote
    int32_t* vote_list;  // work list of the voting chain: int4 entries, one segment per workgroup in evaluation order (irv_plan.h)
    int32_t* vote_evals_arr; // evaluation counter per wave of the chain's grid (statistics), then the entries per workgroup segment
    int32_t* interp_list;     // target list of the interpolation (its own buffers: the voting chain may be CONTINUED after
    int32_t* interp_counters; // the interpolation has run once, and must find its list and control block untouched)
    int32_t* ray_tab;     // [max_search][16] packed ray offsets (dy<<16 | dx&0xffff), NULL when a step is too close to a .5 tie
    int ray_tab_rows;
    int32_t* ray_lin;     // [max_search + ADC_ITP_LPAD][16] linear ray offsets dy * itp_pitch + dx into the padded code map (same allocation as ray_tab)
    int itp_pitch, itp_ms; // row pitch of the code map and the search range it is padded for (adc_device_fn.h)
    uint32_t* cost_rrec;  // [H][rrec_pitch] uint4 {bgrx, census lo, census hi, 0} of the RIGHT image, padded with out-of-image
                          // markers (bgrx = ~0) on both sides; cost_lrec [H][W] the same for the LEFT image (fused cost, k_aggregate.hip)
    uint32_t* cost_lrec;
    int rrec_pitch, rrec_padl;
    int armmax_host[2];   // maximum horizontal / vertical arm of the current pair, read back by the pipeline
    int armmax_valid;     // 0 unknown (debug surface: two launches per pass, the kernels decide), 1 exact (read back),
                          // 2 assumed from the previous Match of this handle (the small-ring kernels verify on the device and
                          // raise armmax[3]; adc_wait then redoes the Match), 3 unknown in the pipeline: full ring (always valid)
    int arm_known;        // armmax_host holds the maxima of the previous Match
    int arm_redos;        // how often the assumption was wrong
    // Streams that alternate between short-arm and long-arm images: instead of assuming ONE ring depth and redoing the
    // Match when it is wrong, the aggregation is enqueued as TWO plans (assumed small rings + pass pairs | full ring) and the
    // kernels decide on the device which plan works (agg_gate_skip, k_aggregate_rr.h).
    int agg_dual;         // > 0: enqueue both plans (set to 64 whenever consecutive Matches needed different plans, counts down)
    int agg_dual_last;    // the last aggregation run enqueued both plans (per-launch timings are then not separable)
    int agg_last_plan;    // plan the previous Match's image needed: 0 none yet, 1 small rings, 2 full ring
    int agg_switches;     // how often consecutive Matches needed different plans
    int agg_dual_runs;    // Matches whose aggregation was enqueued as two plans
    int armmax_small[2];  // arm maxima of the last image that fitted the small rings (0 = none seen)
    int agg_gate, agg_gate_thr; // set while a plan of a two-plan run is being enqueued: gate code (3 / 4) and packed depths
    int redo_partial;     // redos that restarted at the aggregation instead of the whole Match
    int match_pending;    // a Match was enqueued and adc_wait has not yet looked at its arm maxima / speculation flags
    int fuse_cost;        // set by the pipeline: the first aggregation pass computes the matching cost itself
    int agg_first_fused;  // the last aggregation run did so (pass timings: the regular passes are 1..)
    int fuse_wta;         // set by the pipeline: the last scanline pass also writes the left-view disparity map
    int fuse_agg_so;      // set by the pipeline: the last aggregation pass may move into the first scanline pass (short-arm plan)
    int so_agg_fused;     // adc_launch_aggregate did so: vol_a holds the volume BEFORE that pass (consumed by the scanline stage)
    int agg_so_fusions;   // Matches that ran that way
    int wta_left_done;    // the scanline stage did so: adc_launch_wta only runs the right view
    float* med_hand;      // banded median: per-band hand-off rows [bands][med_hpitch], indexed by wavefront level
    int med_hpitch;
    float* med_sink;      // banded median: 16-byte store sink per lane of every wave (bands + speculative copies)
    int med_spec_off;     // > 0: the banded median runs in its chained form (a speculative seam failed; counts down per Match)
    int med_spec_last;    // the last banded launch used speculative bands
    int med_seg_last;     // ... and this many column segments per band link
    int med_seg_off;      // > 0: whole rows (a seam failed while column segments were on; counts down per Match)
    int med_spec_fails;   // how often adc_wait had to redo the median because a speculative seam differed
    int force_median_fallback; // test hook (ADC_DEBUG_FORCE_MEDIAN_FALLBACK via adc_debug_run): adc_wait takes the fallback path
    int median_fallbacks;      // how often adc_wait had to redo the median
    int32_t* pin_flags;   // pinned host word: error flag of the banded median's hand-off (read back after every Match)
    int bgrx_valid;       // bgrx_l holds the packed left image of the current pair (written by the arms stage)
    uint32_t* bgrx_l;     // left image packed B | G<<8 | R<<16 per pixel (interpolation gathers)
    uint16_t* st16;      // region voting: 16-bit state map [H][st16_pitch] {bin:11 | iteration of the fill:3 | list:2} (irv_plan.h)
    int st16_pitch;      // row pitch of st16 in elements (multiple of 8: rows start 16-byte aligned)
    float* disp_vote;    // the map the voting chain works on (copy of the LR-checked map, copied back when the chain ends)
    int irv_xcd_mode;    // 1: the chain's work list uses the band -> XCD sweep layout (the device's workgroup -> XCD mapping was probed)
    int irv_grid;        // workgroups of the voting chain (adc_irv_grid, fixed per handle: the work-list layout depends on it)
    int irv_budget;      // kernels the next Match enqueues for the voting chain (adapted from the last Matches)
    int irv_used_hist[8]; // kernels the last 8 Matches needed
    unsigned irv_used_pos;
    int irv_chain;       // kernels of the chain enqueued so far (continuation starts here)
    int irv_pending;     // a chain was enqueued and its final state has not been looked at yet
    int irv_overflows;   // how often adc_wait had to continue the chain (budget too small)
    float *tail_disp_l, *tail_disp_tmp; // buffer roles at the start of the stages behind the voting (for a redo)
    void* irv_cold;                  // device block of the chain's rarely used kernel arguments (64 bytes)
    unsigned char irv_cold_host[64]; // the block as it was last sent (k_voting.hip: IrvCold; staged through pin_flags[32..63])
    int irv_cold_valid, irv_cold_flip;
    int32_t* vote_counters; // voting chain control block (state slots + accumulator ring), median progress words at [160..]
    uint32_t* irv_px;    // per-pixel change bitmap of the voting rounds, IRV_PX_PLANES planes (slack budgets, irv_plan.h)
    uint8_t* chg_a;      // change-tile map of the voting rounds: one byte stamp per 8x8 tile, row pitch chg_pitch
    int chg_pitch;
    uint8_t* edge;       // discontinuity adjustment edge mask
    uint8_t* itp_cells;  // interpolation: 3 byte maps of 2x2-pixel cells (cell has a valid pixel / row distance / Chebyshev distance
                         // in cells to the nearest cell with a valid pixel): empty-space skipping of the ray walk (k_refine.hip);
                         // behind them the padded code map of the walk (adc_device_fn.h: ADC_ITP_VALID ...)
    // pinned staging for adc_match / adc_match_async
    uint8_t* pin_in;  // 2 * 3*W*H
    float* pin_out;   // W*H
    float* async_dst;
    int async_dst_direct; // 0 = via pin_out + host copy, 1 = DMA into the caller's page-locked map, 2 = pageable copy by adc_wait (ADC_HOST_DIRECT)
    void* device_dst;  // adc_match_device: the caller's device buffer (re-filled by the median fallback)
    // profiling
    int profiling, verbose;
    hipEvent_t ev[ADC_STAGE_COUNT + 1];
    hipEvent_t ev_agg[9];
    float stage_ms[ADC_STAGE_COUNT];
    float agg_pass_ms;
    int agg_launches;
    int agg_passes;    // algorithmic passes those launches covered (a pair launch covers two)
    const char* agg_kernel; // kernel family of the last regular aggregation launch (static string)
    // opt-in paper modes (k_paper.hip; 0 = the reference's behaviour)
    uint32_t paper;
    uint8_t* arms_r;      // arms built on the RIGHT image (ADC_PAPER_RIGHT_ARMS), with its packed pixels and a scratch armmax
    uint32_t* bgrx_r;
    int* armmax_r;
    float* vol_c;         // third volume: accumulator of the averaged scanline paths (ADC_PAPER_SO_SUM)
    bool timings_pending;
    // region voting statistics of the last run
    int64_t vote_rounds, vote_evals;
    // which volume holds the latest result (always vol_a at stage boundaries)
};

// ------------------------------------------------------------------ kernel launchers (one per stage)
// All launch on h->stream, asynchronous, return hipError_t of the launch.
hipError_t adc_launch_gray_census(adc_handle* h);
hipError_t adc_launch_cost(adc_handle* h, float* vol_out);
hipError_t adc_launch_cost_records(adc_handle* h);
int adc_agg_small_L(const adc_handle* h);
hipError_t adc_launch_arms(adc_handle* h); // arms, support counts, colour-difference maps (= _left + _rest)
hipError_t adc_launch_arms_left(adc_handle* h); // what needs only the left image: packed pixels, arms, maxima, support counts
hipError_t adc_launch_sup_counts(adc_handle* h); // support counts + region boxes from the arms in HBM (debug surface)
hipError_t adc_launch_arms_rest(adc_handle* h); // what reads both images: colour-step maps (+ paper mode: right-image arms)
hipError_t adc_launch_aggregate_tail(adc_handle* h); // the dividing H pass adc_launch_aggregate left to the scanline stage, as a launch of its own
hipError_t adc_launch_records(adc_handle* h); // arms + counts -> packed aggregation records
hipError_t adc_launch_aggregate(adc_handle* h, int iterations); // vol_a -> vol_a via vol_b
hipError_t adc_launch_so_classes(adc_handle* h, hipStream_t stream);
size_t adc_so_cls_bytes(int W, int H);
hipError_t adc_launch_scanline(adc_handle* h, int passes);      // vol_a -> vol_a via vol_b (passes=4)
int adc_so_segments(const adc_handle* h, int* warm_out);        // verified segments per row of the next scanline run
bool adc_so_can_fuse_agg(const adc_handle* h);                  // the next scanline run can take over the last aggregation pass
size_t adc_so_seam_bytes(int W, int H, int Dp);
hipError_t adc_launch_wta(adc_handle* h);                       // vol_a -> disp_l, disp_r
hipError_t adc_launch_wta_left(adc_handle* h);                  // vol_a -> disp_l only (debug form of the left view)
hipError_t adc_paper_aggregate(adc_handle* h, int iterations);  // k_paper.hip: aggregation limited by both images' arms
hipError_t adc_paper_accumulate(adc_handle* h, float* acc, const float* src, int first, int last);
hipError_t adc_launch_lrcheck(adc_handle* h);
size_t adc_itp_cell_bytes(int W, int H, int ms);
#define ADC_MEDB_MAX_SEG 12                    // column segments per band link of the median, at most (k_refine.hip; sizes the hand-off / sink / seam buffers)
size_t adc_median_hand_rows(int H);             // hand-off rows / store-sink blocks of the banded median (k_refine.hip)       // byte maps of the interpolation's empty-space skipping (k_refine.hip)
int adc_irv_probe_xcd_mode(int device);         // 1 iff workgroup g of a launch runs on XCD g % 8 on this device (probed once)
int adc_irv_grid(size_t pixels);                // workgroups of the voting chain for an image of this size
size_t adc_irv_waves(int grid);                // ints of the chain's statistics block (per-wave counters + per-workgroup segment lengths)
size_t adc_irv_px_words(int W, int H);          // dwords of the per-pixel change bitmap (all planes)
size_t adc_irv_list_entries(int W, int H, int D, int grid);     // capacity of the voting work list (one segment per workgroup)
hipError_t adc_run_region_voting(adc_handle* h); // enqueue only (device-driven chain with a launch budget)
hipError_t adc_voting_finish(adc_handle* h, int* continued); // after a sync: continue the chain if the budget was too small
hipError_t adc_launch_interpolation(adc_handle* h);
hipError_t adc_launch_discontinuity(adc_handle* h);
hipError_t adc_launch_median(adc_handle* h);
hipError_t adc_median_fallback(adc_handle* h); // after a hand-off time-out of the banded filter (adc_wait)
// layout helpers for the debug surface
hipError_t adc_launch_pad_volume(adc_handle* h, const float* src_HWD, float* dst_HWDp);
hipError_t adc_launch_unpad_volume(adc_handle* h, const float* src_HWDp, float* dst_HWD);
