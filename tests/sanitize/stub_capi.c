/* Stub of the C ABI (include/adcensus_c_api.h) for the SANITIZER builds of the host C++ layer (facade + CLI): no HIP, no
 * device.  adc_match fills a synthetic disparity map (a ramp with invalid holes and negative values) so that every writer of
 * examples/adcensus_cli.cpp -- the PNG encoder, the JET mapping, the point cloud, the PFM -- runs under ASAN / UBSAN on a real
 * map shape.  Test infrastructure (adcensus_amd/host/Makefile: `make asan`), never part of the product. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "adcensus_c_api.h"

struct adc_handle { int w, h; adc_option opt; };
static const char* g_err = "";

void adc_option_default(adc_option* o)
{
    memset(o, 0, sizeof(*o));
    o->max_disparity = 64; o->lambda_ad = 10.0f; o->lambda_census = 30.0f; o->cross_L1 = 34; o->cross_L2 = 17; o->cross_t1 = 20; o->cross_t2 = 6;
    o->so_p1 = 1.0f; o->so_p2 = 3.0f; o->so_tso = 15; o->irv_ts = 20; o->irv_th = 0.4f; o->lrcheck_thres = 1.0f;
    o->do_lr_check = 1; o->do_filling = 1;
}
adc_handle* adc_create(int32_t w, int32_t h, const adc_option* opt, int device)
{
    (void)device;
    if (w <= 0 || h <= 0 || !opt || opt->max_disparity - opt->min_disparity <= 0) { g_err = "adc_create: bad geometry"; return NULL; }
    adc_handle* p = (adc_handle*)malloc(sizeof(adc_handle));
    if (p) { p->w = w; p->h = h; p->opt = *opt; }
    return p;
}
void adc_destroy(adc_handle* h) { free(h); }
int adc_match(adc_handle* h, const uint8_t* l, const uint8_t* r, float* d)
{
    if (!h || !l || !r || !d) return 1;
    const float inv = INFINITY;
    for (int y = 0; y < h->h; y++)
        for (int x = 0; x < h->w; x++) {
            const size_t i = (size_t)y * h->w + x;
            const int v = (l[3 * i] + r[3 * i + 1] + x + 2 * y) % 97;
            d[i] = v == 0 ? inv : (v % 13 == 0 ? -(float)v * 0.5f : (float)h->opt.min_disparity + (float)v * 0.37f);
        }
    return 0;
}
int adc_match_async(adc_handle* h, const uint8_t* l, const uint8_t* r, float* d) { return adc_match(h, l, r, d); }
int adc_wait(adc_handle* h) { return h ? 0 : 1; }
void adc_set_profiling(adc_handle* h, int on) { (void)h; (void)on; }
void adc_set_verbose(adc_handle* h, int on) { (void)h; (void)on; }
int adc_set_paper_modes(adc_handle* h, uint32_t m) { (void)h; (void)m; return 0; }
int adc_get_stage_ms(adc_handle* h, float* ms, int n) { if (!h || !ms) return 1; for (int i = 0; i < n; i++) ms[i] = 0.f; return 0; }
const char* adc_last_error(void) { return g_err; }
