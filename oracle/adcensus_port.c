/*
 * adcensus_port.c -- TEST INFRASTRUCTURE ONLY (kind "port").
 *
 * Plain-C, single-threaded CPU restatement of the reference's ADCensusStereo::Match path, written
 * from the algorithm description in SURVEY.md Appendix A.  Every function cites the reference
 * file:line it follows (paths relative to AD-Census/ in the reference checkout).  It exists so
 * that parity tests have an oracle on machines where /root/reference is absent, and it is itself
 * pinned bit-for-bit against the real reference build (oracle/_ref) by tests/test_oracle.py and by
 * the committed SHA-256 goldens in tests/golden/ (generated from oracle/_ref).
 *
 * Arithmetic rules: f32 where the reference uses float32, no FMA contraction
 * (compiled with -ffp-contract=off), glibc expf/sin/cos/lroundf/lround, sequential accumulation
 * order exactly as the reference loops.
 *
 * Nothing under adcensus_amd/ or include/ may link or call this file.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "oracle_abi.h"

#define LARGE_FLOAT 99999.0f          /* adcensus_types.h:35 */
#define INVALID_FLOAT ((float)INFINITY) /* adcensus_types.h:33 */

typedef struct {
    int w, h, dmin, dmax, D;
    adc_option opt;
    const uint8_t *left, *right;
    uint8_t *gray_l, *gray_r;
    uint64_t *census_l, *census_r;
    float *cost_init, *cost_aggr;
    uint8_t* arms; /* [P][4] left,right,top,bottom */
    uint16_t *sup_h, *sup_v, *sup_tmp;
    float *tmp0, *tmp1;
    float *disp_l, *disp_r;
    uint8_t* label;  /* 0 valid, 1 mismatch, 2 occlusion */
    int32_t *mis, *occ; /* raster-ordered pixel indices */
    size_t n_mis, n_occ;
    /* opt-in PAPER modes (features of the AD-Census paper the reference declares or stores but does not implement; NOT
     * the reference's behaviour -- oracle of the product's adc_set_paper_modes only): see adc_oracle_run_paper below */
    uint32_t paper;
    uint8_t* arms_r; /* [P][4] arms built on the RIGHT image (ADC_PAPER_RIGHT_ARMS) */
} port_ctx;

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static float fminf2(float a, float b) { return b < a ? b : a; } /* std::min(a,b): b<a ? b : a */

/* ---------------------------------------------------------------- gray (cost_computor.cpp:58-73) */
static void port_gray(const uint8_t* bgr, uint8_t* gray, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t b = bgr[(size_t)y * w * 3 + 3 * x];
            const uint8_t g = bgr[(size_t)y * w * 3 + 3 * x + 1];
            const uint8_t r = bgr[(size_t)y * w * 3 + 3 * x + 2];
            /* operands promote to double; left-to-right; truncation */
            gray[(size_t)y * w + x] = (uint8_t)(r * 0.299 + g * 0.587 + b * 0.114);
        }
}

/* ------------------------------------------------------- census 9x7 (adcensus_util.cpp:10-39) */
static void port_census(const uint8_t* gray, uint64_t* census, int w, int h)
{
    if (w <= 9 || h <= 7) return; /* :12 -- whole transform skipped, census stays 0 */
    for (int i = 4; i < h - 4; i++)
        for (int j = 3; j < w - 3; j++) {
            const uint8_t c = gray[(size_t)i * w + j];
            uint64_t v = 0;
            for (int r = -4; r <= 4; r++)
                for (int cc = -3; cc <= 3; cc++) {
                    v <<= 1;
                    if (gray[(size_t)(i + r) * w + j + cc] < c) v += 1;
                }
            census[(size_t)i * w + j] = v;
        }
}

/* PAPER mode: 5x5 census (adcensus_types.h:39-42 declares CensusSize::Census5x5, nothing implements it): the 9x7 transform
 * restated for a 5x5 window -- 25 bits, MSB first, centre bit included (= 0), interior pixels only, whole transform
 * skipped for images not larger than the window (same conventions as adcensus_util.cpp:10-39). */
static void port_census5x5(const uint8_t* gray, uint64_t* census, int w, int h)
{
    if (w <= 5 || h <= 5) return;
    for (int i = 2; i < h - 2; i++)
        for (int j = 2; j < w - 2; j++) {
            const uint8_t c = gray[(size_t)i * w + j];
            uint64_t v = 0;
            for (int r = -2; r <= 2; r++)
                for (int cc = -2; cc <= 2; cc++) {
                    v <<= 1;
                    if (gray[(size_t)(i + r) * w + j + cc] < c) v += 1;
                }
            census[(size_t)i * w + j] = v;
        }
}

/* ------------------------------------------------------------ Hamming64 (adcensus_util.cpp:42-53) */
static int port_hamming(uint64_t a, uint64_t b)
{
    uint64_t v = a ^ b;
    int n = 0;
    while (v) {
        ++n;
        v &= v - 1;
    }
    return n;
}

/* ---------------------------------------------------------- cost (cost_computor.cpp:82-121) */
static void port_cost(port_ctx* c)
{
    const int w = c->w, h = c->h, D = c->D;
    const int lambda_ad = c->opt.lambda_ad, lambda_census = c->opt.lambda_census;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* pl = c->left + (size_t)y * w * 3 + 3 * x;
            const uint64_t cl = c->census_l[(size_t)y * w + x];
            float* out = c->cost_init + ((size_t)y * w + x) * D;
            for (int d = c->dmin; d < c->dmax; d++) {
                const int xr = x - d;
                if (xr < 0 || xr >= w) { /* :101-104 */
                    out[d - c->dmin] = 1.0f;
                    continue;
                }
                const uint8_t* pr = c->right + (size_t)y * w * 3 + 3 * xr;
                const float cost_ad = (float)(abs(pl[0] - pr[0]) + abs(pl[1] - pr[1]) + abs(pl[2] - pr[2])) / 3.0f; /* :110 */
                const float cost_census = (float)port_hamming(cl, c->census_r[(size_t)y * w + xr]);                 /* :113-114 */
                /* :117  1 - exp(-cost_ad/lambda_ad) + 1 - exp(-cost_census/lambda_census), all f32 */
                const float ea = expf(-cost_ad / (float)lambda_ad);
                const float ec = expf(-cost_census / (float)lambda_census);
                out[d - c->dmin] = ((1.0f - ea) + 1.0f) - ec;
            }
        }
}

/* ------------------------------------------------ arms (cross_aggregator.cpp:135-269, h:78-80) */
static int color_dist_max(const uint8_t* a, const uint8_t* b)
{
    return imax(abs(a[2] - b[2]), imax(abs(a[1] - b[1]), abs(a[0] - b[0])));
}

/* One arm: walk from (x,y) in direction (dx,dy); returns its length. */
static uint8_t port_arm_img(const port_ctx* c, const uint8_t* img, int x, int y, int dx, int dy);
static uint8_t port_arm(const port_ctx* c, int x, int y, int dx, int dy) { return port_arm_img(c, c->left, x, y, dx, dy); }
static uint8_t port_arm_img(const port_ctx* c, const uint8_t* img, int x, int y, int dx, int dy)
{
    const int w = c->w, h = c->h;
    const int L1 = c->opt.cross_L1, L2 = c->opt.cross_L2, t1 = c->opt.cross_t1, t2 = c->opt.cross_t2;
    const uint8_t* p0 = img + ((size_t)y * w + x) * 3;
    const uint8_t* last = p0;
    uint8_t len = 0;
    int xn = x + dx, yn = y + dy;
    for (int n = 0; n < imin(L1, 255); n++) { /* MAX_ARM_LENGTH 255, cross_aggregator.h:22 */
        if (xn < 0 || xn >= w || yn < 0 || yn >= h) break; /* :154-163 */
        const uint8_t* p = img + ((size_t)yn * w + xn) * 3;
        const int d1 = color_dist_max(p, p0);
        if (d1 >= t1) break;                                  /* :169-172 */
        if (n > 0 && color_dist_max(p, last) >= t1) break;    /* :175-180 */
        if (n + 1 > L2 && d1 >= t2) break;                    /* :183-187 */
        len++;
        last = p;
        xn += dx;
        yn += dy;
    }
    return len;
}

static void port_build_arms(port_ctx* c) /* cross_aggregator.cpp:76-86 */
{
    for (int y = 0; y < c->h; y++)
        for (int x = 0; x < c->w; x++) {
            uint8_t* a = c->arms + ((size_t)y * c->w + x) * 4;
            a[0] = port_arm(c, x, y, -1, 0);
            a[1] = port_arm(c, x, y, +1, 0);
            a[2] = port_arm(c, x, y, 0, -1);
            a[3] = port_arm(c, x, y, 0, +1);
        }
}

/* ------------------------------------------- support counts (cross_aggregator.cpp:271-325) */
static void port_sup_counts(port_ctx* c)
{
    const int w = c->w, h = c->h;
    for (int id = 0; id < 2; id++) {
        uint16_t* dst = id == 0 ? c->sup_h : c->sup_v;
        for (int k = 0; k < 2; k++)
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    const uint8_t* a = c->arms + ((size_t)y * w + x) * 4;
                    int count = 0;
                    if (id == 0) {
                        if (k == 0) count = a[0] + a[1] + 1;
                        else for (int t = -a[2]; t <= a[3]; t++) count += c->sup_tmp[(size_t)(y + t) * w + x];
                    } else {
                        if (k == 0) count = a[2] + a[3] + 1;
                        else for (int t = -a[0]; t <= a[1]; t++) count += c->sup_tmp[(size_t)y * w + x + t];
                    }
                    if (k == 0) c->sup_tmp[(size_t)y * w + x] = (uint16_t)count;
                    else dst[(size_t)y * w + x] = (uint16_t)count;
                }
    }
}

/* -------------------------------------------- aggregation (cross_aggregator.cpp:89-118,327-394) */
static void port_aggregate_plane(port_ctx* c, int di, int horizontal_first)
{
    const int w = c->w, h = c->h, D = c->D;
    const size_t P = (size_t)w * h;
    for (size_t p = 0; p < P; p++) c->tmp0[p] = c->cost_aggr[p * D + di]; /* :342-346 */
    const uint16_t* cnt = horizontal_first ? c->sup_h : c->sup_v;
    for (int k = 0; k < 2; k++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                const uint8_t* a = c->arms + ((size_t)y * w + x) * 4;
                const float* src = k == 0 ? c->tmp0 : c->tmp1;
                const int horizontal = horizontal_first ? (k == 0) : (k == 1);
                float cost = 0.0f; /* accumulated from 0 in order t=-arm..+arm (:358-383) */
                if (horizontal) for (int t = -a[0]; t <= a[1]; t++) cost += src[(size_t)y * w + x + t];
                else for (int t = -a[2]; t <= a[3]; t++) cost += src[(size_t)(y + t) * w + x];
                if (k == 0) c->tmp1[(size_t)y * w + x] = cost;
                else c->cost_aggr[((size_t)y * w + x) * D + di] = cost / (float)cnt[(size_t)y * w + x]; /* :389 */
            }
}

/* PAPER mode: the support region of (p, d) is limited by BOTH images (the reference stores img_right_ for this,
 * cross_aggregator.h:91, and never reads it): arm(p, d) = min(left arm at p, right arm at (x - d, y)) per direction when
 * the right pixel lies in the image, else the left arm.  Same two-pass ordered sums; the divisor is the number of cost
 * values that contributed = sum over the second pass's span of the first pass's span lengths (as float). */
static void port_arms_at(const port_ctx* c, int x, int y, int d, int a[4])
{
    const uint8_t* al = c->arms + ((size_t)y * c->w + x) * 4;
    const int xr = x - d;
    for (int k = 0; k < 4; k++) a[k] = al[k];
    if (xr >= 0 && xr < c->w) {
        const uint8_t* ar = c->arms_r + ((size_t)y * c->w + xr) * 4;
        for (int k = 0; k < 4; k++) a[k] = imin(a[k], ar[k]);
    }
}
static void port_aggregate_plane_rarms(port_ctx* c, int di, int horizontal_first)
{
    const int w = c->w, h = c->h, D = c->D, d = di + c->dmin;
    const size_t P = (size_t)w * h;
    for (size_t p = 0; p < P; p++) c->tmp0[p] = c->cost_aggr[p * D + di];
    for (int k = 0; k < 2; k++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int a[4];
                port_arms_at(c, x, y, d, a);
                const float* src = k == 0 ? c->tmp0 : c->tmp1;
                const int horizontal = horizontal_first ? (k == 0) : (k == 1);
                float cost = 0.0f;
                int cnt = 0;
                if (horizontal) {
                    for (int t = -a[0]; t <= a[1]; t++) {
                        cost += src[(size_t)y * w + x + t];
                        if (k == 1) { int b[4]; port_arms_at(c, x + t, y, d, b); cnt += b[2] + b[3] + 1; }
                    }
                } else {
                    for (int t = -a[2]; t <= a[3]; t++) {
                        cost += src[(size_t)(y + t) * w + x];
                        if (k == 1) { int b[4]; port_arms_at(c, x, y + t, d, b); cnt += b[0] + b[1] + 1; }
                    }
                }
                if (k == 0) c->tmp1[(size_t)y * w + x] = cost;
                else c->cost_aggr[((size_t)y * w + x) * D + di] = cost / (float)cnt;
            }
}

static void port_aggregate(port_ctx* c, int iters)
{
    const size_t n = (size_t)c->w * c->h * c->D;
    port_build_arms(c);
    port_sup_counts(c);
    if (c->paper & 4u) {
        for (int y = 0; y < c->h; y++)
            for (int x = 0; x < c->w; x++) {
                uint8_t* a = c->arms_r + ((size_t)y * c->w + x) * 4;
                a[0] = port_arm_img(c, c->right, x, y, -1, 0);
                a[1] = port_arm_img(c, c->right, x, y, +1, 0);
                a[2] = port_arm_img(c, c->right, x, y, 0, -1);
                a[3] = port_arm_img(c, c->right, x, y, 0, +1);
            }
    }
    memcpy(c->cost_aggr, c->cost_init, n * sizeof(float)); /* :108 */
    int horizontal_first = 1;
    for (int k = 0; k < iters; k++) {
        for (int di = 0; di < c->D; di++) {
            if (c->paper & 4u) port_aggregate_plane_rarms(c, di, horizontal_first);
            else port_aggregate_plane(c, di, horizontal_first);
        }
        horizontal_first = !horizontal_first;
    }
}

/* ------------------------------------------------ scanline (scanline_optimizer.cpp:63-279) */
/* One path: npix pixels, pixel k at volume offset vol_off(k), left colour at lcol(k).  For
 * disparity index d the right-image pixel pair compared for d2 is (rpix(k,xr), rprev(k,xr)). */
static void port_so_pass(port_ctx* c, const float* src, float* dst, int vertical, int forward)
{
    const int w = c->w, h = c->h, D = c->D, dmin = c->dmin;
    const float p1 = c->opt.so_p1, p2 = c->opt.so_p2;
    const int tso = c->opt.so_tso;
    const int dir = forward ? 1 : -1;
    const int npaths = vertical ? w : h;
    const int plen = vertical ? h : w;
    float* last = (float*)malloc(sizeof(float) * (D + 2));
    for (int path = 0; path < npaths; path++) {
        int x = vertical ? path : (forward ? 0 : w - 1);
        int y = vertical ? (forward ? 0 : h - 1) : path;
        const int sx = vertical ? 0 : dir, sy = vertical ? dir : 0;
        size_t off = ((size_t)y * w + x) * D;
        const uint8_t* col_last = c->left + ((size_t)y * w + x) * 3;
        for (int i = 0; i < D + 2; i++) last[i] = LARGE_FLOAT; /* :96 */
        memcpy(dst + off, src + off, D * sizeof(float));       /* :99 */
        memcpy(last + 1, dst + off, D * sizeof(float));
        float min_last = LARGE_FLOAT;
        for (int i = 0; i < D + 2; i++) min_last = fminf2(min_last, last[i]); /* :107-110 */
        x += sx;
        y += sy;
        for (int j = 0; j < plen - 1; j++) {
            off = ((size_t)y * w + x) * D;
            const uint8_t* col = c->left + ((size_t)y * w + x) * 3;
            const uint8_t d1 = (uint8_t)color_dist_max(col, col_last); /* :114-115 */
            uint8_t d2 = d1;                                          /* sticky across d (:116) */
            float min_cost = LARGE_FLOAT;
            for (int d = 0; d < D; d++) {
                const int xr = x - d - dmin;
                if (xr > 0 && xr < w - 1) { /* :119, :228 */
                    const uint8_t* r0 = c->right + ((size_t)y * w + xr) * 3;
                    const uint8_t* r1 = vertical ? c->right + ((size_t)(y - dir) * w + xr) * 3
                                                 : c->right + ((size_t)y * w + (xr - dir)) * 3;
                    d2 = (uint8_t)color_dist_max(r0, r1);
                }
                float P1, P2; /* :129-141 */
                if (d1 < tso && d2 < tso) { P1 = p1; P2 = p2; }
                else if (d1 >= tso && d2 >= tso) { P1 = p1 / 10; P2 = p2 / 10; }
                else { P1 = p1 / 4; P2 = p2 / 4; }
                const float cost = src[off + d];
                const float l1 = last[d + 1];
                const float l2 = last[d] + P1;
                const float l3 = last[d + 2] + P1;
                const float l4 = min_last + P2;
                float cost_s = cost + fminf2(fminf2(l1, l2), fminf2(l3, l4)); /* :150 */
                cost_s /= 2;                                                 /* :151 */
                dst[off + d] = cost_s;
                min_cost = fminf2(min_cost, cost_s);
            }
            min_last = min_cost;
            memcpy(last + 1, dst + off, D * sizeof(float));
            col_last = col;
            x += sx;
            y += sy;
        }
    }
    free(last);
}

/* PAPER mode: the four path costs are computed INDEPENDENTLY from the aggregated volume and averaged (AD-Census paper,
 * eq. 10: C2 = 1/4 sum_r C_r), where the reference chains them (scanline_optimizer.cpp:54-60).  Each path keeps the
 * reference's recurrence (incl. its division by 2); sum order L->R, R->L, T->B, B->T, then * 0.25f. */
static void port_scanline_sum(port_ctx* c)
{
    const size_t n = (size_t)c->w * c->h * c->D;
    float* acc = (float*)malloc(n * sizeof(float));
    for (int r = 0; r < 4; r++) {
        port_so_pass(c, c->cost_aggr, c->cost_init, r >= 2, (r & 1) == 0);
        for (size_t i = 0; i < n; i++) acc[i] = r == 0 ? c->cost_init[i] : acc[i] + c->cost_init[i];
    }
    for (size_t i = 0; i < n; i++) c->cost_aggr[i] = acc[i] * 0.25f;
    free(acc);
}

static void port_scanline(port_ctx* c) /* scanline_optimizer.cpp:40-61: chained, ping-pong */
{
    if (c->paper & 2u) { port_scanline_sum(c); return; }
    port_so_pass(c, c->cost_aggr, c->cost_init, 0, 1);
    port_so_pass(c, c->cost_init, c->cost_aggr, 0, 0);
    port_so_pass(c, c->cost_aggr, c->cost_init, 1, 1);
    port_so_pass(c, c->cost_init, c->cost_aggr, 1, 0);
}

/* ---------------------------------------------------------- WTA (ADCensusStereo.cpp:188-310) */
static float port_subpixel(const float* cl, int best, int dmin, float min_cost)
{
    const float c1 = cl[best - 1 - dmin], c2 = cl[best + 1 - dmin];
    const float denom = c1 + c2 - 2 * min_cost; /* :233 */
    if (denom != 0.0f) return (float)best + (c1 - c2) / (denom * 2.0f);
    return (float)best;
}

static void port_wta_left(port_ctx* c)
{
    const int w = c->w, h = c->h, D = c->D, dmin = c->dmin, dmax = c->dmax;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const float* cl = c->cost_aggr + ((size_t)i * w + j) * D;
            float min_cost = LARGE_FLOAT;
            int best = 0;
            for (int d = dmin; d < dmax; d++)
                if (min_cost > cl[d - dmin]) { min_cost = cl[d - dmin]; best = d; } /* strict: lowest d wins */
            if (best == dmin || best == dmax - 1) { c->disp_l[(size_t)i * w + j] = INVALID_FLOAT; continue; } /* :222-225 */
            c->disp_l[(size_t)i * w + j] = port_subpixel(cl, best, dmin, min_cost);
        }
}

static void port_wta_right(port_ctx* c)
{
    const int w = c->w, h = c->h, D = c->D, dmin = c->dmin, dmax = c->dmax;
    float* cl = (float*)calloc(D, sizeof(float)); /* cost_local persists across pixels (:262) */
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            float min_cost = LARGE_FLOAT;
            int best = 0;
            for (int d = dmin; d < dmax; d++) {
                const int col = j + d;
                if (col >= 0 && col < w) { /* cost(xr,yr,d) = cost(xr+d,yl,d) */
                    const float v = cl[d - dmin] = c->cost_aggr[((size_t)i * w + col) * D + (d - dmin)];
                    if (min_cost > v) { min_cost = v; best = d; }
                } else cl[d - dmin] = LARGE_FLOAT; /* :281-283 */
            }
            if (best == dmin || best == dmax - 1) { c->disp_r[(size_t)i * w + j] = (float)best; continue; } /* :290-293 */
            if (best - 1 - dmin < 0 || best + 1 - dmin >= D) { /* only reachable when every candidate is out of the image
                                                                  and dmin != 0 (reference reads out of bounds there) */
                c->disp_r[(size_t)i * w + j] = (float)best;
                continue;
            }
            c->disp_r[(size_t)i * w + j] = port_subpixel(cl, best, dmin, min_cost);
        }
    free(cl);
}

/* ---------------------------------------- outlier detection (multistep_refiner.cpp:90-151) */
static void port_outlier(port_ctx* c)
{
    const int w = c->w, h = c->h;
    const float thres = c->opt.lrcheck_thres;
    c->n_mis = c->n_occ = 0;
    memset(c->label, 0, (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float* disp = &c->disp_l[(size_t)y * w + x];
            const int32_t pix = y * w + x;
            if (*disp == INVALID_FLOAT) { c->mis[c->n_mis++] = pix; continue; }
            const long col_right = lroundf((float)x - *disp); /* :114 (int - float -> float) */
            if (col_right >= 0 && col_right < w) {
                const float disp_r = c->disp_r[(size_t)y * w + col_right];
                if (fabsf(*disp - disp_r) > thres) {
                    const int col_rl = (int)lroundf((float)col_right + disp_r); /* :127 */
                    if (col_rl > 0 && col_rl < w) {
                        const float disp_l = c->disp_l[(size_t)y * w + col_rl]; /* in place: may already be inf */
                        if (disp_l > *disp) c->occ[c->n_occ++] = pix;
                        else c->mis[c->n_mis++] = pix;
                    } else c->mis[c->n_mis++] = pix;
                    *disp = INVALID_FLOAT;
                }
            } else {
                *disp = INVALID_FLOAT;
                c->mis[c->n_mis++] = pix;
            }
        }
    for (size_t i = 0; i < c->n_mis; i++) c->label[c->mis[i]] = 1;
    for (size_t i = 0; i < c->n_occ; i++) c->label[c->occ[i]] = 2;
}

/* ------------------------------------ iterative region voting (multistep_refiner.cpp:153-227) */
static void port_region_voting(port_ctx* c)
{
    const int w = c->w, D = c->D, dmin = c->dmin;
    int32_t* hist = (int32_t*)malloc(sizeof(int32_t) * D);
    for (int it = 0; it < 5; it++)
        for (int k = 0; k < 2; k++) {
            int32_t* list = k == 0 ? c->mis : c->occ;
            size_t* n = k == 0 ? &c->n_mis : &c->n_occ;
            for (size_t i = 0; i < *n; i++) {
                const int x = list[i] % w, y = list[i] / w;
                float* disp = &c->disp_l[list[i]];
                if (*disp != INVALID_FLOAT) continue;
                memset(hist, 0, sizeof(int32_t) * D);
                const uint8_t* arm = c->arms + (size_t)list[i] * 4;
                for (int t = -arm[2]; t <= arm[3]; t++) {
                    const int yt = y + t;
                    const uint8_t* arm2 = c->arms + ((size_t)yt * w + x) * 4;
                    for (int s = -arm2[0]; s <= arm2[1]; s++) {
                        const float d = c->disp_l[(size_t)yt * w + x + s];
                        if (d != INVALID_FLOAT) hist[lroundf(d) - dmin]++; /* :193-196 */
                    }
                }
                int best = 0, count = 0, max_ht = 0;
                for (int d = 0; d < D; d++) {
                    if (max_ht < hist[d]) { max_ht = hist[d]; best = d; }
                    count += hist[d];
                }
                if (max_ht > 0 && count > c->opt.irv_ts && max_ht * 1.0f / count > c->opt.irv_th)
                    *disp = (float)(best + dmin); /* in place: later pixels of this pass see it */
            }
            /* erase filled pixels, order preserved (:217-224) */
            size_t m = 0;
            for (size_t i = 0; i < *n; i++)
                if (c->disp_l[list[i]] == INVALID_FLOAT) list[m++] = list[i];
            *n = m;
        }
    free(hist);
}

/* --------------------------------------- proper interpolation (multistep_refiner.cpp:229-305) */
static void port_interpolation(port_ctx* c)
{
    const int w = c->w, h = c->h;
    const float pi = 3.1415926f;
    const int max_search = imax(abs(c->dmax), abs(c->dmin));
    for (int k = 0; k < 2; k++) {
        const int32_t* list = k == 0 ? c->mis : c->occ;
        const size_t n = k == 0 ? c->n_mis : c->n_occ;
        if (n == 0) continue;
        float* fill = (float*)calloc(n, sizeof(float)); /* value-initialised: no hit -> 0.0f */
        for (size_t i = 0; i < n; i++) {
            const int x = list[i] % w, y = list[i] / w;
            int32_t src_off[16];
            float src_d[16];
            int ncol = 0;
            double ang = 0.0;
            for (int s = 0; s < 16; s++) {
                const double sina = sin(ang), cosa = cos(ang);
                for (int m = 1; m < max_search; m++) {
                    const int yy = (int)lround(y + m * sina);
                    const int xx = (int)lround(x + m * cosa);
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) break;
                    const float d = c->disp_l[(size_t)yy * w + xx];
                    if (d != INVALID_FLOAT) {
                        src_off[ncol] = yy * w * 3 + 3 * xx;
                        src_d[ncol++] = d;
                        break;
                    }
                }
                ang += pi / 16; /* float divide, widened on += */
            }
            if (ncol == 0) continue;
            if (k == 0) { /* mismatch: colour-nearest, first minimum */
                int min_dist = 9999;
                float d = 0.0f;
                const uint8_t* p = c->left + ((size_t)y * w + x) * 3;
                for (int q = 0; q < ncol; q++) {
                    const uint8_t* p2 = c->left + src_off[q];
                    const int dist = abs(p[2] - p2[2]) + abs(p[1] - p2[1]) + abs(p[0] - p2[0]);
                    if (min_dist > dist) { min_dist = dist; d = src_d[q]; }
                }
                fill[i] = d;
            } else { /* occlusion: smallest disparity */
                float md = LARGE_FLOAT;
                for (int q = 0; q < ncol; q++) md = fminf2(md, src_d[q]);
                fill[i] = md;
            }
        }
        for (size_t i = 0; i < n; i++) c->disp_l[list[i]] = fill[i]; /* deferred write-back (:298-303) */
        free(fill);
    }
}

/* --------------------------- discontinuity adjustment (multistep_refiner.cpp:307-371), default off */
static void port_dda(port_ctx* c)
{
    const int w = c->w, h = c->h, D = c->D;
    uint8_t* edge = (uint8_t*)calloc((size_t)w * h, 1);
    const float* dp = c->disp_l;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
#define DP(yy, xx) dp[(size_t)(yy)*w + (xx)]
            const float gx = (-DP(y - 1, x - 1) + DP(y - 1, x + 1)) + (-2 * DP(y, x - 1) + 2 * DP(y, x + 1)) +
                             (-DP(y + 1, x - 1) + DP(y + 1, x + 1));
            const float gy = (-DP(y - 1, x - 1) - 2 * DP(y - 1, x) - DP(y - 1, x + 1)) +
                             (DP(y + 1, x - 1) + 2 * DP(y + 1, x) + DP(y + 1, x + 1));
#undef DP
            if (fabsf(gx) + fabsf(gy) > 5.0f) edge[(size_t)y * w + x] = 1;
        }
    for (int y = 0; y < h; y++)
        for (int x = 1; x < w - 1; x++) {
            if (edge[(size_t)y * w + x] != 1) continue;
            float* row = c->disp_l + (size_t)y * w;
            if (row[x] == INVALID_FLOAT) continue;
            const long di = lroundf(row[x]); /* NOTE: not offset by min_disparity (:329-331), kept */
            const float* cp = c->cost_aggr + ((size_t)y * w + x) * D;
            float c0 = cp[di];
            for (int k = 0; k < 2; k++) {
                const int x2 = k == 0 ? x - 1 : x + 1;
                const float d2 = row[x2];
                if (d2 == INVALID_FLOAT) continue;
                const long d2i = lroundf(d2);
                const float cc = k == 0 ? cp[-D + d2i] : cp[D + d2i];
                if (cc < c0) { row[x] = d2; c0 = cc; }
            }
        }
    free(edge);
}

/* ------------------------------------ 3x3 median, in == out (adcensus_util.cpp:55-81) */
static void port_median3_inplace(float* disp, int w, int h)
{
    float wnd[9];
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int n = 0;
            for (int r = -1; r <= 1; r++)
                for (int cc = -1; cc <= 1; cc++) {
                    const int row = y + r, col = x + cc;
                    if (row >= 0 && row < h && col >= 0 && col < w) wnd[n++] = disp[(size_t)row * w + col];
                }
            for (int i = 1; i < n; i++) { /* insertion sort == std::sort result on values */
                const float v = wnd[i];
                int j = i - 1;
                while (j >= 0 && wnd[j] > v) { wnd[j + 1] = wnd[j]; j--; }
                wnd[j + 1] = v;
            }
            if (n) disp[(size_t)y * w + x] = wnd[n / 2]; /* recursive: already-filtered neighbours are read */
        }
}

/* ------------------------------------------------------------------------------ driver */
static void port_free(port_ctx* c)
{
    free(c->gray_l); free(c->gray_r); free(c->census_l); free(c->census_r);
    free(c->cost_init); free(c->cost_aggr); free(c->arms);
    free(c->sup_h); free(c->sup_v); free(c->sup_tmp); free(c->tmp0); free(c->tmp1);
    free(c->disp_l); free(c->disp_r); free(c->label); free(c->mis); free(c->occ); free(c->arms_r);
}

static int port_init(port_ctx* c, int w, int h, const adc_option* opt)
{
    memset(c, 0, sizeof(*c));
    if (w <= 0 || h <= 0) return 1;                         /* ADCensusStereo.cpp:31-33 */
    if (opt->max_disparity - opt->min_disparity <= 0) return 1; /* :38-40 */
    c->w = w; c->h = h; c->opt = *opt;
    c->dmin = opt->min_disparity; c->dmax = opt->max_disparity; c->D = c->dmax - c->dmin;
    const size_t P = (size_t)w * h;
    c->gray_l = calloc(P, 1); c->gray_r = calloc(P, 1);
    c->census_l = calloc(P, 8); c->census_r = calloc(P, 8); /* zero-filled once (cost_computor.cpp:37-38) */
    c->cost_init = calloc(P * c->D, 4); c->cost_aggr = calloc(P * c->D, 4);
    c->arms = calloc(P, 4); c->arms_r = calloc(P, 4);
    c->sup_h = calloc(P, 2); c->sup_v = calloc(P, 2); c->sup_tmp = calloc(P, 2);
    c->tmp0 = calloc(P, 4); c->tmp1 = calloc(P, 4);
    c->disp_l = calloc(P, 4); c->disp_r = calloc(P, 4);
    c->label = calloc(P, 1); c->mis = calloc(P, 4); c->occ = calloc(P, 4);
    return 0;
}

#define DUMP(dst, src, bytes) do { if (dst) memcpy((dst), (src), (bytes)); } while (0)

static void port_pipeline(port_ctx* c, const uint8_t* left, const uint8_t* right, adc_oracle_dump* out)
{
    const size_t P = (size_t)c->w * c->h, D = (size_t)c->D;
    c->left = left; c->right = right;
    /* ComputeCost (ADCensusStereo.cpp:147-155; cost_computor.cpp:123-137) */
    port_gray(left, c->gray_l, c->w, c->h);
    port_gray(right, c->gray_r, c->w, c->h);
    if (c->paper & 1u) {
        port_census5x5(c->gray_l, c->census_l, c->w, c->h);
        port_census5x5(c->gray_r, c->census_r, c->w, c->h);
    } else {
        port_census(c->gray_l, c->census_l, c->w, c->h);
        port_census(c->gray_r, c->census_r, c->w, c->h);
    }
    port_cost(c);
    if (out) {
        DUMP(out->gray_left, c->gray_l, P); DUMP(out->gray_right, c->gray_r, P);
        DUMP(out->census_left, c->census_l, P * 8); DUMP(out->census_right, c->census_r, P * 8);
        DUMP(out->cost_init, c->cost_init, P * D * 4);
    }
    port_aggregate(c, 4); /* ADCensusStereo.cpp:164 */
    if (out) {
        DUMP(out->arms, c->arms, P * 4);
        DUMP(out->sup_count_h, c->sup_h, P * 2); DUMP(out->sup_count_v, c->sup_v, P * 2);
        DUMP(out->cost_aggr, c->cost_aggr, P * D * 4);
    }
    port_scanline(c);
    if (out) DUMP(out->cost_so, c->cost_aggr, P * D * 4);
    port_wta_left(c);
    port_wta_right(c);
    if (out) { DUMP(out->disp_left_wta, c->disp_l, P * 4); DUMP(out->disp_right_wta, c->disp_r, P * 4); }
    /* MultiStepRefiner::Refine (multistep_refiner.cpp:60-87); do_filling drives voting AND interpolation
     * (ADCensusStereo.cpp:182-183) */
    c->n_mis = c->n_occ = 0;
    memset(c->label, 0, P);
    if (c->opt.do_lr_check) port_outlier(c);
    if (out) { DUMP(out->outlier_label, c->label, P); DUMP(out->disp_after_lr, c->disp_l, P * 4); }
    if (c->opt.do_filling) port_region_voting(c);
    if (out) DUMP(out->disp_after_irv, c->disp_l, P * 4);
    if (c->opt.do_filling) port_interpolation(c);
    if (out) DUMP(out->disp_after_interp, c->disp_l, P * 4);
    if (c->opt.do_discontinuity_adjustment) port_dda(c);
    if (out) DUMP(out->disp_after_dda, c->disp_l, P * 4);
    port_median3_inplace(c->disp_l, c->w, c->h);
    if (out) DUMP(out->disp_final, c->disp_l, P * 4);
}

const char* adc_oracle_kind(void) { return "port"; }
/* compiler + flags this checker was built with (reported next to the CPU baseline by bench.py) */
#ifndef ADC_ORACLE_FLAGS
#define ADC_ORACLE_FLAGS "?"
#endif
const char* adc_oracle_build_info(void) { return "compiler " __VERSION__ ", flags " ADC_ORACLE_FLAGS; }

int adc_oracle_run(int32_t width, int32_t height, const adc_option* opt, const uint8_t* bgr_left,
                   const uint8_t* bgr_right, adc_oracle_dump* dump)
{
    port_ctx c;
    if (port_init(&c, width, height, opt)) return 1;
    if (!bgr_left || !bgr_right) { port_free(&c); return 2; }
    port_pipeline(&c, bgr_left, bgr_right, dump);
    port_free(&c);
    return 0;
}

/* PORT ONLY (the reference has no such modes): the pipeline with the opt-in paper features of adc_set_paper_modes
 * (bit 0: 5x5 census, bit 1: averaged instead of chained scanline paths, bit 2: right-image arms in the aggregation). */
int adc_oracle_run_paper(int32_t width, int32_t height, const adc_option* opt, uint32_t paper_modes, const uint8_t* bgr_left,
                         const uint8_t* bgr_right, adc_oracle_dump* dump)
{
    port_ctx c;
    if (port_init(&c, width, height, opt)) return 1;
    if (!bgr_left || !bgr_right) { port_free(&c); return 2; }
    c.paper = paper_modes;
    port_pipeline(&c, bgr_left, bgr_right, dump);
    port_free(&c);
    return 0;
}

int adc_oracle_match(int32_t width, int32_t height, const adc_option* opt, const uint8_t* bgr_left,
                     const uint8_t* bgr_right, float* disp_left, double* seconds_match)
{
    port_ctx c;
    if (port_init(&c, width, height, opt)) return 1;
    if (!bgr_left || !bgr_right || !disp_left) { port_free(&c); return 2; }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    port_pipeline(&c, bgr_left, bgr_right, NULL);
    memcpy(disp_left, c.disp_l, (size_t)width * height * sizeof(float));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (seconds_match) *seconds_match = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    port_free(&c);
    return 0;
}

void adc_oracle_median3_inplace(float* disp, int32_t width, int32_t height)
{
    port_median3_inplace(disp, width, height);
}
