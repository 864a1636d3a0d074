/* oracle/sanitize_main.c -- TEST INFRASTRUCTURE ONLY: a small structured pair through adc_oracle_run (all stages dumped) and
 * adc_oracle_match of whichever oracle it is linked with, for the ASAN / UBSAN builds (`make -C oracle asan`): the port
 * (adcensus_port.c) and, where /root/reference exists, the reference's own sources behind ref_driver.cpp.  Exit code 0 and no
 * sanitizer report = clean on this input (min_disparity = 0; the reference's documented out-of-bounds read for
 * min_disparity > 0, ADCensusStereo.cpp:296-300, is not provoked). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_abi.h"

int main(void)
{
    const int W = 96, H = 64, D = 24;
    uint8_t* l = (uint8_t*)malloc((size_t)W * H * 3);
    uint8_t* r = (uint8_t*)malloc((size_t)W * H * 3);
    unsigned s = 12345u;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W + D; x++) { /* a textured strip; the left view is the right one shifted by a piecewise-constant disparity */
            s = s * 1664525u + 1013904223u;
            const int base = 128 + (int)(60.0 * ((x / 9 + y / 7) % 3 - 1)) + (int)((s >> 24) % 17) - 8;
            for (int c = 0; c < 3; c++) {
                const int v = base + 9 * c;
                const uint8_t px = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
                if (x >= D && x - D < W) r[((size_t)y * W + (x - D)) * 3 + c] = px;
                const int d = 4 + ((y / 16) % 2) * 6, xl = x - D + d;
                if (xl >= 0 && xl < W) l[((size_t)y * W + xl) * 3 + c] = px;
            }
        }
    for (int y = 0; y < H; y++) for (int x = 0; x < 12; x++) for (int c = 0; c < 3; c++) l[((size_t)y * W + x) * 3 + c] = (uint8_t)(40 + x);
    adc_option opt;
    memset(&opt, 0, sizeof(opt));
    opt.min_disparity = 0; opt.max_disparity = D; opt.lambda_ad = 10.0f; opt.lambda_census = 30.0f; opt.cross_L1 = 34; opt.cross_L2 = 17;
    opt.cross_t1 = 20; opt.cross_t2 = 6; opt.so_p1 = 1.0f; opt.so_p2 = 3.0f; opt.so_tso = 15; opt.irv_ts = 20; opt.irv_th = 0.4f;
    opt.lrcheck_thres = 1.0f; opt.do_lr_check = 1; opt.do_filling = 1; opt.do_discontinuity_adjustment = 1;
    const size_t P = (size_t)W * H;
    adc_oracle_dump d;
    memset(&d, 0, sizeof(d));
    d.gray_left = malloc(P); d.gray_right = malloc(P); d.census_left = malloc(P * 8); d.census_right = malloc(P * 8);
    d.cost_init = malloc(P * D * 4); d.arms = malloc(P * 4); d.sup_count_h = malloc(P * 2); d.sup_count_v = malloc(P * 2);
    d.cost_aggr = malloc(P * D * 4); d.cost_so = malloc(P * D * 4); d.disp_left_wta = malloc(P * 4); d.disp_right_wta = malloc(P * 4);
    d.outlier_label = malloc(P); d.disp_after_lr = malloc(P * 4); d.disp_after_irv = malloc(P * 4); d.disp_after_interp = malloc(P * 4);
    d.disp_after_dda = malloc(P * 4); d.disp_final = malloc(P * 4);
    if (adc_oracle_run(W, H, &opt, l, r, &d) != 0) { printf("adc_oracle_run failed\n"); return 2; }
    float* m = malloc(P * 4);
    double secs = 0;
    if (adc_oracle_match(W, H, &opt, l, r, m, &secs) != 0) { printf("adc_oracle_match failed\n"); return 2; }
    const int same = memcmp(m, d.disp_final, P * 4) == 0;
    printf("%s oracle under the sanitizers: run + match ok, match == staged run: %s\n", adc_oracle_kind(), same ? "yes" : "NO");
    free(l); free(r); free(m);
    free(d.gray_left); free(d.gray_right); free(d.census_left); free(d.census_right); free(d.cost_init); free(d.arms); free(d.sup_count_h);
    free(d.sup_count_v); free(d.cost_aggr); free(d.cost_so); free(d.disp_left_wta); free(d.disp_right_wta); free(d.outlier_label);
    free(d.disp_after_lr); free(d.disp_after_irv); free(d.disp_after_interp); free(d.disp_after_dda); free(d.disp_final);
    return same ? 0 : 3;
}
