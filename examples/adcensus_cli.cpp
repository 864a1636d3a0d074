// adcensus_cli.cpp -- counterpart of the reference's demo (main.cpp:34-145) without OpenCV:
//   adcensus_cli left.{png,ppm} right.{png,ppm} [min_disparity] [max_disparity] [out_prefix]
// loads the pair (8-bit PNG: gray / RGB / palette / RGBA, non-interlaced; or binary PPM), runs
// ADCensusStereo::Initialize / Match exactly like main.cpp:80-118 and writes what SaveDisparityMap /
// SaveDisparityCloud write (main.cpp:120-128,180-230):
//   <out>-d.png      min-max normalised 8-bit disparity: uchar((|d| - min) / (max - min) * 255), invalid -> 0
//   <out>-c.png      cv::applyColorMap(<out>-d, COLORMAP_JET)
//   <out>-cloud.txt  "x y |d| r g b" per valid pixel ("%f %f %f %d %d %d", colours of the LEFT image)
// plus <out>.pfm (raw float32 disparity, for bit-exact comparisons).  PNG coding uses zlib (the image has no OpenCV).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ADCensusStereo.h"
#include "adc_image_io.h"

int main(int argc, char** argv)
{
    // file-format helpers that need no GPU (used by the CPU test tier):
    //   --convert in.{png,ppm} out.png       decode + re-encode (R,G,B)
    //   --colormap in-d.png out-c.png        the JET mapping SaveDisparityMap applies to the grey disparity image
    if (argc == 4 && (!strcmp(argv[1], "--convert") || !strcmp(argv[1], "--colormap"))) {
        std::vector<uint8> bgr;
        int cw = 0, chh = 0;
        if (!load_image(argv[2], bgr, cw, chh)) { printf("cannot read %s\n", argv[2]); return -1; }
        std::vector<uint8> rgb(bgr.size());
        for (size_t i = 0; i < (size_t)cw * chh; i++) {
            if (!strcmp(argv[1], "--convert")) { rgb[3 * i] = bgr[3 * i + 2]; rgb[3 * i + 1] = bgr[3 * i + 1]; rgb[3 * i + 2] = bgr[3 * i]; }
            else { const uint8 g = bgr[3 * i + 1]; rgb[3 * i] = kJet[g][0]; rgb[3 * i + 1] = kJet[g][1]; rgb[3 * i + 2] = kJet[g][2]; }
        }
        return write_png(argv[3], rgb.data(), cw, chh, 3) ? 0 : -1;
    }
    if (argc < 3) {
        printf("usage: %s left.png right.png [min_disparity] [max_disparity] [out_prefix]\n", argv[0]);
        return -1;
    }
    printf("Image Loading...");
    std::vector<uint8> left, right;
    int w = 0, h = 0, w2 = 0, h2 = 0;
    if (!load_image(argv[1], left, w, h) || !load_image(argv[2], right, w2, h2)) {
        printf("cannot read the image pair (8-bit PNG or binary PPM)\n"); // main.cpp:50-53
        return -1;
    }
    if (w != w2 || h != h2) {
        printf("the two images differ in size\n"); // main.cpp:54-57
        return -1;
    }
    printf("Done!\n");
    ADCensusOption ad_option;                               // main.cpp:80-92
    ad_option.min_disparity = argc < 4 ? 0 : atoi(argv[3]);
    ad_option.max_disparity = argc < 5 ? 64 : atoi(argv[4]);
    ad_option.lrcheck_thres = 1.0f;
    ad_option.do_lr_check = true;
    ad_option.do_filling = true;
    std::string out = argc < 6 ? std::string(argv[1]) : std::string(argv[5]);
    if (argc < 6 && out.size() > 4 && out[out.size() - 4] == '.') out.resize(out.size() - 4);
    printf("w = %d, h = %d, d = [%d,%d]\n\n", w, h, ad_option.min_disparity, ad_option.max_disparity);

    ADCensusStereo ad_census;
    ad_census.SetVerbose(true);
    printf("AD-Census Initializing...\n");
    auto t0 = std::chrono::steady_clock::now();
    if (!ad_census.Initialize(w, h, ad_option)) { printf("AD-Census initialisation failed: %s\n", ad_census.LastError()); return -2; }
    auto t1 = std::chrono::steady_clock::now();
    printf("AD-Census Initializing Done! Timing :	%lf s\n\n", std::chrono::duration<double>(t1 - t0).count());
    printf("AD-Census Matching...\n");
    std::vector<float32> disparity((size_t)w * h, 0.0f);
    t0 = std::chrono::steady_clock::now();
    if (!ad_census.Match(left.data(), right.data(), disparity.data())) { printf("AD-Census matching failed: %s\n", ad_census.LastError()); return -2; }
    t1 = std::chrono::steady_clock::now();
    printf("\nAD-Census Matching...Done! Timing :	%lf s\n", std::chrono::duration<double>(t1 - t0).count());

    // SaveDisparityMap (main.cpp:180-206): min-max over the valid |d|, uchar((|d| - min) / (max - min) * 255), invalid -> 0
    float32 mn = float32(w), mx = -float32(w);
    for (float32 d : disparity) { const float32 a = fabsf(d); if (a != Invalid_Float) { mn = a < mn ? a : mn; mx = a > mx ? a : mx; } }
    std::vector<uint8> gray((size_t)w * h, 0), col((size_t)w * h * 3, 0);
    for (size_t i = 0; i < gray.size(); i++) {
        const float32 a = fabsf(disparity[i]);
        // (a constant map has max == min: the reference divides 0 by 0 there, main.cpp:196 -- written as 0 instead of casting a NaN)
        gray[i] = (a == Invalid_Float || !(mx > mn)) ? 0 : static_cast<uint8>((a - mn) / (mx - mn) * 255);
        col[3 * i] = kJet[gray[i]][0]; col[3 * i + 1] = kJet[gray[i]][1]; col[3 * i + 2] = kJet[gray[i]][2];
    }
    if (!write_png(out + "-d.png", gray.data(), w, h, 1) || !write_png(out + "-c.png", col.data(), w, h, 3)) printf("cannot write %s-d.png / -c.png\n", out.c_str());
    // SaveDisparityCloud (main.cpp:212-230): x y |d| r g b, colours of the left image (stored B,G,R)
    FILE* f = fopen((out + "-cloud.txt").c_str(), "w");
    if (f) {
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++) {
                const float32 a = fabsf(disparity[(size_t)i * w + j]);
                if (a == Invalid_Float) continue;
                const uint8* p = &left[((size_t)i * w + j) * 3];
                fprintf(f, "%f %f %f %d %d %d\n", float32(j), float32(i), a, p[2], p[1], p[0]);
            }
        fclose(f);
    }
    f = fopen((out + ".pfm").c_str(), "wb");
    if (f) { fprintf(f, "Pf\n%d %d\n-1.0\n", w, h); for (int y = h - 1; y >= 0; y--) fwrite(&disparity[(size_t)y * w], 4, w, f); fclose(f); }
    return 0;
}
