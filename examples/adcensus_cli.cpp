// adcensus_cli.cpp -- counterpart of the reference's demo (main.cpp:34-145) without OpenCV:
//   adcensus_cli left.ppm right.ppm [dmin] [dmax] [out_prefix]
// reads two binary PPM (P6, 8-bit RGB) images, runs ADCensusStereo::Initialize / Match exactly like
// main.cpp:80-118 and writes
//   <out>-d.pgm   min-max normalised 8-bit disparity (SaveDisparityMap, main.cpp:180-206)
//   <out>-c.ppm   JET colour map of it (cv::applyColorMap(..., COLORMAP_JET), main.cpp:207)
//   <out>.pfm     raw float32 disparity
//   <out>.txt     x y disparity point list (SaveDisparityCloud without the Q-matrix, main.cpp:212-230)
// PNG <-> PPM conversion: tools/png2ppm.py (PIL).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ADCensusStereo.h"

static bool read_ppm(const char* path, std::vector<uint8>& bgr, int& w, int& h)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0};
    int maxv = 0;
    auto skip = [&]() { int c; while ((c = fgetc(f)) != EOF) { if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} } else if (!isspace(c)) { ungetc(c, f); break; } } };
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) { fclose(f); return false; }
    skip(); if (fscanf(f, "%d", &w) != 1) { fclose(f); return false; }
    skip(); if (fscanf(f, "%d", &h) != 1) { fclose(f); return false; }
    skip(); if (fscanf(f, "%d", &maxv) != 1 || maxv != 255) { fclose(f); return false; }
    fgetc(f);
    std::vector<uint8> rgb((size_t)w * h * 3);
    const bool ok = fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
    fclose(f);
    bgr.resize(rgb.size());
    for (size_t i = 0; i < (size_t)w * h; i++) { bgr[3 * i] = rgb[3 * i + 2]; bgr[3 * i + 1] = rgb[3 * i + 1]; bgr[3 * i + 2] = rgb[3 * i]; } // main.cpp:69-74
    return ok;
}

static void jet(uint8 v, uint8 rgb[3])
{
    const float t = v / 255.0f;
    auto ch = [](float x) { x = x < 0 ? 0 : (x > 1 ? 1 : x); return (uint8)lroundf(x * 255.0f); };
    rgb[0] = ch(1.5f - fabsf(4.0f * t - 3.0f));
    rgb[1] = ch(1.5f - fabsf(4.0f * t - 2.0f));
    rgb[2] = ch(1.5f - fabsf(4.0f * t - 1.0f));
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        printf("usage: %s left.ppm right.ppm [min_disparity] [max_disparity] [out_prefix]\n", argv[0]);
        return -1;
    }
    std::vector<uint8> left, right;
    int w = 0, h = 0, w2 = 0, h2 = 0;
    if (!read_ppm(argv[1], left, w, h) || !read_ppm(argv[2], right, w2, h2) || w != w2 || h != h2) {
        printf("cannot read the image pair (binary PPM, equal sizes)\n");
        return -1;
    }
    ADCensusOption ad_option;                               // main.cpp:80-92
    ad_option.min_disparity = argc < 4 ? 0 : atoi(argv[3]);
    ad_option.max_disparity = argc < 5 ? 64 : atoi(argv[4]);
    ad_option.lrcheck_thres = 1.0f;
    ad_option.do_lr_check = true;
    ad_option.do_filling = true;
    const std::string out = argc < 6 ? std::string(argv[1]) : std::string(argv[5]);
    printf("w = %d, h = %d, d = [%d,%d]\n\n", w, h, ad_option.min_disparity, ad_option.max_disparity);

    ADCensusStereo ad_census;
    ad_census.SetVerbose(true);
    auto t0 = std::chrono::steady_clock::now();
    if (!ad_census.Initialize(w, h, ad_option)) { printf("AD-Census initialisation failed: %s\n", ad_census.LastError()); return -2; }
    auto t1 = std::chrono::steady_clock::now();
    printf("AD-Census Initializing Done! Timing :	%lf s\n\n", std::chrono::duration<double>(t1 - t0).count());
    std::vector<float32> disparity((size_t)w * h, 0.0f);
    t0 = std::chrono::steady_clock::now();
    if (!ad_census.Match(left.data(), right.data(), disparity.data())) { printf("AD-Census matching failed: %s\n", ad_census.LastError()); return -2; }
    t1 = std::chrono::steady_clock::now();
    printf("\nAD-Census Matching...Done! Timing :	%lf s\n", std::chrono::duration<double>(t1 - t0).count());

    // SaveDisparityMap (main.cpp:180-206): min-max over valid |d|, uchar((|d|-min)/(max-min)*255)
    float mn = (float)w, mx = -(float)w;
    for (float d : disparity) if (d != Invalid_Float) { const float a = fabsf(d); mn = a < mn ? a : mn; mx = a > mx ? a : mx; }
    std::vector<uint8> gray((size_t)w * h, 0), col((size_t)w * h * 3, 0);
    for (size_t i = 0; i < gray.size(); i++) {
        if (disparity[i] != Invalid_Float && mx > mn) gray[i] = (uint8)((fabsf(disparity[i]) - mn) / (mx - mn) * 255);
        jet(gray[i], &col[3 * i]);
    }
    FILE* f = fopen((out + "-d.pgm").c_str(), "wb");
    if (f) { fprintf(f, "P5\n%d %d\n255\n", w, h); fwrite(gray.data(), 1, gray.size(), f); fclose(f); }
    f = fopen((out + "-c.ppm").c_str(), "wb");
    if (f) { fprintf(f, "P6\n%d %d\n255\n", w, h); fwrite(col.data(), 1, col.size(), f); fclose(f); }
    f = fopen((out + ".pfm").c_str(), "wb");
    if (f) { fprintf(f, "Pf\n%d %d\n-1.0\n", w, h); for (int y = h - 1; y >= 0; y--) fwrite(&disparity[(size_t)y * w], 4, w, f); fclose(f); }
    f = fopen((out + ".txt").c_str(), "w");
    if (f) { for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const float d = disparity[(size_t)y * w + x]; if (d != Invalid_Float) fprintf(f, "%d %d %f\n", x, y, d); } fclose(f); }
    return 0;
}
