// tests/stubs/integration_b.cpp -- TEST INFRASTRUCTURE: INTEGRATION.md section B compiled as printed.
// tools/integration_b_patch.py puts two files into a scratch directory that is first on the include path:
//   ADCensusStereo.h           the REFERENCE's class declaration (ADCensusStereo.h:14-95) + the two additions section B names
//   integration_b_patch.inc    the code block of section B, verbatim (Initialize / Match / Release over the C ABI)
// What section B leaves as it is in the reference's ADCensusStereo.cpp -- constructor (:11-13), destructor (:15-19), Reset
// (:134-144) -- is restated below (three one-liners); the stage objects the class still owns by value come from the reference's
// own stage sources, compiled in place next to this file (oracle/Makefile, target dropin).
//
//   ref_integration_b W H dmin dmax left.bgr right.bgr out.f32      raw B,G,R in, raw float32 out; exit 0 on success
#include "ADCensusStereo.h"
#include "integration_b_patch.inc"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

ADCensusStereo::ADCensusStereo() : width_(0), height_(0), img_left_(nullptr), img_right_(nullptr), disp_left_(nullptr),
                                   disp_right_(nullptr), is_initialized_(false) {}
ADCensusStereo::~ADCensusStereo() { Release(); is_initialized_ = false; }
bool ADCensusStereo::Reset(const uint32& width, const uint32& height, const ADCensusOption& option)
{
    Release();
    is_initialized_ = false;
    return Initialize(width, height, option);
}

static bool slurp(const char* path, std::vector<uint8>& buf)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    const bool ok = fread(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    return ok;
}

int main(int argc, char** argv)
{
    if (argc != 8) { fprintf(stderr, "usage: %s W H dmin dmax left.bgr right.bgr out.f32\n", argv[0]); return 64; }
    const sint32 w = atoi(argv[1]), h = atoi(argv[2]);
    ADCensusOption opt;
    opt.min_disparity = atoi(argv[3]);
    opt.max_disparity = atoi(argv[4]);
    ADCensusStereo st;
    std::vector<float32> disp((size_t)(w > 0 ? w : 0) * (h > 0 ? h : 0), 0.0f);
    if (st.Match(nullptr, nullptr, nullptr)) return 3; // not initialised -> false (ADCensusStereo.cpp:71-73)
    if (!st.Initialize(w, h, opt)) { fprintf(stderr, "Initialize failed: %s\n", adc_last_error()); return 2; }
    std::vector<uint8> left((size_t)w * h * 3), right((size_t)w * h * 3);
    if (!slurp(argv[5], left) || !slurp(argv[6], right)) return 65;
    if (st.Match(left.data(), nullptr, disp.data())) return 3; // null pointer -> false (:74-76)
    if (!st.Match(left.data(), right.data(), disp.data())) { fprintf(stderr, "Match failed: %s\n", adc_last_error()); return 4; }
    if (!st.Reset(w, h, opt)) return 5; // Release + Initialize (:134-144)
    std::vector<float32> again(disp.size(), 0.0f);
    if (!st.Match(left.data(), right.data(), again.data())) return 4;
    if (memcmp(again.data(), disp.data(), disp.size() * sizeof(float32)) != 0) return 6; // Match is stateless
    FILE* f = fopen(argv[7], "wb");
    if (!f || fwrite(disp.data(), sizeof(float32), disp.size(), f) != disp.size()) return 66;
    fclose(f);
    return 0;
}
