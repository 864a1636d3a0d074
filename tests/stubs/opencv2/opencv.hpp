// tests/stubs/opencv2/opencv.hpp -- TEST INFRASTRUCTURE: the few OpenCV 3.1 names the reference's demo caller uses
// (/root/reference/AD-Census/main.cpp:12,47-48,65-76,150,165-176,183,198-209), so that main.cpp compiles UNMODIFIED against
// include/ADCensusStereo.h and links libadcensus.so (tests/test_reference_caller.py; SURVEY.md 8b "main.cpp:80-118 compiles
// unchanged").  OpenCV itself is not in this image (the reference does not vendor it either: 3rdparty/.gitkeep).
//   cv::Mat / cv::Vec3b       just enough of the reference-counted matrix header (rows, cols, data, at<Vec3b>)
//   cv::imread / cv::imwrite  8-bit PNG through the demo programs' own codec (examples/adc_image_io.h, zlib)
//   cv::applyColorMap         COLORMAP_JET only: the table of examples/adc_image_io.h
//   cv::imshow / cv::waitKey  no-ops (no display)
//   fopen_s / fprintf_s       MSVC CRT names main.cpp:216,224 uses (not OpenCV; declared here so ONE -I switch suffices)
#pragma once

// (like the real header: the C and C++ standard library, incl. the float overloads of abs -- main.cpp:154 takes abs(float))
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "../../../examples/adc_image_io.h"

typedef unsigned char uchar;
#define CV_8UC1 0
#define CV_8UC3 16

namespace cv {

struct Vec3b {
    uchar val[3];
    uchar& operator[](int i) { return val[i]; }
    const uchar& operator[](int i) const { return val[i]; }
};

enum { IMREAD_COLOR = 1 };
enum { COLORMAP_JET = 2 };

class Mat {
public:
    int rows, cols;
    uchar* data;
    Mat() : rows(0), cols(0), data(nullptr), type_(CV_8UC1) {}
    Mat(int r, int c, int type) : rows(r), cols(c), data(nullptr), type_(type) { create(); }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    template <typename T> T& at(int i, int j) { return *reinterpret_cast<T*>(data + ((size_t)i * cols + j) * sizeof(T)); }
    template <typename T> const T& at(int i, int j) const { return *reinterpret_cast<const T*>(data + ((size_t)i * cols + j) * sizeof(T)); }
    void create()
    {   // shared storage: copies of a Mat are headers onto the same pixels, like OpenCV's
        store_ = std::make_shared<std::vector<uchar>>((size_t)rows * cols * channels(), (uchar)0);
        data = store_->data();
    }
private:
    int type_;
    std::shared_ptr<std::vector<uchar>> store_;
};

inline Mat imread(const std::string& path, int /*flags: IMREAD_COLOR*/)
{
    std::vector<uint8> bgr;
    int w = 0, h = 0;
    if (!load_image(path.c_str(), bgr, w, h)) return Mat(); // data == nullptr, what main.cpp:50 tests
    Mat m(h, w, CV_8UC3);
    memcpy(m.data, bgr.data(), bgr.size());
    return m;
}

inline bool imwrite(const std::string& path, const Mat& m)
{
    if (!m.data) return false;
    if (m.channels() == 1) return write_png(path, m.data, m.cols, m.rows, 1);
    std::vector<uint8> rgb((size_t)m.rows * m.cols * 3); // Mat holds B,G,R; PNG wants R,G,B
    for (size_t i = 0; i < (size_t)m.rows * m.cols; i++) { rgb[3 * i] = m.data[3 * i + 2]; rgb[3 * i + 1] = m.data[3 * i + 1]; rgb[3 * i + 2] = m.data[3 * i]; }
    return write_png(path, rgb.data(), m.cols, m.rows, 3);
}

inline void applyColorMap(const Mat& src, Mat& dst, int /*colormap: COLORMAP_JET*/)
{
    dst = Mat(src.rows, src.cols, CV_8UC3);
    for (size_t i = 0; i < (size_t)src.rows * src.cols; i++) {
        const uint8* c = kJet[src.data[i]]; // {R,G,B}
        dst.data[3 * i] = c[2]; dst.data[3 * i + 1] = c[1]; dst.data[3 * i + 2] = c[0];
    }
}

inline void imshow(const std::string&, const Mat&) {}
inline int waitKey(int = 0) { return -1; }

} // namespace cv

#ifndef _MSC_VER
inline int fopen_s(FILE** f, const char* path, const char* mode) { *f = fopen(path, mode); return *f ? 0 : 1; }
#define fprintf_s fprintf
#endif
