// adc_image_io.h -- the image-file helpers of the demo programs: 8-bit PNG (zlib) / binary PPM reader into tightly packed
// B,G,R (the layout the reference's main.cpp:65-76 builds from cv::Vec3b), PNG writer, and OpenCV 3.1's COLORMAP_JET table.
// Shared by examples/adcensus_cli.cpp and by tests/stubs/opencv2/opencv.hpp (the stand-in that lets the reference's own
// main.cpp compile unmodified against include/ADCensusStereo.h).  Header-only, static functions; link with -lz.
#pragma once

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "adcensus_types.h"

// ------------------------------------------------------------------------------------------------ image files
static bool read_file(const char* path, std::vector<uint8>& buf)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    return ok;
}
static uint32 be32(const uint8* p) { return ((uint32)p[0] << 24) | ((uint32)p[1] << 16) | ((uint32)p[2] << 8) | p[3]; }

// 8-bit, non-interlaced PNG -> tightly packed B,G,R (the layout main.cpp:65-76 builds from cv::Vec3b)
static bool decode_png(const std::vector<uint8>& file, std::vector<uint8>& bgr, int& w, int& h)
{
    static const uint8 sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8> idat, plte;
    while (pos + 12 <= file.size()) {
        const uint32 len = be32(&file[pos]);
        const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
        if (pos + 12 + len > file.size()) return false;
        const uint8* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return false; // (a short header chunk would be read past its end)
            w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) return false;
    const int ch = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : (ctype == 4 ? 2 : (ctype == 6 ? 4 : 0))));
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    // a deflate stream expands at most ~1032x: bound the dimensions the header claims by what the IDAT data can hold
    if (w > (1 << 20) || h > (1 << 20) || (stride + 1) * (size_t)h > idat.size() * 1040 + 65536) return false;
    std::vector<uint8> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<uint8> img(stride * h), zero(stride, 0);
    for (int y = 0; y < h; y++) { // undo the scanline filters (PNG specification, section 9)
        const uint8 ft = raw[(stride + 1) * y];
        const uint8* in = &raw[(stride + 1) * y + 1];
        uint8* out = &img[stride * y];
        const uint8* up = y ? &img[stride * (y - 1)] : zero.data();
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? out[i - ch] : 0, b = up[i], c = i >= (size_t)ch ? up[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) return false;
            out[i] = (uint8)(in[i] + pred);
        }
    }
    bgr.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        uint8 r, g, b;
        const uint8* p = &img[i * ch];
        if (ctype == 0 || ctype == 4) r = g = b = p[0];
        else if (ctype == 3) { if ((size_t)p[0] * 3 + 2 >= plte.size()) return false; r = plte[p[0] * 3]; g = plte[p[0] * 3 + 1]; b = plte[p[0] * 3 + 2]; }
        else { r = p[0]; g = p[1]; b = p[2]; }
        bgr[3 * i] = b; bgr[3 * i + 1] = g; bgr[3 * i + 2] = r;
    }
    return true;
}
static bool decode_ppm(const std::vector<uint8>& file, std::vector<uint8>& bgr, int& w, int& h)
{
    size_t pos = 0;
    auto token = [&](std::string& t) {
        t.clear();
        while (pos < file.size()) {
            if (file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; }
            else if (isspace(file[pos])) pos++;
            else break;
        }
        while (pos < file.size() && !isspace(file[pos])) t.push_back((char)file[pos++]);
        return !t.empty();
    };
    std::string t;
    if (!token(t) || t != "P6") return false;
    if (!token(t)) return false;
    w = atoi(t.c_str());
    if (!token(t)) return false;
    h = atoi(t.c_str());
    if (!token(t) || atoi(t.c_str()) != 255) return false;
    pos++; // the single whitespace byte behind maxval
    if (w <= 0 || h <= 0 || pos + (size_t)w * h * 3 > file.size()) return false;
    bgr.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) { bgr[3 * i] = file[pos + 3 * i + 2]; bgr[3 * i + 1] = file[pos + 3 * i + 1]; bgr[3 * i + 2] = file[pos + 3 * i]; }
    return true;
}
static bool load_image(const char* path, std::vector<uint8>& bgr, int& w, int& h)
{
    std::vector<uint8> file;
    if (!read_file(path, file)) return false;
    return decode_png(file, bgr, w, h) || decode_ppm(file, bgr, w, h);
}

// 8-bit PNG, channels = 1 (gray) or 3 (R,G,B): filter 0 on every line, one zlib stream, one IDAT
static bool write_png(const std::string& path, const uint8* px, int w, int h, int channels)
{
    const size_t stride = (size_t)w * channels;
    std::vector<uint8> raw((stride + 1) * h);
    for (int y = 0; y < h; y++) { raw[(stride + 1) * y] = 0; memcpy(&raw[(stride + 1) * y + 1], px + stride * y, stride); }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    auto chunk = [&](const char* type, const uint8* data, uint32 len) {
        uint8 hdr[8] = {(uint8)(len >> 24), (uint8)(len >> 16), (uint8)(len >> 8), (uint8)len, (uint8)type[0], (uint8)type[1], (uint8)type[2], (uint8)type[3]};
        fwrite(hdr, 1, 8, f);
        if (len) fwrite(data, 1, len, f);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (len) crc = crc32(crc, data, len);
        const uint8 c4[4] = {(uint8)(crc >> 24), (uint8)(crc >> 16), (uint8)(crc >> 8), (uint8)crc};
        fwrite(c4, 1, 4, f);
    };
    static const uint8 sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    fwrite(sig, 1, 8, f);
    const uint8 ihdr[13] = {(uint8)(w >> 24), (uint8)(w >> 16), (uint8)(w >> 8), (uint8)w, (uint8)(h >> 24), (uint8)(h >> 16), (uint8)(h >> 8), (uint8)h,
                            8, (uint8)(channels == 1 ? 0 : 2), 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32)clen);
    chunk("IEND", nullptr, 0);
    fclose(f);
    return true;
}

// cv::COLORMAP_JET of OpenCV 3.1 as {R,G,B} per grey level.  The reference writes <name>-d.png and its colour-mapped
// copy <name>-c.png (main.cpp:203-209); this table is that mapping, read off the reference's own result images
// (doc/exp/res/{cone,cloth,piano}-{d,c}.png: 239 of the 256 levels occur, each with ONE colour); the 17 levels that do
// not occur (2-9, 13-19, 22, 24) lie on the first linear segment (0, 0, 128 + 4*level).
static const uint8 kJet[256][3] = {
    {0,0,128}, {0,0,132}, {0,0,136}, {0,0,140}, {0,0,144}, {0,0,148}, {0,0,152}, {0,0,156},
    {0,0,160}, {0,0,164}, {0,0,168}, {0,0,172}, {0,0,176}, {0,0,180}, {0,0,184}, {0,0,188},
    {0,0,192}, {0,0,196}, {0,0,200}, {0,0,204}, {0,0,208}, {0,0,212}, {0,0,216}, {0,0,220},
    {0,0,224}, {0,0,228}, {0,0,232}, {0,0,236}, {0,0,240}, {0,0,244}, {0,0,248}, {0,0,252},
    {0,0,255}, {0,4,255}, {0,8,255}, {0,12,255}, {0,16,255}, {0,20,255}, {0,24,255}, {0,28,255},
    {0,32,255}, {0,36,255}, {0,40,255}, {0,44,255}, {0,48,255}, {0,52,255}, {0,56,255}, {0,60,255},
    {0,64,255}, {0,68,255}, {0,72,255}, {0,76,255}, {0,80,255}, {0,84,255}, {0,88,255}, {0,92,255},
    {0,96,255}, {0,100,255}, {0,104,255}, {0,108,255}, {0,112,255}, {0,116,255}, {0,120,255}, {0,124,255},
    {0,128,255}, {0,132,255}, {0,136,255}, {0,140,255}, {0,144,255}, {0,148,255}, {0,152,255}, {0,156,255},
    {0,160,255}, {0,164,255}, {0,168,255}, {0,172,255}, {0,176,255}, {0,180,255}, {0,184,255}, {0,188,255},
    {0,192,255}, {0,196,255}, {0,200,255}, {0,204,255}, {0,208,255}, {0,212,255}, {0,216,255}, {0,220,255},
    {0,224,255}, {0,228,255}, {0,232,255}, {0,236,255}, {0,240,255}, {0,244,255}, {0,248,255}, {0,252,255},
    {2,255,254}, {6,255,250}, {10,255,246}, {14,255,242}, {18,255,238}, {22,255,234}, {26,255,230}, {30,255,226},
    {34,255,222}, {38,255,218}, {42,255,214}, {46,255,210}, {50,255,206}, {54,255,202}, {58,255,198}, {62,255,194},
    {66,255,190}, {70,255,186}, {74,255,182}, {78,255,178}, {82,255,174}, {86,255,170}, {90,255,166}, {94,255,162},
    {98,255,158}, {102,255,154}, {106,255,150}, {110,255,146}, {114,255,142}, {118,255,138}, {122,255,134}, {126,255,130},
    {130,255,126}, {134,255,122}, {138,255,118}, {142,255,114}, {146,255,110}, {150,255,106}, {154,255,102}, {158,255,98},
    {162,255,94}, {166,255,90}, {170,255,86}, {174,255,82}, {178,255,78}, {182,255,74}, {186,255,70}, {190,255,66},
    {194,255,62}, {198,255,58}, {202,255,54}, {206,255,50}, {210,255,46}, {214,255,42}, {218,255,38}, {222,255,34},
    {226,255,30}, {230,255,26}, {234,255,22}, {238,255,18}, {242,255,14}, {246,255,10}, {250,255,6}, {254,255,1},
    {255,252,0}, {255,248,0}, {255,244,0}, {255,240,0}, {255,236,0}, {255,232,0}, {255,228,0}, {255,224,0},
    {255,220,0}, {255,216,0}, {255,212,0}, {255,208,0}, {255,204,0}, {255,200,0}, {255,196,0}, {255,192,0},
    {255,188,0}, {255,184,0}, {255,180,0}, {255,176,0}, {255,172,0}, {255,168,0}, {255,164,0}, {255,160,0},
    {255,156,0}, {255,152,0}, {255,148,0}, {255,144,0}, {255,140,0}, {255,136,0}, {255,132,0}, {255,128,0},
    {255,124,0}, {255,120,0}, {255,116,0}, {255,112,0}, {255,108,0}, {255,104,0}, {255,100,0}, {255,96,0},
    {255,92,0}, {255,88,0}, {255,84,0}, {255,80,0}, {255,76,0}, {255,72,0}, {255,68,0}, {255,64,0},
    {255,60,0}, {255,56,0}, {255,52,0}, {255,48,0}, {255,44,0}, {255,40,0}, {255,36,0}, {255,32,0},
    {255,28,0}, {255,24,0}, {255,20,0}, {255,16,0}, {255,12,0}, {255,8,0}, {255,4,0}, {255,0,0},
    {252,0,0}, {248,0,0}, {244,0,0}, {240,0,0}, {236,0,0}, {232,0,0}, {228,0,0}, {224,0,0},
    {220,0,0}, {216,0,0}, {212,0,0}, {208,0,0}, {204,0,0}, {200,0,0}, {196,0,0}, {192,0,0},
    {188,0,0}, {184,0,0}, {180,0,0}, {176,0,0}, {172,0,0}, {168,0,0}, {164,0,0}, {160,0,0},
    {156,0,0}, {152,0,0}, {148,0,0}, {144,0,0}, {140,0,0}, {136,0,0}, {132,0,0}, {128,0,0}};
