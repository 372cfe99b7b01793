// Image files of the host layer: Radiance .hdr (RGBE) reader for the environment map the reference loads
// (RayTracingLayer.cpp:50-54, "assets/hdr/graveyard_pathways_4k.hdr", through the absent engine's TextureCube)
// and a binary PPM writer for post-processed frames.
//
// RGBE decoding follows the published format (G. Ward, "Real Pixels", Graphics Gems II): header lines up to an
// empty line, a resolution line "-Y <h> +X <w>", then scanlines either flat (4 bytes per pixel) or run-length
// encoded per channel (marker 2 2 <w hi> <w lo>).  A pixel (r, g, b, e) is (r, g, b) * 2^(e - 136), or 0 if e == 0.
#include "host.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace ptamd {

namespace {

struct File {
    FILE* f;
    explicit File(const std::string& path, const char* mode) : f(fopen(path.c_str(), mode)) {}
    ~File() { if (f) fclose(f); }
};

bool read_line(FILE* f, std::string& line)
{
    line.clear();
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '\n') return true;
        if (c != '\r') line.push_back((char)c);
        if (line.size() > 4096) throw std::runtime_error("hdr: header line too long");
    }
    return !line.empty();
}

void decode_pixel(const uint8_t* rgbe, float* out)
{
    if (rgbe[3] == 0) { out[0] = out[1] = out[2] = 0.0f; }
    else {
        float f = std::ldexp(1.0f, (int)rgbe[3] - 136);
        out[0] = (float)rgbe[0] * f; out[1] = (float)rgbe[1] * f; out[2] = (float)rgbe[2] * f;
    }
    out[3] = 1.0f;
}

} // namespace

HdrImage LoadHdr(const std::string& path)
{
    File file(path, "rb");
    if (!file.f) throw std::runtime_error("hdr: cannot open " + path);
    std::string line;
    if (!read_line(file.f, line) || (line != "#?RADIANCE" && line != "#?RGBE")) throw std::runtime_error("hdr: not a Radiance file: " + path);
    bool format_ok = false;
    for (;;) {
        if (!read_line(file.f, line) || line.empty()) break;   // an empty line ends the header
        if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
    }
    if (!format_ok) throw std::runtime_error("hdr: FORMAT=32-bit_rle_rgbe expected");
    if (!read_line(file.f, line)) throw std::runtime_error("hdr: resolution line missing");
    int h = 0, w = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (long long)w * h > (1ll << 28))
        throw std::runtime_error("hdr: only the '-Y h +X w' layout is supported");
    HdrImage img;
    img.Width = (uint32_t)w;
    img.Height = (uint32_t)h;
    img.Pixels.resize((size_t)w * h * 4);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        uint8_t head[4];
        if (fread(head, 1, 4, file.f) != 4) throw std::runtime_error("hdr: file ends inside the image");
        bool rle = w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && (head[2] & 0x80) == 0 && ((head[2] << 8) | head[3]) == w;
        if (!rle) {
            // flat scanline: the four bytes were its first pixel
            std::memcpy(scan.data(), head, 4);
            if (w > 1 && fread(scan.data() + 4, 4, (size_t)w - 1, file.f) != (size_t)w - 1) throw std::runtime_error("hdr: file ends inside the image");
        } else {
            for (int ch = 0; ch < 4; ch++) {
                int x = 0;
                while (x < w) {
                    int count = fgetc(file.f);
                    if (count == EOF) throw std::runtime_error("hdr: file ends inside a run");
                    if (count > 128) {
                        count -= 128;
                        int value = fgetc(file.f);
                        if (value == EOF || count == 0 || x + count > w) throw std::runtime_error("hdr: bad run");
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + ch] = (uint8_t)value;
                    } else {
                        if (count == 0 || x + count > w) throw std::runtime_error("hdr: bad literal run");
                        for (int k = 0; k < count; k++) {
                            int value = fgetc(file.f);
                            if (value == EOF) throw std::runtime_error("hdr: file ends inside a run");
                            scan[(size_t)(x++) * 4 + ch] = (uint8_t)value;
                        }
                    }
                }
            }
        }
        for (int x = 0; x < w; x++) decode_pixel(&scan[(size_t)x * 4], &img.Pixels[((size_t)y * w + x) * 4]);
    }
    return img;
}

void WritePpm(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgba8)
{
    File file(path, "wb");
    if (!file.f) throw std::runtime_error("ppm: cannot create " + path);
    fprintf(file.f, "P6\n%u %u\n255\n", width, height);
    std::vector<uint8_t> row((size_t)width * 3);
    for (uint32_t y = 0; y < height; y++) {
        for (uint32_t x = 0; x < width; x++) std::memcpy(&row[(size_t)x * 3], rgba8 + ((size_t)y * width + x) * 4, 3);
        if (fwrite(row.data(), 1, row.size(), file.f) != row.size()) throw std::runtime_error("ppm: write failed");
    }
}

} // namespace ptamd
