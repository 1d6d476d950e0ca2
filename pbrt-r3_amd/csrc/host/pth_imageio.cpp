// pth_imageio.cpp -- image output of the front end: what Film::write_image -> write_image
// (src/core/film/film.rs:440-484, src/core/imageio/write_image.rs:16-84) does with the resolved RGB buffer.
//   .exr  32-bit float scanline OpenEXR, uncompressed (the reference writes through the `exr` crate; the pixel
//         values are the same floats, the container bytes are not meant to match)
//   .png  8-bit RGB; to_byte = clamp(255 * gamma_correct(v), 0, 255) truncated (write_image.rs:16-18,
//         core/base/functions.rs:13-19), deflate through zlib
//   .pfm  little-endian float RGB
#include <zlib.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/pbrtgpu_host.h"

namespace {

float gamma_correct(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f; }
uint8_t to_byte(float v) {
    float b = 255.0f * gamma_correct(v);
    if (!(b > 0.0f)) return 0;              // clamp(.., 0, 255) as u8; NaN -> 0 (Rust's saturating cast)
    return b >= 255.0f ? 255 : (uint8_t)b;
}
bool has_ext(const std::string& p, const char* e) {
    size_t n = std::strlen(e);
    if (p.size() < n) return false;
    for (size_t i = 0; i < n; i++)
        if (std::tolower((unsigned char)p[p.size() - n + i]) != e[i]) return false;
    return true;
}

void put32(std::vector<uint8_t>& v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
void put64(std::vector<uint8_t>& v, uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i))); }
void putf(std::vector<uint8_t>& v, float f) { uint32_t u; std::memcpy(&u, &f, 4); put32(v, u); }
void puts0(std::vector<uint8_t>& v, const char* s) { while (*s) v.push_back((uint8_t)*s++); v.push_back(0); }
void attr(std::vector<uint8_t>& v, const char* name, const char* type, const std::vector<uint8_t>& data) {
    puts0(v, name); puts0(v, type); put32(v, (uint32_t)data.size());
    v.insert(v.end(), data.begin(), data.end());
}

pt_status write_exr(const char* path, const float* rgb, int w, int h, int x0, int y0, int full_w, int full_h) {
    std::vector<uint8_t> hd;
    put32(hd, 20000630u);                   // magic
    put32(hd, 2u);                          // version 2, single-part scanline
    {   // channels, alphabetical: B G R, 32-bit float
        std::vector<uint8_t> ch;
        for (const char* c : {"B", "G", "R"}) { puts0(ch, c); put32(ch, 2u); put32(ch, 0u); put32(ch, 1u); put32(ch, 1u); }
        ch.push_back(0);
        attr(hd, "channels", "chlist", ch);
    }
    { std::vector<uint8_t> d; d.push_back(0); attr(hd, "compression", "compression", d); }
    { std::vector<uint8_t> d; put32(d, (uint32_t)x0); put32(d, (uint32_t)y0); put32(d, (uint32_t)(x0 + w - 1)); put32(d, (uint32_t)(y0 + h - 1)); attr(hd, "dataWindow", "box2i", d); }
    { std::vector<uint8_t> d; put32(d, 0); put32(d, 0); put32(d, (uint32_t)(full_w - 1)); put32(d, (uint32_t)(full_h - 1)); attr(hd, "displayWindow", "box2i", d); }
    { std::vector<uint8_t> d; d.push_back(0); attr(hd, "lineOrder", "lineOrder", d); }
    { std::vector<uint8_t> d; putf(d, 1.0f); attr(hd, "pixelAspectRatio", "float", d); }
    { std::vector<uint8_t> d; putf(d, 0.0f); putf(d, 0.0f); attr(hd, "screenWindowCenter", "v2f", d); }
    { std::vector<uint8_t> d; putf(d, 1.0f); attr(hd, "screenWindowWidth", "float", d); }
    hd.push_back(0);
    const uint64_t line_bytes = 8 + (uint64_t)w * 12;
    uint64_t off = hd.size() + (uint64_t)h * 8;
    for (int y = 0; y < h; y++) { put64(hd, off); off += line_bytes; }
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_ERR_INVALID_ARGUMENT;
    bool ok = std::fwrite(hd.data(), 1, hd.size(), f) == hd.size();
    std::vector<float> line((size_t)w * 3);
    for (int y = 0; y < h && ok; y++) {
        const float* row = rgb + (size_t)y * w * 3;
        for (int x = 0; x < w; x++) { line[x] = row[3 * x + 2]; line[w + x] = row[3 * x + 1]; line[2 * w + x] = row[3 * x]; }
        int32_t head[2] = {y0 + y, (int32_t)((size_t)w * 12)};
        ok = std::fwrite(head, 4, 2, f) == 2 && std::fwrite(line.data(), 4, line.size(), f) == line.size();
    }
    std::fclose(f);
    return ok ? PT_OK : PT_ERR_INVALID_ARGUMENT;
}

pt_status write_png(const char* path, const float* rgb, int w, int h) {
    std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 3));
    for (int y = 0; y < h; y++) {
        uint8_t* row = &raw[(size_t)y * (1 + (size_t)w * 3)];
        row[0] = 0;                          // filter type None
        for (int i = 0; i < 3 * w; i++) row[1 + i] = to_byte(rgb[(size_t)y * w * 3 + i]);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return PT_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_ERR_INVALID_ARGUMENT;
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t n) {
        uint8_t len[4] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n};
        std::fwrite(len, 1, 4, f);
        std::fwrite(type, 1, 4, f);
        if (n) std::fwrite(data, 1, n, f);
        uLong c = crc32(0L, (const Bytef*)type, 4);
        if (n) c = crc32(c, data, n);
        uint8_t cb[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
        std::fwrite(cb, 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                        8, 2, 0, 0, 0};     // 8 bits, colour type 2 (RGB)
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zn);
    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return PT_OK;
}

}  // namespace

extern "C" {

pt_status pth_write_pfm(const char* path, const float* rgb, int w, int h) {
    if (!path || !rgb || w <= 0 || h <= 0) return PT_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PT_ERR_INVALID_ARGUMENT;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", w, h);                  // little-endian, bottom row first
    for (int y = h - 1; y >= 0; y--) std::fwrite(rgb + (size_t)y * w * 3, sizeof(float), (size_t)w * 3, f);
    std::fclose(f);
    return PT_OK;
}

pt_status pth_write_image(const char* path, const float* rgb, int w, int h, int x0, int y0, int full_w, int full_h) {
    if (!path || !rgb || w <= 0 || h <= 0) return PT_ERR_INVALID_ARGUMENT;
    const std::string p = path;
    if (has_ext(p, ".exr")) return write_exr(path, rgb, w, h, x0, y0, full_w > 0 ? full_w : w, full_h > 0 ? full_h : h);
    if (has_ext(p, ".png")) return write_png(path, rgb, w, h);
    if (has_ext(p, ".pfm")) return pth_write_pfm(path, rgb, w, h);
    return PT_ERR_UNSUPPORTED;
}

}  // extern "C"
