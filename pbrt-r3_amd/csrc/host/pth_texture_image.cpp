// pth_texture_image.cpp -- see pth_texture_image.h.
#include "pth_texture_image.h"
#include "pth_exr_codecs.h"
#include "pth_jpeg.h"
#include <zlib.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>

namespace pth {
namespace {

bool read_all(const std::string& path, std::vector<uint8_t>* out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    size_t got = out->empty() ? 0 : std::fread(out->data(), 1, out->size(), f);
    std::fclose(f);
    return got == out->size();
}
bool ends_with(const std::string& s, const char* ext) {
    size_t n = std::strlen(ext);
    return s.size() >= n && s.compare(s.size() - n, n, ext) == 0;
}

// ---- PFM (read_image_pfm.rs:26-110): rows bottom-up in the file, scale sign = endianness
bool read_pfm(const std::vector<uint8_t>& b, RgbImage* out, std::string* err) {
    size_t pos = 0;
    auto word = [&](std::string* w) {
        while (pos < b.size() && std::isspace(b[pos])) pos++;
        size_t s = pos;
        while (pos < b.size() && !std::isspace(b[pos])) pos++;
        w->assign((const char*)b.data() + s, pos - s);
        return pos > s;
    };
    std::string cc, sw, sh, ss;
    if (!word(&cc) || !word(&sw) || !word(&sh) || !word(&ss)) { *err = "truncated PFM header"; return false; }
    int nc = cc == "Pf" ? 1 : (cc == "PF" ? 3 : 0);
    if (!nc) { *err = "not a PFM file"; return false; }
    pos++;                                               // the single whitespace byte after the scale
    long w = std::atol(sw.c_str()), h = std::atol(sh.c_str());
    float scale = std::strtof(ss.c_str(), nullptr);
    if (w <= 0 || h <= 0 || pos + (size_t)w * h * nc * 4 > b.size()) { *err = "truncated PFM data"; return false; }
    const bool little = scale < 0.0f;
    scale = std::fabs(scale);
    out->width = (int)w; out->height = (int)h;
    out->rgb.resize((size_t)3 * w * h);
    for (long y = 0; y < h; y++) {
        long yy = h - y - 1;
        for (long x = 0; x < w; x++)
            for (int c = 0; c < nc; c++) {
                uint8_t q[4];
                std::memcpy(q, b.data() + pos, 4);
                pos += 4;
                uint32_t u = little ? ((uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24)
                                    : ((uint32_t)q[3] | (uint32_t)q[2] << 8 | (uint32_t)q[1] << 16 | (uint32_t)q[0] << 24);
                float f;
                std::memcpy(&f, &u, 4);
                if (scale != 1.0f) f *= scale;
                float* px = &out->rgb[3 * ((size_t)yy * w + x)];
                if (nc == 1) px[0] = px[1] = px[2] = f; else px[c] = f;
            }
    }
    return true;
}

// ---- PNG: zlib stream of filtered scanlines (no interlace)
bool read_png(const std::vector<uint8_t>& b, RgbImage* out, std::string* err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 8 || std::memcmp(b.data(), sig, 8) != 0) { *err = "not a PNG file"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    auto be32 = [&](size_t p) { return (uint32_t)b[p] << 24 | (uint32_t)b[p + 1] << 16 | (uint32_t)b[p + 2] << 8 | (uint32_t)b[p + 3]; };
    while (pos + 12 <= b.size()) {
        uint32_t len = be32(pos);
        std::string type((const char*)b.data() + pos + 4, 4);
        if (pos + 12 + len > b.size()) { *err = "truncated PNG chunk"; return false; }
        const uint8_t* d = b.data() + pos + 8;
        if (type == "IHDR" && len >= 13) { w = be32(pos + 8); h = be32(pos + 12); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (type == "PLTE") plte.assign(d, d + len);
        else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (type == "IEND") break;
        pos += 12 + len;
    }
    if (!w || !h || idat.empty()) { *err = "PNG without image data"; return false; }
    if (interlace) { *err = "interlaced PNG is not supported"; return false; }
    const int nch = ctype == 0 ? 1 : (ctype == 2 ? 3 : (ctype == 3 ? 1 : (ctype == 4 ? 2 : (ctype == 6 ? 4 : 0))));
    // the reference accepts Luma8, LumaA8, Rgb8, Rgba8 (palettes expand to those) and Rgb16 (read_image.rs:145-181)
    const bool ok = nch && ((depth == 8) || (depth == 16 && ctype == 2));
    if (!ok) { *err = "PNG colour type / bit depth not supported (8-bit gray, gray+alpha, RGB, RGBA, palette; 16-bit RGB)"; return false; }
    const size_t bpp = (size_t)nch * depth / 8, stride = bpp * w;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { *err = "PNG data does not inflate to the image size"; return false; }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* src = raw.data() + (stride + 1) * y;
        uint8_t* cur = img.data() + stride * y;
        const uint8_t* up = y ? cur - stride : nullptr;
        const int filter = src[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? cur[i - bpp] : 0, bb = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, x = src[1 + i], v;
            switch (filter) {
                case 0: v = x; break;
                case 1: v = x + a; break;
                case 2: v = x + bb; break;
                case 3: v = x + ((a + bb) >> 1); break;
                case 4: { int p = a + bb - c, pa = std::abs(p - a), pb = std::abs(p - bb), pc = std::abs(p - c); v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c)); break; }
                default: *err = "bad PNG filter type"; return false;
            }
            cur[i] = (uint8_t)v;
        }
    }
    out->width = (int)w; out->height = (int)h;
    out->rgb.resize((size_t)3 * w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* p = img.data() + i * bpp;
        float* o = &out->rgb[3 * i];
        if (depth == 16) { for (int c = 0; c < 3; c++) o[c] = (float)(uint16_t)((uint16_t)p[2 * c] << 8 | p[2 * c + 1]) / 65535.0f; }
        else if (ctype == 3) {
            if ((size_t)p[0] * 3 + 2 >= plte.size()) { *err = "PNG palette index out of range"; return false; }
            for (int c = 0; c < 3; c++) o[c] = (float)plte[(size_t)p[0] * 3 + c] / 255.0f;
        } else if (nch <= 2) o[0] = o[1] = o[2] = (float)p[0] / 255.0f;
        else for (int c = 0; c < 3; c++) o[c] = (float)p[c] / 255.0f;
    }
    return true;
}

// ---- TGA: types 2 / 3 (raw) and 10 / 11 (RLE), 8-bit gray or 24 / 32-bit BGR(A)
bool read_tga(const std::vector<uint8_t>& b, RgbImage* out, std::string* err) {
    if (b.size() < 18) { *err = "truncated TGA header"; return false; }
    const int id_len = b[0], cmap = b[1], type = b[2], w = b[12] | b[13] << 8, h = b[14] | b[15] << 8, bpp = b[16], desc = b[17];
    const bool rle = type == 10 || type == 11, gray = type == 3 || type == 11;
    if (cmap != 0 || !(type == 2 || type == 3 || rle) || (gray ? bpp != 8 : (bpp != 24 && bpp != 32)) || w <= 0 || h <= 0) { *err = "TGA variant not supported (true-colour 24/32-bit or 8-bit gray, raw or RLE)"; return false; }
    const size_t px = (size_t)bpp / 8, n = (size_t)w * h;
    std::vector<uint8_t> data(n * px);
    size_t pos = 18 + (size_t)id_len, at = 0;
    if (!rle) {
        if (pos + data.size() > b.size()) { *err = "truncated TGA data"; return false; }
        std::memcpy(data.data(), b.data() + pos, data.size());
    } else {
        while (at < n) {
            if (pos >= b.size()) { *err = "truncated TGA data"; return false; }
            const int hd = b[pos++], cnt = (hd & 127) + 1;
            if (hd & 128) {
                if (pos + px > b.size()) { *err = "truncated TGA data"; return false; }
                for (int k = 0; k < cnt && at < n; k++, at++) std::memcpy(&data[at * px], &b[pos], px);
                pos += px;
            } else {
                if (pos + px * cnt > b.size()) { *err = "truncated TGA data"; return false; }
                for (int k = 0; k < cnt && at < n; k++, at++, pos += px) std::memcpy(&data[at * px], &b[pos], px);
            }
        }
    }
    const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
    out->width = w; out->height = h;
    out->rgb.resize(3 * n);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t* p = &data[((size_t)y * w + x) * px];
            const int oy = top_down ? y : h - 1 - y, ox = right_left ? w - 1 - x : x;
            float* o = &out->rgb[3 * ((size_t)oy * w + ox)];
            if (gray) o[0] = o[1] = o[2] = (float)p[0] / 255.0f;
            else { o[0] = (float)p[2] / 255.0f; o[1] = (float)p[1] / 255.0f; o[2] = (float)p[0] / 255.0f; }
        }
    return true;
}

// ---- OpenEXR (image::open -> codecs::openexr: the R, G, B channels of a single-part scan-line or tiled file as f32; alpha is read and
// dropped by convert_from_rgba32f, read_image.rs:112-142).  Pixel types half and float; compression none, RLE, ZIPS, ZIP (the
// byte-stream schemes, below), PIZ and PXR24 (pth_exr_codecs.cpp); B44 / DWA files are reported, not approximated.
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                   // subnormal half: normalise
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

struct ExrChannel { std::string name; int type = 0; };

bool exr_unpack_block(const uint8_t* src, size_t n_src, int compression, std::vector<uint8_t>* raw, std::string* err) {
    const size_t want = raw->size();
    if (n_src == want) { std::memcpy(raw->data(), src, want); return true; }      // stored: compression did not pay
    std::vector<uint8_t> tmp(want);
    if (compression == 1) {                      // RLE: count < 0 -> -count literal bytes, else count + 1 copies of the next byte
        size_t i = 0, o = 0;
        while (i < n_src) {
            const int c = (int8_t)src[i++];
            if (c < 0) {
                const size_t k = (size_t)(-c);
                if (i + k > n_src || o + k > want) { *err = "EXR: corrupt RLE block"; return false; }
                std::memcpy(&tmp[o], src + i, k); i += k; o += k;
            } else {
                const size_t k = (size_t)c + 1;
                if (i >= n_src || o + k > want) { *err = "EXR: corrupt RLE block"; return false; }
                std::memset(&tmp[o], src[i++], k); o += k;
            }
        }
        if (o != want) { *err = "EXR: short RLE block"; return false; }
    } else {                                     // ZIPS / ZIP: one zlib stream per block
        uLongf got = (uLongf)want;
        if (uncompress(tmp.data(), &got, src, (uLong)n_src) != Z_OK || got != want) { *err = "EXR: corrupt ZIP block"; return false; }
    }
    for (size_t i = 1; i < want; i++) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);      // undo the byte delta predictor
    const size_t half = (want + 1) / 2;                                                     // undo the even / odd byte split
    for (size_t i = 0; i < want; i++) (*raw)[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
    return true;
}

bool read_exr(const std::vector<uint8_t>& b, RgbImage* out, std::string* err) {
    size_t pos = 0;
    auto need = [&](size_t n) { return pos + n <= b.size(); };
    auto rd_i32 = [&](size_t at) { int32_t v; std::memcpy(&v, &b[at], 4); return v; };
    auto rd_str = [&](std::string* sv) {
        const size_t start = pos;
        while (pos < b.size() && b[pos] != 0) pos++;
        if (pos >= b.size()) return false;
        sv->assign((const char*)&b[start], pos - start);
        pos++;
        return true;
    };
    if (b.size() < 8 || rd_i32(0) != 20000630) { *err = "not an OpenEXR file"; return false; }
    const uint32_t version = (uint32_t)rd_i32(4);
    if ((version & 0xffu) != 2 || (version & 0x1800u)) { *err = "EXR: deep and multi-part files are not supported (single-part scan-line and tiled files are)"; return false; }
    const bool tiled = (version & 0x200u) != 0;
    int64_t tile_w = 0, tile_h = 0;
    pos = 8;
    std::vector<ExrChannel> channels;
    int compression = -1, dw[4] = {0, 0, -1, -1};
    for (;;) {
        std::string name, type;
        if (!rd_str(&name)) { *err = "EXR: truncated header"; return false; }
        if (name.empty()) break;
        if (!rd_str(&type) || !need(4)) { *err = "EXR: truncated header"; return false; }
        const int32_t size = rd_i32(pos);
        pos += 4;
        if (size < 0 || !need((size_t)size)) { *err = "EXR: truncated header"; return false; }
        const size_t end = pos + (size_t)size;
        if (name == "channels") {
            while (pos < end && b[pos] != 0) {
                ExrChannel ch;
                if (!rd_str(&ch.name) || pos + 16 > end) { *err = "EXR: bad channel list"; return false; }
                ch.type = rd_i32(pos);
                if (rd_i32(pos + 8) != 1 || rd_i32(pos + 12) != 1) { *err = "EXR: subsampled channels are not supported"; return false; }
                pos += 16;
                channels.push_back(ch);
            }
        } else if (name == "compression" && size == 1) compression = b[pos];
        else if (name == "dataWindow" && size == 16) for (int i = 0; i < 4; i++) dw[i] = rd_i32(pos + 4 * (size_t)i);
        else if (name == "tiles" && size == 9) { tile_w = (uint32_t)rd_i32(pos); tile_h = (uint32_t)rd_i32(pos + 4); }      // the level mode does not matter: level (0, 0) is read
        pos = end;
    }
    if (channels.empty() || compression < 0 || dw[2] < dw[0] || dw[3] < dw[1]) { *err = "EXR: header lacks channels / compression / dataWindow"; return false; }
    if (compression > 5) {
        static const char* names[] = {"none", "RLE", "ZIPS", "ZIP", "PIZ", "PXR24", "B44", "B44A", "DWAA", "DWAB"};
        *err = std::string("EXR compression ") + (compression < 10 ? names[compression] : "?") + " is not supported (none, RLE, ZIPS, ZIP, PIZ, PXR24 are)";
        return false;
    }
    const int64_t w = (int64_t)dw[2] - dw[0] + 1, h = (int64_t)dw[3] - dw[1] + 1;
    if (w > 65536 || h > 65536 || w * h > kMaxImagePixels) { *err = "EXR: image too large"; return false; }
    int rgb_at[3] = {-1, -1, -1};
    size_t line_bytes = 0;
    std::vector<size_t> ch_off(channels.size());
    for (size_t c = 0; c < channels.size(); c++) {
        if (channels[c].type != 1 && channels[c].type != 2) { *err = "EXR: uint channels are not supported (half and float are)"; return false; }
        ch_off[c] = line_bytes;
        line_bytes += (size_t)w * (channels[c].type == 1 ? 2 : 4);
        if (channels[c].name == "R") rgb_at[0] = (int)c;
        if (channels[c].name == "G") rgb_at[1] = (int)c;
        if (channels[c].name == "B") rgb_at[2] = (int)c;
    }
    if (rgb_at[0] < 0 || rgb_at[1] < 0 || rgb_at[2] < 0) { *err = "EXR: no R, G, B channels (the reference's decoder needs an RGB layer)"; return false; }
    const int lines_per_block = compression == 4 ? 32 : (compression == 3 || compression == 5 ? 16 : 1);
    std::vector<ExrPlane> planes;
    for (const ExrChannel& ch : channels) planes.push_back(ExrPlane{ch.type == 1 ? 1 : 2});
    const size_t n_blocks = (size_t)((h + lines_per_block - 1) / lines_per_block);
    if (!tiled && !need(n_blocks * 8)) { *err = "EXR: truncated offset table"; return false; }
    const size_t table = pos;
    out->width = (int)w; out->height = (int)h;
    out->rgb.assign((size_t)w * (size_t)h * 3, 0.0f);
    std::vector<uint8_t> raw;
    if (tiled) {
        // Tiled files (the `exr` crate reads the largest level): the offset table starts with the tiles of level (0, 0), row by row;
        // a chunk is tile x, tile y, level x, level y, byte count, then the tile's lines coded as one block.
        if (tile_w <= 0 || tile_h <= 0 || tile_w > 65536 || tile_h > 65536) { *err = "EXR: tiled file without a valid tile description"; return false; }
        const int64_t ntx = (w + tile_w - 1) / tile_w, nty = (h + tile_h - 1) / tile_h;
        if (!need((size_t)(ntx * nty) * 8)) { *err = "EXR: truncated offset table"; return false; }
        std::vector<uint8_t> tile_raw;
        for (int64_t t = 0; t < ntx * nty; t++) {
            uint64_t off;
            std::memcpy(&off, &b[table + 8 * (size_t)t], 8);
            if (off > b.size() || b.size() - off < 20) { *err = "EXR: block offset outside the file"; return false; }      // no sum on the file's own 64-bit value: it wraps
            const int64_t tx = rd_i32((size_t)off), ty = rd_i32((size_t)off + 4);
            const int32_t lx = rd_i32((size_t)off + 8), ly = rd_i32((size_t)off + 12), n_src = rd_i32((size_t)off + 16);
            if (tx < 0 || tx >= ntx || ty < 0 || ty >= nty || lx != 0 || ly != 0 || n_src < 0 || (uint64_t)n_src > b.size() - off - 20) { *err = "EXR: bad tile header"; return false; }
            const int64_t tw = std::min<int64_t>(tile_w, w - tx * tile_w), th = std::min<int64_t>(tile_h, h - ty * tile_h);
            size_t tile_line_bytes = 0;
            std::vector<size_t> t_off(channels.size());
            for (size_t c = 0; c < channels.size(); c++) { t_off[c] = tile_line_bytes; tile_line_bytes += (size_t)tw * (channels[c].type == 1 ? 2 : 4); }
            tile_raw.assign((size_t)th * tile_line_bytes, 0);
            const uint8_t* src = &b[(size_t)off + 20];
            if (compression == 0 || (size_t)n_src == tile_raw.size()) {
                if ((size_t)n_src != tile_raw.size()) { *err = "EXR: bad uncompressed tile size"; return false; }
                std::memcpy(tile_raw.data(), src, tile_raw.size());
            } else if (compression == 4) {
                if (!exr_unpack_piz(src, (size_t)n_src, planes, (size_t)tw, (size_t)th, &tile_raw, err)) return false;
            } else if (compression == 5) {
                if (!exr_unpack_pxr24(src, (size_t)n_src, planes, (size_t)tw, (size_t)th, &tile_raw, err)) return false;
            } else if (!exr_unpack_block(src, (size_t)n_src, compression, &tile_raw, err)) return false;
            for (int64_t yy = 0; yy < th; yy++)
                for (int c = 0; c < 3; c++) {
                    const ExrChannel& ch = channels[(size_t)rgb_at[c]];
                    const uint8_t* line = tile_raw.data() + (size_t)yy * tile_line_bytes + t_off[(size_t)rgb_at[c]];
                    float* dst = &out->rgb[((size_t)(ty * tile_h + yy) * (size_t)w + (size_t)(tx * tile_w)) * 3 + (size_t)c];
                    for (int64_t x = 0; x < tw; x++) {
                        if (ch.type == 1) { uint16_t hv; std::memcpy(&hv, line + 2 * x, 2); dst[3 * x] = half_to_float(hv); }
                        else std::memcpy(&dst[3 * x], line + 4 * x, 4);
                    }
                }
        }
        return true;
    }
    for (size_t k = 0; k < n_blocks; k++) {
        uint64_t off;
        std::memcpy(&off, &b[table + 8 * k], 8);
        if (off > b.size() || b.size() - off < 8) { *err = "EXR: block offset outside the file"; return false; }
        const int64_t y0 = (int64_t)rd_i32((size_t)off) - dw[1];
        const int32_t n_src = rd_i32((size_t)off + 4);
        if (y0 < 0 || y0 >= h || n_src < 0 || (uint64_t)n_src > b.size() - off - 8) { *err = "EXR: bad block header"; return false; }
        const int64_t lines = std::min<int64_t>(lines_per_block, h - y0);
        raw.assign((size_t)lines * line_bytes, 0);
        if (compression == 0) {
            if ((size_t)n_src != raw.size()) { *err = "EXR: bad uncompressed block size"; return false; }
            std::memcpy(raw.data(), &b[(size_t)off + 8], raw.size());
        } else if (compression >= 4 && (size_t)n_src == raw.size()) std::memcpy(raw.data(), &b[(size_t)off + 8], raw.size());      // stored
        else if (compression == 4) {
            if (!exr_unpack_piz(&b[(size_t)off + 8], (size_t)n_src, planes, (size_t)w, (size_t)lines, &raw, err)) return false;
        } else if (compression == 5) {
            if (!exr_unpack_pxr24(&b[(size_t)off + 8], (size_t)n_src, planes, (size_t)w, (size_t)lines, &raw, err)) return false;
        } else if (!exr_unpack_block(&b[(size_t)off + 8], (size_t)n_src, compression, &raw, err)) return false;
        for (int64_t ly = 0; ly < lines; ly++) {
            for (int c = 0; c < 3; c++) {
                const ExrChannel& ch = channels[(size_t)rgb_at[c]];
                const uint8_t* src = raw.data() + (size_t)ly * line_bytes + ch_off[(size_t)rgb_at[c]];
                float* dst = &out->rgb[((size_t)(y0 + ly) * (size_t)w) * 3 + (size_t)c];
                for (int64_t x = 0; x < w; x++) {
                    if (ch.type == 1) { uint16_t hv; std::memcpy(&hv, src + 2 * x, 2); dst[3 * x] = half_to_float(hv); }
                    else std::memcpy(&dst[3 * x], src + 4 * x, 4);
                }
            }
        }
    }
    return true;
}

float inverse_gamma_correct(float v) {                    // core/base/functions.rs:22-28
    if (v <= 0.04045f) return v * 1.0f / 12.92f;
    return std::pow((v + 0.055f) * 1.0f / 1.055f, 2.4f);
}
float lanczos(float x, float tau) {                       // core/texture/noise.rs:152-162
    x = std::fabs(x);
    if (x < 1e-5f) return 1.0f;
    if (x > 1.0f) return 0.0f;
    x *= 3.14159265358979323846f;
    float s = std::sin(x * tau) / (x * tau);
    float l = std::sin(x) / x;
    return s * l;
}
int math_mod(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
uint32_t round_up_pow2(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
struct ResampleWeight { int first_texel; float weight[4]; };
std::vector<ResampleWeight> resample_weights(int old_res, int new_res) {       // mipmap.rs:291-314
    std::vector<ResampleWeight> wt((size_t)new_res);
    const float filter_width = 2.0f;
    for (int i = 0; i < new_res; i++) {
        float center = ((float)i + 0.5f) * ((float)old_res / (float)new_res);
        float first = std::floor((center - filter_width) + 0.5f);
        for (int j = 0; j < 4; j++) {
            float pos = first + (float)j + 0.5f;
            wt[i].weight[j] = lanczos((pos - center) / filter_width, filter_width);
        }
        float inv = 1.0f / (wt[i].weight[0] + wt[i].weight[1] + wt[i].weight[2] + wt[i].weight[3]);
        for (int j = 0; j < 4; j++) wt[i].weight[j] *= inv;
        wt[i].first_texel = (int)first;
    }
    return wt;
}
// resample_image (mipmap.rs:316-404): s pass into the widened image, then t pass column by column, clamp at zero
void resample(const std::vector<float>& img, int c, int w, int h, int swrap, int twrap, int* nw, int* nh, std::vector<float>* out) {
    auto pow2 = [](int v) { return (v & (v - 1)) == 0; };
    if (pow2(w) && pow2(h)) { *nw = w; *nh = h; *out = img; return; }
    const int pw = (int)round_up_pow2((uint32_t)w), ph = (int)round_up_pow2((uint32_t)h);
    std::vector<float> r((size_t)c * pw * ph, 0.0f);
    {
        std::vector<ResampleWeight> sw = resample_weights(w, pw);
        for (int t = 0; t < h; t++)
            for (int s = 0; s < pw; s++)
                for (int j = 0; j < 4; j++) {
                    int os = sw[s].first_texel + j;
                    if (swrap == PT_WRAP_REPEAT) os = math_mod(os, w);
                    else if (swrap == PT_WRAP_CLAMP) os = os < 0 ? 0 : (os > w - 1 ? w - 1 : os);
                    if (os >= 0 && os < w) {
                        float wgt = sw[s].weight[j];
                        size_t src = (size_t)t * w + os, dst = (size_t)t * pw + s;
                        for (int k = 0; k < c; k++) r[c * dst + k] += img[c * src + k] * wgt;
                    }
                }
    }
    {
        std::vector<ResampleWeight> tw = resample_weights(h, ph);
        std::vector<float> buffer((size_t)c * h);
        for (int s = 0; s < pw; s++) {
            for (int t = 0; t < h; t++)
                for (int k = 0; k < c; k++) buffer[(size_t)c * t + k] = r[c * ((size_t)t * pw + s) + k];
            for (int t = 0; t < ph; t++) {
                float l[3] = {0.0f, 0.0f, 0.0f};
                for (int j = 0; j < 4; j++) {
                    int ot = tw[t].first_texel + j;
                    if (twrap == PT_WRAP_REPEAT) ot = math_mod(ot, h);
                    else if (twrap == PT_WRAP_CLAMP) ot = ot < 0 ? 0 : (ot > h - 1 ? h - 1 : ot);
                    if (ot >= 0 && ot < h) {
                        float wgt = tw[t].weight[j];
                        for (int k = 0; k < c; k++) l[k] += buffer[(size_t)c * ot + k] * wgt;
                    }
                }
                for (int k = 0; k < c; k++) r[c * ((size_t)t * pw + s) + k] = l[k];
            }
        }
    }
    for (float& v : r) v = std::fmax(v, 0.0f);             // f32::max: a NaN becomes 0
    *nw = pw; *nh = ph;
    out->swap(r);
}

}  // namespace

static bool read_image_file_checked(const std::string& path, RgbImage* out, std::string* err);

// Files come from outside: a header may claim any size, so the readers bound what they allocate (kMaxImagePixels) and whatever
// still cannot be allocated comes back as an error string -- nothing is thrown across the extern "C" entry points.
bool read_image_file(const std::string& path, RgbImage* out, std::string* err) {
    try {
        return read_image_file_checked(path, out, err);
    } catch (const std::bad_alloc&) {
        *err = "\"" + path + "\": not enough memory for the image its header describes";
    } catch (const std::length_error&) {
        *err = "\"" + path + "\": the image its header describes is too large";
    }
    return false;
}

static bool read_image_file_checked(const std::string& path, RgbImage* out, std::string* err) {
    std::vector<uint8_t> bytes;
    if (!read_all(path, &bytes)) { *err = "File not found: " + path; return false; }
    std::string lower = path;
    for (char& ch : lower) ch = (char)std::tolower((unsigned char)ch);
    if (ends_with(path, ".pfm")) return read_pfm(bytes, out, err);      // has_extension is case-sensitive (read_image.rs:184-186)
    if (ends_with(lower, ".png")) return read_png(bytes, out, err);
    if (ends_with(lower, ".tga")) return read_tga(bytes, out, err);
    if (ends_with(lower, ".exr")) return read_exr(bytes, out, err);
    if (ends_with(lower, ".jpg") || ends_with(lower, ".jpeg")) {      // image::open -> Luma8 / Rgb8 -> convert_from_luma8 / _rgb8 (read_image.rs:11-71)
        int w = 0, h = 0, nch = 0;
        std::vector<uint8_t> px;
        if (!decode_jpeg(bytes, &w, &h, &nch, &px, err)) return false;
        out->width = w; out->height = h;
        out->rgb.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; i++)
            for (int c = 0; c < 3; c++) out->rgb[3 * i + c] = (float)px[i * (size_t)nch + (nch == 1 ? 0 : c)] / 255.0f;
        return true;
    }
    *err = "image format of \"" + path + "\" is not on the accelerated path (pfm, png, tga, exr, jpg are)";
    return false;
}

void build_pyramid(const RgbImage& img, int channels, float scale, bool gamma, int swrap, int twrap, Pyramid* out) {
    const int w = img.width, h = img.height;
    // ImageTexture::convert_in (imagemap.rs:33-61) and flip_y (:136-149)
    std::vector<float> data((size_t)channels * w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float* p = &img.rgb[3 * ((size_t)y * w + x)];
            float* o = &data[(size_t)channels * ((size_t)(h - 1 - y) * w + x)];
            if (channels == 1) {
                float lum = 0.212671f * p[0] + 0.715160f * p[1] + 0.072169f * p[2];          // RGBSpectrum::y (rgb.rs:46-50)
                o[0] = scale * (gamma ? inverse_gamma_correct(lum) : lum);
            } else {
                for (int k = 0; k < 3; k++) o[k] = (gamma ? inverse_gamma_correct(p[k]) : p[k]) * scale;
            }
        }
    int cw, ch;
    std::vector<float> level;
    resample(data, channels, w, h, swrap, twrap, &cw, &ch, &level);
    out->desc.width = (uint32_t)cw; out->desc.height = (uint32_t)ch; out->desc.channels = (uint32_t)channels;
    out->texels.clear();
    uint32_t n_levels = 0;
    for (;;) {                                             // make_pyramid (mipmap.rs:406-441): halve in s, then in t
        out->texels.insert(out->texels.end(), level.begin(), level.end());
        n_levels++;
        if ((size_t)cw * ch == 1) break;
        std::vector<float> next;
        int nw = cw, nh = ch;
        if (cw > 1) {
            nw = cw / 2;
            next.resize((size_t)channels * nw * ch);
            for (int y = 0; y < ch; y++)
                for (int x = 0; x < nw; x++)
                    for (int k = 0; k < channels; k++)
                        next[(size_t)channels * ((size_t)y * nw + x) + k] =
                            level[(size_t)channels * ((size_t)y * cw + 2 * x) + k] * 0.5f + level[(size_t)channels * ((size_t)y * cw + 2 * x + 1) + k] * 0.5f;
        } else next = level;
        if (ch > 1) {
            nh = ch / 2;
            std::vector<float> n2((size_t)channels * nw * nh);
            for (int y = 0; y < nh; y++)
                for (int x = 0; x < nw; x++)
                    for (int k = 0; k < channels; k++)
                        n2[(size_t)channels * ((size_t)y * nw + x) + k] =
                            next[(size_t)channels * ((size_t)(2 * y) * nw + x) + k] * 0.5f + next[(size_t)channels * ((size_t)(2 * y + 1) * nw + x) + k] * 0.5f;
            next.swap(n2);
        }
        level.swap(next);
        cw = nw; ch = nh;
    }
    out->desc.n_levels = n_levels;
    out->desc.texels = out->texels.data();
}

}  // namespace pth
