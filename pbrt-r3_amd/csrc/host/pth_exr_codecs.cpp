// pth_exr_codecs.cpp -- see pth_exr_codecs.h.  Written from the OpenEXR file-format description of the two schemes:
//   PIZ    u16 lo, u16 hi (first / last non-zero byte of a 65 536-bit "value is used" map), the map bytes lo..hi, i32 n, then n bytes
//          of Huffman-coded 16-bit symbols.  The symbols are the block's samples, one plane per channel (all lines of a channel
//          together; a float is two interleaved planes), each plane 2-D Haar-wavelet transformed on the ranks of the used values.
//   PXR24  one zlib stream; per line and channel the most significant bytes of all samples first, then the next bytes ...
//          (2 byte planes for half, 3 for float: the low mantissa byte is not stored), each sample a difference to its left neighbour.
#include "pth_exr_codecs.h"
#include <zlib.h>
#include <algorithm>
#include <cstring>

namespace pth {
namespace {

// ------------------------------------------------------------------ Huffman part of PIZ
// 20-byte header (u32 first symbol, u32 last symbol, u32 table bytes, u32 data bits, u32 0), the code lengths of first..last packed
// in 6-bit fields (59..62 = 2..5 zero lengths, 63 + 8 bits = 6..261 zero lengths), then the codes, most significant bit first.
// Codes are canonical: within a length by symbol, and the longest length owns the smallest code values.  The last symbol is not a
// sample: it is followed by an 8-bit count and repeats the previous sample that many times.
constexpr uint32_t kHufSymbols = 65537, kMaxCodeLen = 58, kFastBits = 12;

struct BitReader {
    const uint8_t* p;
    size_t n_bytes;
    uint64_t peek(uint64_t bit, unsigned n) const {          // n <= 32 bits starting at `bit`, zeros past the end
        uint64_t v = 0;
        const size_t byte = (size_t)(bit >> 3);
        for (size_t k = 0; k < 6; k++) v = (v << 8) | (byte + k < n_bytes ? p[byte + k] : 0u);
        return (v >> (48u - (unsigned)(bit & 7u) - n)) & ((1ull << n) - 1ull);
    }
};

bool huf_uncompress(const uint8_t* in, size_t n_in, uint16_t* out, size_t n_out, std::string* err) {
    auto bad = [&](const char* what) { *err = std::string("EXR: corrupt PIZ block (") + what + ")"; return false; };
    if (n_in == 0) return n_out == 0 ? true : bad("no data");
    if (n_in < 20) return bad("short Huffman header");
    auto u32 = [&](size_t at) { return (uint32_t)in[at] | (uint32_t)in[at + 1] << 8 | (uint32_t)in[at + 2] << 16 | (uint32_t)in[at + 3] << 24; };
    const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
    if (im >= kHufSymbols || iM >= kHufSymbols || im > iM) return bad("symbol range");
    BitReader br{in + 20, n_in - 20};
    std::vector<uint8_t> len(kHufSymbols, 0);
    uint64_t bit = 0;
    const uint64_t avail = (uint64_t)br.n_bytes * 8u;
    for (uint32_t s = im; s <= iM; s++) {
        if (bit + 6 > avail) return bad("code table");
        const uint32_t l = (uint32_t)br.peek(bit, 6);
        bit += 6;
        uint32_t zeros = 0;
        if (l == 63) {
            if (bit + 8 > avail) return bad("code table");
            zeros = (uint32_t)br.peek(bit, 8) + 6u;
            bit += 8;
        } else if (l >= 59) zeros = l - 59u + 2u;
        else len[s] = (uint8_t)l;
        if (zeros) {
            if (s + zeros > iM + 1u) return bad("zero run in the code table");
            s += zeros - 1u;
        }
    }
    const size_t table_bytes = (size_t)((bit + 7) >> 3);
    br.p += table_bytes;
    br.n_bytes -= table_bytes;
    if ((uint64_t)n_bits > (uint64_t)br.n_bytes * 8u) return bad("bit count");
    // canonical codes
    uint64_t count[kMaxCodeLen + 1] = {0}, base[kMaxCodeLen + 1] = {0};
    for (uint32_t s = im; s <= iM; s++) count[len[s]]++;
    count[0] = 0;
    {
        uint64_t c = 0;
        for (uint32_t l = kMaxCodeLen; l > 0; l--) { const uint64_t nc = (c + count[l]) >> 1; base[l] = c; c = nc; }
    }
    size_t first[kMaxCodeLen + 2] = {0};
    for (uint32_t l = 1; l <= kMaxCodeLen; l++) first[l + 1] = first[l] + (size_t)count[l];
    std::vector<uint32_t> by_len(first[kMaxCodeLen + 1]);
    struct Fast { uint32_t sym; uint32_t len; };
    std::vector<Fast> fast((size_t)1 << kFastBits, Fast{0, 0});
    {
        size_t fill[kMaxCodeLen + 1];
        for (uint32_t l = 0; l <= kMaxCodeLen; l++) fill[l] = first[l];
        for (uint32_t s = im; s <= iM; s++) {
            const uint32_t l = len[s];
            if (!l) continue;
            const uint64_t code = base[l] + (fill[l] - first[l]);
            by_len[fill[l]++] = s;
            if (l < 64 && (code >> l) != 0) return bad("code lengths");          // over-subscribed table
            if (l <= kFastBits) {
                const size_t lo = (size_t)(code << (kFastBits - l)), hi = (size_t)((code + 1) << (kFastBits - l));
                if (hi > fast.size()) return bad("code lengths");
                for (size_t k = lo; k < hi; k++) fast[k] = Fast{s, l};
            }
        }
    }
    const uint32_t run_symbol = iM;
    size_t o = 0;
    bit = 0;
    while (bit < n_bits) {
        uint32_t sym, l;
        const Fast f = fast[(size_t)br.peek(bit, kFastBits)];
        if (f.len) { sym = f.sym; l = f.len; }
        else {
            uint64_t code = br.peek(bit, kFastBits);
            l = kFastBits;
            bool found = false;
            while (l < kMaxCodeLen) {
                code = (code << 1) | br.peek(bit + l, 1);
                l++;
                if (count[l] && code >= base[l] && code - base[l] < count[l]) { sym = by_len[first[l] + (size_t)(code - base[l])]; found = true; break; }
            }
            if (!found) return bad("unknown code");
        }
        bit += l;
        if (bit > n_bits) return bad("code past the end");
        if (sym == run_symbol) {
            if (bit + 8 > n_bits) return bad("run past the end");
            const size_t run = (size_t)br.peek(bit, 8);
            bit += 8;
            if (o == 0 || o + run > n_out) return bad("run");
            const uint16_t v = out[o - 1];
            for (size_t k = 0; k < run; k++) out[o++] = v;
        } else {
            if (o >= n_out) return bad("too many samples");
            out[o++] = (uint16_t)sym;
        }
    }
    return o == n_out ? true : bad("too few samples");
}

// ------------------------------------------------------------------ wavelet part of PIZ
// Inverse of the 2-D transform: per 2x2 cell (a, b / c, d at stride p) average-and-difference pairs are undone along x then y, from
// the coarsest level down.  Two pair codings: values below 2^14 use plain signed average / difference, otherwise arithmetic modulo 2^16.
inline void undo14(uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    *a = (uint16_t)(int16_t)ai;
    *b = (uint16_t)(int16_t)(ai - hs);
}
inline void undo16(uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    *b = (uint16_t)bb;
    *a = (uint16_t)aa;
}
void wavelet_undo(uint16_t* in, ptrdiff_t nx, ptrdiff_t ox, ptrdiff_t ny, ptrdiff_t oy, uint32_t max_value) {
    const bool small = max_value < (1u << 14);
    const ptrdiff_t n = std::min(nx, ny);
    ptrdiff_t p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    ptrdiff_t p2 = p;
    p >>= 1;
    auto undo = [&](uint16_t l, uint16_t h, uint16_t* a, uint16_t* b) { if (small) undo14(l, h, a, b); else undo16(l, h, a, b); };
    while (p >= 1) {
        uint16_t* py = in;
        uint16_t* const ey = in + oy * (ny - p2);
        const ptrdiff_t oy1 = oy * p, oy2 = oy * p2, ox1 = ox * p, ox2 = ox * p2;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py;
            uint16_t* const ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
                uint16_t i00, i01, i10, i11;
                undo(*px, *p10, &i00, &i10);
                undo(*p01, *p11, &i01, &i11);
                undo(i00, i01, px, p01);
                undo(i10, i11, p10, p11);
            }
            if (nx & p) {                       // odd column left over at this level
                uint16_t* p10 = px + oy1;
                uint16_t i00;
                undo(*px, *p10, &i00, p10);
                *px = i00;
            }
        }
        if (ny & p) {                           // odd line left over
            uint16_t* px = py;
            uint16_t* const ex = py + ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1;
                uint16_t i00;
                undo(*px, *p01, &i00, p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

}  // namespace

bool exr_unpack_piz(const uint8_t* src, size_t n_src, const std::vector<ExrPlane>& planes, size_t width, size_t lines, std::vector<uint8_t>* raw, std::string* err) {
    size_t words_per_line = 0;
    for (const ExrPlane& pl : planes) words_per_line += width * (size_t)pl.words;
    const size_t n_words = words_per_line * lines;
    if (raw->size() != n_words * 2) { *err = "EXR: PIZ block size mismatch"; return false; }
    if (n_src < 4) { *err = "EXR: corrupt PIZ block (header)"; return false; }
    const uint32_t lo = (uint32_t)src[0] | (uint32_t)src[1] << 8, hi = (uint32_t)src[2] | (uint32_t)src[3] << 8;
    if (hi >= 8192) { *err = "EXR: corrupt PIZ block (value map)"; return false; }
    std::vector<uint8_t> used(8192, 0);
    size_t pos = 4;
    if (lo <= hi) {
        if (pos + (hi - lo + 1) > n_src) { *err = "EXR: corrupt PIZ block (value map)"; return false; }
        std::memcpy(&used[lo], src + pos, hi - lo + 1);
        pos += hi - lo + 1;
    }
    std::vector<uint16_t> value_of_rank(65536, 0);          // rank -> value; zero is always rank 0
    uint32_t n_used = 0;
    for (uint32_t v = 0; v < 65536; v++)
        if (v == 0 || (used[v >> 3] & (1u << (v & 7u)))) value_of_rank[n_used++] = (uint16_t)v;
    const uint32_t max_value = n_used - 1;
    if (pos + 4 > n_src) { *err = "EXR: corrupt PIZ block (length)"; return false; }
    const int32_t n_huf = (int32_t)((uint32_t)src[pos] | (uint32_t)src[pos + 1] << 8 | (uint32_t)src[pos + 2] << 16 | (uint32_t)src[pos + 3] << 24);
    pos += 4;
    if (n_huf < 0 || pos + (size_t)n_huf > n_src) { *err = "EXR: corrupt PIZ block (length)"; return false; }
    std::vector<uint16_t> tmp(n_words);
    if (!huf_uncompress(src + pos, (size_t)n_huf, tmp.data(), n_words, err)) return false;
    std::vector<size_t> start(planes.size());
    size_t at = 0;
    for (size_t c = 0; c < planes.size(); c++) {
        start[c] = at;
        const size_t w = (size_t)planes[c].words;
        for (size_t j = 0; j < w; j++) wavelet_undo(tmp.data() + at + j, (ptrdiff_t)width, (ptrdiff_t)w, (ptrdiff_t)lines, (ptrdiff_t)(width * w), max_value);
        at += width * w * lines;
    }
    for (size_t i = 0; i < n_words; i++) tmp[i] = value_of_rank[tmp[i]];
    uint8_t* dst = raw->data();
    for (size_t y = 0; y < lines; y++)
        for (size_t c = 0; c < planes.size(); c++) {
            const size_t n = width * (size_t)planes[c].words;
            std::memcpy(dst, tmp.data() + start[c] + y * n, n * 2);          // host and file are both little-endian
            dst += n * 2;
        }
    return true;
}

bool exr_unpack_pxr24(const uint8_t* src, size_t n_src, const std::vector<ExrPlane>& planes, size_t width, size_t lines, std::vector<uint8_t>* raw, std::string* err) {
    size_t stored_per_line = 0, bytes_per_line = 0;
    for (const ExrPlane& pl : planes) {
        stored_per_line += width * (pl.words == 1 ? 2u : 3u);
        bytes_per_line += width * (size_t)pl.words * 2u;
    }
    if (raw->size() != bytes_per_line * lines) { *err = "EXR: PXR24 block size mismatch"; return false; }
    std::vector<uint8_t> tmp(stored_per_line * lines);
    uLongf got = (uLongf)tmp.size();
    if (uncompress(tmp.data(), &got, src, (uLong)n_src) != Z_OK || got != tmp.size()) { *err = "EXR: corrupt PXR24 block"; return false; }
    const uint8_t* in = tmp.data();
    uint8_t* dst = raw->data();
    for (size_t y = 0; y < lines; y++)
        for (const ExrPlane& pl : planes) {
            uint32_t v = 0;
            if (pl.words == 1) {
                const uint8_t *b0 = in, *b1 = in + width;
                in += 2 * width;
                for (size_t x = 0; x < width; x++) {
                    v += ((uint32_t)b0[x] << 8) | (uint32_t)b1[x];
                    const uint16_t hv = (uint16_t)v;
                    std::memcpy(dst, &hv, 2);
                    dst += 2;
                }
            } else {
                const uint8_t *b0 = in, *b1 = in + width, *b2 = in + 2 * width;
                in += 3 * width;
                for (size_t x = 0; x < width; x++) {
                    v += ((uint32_t)b0[x] << 24) | ((uint32_t)b1[x] << 16) | ((uint32_t)b2[x] << 8);
                    std::memcpy(dst, &v, 4);
                    dst += 4;
                }
            }
        }
    return true;
}

}  // namespace pth
