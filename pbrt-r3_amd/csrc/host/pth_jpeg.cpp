// pth_jpeg.cpp -- see pth_jpeg.h.
#include "pth_jpeg.h"
#include <algorithm>
#include <cstring>

namespace pth {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t fast_len[512];          // 9-bit prefix -> code length (0: longer than 9 bits)
    uint8_t fast_sym[512];
    int32_t maxcode[18];            // largest code of each length, -1 if none; [17] = sentinel
    int32_t valptr[17], mincode[17];
    uint8_t values[256];
};

bool build_table(const uint8_t counts[16], const uint8_t* symbols, int n_symbols, HuffTable* t) {
    std::memset(t->fast_len, 0, sizeof t->fast_len);
    std::memcpy(t->values, symbols, (size_t)n_symbols);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valptr[l] = k;
        t->mincode[l] = code;
        for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
            if (code >= (1 << l)) return false;
            if (l <= 9) {
                const int lo = code << (9 - l), hi = (code + 1) << (9 - l);
                for (int f = lo; f < hi; f++) { t->fast_len[f] = (uint8_t)l; t->fast_sym[f] = symbols[k]; }
            }
        }
        t->maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    t->present = true;
    return true;
}

// Entropy-coded segment reader: 0xFF 0x00 is a data byte 0xFF, any other marker ends the data (zeros are fed from there on).
struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    void fill() {
        while (n <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                if (*p == 0xff) {
                    if (p + 1 < end && p[1] == 0x00) { byte = 0xff; p += 2; }
                    else hit_marker = true;
                } else byte = *p++;
            } else hit_marker = true;
            acc |= byte << (24 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)(acc >> (32 - k)); }
    void skip(int k) { acc <<= k; n -= k; }
    int get(int k) { if (!k) return 0; const int v = peek(k); skip(k); return v; }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

int decode_symbol(Bits& b, const HuffTable& t) {
    const int f = b.peek(9);
    if (t.fast_len[f]) { b.skip(t.fast_len[f]); return t.fast_sym[f]; }
    int code = b.peek(16);
    for (int l = 10; l <= 16; l++) {
        const int c = code >> (16 - l);
        if (t.maxcode[l] >= 0 && c <= t.maxcode[l] && c >= t.mincode[l]) { b.skip(l); return t.values[t.valptr[l] + c - t.mincode[l]]; }
    }
    return -1;
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// Dequantised coefficients -> 8x8 samples: two passes of a 1-D integer IDCT with 12-bit constants, two extra bits kept between the
// passes, + 128 and clamp at the end.
// Arithmetic wraps modulo 2^32 (a damaged file can carry coefficients far outside the range a real image produces).
struct W32 { int32_t v; };
inline W32 operator+(W32 a, W32 b) { return W32{(int32_t)((uint32_t)a.v + (uint32_t)b.v)}; }
inline W32 operator-(W32 a, W32 b) { return W32{(int32_t)((uint32_t)a.v - (uint32_t)b.v)}; }
inline W32 operator*(W32 a, W32 b) { return W32{(int32_t)((uint32_t)a.v * (uint32_t)b.v)}; }
inline W32 f2f(float x) { return W32{(int32_t)(x * 4096.0f + 0.5f)}; }
struct Idct1D { W32 x0, x1, x2, x3, t0, t1, t2, t3; };
inline Idct1D idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
    W32 p2{s2}, p3{s6};
    W32 p1 = (p2 + p3) * f2f(0.5411961f);
    W32 t2 = p1 + p3 * f2f(-1.847759065f);
    W32 t3 = p1 + p2 * f2f(0.765366865f);
    p2 = W32{s0}; p3 = W32{s4};
    W32 t0 = (p2 + p3) * W32{4096};
    W32 t1 = (p2 - p3) * W32{4096};
    Idct1D r;
    r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
    t0 = W32{s7}; t1 = W32{s5}; t2 = W32{s3}; t3 = W32{s1};
    p3 = t0 + t2;
    W32 p4 = t1 + t3;
    p1 = t0 + t3;
    p2 = t1 + t2;
    const W32 p5 = (p3 + p4) * f2f(1.175875602f);
    t0 = t0 * f2f(0.298631336f);
    t1 = t1 * f2f(2.053119869f);
    t2 = t2 * f2f(3.072711026f);
    t3 = t3 * f2f(1.501321110f);
    p1 = p5 + p1 * f2f(-0.899976223f);
    p2 = p5 + p2 * f2f(-2.562915447f);
    p3 = p3 * f2f(-1.961570560f);
    p4 = p4 * f2f(-0.390180644f);
    r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
    return r;
}
inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idct_block(const int d[64], uint8_t* out, size_t stride) {
    int val[64];
    for (int i = 0; i < 8; i++) {
        const int* c = d + i;
        int* v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            const int dc = (W32{c[0]} * W32{4}).v;
            for (int k = 0; k < 8; k++) v[8 * k] = dc;
        } else {
            Idct1D r = idct_1d(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
            const W32 half{512};
            r.x0 = r.x0 + half; r.x1 = r.x1 + half; r.x2 = r.x2 + half; r.x3 = r.x3 + half;
            v[0] = (r.x0 + r.t3).v >> 10; v[56] = (r.x0 - r.t3).v >> 10;
            v[8] = (r.x1 + r.t2).v >> 10; v[48] = (r.x1 - r.t2).v >> 10;
            v[16] = (r.x2 + r.t1).v >> 10; v[40] = (r.x2 - r.t1).v >> 10;
            v[24] = (r.x3 + r.t0).v >> 10; v[32] = (r.x3 - r.t0).v >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {
        const int* v = val + 8 * i;
        uint8_t* o = out + (size_t)i * stride;
        Idct1D r = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const W32 bias{65536 + (128 << 17)};
        r.x0 = r.x0 + bias; r.x1 = r.x1 + bias; r.x2 = r.x2 + bias; r.x3 = r.x3 + bias;
        o[0] = clamp8((r.x0 + r.t3).v >> 17); o[7] = clamp8((r.x0 - r.t3).v >> 17);
        o[1] = clamp8((r.x1 + r.t2).v >> 17); o[6] = clamp8((r.x1 - r.t2).v >> 17);
        o[2] = clamp8((r.x2 + r.t1).v >> 17); o[5] = clamp8((r.x2 - r.t1).v >> 17);
        o[3] = clamp8((r.x3 + r.t0).v >> 17); o[4] = clamp8((r.x3 - r.t0).v >> 17);
    }
}

// One block of one progressive scan (T.81 annex G): DC first / refinement, AC first / refinement with end-of-band runs.
// Returns nullptr or what is wrong.
const char* progressive_block(Bits& bits, int* blk, int Ss, int Se, int Ah, int Al, const HuffTable& dct, const HuffTable& act, int* pred, int* eobrun) {
    if (Ss == 0) {
        if (Ah == 0) {
            const int t = decode_symbol(bits, dct);
            if (t < 0 || t > 11) return "corrupt entropy-coded data (DC)";
            *pred = (int)((uint32_t)*pred + (uint32_t)(t ? extend(bits.get(t), t) : 0));
            blk[0] = (int)((uint32_t)*pred << Al);
        } else if (bits.get(1)) blk[0] |= 1 << Al;
        return nullptr;
    }
    if (Ah == 0) {
        if (*eobrun > 0) { (*eobrun)--; return nullptr; }
        for (int k = Ss; k <= Se;) {
            const int rs = decode_symbol(bits, act);
            if (rs < 0) return "corrupt entropy-coded data (AC)";
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r < 15) {
                    *eobrun = (1 << r) - 1;
                    if (r) *eobrun += bits.get(r);
                    break;
                }
                k += 16;
                continue;
            }
            k += r;
            if (k > Se) return "corrupt entropy-coded data (run)";
            blk[kZigzag[k]] = extend(bits.get(sz), sz) * (1 << Al);
            k++;
        }
        return nullptr;
    }
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto correct = [&](int* c) {
        if (bits.get(1) && (*c & p1) == 0) *c += *c >= 0 ? p1 : m1;
    };
    int k = Ss;
    if (*eobrun == 0) {
        for (; k <= Se; k++) {
            const int rs = decode_symbol(bits, act);
            if (rs < 0) return "corrupt entropy-coded data (AC)";
            int r = rs >> 4, v = rs & 15;
            if (v) v = bits.get(1) ? p1 : m1;
            else if (r != 15) {
                *eobrun = 1 << r;
                if (r) *eobrun += bits.get(r);
                break;
            }
            do {                                   // pass the coefficients that have a history, and r that do not
                int* c = &blk[kZigzag[k]];
                if (*c != 0) correct(c);
                else if (--r < 0) break;
                k++;
            } while (k <= Se);
            if (v && k <= Se) blk[kZigzag[k]] = v;
        }
    }
    if (*eobrun > 0) {
        for (; k <= Se; k++) {
            int* c = &blk[kZigzag[k]];
            if (*c != 0) correct(c);
        }
        (*eobrun)--;
    }
    return nullptr;
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int dc_table = 0, ac_table = 0, pred = 0;
    int width = 0, height = 0;                 // true size of the plane (ceil of the scaled image size)
    int blocks_w = 0, blocks_h = 0;            // allocated size in blocks (whole MCUs)
    std::vector<uint8_t> plane;                // blocks_w * 8 bytes per row
    std::vector<int> coefs;                    // progressive files: 64 per block, natural order, built up scan by scan
};

// One output row of a component stretched to the image width.  2:1 ratios use a triangle filter (3/4 nearer + 1/4 farther sample),
// other integer ratios repeat samples.
void upsample_row(const Component& c, int hs, int vs, int row, int out_w, uint8_t* out) {
    const size_t stride = (size_t)c.blocks_w * 8;
    const int iw = c.width, ih = c.height;
    if (hs == 1 && vs == 1) { std::memcpy(out, &c.plane[(size_t)row * stride], (size_t)out_w); return; }
    if ((hs == 2 || hs == 1) && (vs == 2 || vs == 1)) {
        const uint8_t *near_row, *far_row = nullptr;
        if (vs == 2) {
            const int rn = row >> 1;
            int rf = (row & 1) ? rn + 1 : rn - 1;
            rf = std::min(std::max(rf, 0), ih - 1);
            near_row = &c.plane[(size_t)rn * stride];
            far_row = &c.plane[(size_t)rf * stride];
        } else near_row = &c.plane[(size_t)row * stride];
        if (hs == 1) {                          // vertical only
            for (int i = 0; i < out_w; i++) out[i] = (uint8_t)((3u * near_row[i] + far_row[i] + 2u) >> 2);
            return;
        }
        std::vector<uint8_t> wide((size_t)iw * 2);
        if (vs == 1) {                          // horizontal only
            const uint8_t* in = near_row;
            if (iw == 1) { wide[0] = wide[1] = in[0]; }
            else {
                wide[0] = in[0];
                wide[1] = (uint8_t)((in[0] * 3u + in[1] + 2u) >> 2);
                for (int i = 1; i < iw - 1; i++) {
                    const uint32_t s = 3u * in[i] + 2u;
                    wide[(size_t)i * 2] = (uint8_t)((s + in[i - 1]) >> 2);
                    wide[(size_t)i * 2 + 1] = (uint8_t)((s + in[i + 1]) >> 2);
                }
                wide[(size_t)(iw - 1) * 2] = (uint8_t)((in[iw - 1] * 3u + in[iw - 2] + 2u) >> 2);
                wide[(size_t)(iw - 1) * 2 + 1] = in[iw - 1];
            }
        } else {                                // both
            if (iw == 1) { wide[0] = wide[1] = (uint8_t)((3u * near_row[0] + far_row[0] + 2u) >> 2); }
            else {
                uint32_t t0 = 3u * near_row[0] + far_row[0], t1 = 3u * near_row[1] + far_row[1];
                wide[0] = (uint8_t)((4u * t0 + 8u) >> 4);
                wide[1] = (uint8_t)((3u * t0 + t1 + 8u) >> 4);
                for (int i = 2; i < iw; i++) {
                    const uint32_t t2 = 3u * near_row[i] + far_row[i];
                    wide[(size_t)i * 2 - 2] = (uint8_t)((3u * t1 + t0 + 8u) >> 4);
                    wide[(size_t)i * 2 - 1] = (uint8_t)((3u * t1 + t2 + 8u) >> 4);
                    t0 = t1; t1 = t2;
                }
                wide[(size_t)iw * 2 - 2] = (uint8_t)((3u * t1 + t0 + 8u) >> 4);
                wide[(size_t)iw * 2 - 1] = (uint8_t)((4u * t1 + 8u) >> 4);
            }
        }
        std::memcpy(out, wide.data(), (size_t)std::min(out_w, iw * 2));
        return;
    }
    const uint8_t* in = &c.plane[(size_t)std::min(row / vs, ih - 1) * stride];
    for (int i = 0; i < out_w; i++) out[i] = in[std::min(i / hs, iw - 1)];
}

inline uint8_t clamp_fixed(int v) { v >>= 20; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int fix20(float x) { return (int)(x * (float)(1 << 20) + 0.5f); }

}  // namespace

bool decode_jpeg(const std::vector<uint8_t>& b, int* width, int* height, int* channels, std::vector<uint8_t>* pixels, std::string* err) {
    auto fail = [&](const std::string& what) { *err = "JPEG: " + what; return false; };
    if (b.size() < 4 || b[0] != 0xff || b[1] != 0xd8) return fail("not a JPEG file");
    size_t pos = 2;
    uint16_t quant[4][64];
    bool have_quant[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    std::vector<Component> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0, adobe_transform = -1;
    bool have_frame = false, jfif = false, progressive = false;
    size_t blocks_done = 0;
    for (;;) {
        while (pos < b.size() && b[pos] != 0xff) pos++;          // tolerate stray bytes between segments
        while (pos < b.size() && b[pos] == 0xff) pos++;
        if (pos >= b.size()) return fail("truncated file (no end-of-image marker)");
        const uint8_t m = b[pos++];
        if (m == 0xd9) break;                                     // EOI
        if (m == 0x00 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;     // stuffed byte, TEM, stray RSTn
        if (pos + 2 > b.size()) return fail("truncated segment");
        const size_t len = ((size_t)b[pos] << 8) | b[pos + 1];
        if (len < 2 || pos + len > b.size()) return fail("truncated segment");
        const uint8_t* s = &b[pos + 2];
        const size_t n = len - 2;
        pos += len;
        if (m == 0xc0 || m == 0xc1 || m == 0xc2) {                // SOF0 / SOF1 / SOF2
            progressive = m == 0xc2;
            if (have_frame) return fail("more than one frame");
            if (n < 6) return fail("bad frame header");
            if (s[0] != 8) return fail("only 8-bit samples are supported");
            H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4];
            const int nc = s[5];
            if (W <= 0 || H <= 0) return fail("zero-sized image (DNL-defined height is not supported)");
            // every 8x8 block of every scan costs at least one bit of entropy-coded data: a header that claims more blocks than
            // the file has bits is damaged, and is turned away before its planes are allocated
            if ((int64_t)W * H > kMaxImagePixels || (int64_t)((W + 7) / 8) * ((H + 7) / 8) > (int64_t)b.size() * 8) return fail("frame size implausible for the file size");
            if (nc == 4) return fail("four-component (CMYK / YCCK) files are not supported");
            if ((nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) return fail("bad component count");
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                Component& c = comps[(size_t)i];
                c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i];
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component parameters");
            }
            if (nc == 1) comps[0].h = comps[0].v = 1;             // a lone component is never interleaved
            for (const Component& c : comps) { hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v); }
            const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (Component& c : comps) {
                if (hmax % c.h || vmax % c.v) return fail("fractional sampling ratios are not supported");
                c.width = (W * c.h + hmax - 1) / hmax;
                c.height = (H * c.v + vmax - 1) / vmax;
                c.blocks_w = mcux * c.h; c.blocks_h = mcuy * c.v;
                c.plane.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 0);
                if (progressive) c.coefs.assign((size_t)c.blocks_w * (size_t)c.blocks_h * 64, 0);
            }
            have_frame = true;
        } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xc7) || (m >= 0xc9 && m <= 0xcb) || (m >= 0xcd && m <= 0xcf))
            return fail("lossless, hierarchical and arithmetic-coded files are not supported");
        else if (m == 0xc4) {                                     // DHT
            size_t at = 0;
            while (at < n) {
                if (at + 17 > n) return fail("bad Huffman table");
                const int cls = s[at] >> 4, id = s[at] & 15;
                int total = 0;
                for (int i = 0; i < 16; i++) total += s[at + 1 + i];
                if (cls > 1 || id > 3 || total > 256 || at + 17 + (size_t)total > n) return fail("bad Huffman table");
                if (!build_table(&s[at + 1], &s[at + 17], total, cls ? &ac[id] : &dc[id])) return fail("bad Huffman table");
                at += 17 + (size_t)total;
            }
        } else if (m == 0xdb) {                                   // DQT (values arrive in zigzag order)
            size_t at = 0;
            while (at < n) {
                const int pq = s[at] >> 4, id = s[at] & 15;
                if (pq > 1 || id > 3 || at + 1 + 64 * (size_t)(pq + 1) > n) return fail("bad quantisation table");
                at++;
                for (int i = 0; i < 64; i++) {
                    quant[id][kZigzag[i]] = pq ? (uint16_t)((s[at] << 8) | s[at + 1]) : s[at];
                    at += (size_t)pq + 1;
                }
                have_quant[id] = true;
            }
        } else if (m == 0xdd) {                                   // DRI
            if (n < 2) return fail("bad restart interval");
            restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xe0) { if (n >= 5 && !std::memcmp(s, "JFIF\0", 5)) jfif = true; }
        else if (m == 0xee) { if (n >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11]; }
        else if (m == 0xda) {                                     // SOS + entropy-coded data
            if (!have_frame) return fail("scan before the frame header");
            if (n < 1) return fail("bad scan header");
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return fail("bad scan header");
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (!progressive) { if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return fail("spectral selection in a sequential file"); }
            else if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1)) return fail("bad progressive scan parameters");
            const bool need_dc = !progressive || Ss == 0, need_ac = !progressive || Ss > 0;
            std::vector<Component*> sc;
            for (int i = 0; i < ns; i++) {
                Component* c = nullptr;
                for (Component& k : comps) if (k.id == s[1 + 2 * i]) c = &k;
                if (!c) return fail("scan names an unknown component");
                c->dc_table = s[2 + 2 * i] >> 4; c->ac_table = s[2 + 2 * i] & 15;
                if (c->dc_table > 3 || c->ac_table > 3 || (need_dc && Ah == 0 && !dc[c->dc_table].present) || (need_ac && !ac[c->ac_table].present))
                    return fail("scan uses a missing Huffman table");
                if (!have_quant[c->tq]) return fail("scan uses a missing quantisation table");
                c->pred = 0;
                sc.push_back(c);
            }
            Bits bits{b.data() + pos, b.data() + b.size()};      // pos may be b.size(): a scan header that ends the file
            int mcus_x, mcus_y;
            if (ns == 1) { mcus_x = (sc[0]->width + 7) / 8; mcus_y = (sc[0]->height + 7) / 8; }
            else { mcus_x = (W + 8 * hmax - 1) / (8 * hmax); mcus_y = (H + 8 * vmax - 1) / (8 * vmax); }
            int until_restart = restart_interval, next_rst = 0, eobrun = 0;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart_interval && until_restart == 0) {
                        // the bit reader stopped at the marker; it must be the expected RSTn
                        const uint8_t* q = bits.p;
                        while (q < bits.end && *q != 0xff) q++;
                        while (q + 1 < bits.end && q[1] == 0xff) q++;
                        if (q + 1 >= bits.end || q[1] != (uint8_t)(0xd0 + next_rst)) return fail("missing restart marker");
                        bits.p = q + 2;
                        bits.reset();
                        next_rst = (next_rst + 1) & 7;
                        until_restart = restart_interval;
                        for (Component* c : sc) c->pred = 0;
                        eobrun = 0;
                    }
                    for (Component* c : sc) {
                        const int bh = ns == 1 ? 1 : c->h, bv = ns == 1 ? 1 : c->v;
                        for (int by = 0; by < bv; by++)
                            for (int bx = 0; bx < bh; bx++) {
                                if (progressive) {
                                    const int block_x = mx * bh + bx, block_y = my * bv + by;
                                    if (block_x >= c->blocks_w || block_y >= c->blocks_h) return fail("block outside the frame");
                                    int* blk = &c->coefs[((size_t)block_y * (size_t)c->blocks_w + (size_t)block_x) * 64];
                                    const char* what = progressive_block(bits, blk, Ss, Se, Ah, Al, dc[c->dc_table], ac[c->ac_table], &c->pred, &eobrun);
                                    if (what) return fail(what);
                                    blocks_done++;
                                    continue;
                                }
                                int coef[64];
                                std::memset(coef, 0, sizeof coef);
                                const int t = decode_symbol(bits, dc[c->dc_table]);
                                if (t < 0 || t > 11) return fail("corrupt entropy-coded data (DC)");
                                const int diff = t ? extend(bits.get(t), t) : 0;
                                c->pred = (int)((uint32_t)c->pred + (uint32_t)diff);
                                const uint16_t* q = quant[c->tq];
                                coef[0] = (int)((uint32_t)c->pred * (uint32_t)q[0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(bits, ac[c->ac_table]);
                                    if (rs < 0) return fail("corrupt entropy-coded data (AC)");
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r != 15) break;                   // end of block
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) return fail("corrupt entropy-coded data (run)");
                                    const int z = kZigzag[k];
                                    coef[z] = (int)((uint32_t)extend(bits.get(sz), sz) * (uint32_t)q[z]);
                                    k++;
                                }
                                const int block_x = mx * bh + bx, block_y = my * bv + by;
                                if (block_x < c->blocks_w && block_y < c->blocks_h)
                                    idct_block(coef, &c->plane[((size_t)block_y * 8 * (size_t)c->blocks_w + (size_t)block_x) * 8], (size_t)c->blocks_w * 8);
                                blocks_done++;
                            }
                    }
                    if (restart_interval) until_restart--;
                }
            // continue after the entropy-coded data: the next marker
            const uint8_t* q = bits.p;
            while (q + 1 < bits.end && !(q[0] == 0xff && q[1] != 0x00 && q[1] != 0xff)) q++;
            pos = (size_t)(q - b.data());
        }
        // every other segment (APPn, COM, ...) is skipped
    }
    if (!have_frame || blocks_done == 0) return fail("no image data");
    if (progressive)
        for (Component& c : comps) {
            if (!have_quant[c.tq]) return fail("missing quantisation table");
            int coef[64];
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++) {
                    const int* blk = &c.coefs[((size_t)by * (size_t)c.blocks_w + (size_t)bx) * 64];
                    for (int k = 0; k < 64; k++) coef[k] = (int)((uint32_t)blk[k] * (uint32_t)quant[c.tq][k]);
                    idct_block(coef, &c.plane[((size_t)by * 8 * (size_t)c.blocks_w + (size_t)bx) * 8], (size_t)c.blocks_w * 8);
                }
        }
    *width = W; *height = H;
    *channels = (int)comps.size();
    pixels->assign((size_t)W * H * comps.size(), 0);
    if (comps.size() == 1) {
        const Component& c = comps[0];
        for (int y = 0; y < H; y++) std::memcpy(&(*pixels)[(size_t)y * W], &c.plane[(size_t)y * c.blocks_w * 8], (size_t)W);
        return true;
    }
    bool ycc = true;
    if (adobe_transform == 0) ycc = false;
    else if (adobe_transform < 0 && !jfif && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
    std::vector<uint8_t> rows[3];
    for (auto& r : rows) r.resize((size_t)W);
    const int cr_r = fix20(1.40200f), cb_g = fix20(0.34414f), cr_g = fix20(0.71414f), cb_b = fix20(1.77200f);
    for (int y = 0; y < H; y++) {
        for (int k = 0; k < 3; k++) upsample_row(comps[(size_t)k], hmax / comps[(size_t)k].h, vmax / comps[(size_t)k].v, y, W, rows[k].data());
        uint8_t* o = &(*pixels)[(size_t)y * W * 3];
        for (int x = 0; x < W; x++) {
            if (!ycc) { o[3 * x] = rows[0][x]; o[3 * x + 1] = rows[1][x]; o[3 * x + 2] = rows[2][x]; continue; }
            const int yy = (int)rows[0][x] * (1 << 20) + (1 << 19), cb = (int)rows[1][x] - 128, cr = (int)rows[2][x] - 128;
            o[3 * x] = clamp_fixed(yy + cr_r * cr);
            o[3 * x + 1] = clamp_fixed(yy - cb_g * cb - cr_g * cr);
            o[3 * x + 2] = clamp_fixed(yy + cb_b * cb);
        }
    }
    return true;
}

}  // namespace pth
