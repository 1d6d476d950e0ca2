// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_sampling.hpp: PCG32, radical inverse, Sobol' sampler, warps, Distribution1D.
//   follows src/core/rng.rs, src/core/lowdiscrepancy/{radical_inverse.rs,sobol/sobol.rs},
//           src/samplers/sobol.rs, src/core/sampler/{sampler,global_sampler}.rs,
//           src/core/sampling/{sampling,distribution}.rs
#pragma once
#include "orc_math.hpp"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <string>
#include <vector>

namespace orc {

// ---- core/rng.rs:8-67
struct RNG {
    uint64_t state, inc;
    RNG() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    explicit RNG(uint64_t initseq) : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) { set_sequence(initseq); }
    void set_sequence(uint64_t initseq) {
        state = 0;
        inc = (initseq << 1) | 1;
        uniform_uint32();
        state += 0x853c49e6748fea9bULL;
        uniform_uint32();
    }
    uint32_t uniform_uint32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18) ^ oldstate) >> 27);
        uint32_t rot = (uint32_t)(oldstate >> 59);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t uniform_uint32_threshold(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        for (;;) {
            uint32_t r = uniform_uint32();
            if (r >= threshold) return r % b;
        }
    }
    Float uniform_float() {
        Float f = (Float)uniform_uint32() * 2.3283064365386963e-10f;
        return fmin_(kOneMinusEpsilon, f);
    }
};

// ---- core/lowdiscrepancy/radical_inverse.rs:23-58 (first primes of primes.rs:1)
inline uint32_t reverse_bits32(uint32_t n) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    return n;
}
inline uint64_t reverse_bits64(uint64_t n) {
    uint64_t n0 = reverse_bits32((uint32_t)n);
    uint64_t n1 = reverse_bits32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}
inline Float radical_inverse_specialized(uint64_t base, uint64_t a) {
    Float inv_base = 1.0f / (Float)base;
    uint64_t reversed_digits = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversed_digits = reversed_digits * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return fmin_((Float)reversed_digits * inv_base_n, kOneMinusEpsilon);
}
inline Float radical_inverse(uint32_t base_index, uint64_t a) {
    static const uint64_t primes[8] = {2, 3, 5, 7, 11, 13, 17, 19};
    if (base_index == 0) return (Float)reverse_bits64(a) * 5.4210108624275222e-20f;
    return radical_inverse_specialized(primes[base_index], a);
}

// ---- Halton machinery (core/lowdiscrepancy/radical_inverse.rs:60-109, primes.rs, sampling.rs:4-15)
// Places where the reference would have panicked and this restatement went on instead (bit 0: a Halton dimension past the prime
// tables).  A test reads and clears it through orc_reference_panics: the HIP path must report the same condition.
inline std::atomic<uint32_t>& reference_panics() { static std::atomic<uint32_t> v{0}; return v; }
struct HaltonTables {
    std::vector<uint64_t> primes;        // PRIMES: the first 1000 primes (primes.rs:1)
    std::vector<uint64_t> prime_sums;    // prefix sums (primes.rs:68 PRIME_SUMS)
    std::vector<uint16_t> perms;         // compute_radical_inverse_permutations with the default-state RNG (halton.rs:12-20)
    HaltonTables() {
        std::vector<bool> comp(8200, false);
        for (uint64_t i = 2; primes.size() < 1000; i++) {
            if (comp[i]) continue;
            primes.push_back(i);
            for (uint64_t j = i * i; j < comp.size(); j += i) comp[j] = true;
        }
        prime_sums.resize(primes.size() + 1, 0);
        for (size_t i = 0; i < primes.size(); i++) prime_sums[i + 1] = prime_sums[i] + primes[i];
        perms.resize(prime_sums.back());
        RNG rng;
        size_t offset = 0;
        for (size_t i = 0; i < primes.size(); i++) {
            size_t len = primes[i];
            for (size_t j = 0; j < len; j++) perms[offset + j] = (uint16_t)j;
            // shuffle_array(&mut perms[offset..], len, 1, rng)
            for (size_t j = 0; j < len; j++) {
                size_t other = j + rng.uniform_uint32_threshold((uint32_t)(len - j));
                std::swap(perms[offset + j], perms[offset + other]);
            }
            offset += len;
        }
    }
    static const HaltonTables& get() { static HaltonTables t; return t; }
};
inline uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, size_t ndigits) {
    uint64_t index = 0;
    for (size_t i = 0; i < ndigits; i++) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
inline Float scrambled_radical_inverse(uint64_t base, const uint16_t* perm, uint64_t a) {
    Float inv_base = 1.0f / (Float)base;
    uint64_t reverse_digits = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reverse_digits = reverse_digits * base + perm[digit];
        inv_base_n *= inv_base;
        a = next;
    }
    return fmin_(inv_base_n * ((Float)reverse_digits + inv_base * (Float)perm[0] / (1.0f - inv_base)), kOneMinusEpsilon);
}
inline uint64_t math_mod(int64_t a, int64_t b) {     // halton.rs:22-29
    int64_t result = a - (a / b) * b;
    return result < 0 ? (uint64_t)(result + b) : (uint64_t)result;
}
inline void extended_gcd(uint64_t a, uint64_t b, int64_t* x, int64_t* y) {   // halton.rs:31-38
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = (int64_t)(a / b), xp, yp;
    extended_gcd(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - (d * yp);
}
inline uint64_t multiplicative_inverse(int64_t a, int64_t n) {
    int64_t x, y;
    extended_gcd((uint64_t)a, (uint64_t)n, &x, &y);
    return math_mod(x, n);
}

// ---- Sobol' tables (data file written by tools/extract_sobol_tables.py)
struct SobolTables {
    uint32_t n_dims = 0, msize = 0, n_vdc = 0, n_inv = 0;
    std::vector<uint32_t> m32;
    std::vector<uint64_t> vdc, inv;
    bool load(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        char magic[8];
        uint32_t hdr[4];
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTSOBOL1", 8) == 0 && std::fread(hdr, 4, 4, f) == 4;
        if (ok) {
            n_dims = hdr[0]; msize = hdr[1]; n_vdc = hdr[2]; n_inv = hdr[3];
            m32.resize((size_t)n_dims * msize);
            vdc.resize((size_t)n_vdc * msize);
            inv.resize((size_t)n_inv * msize);
            ok = std::fread(m32.data(), 4, m32.size(), f) == m32.size() && std::fread(vdc.data(), 8, vdc.size(), f) == vdc.size() &&
                 std::fread(inv.data(), 8, inv.size(), f) == inv.size();
        }
        std::fclose(f);
        return ok;
    }
};

// core/lowdiscrepancy/sobol/sobol.rs:5-32
inline uint64_t sobol_interval_to_index(const SobolTables& T, uint32_t m, uint64_t frame, int32_t px, int32_t py) {
    if (m == 0) return 0;
    uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    for (int c = 0; frame != 0; frame >>= 1, c++)
        if (frame & 1) delta ^= T.vdc[(size_t)(m - 1) * T.msize + c];
    uint64_t b = ((((uint64_t)(uint32_t)px) << m) | (uint64_t)(int64_t)py) ^ delta;
    for (int c = 0; b != 0; b >>= 1, c++)
        if (b & 1) index ^= T.inv[(size_t)(m - 1) * T.msize + c];
    return index;
}
// sobol.rs:39-56: double scale then narrowing to f32 (not pbrt-v3's float multiply)
inline Float sobol_sample_float(const SobolTables& T, int64_t a, uint32_t dimension, uint32_t scramble) {
    uint32_t v = scramble;
    size_t len = T.m32.size();
    size_t i = (size_t)dimension * T.msize;
    if (i > len - 1) i = len - 1;
    while (a != 0) {
        if (a & 1) v ^= T.m32[i];
        a >>= 1;
        i += 1;
        i %= len;
    }
    Float fv = (Float)((double)v * 2.3283064365386963e-10);
    return fmin_(fv, kOneMinusEpsilon);
}

inline uint32_t round_up_pow2(uint32_t v) {
    v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}
inline uint32_t log2int(uint32_t v) { return 31 - (uint32_t)__builtin_clz(v); }

// samplers/sobol.rs:7-190 with core/sampler/global_sampler.rs (no sample arrays are
// requested by PathIntegrator, so array_end_dim == array_start_dim == 5).
struct SobolSampler {
    // kind 0: SobolSampler (samplers/sobol.rs); kind 1: HaltonSampler (samplers/halton.rs:46-273, with the
    // per-pixel offset evaluated as the pure function it is meant to be -- quirk Q9)
    int kind = 0;
    int32_t base_scales[2] = {1, 1}, base_exponents[2] = {0, 0}, sample_stride = 1, mult_inverse[2] = {0, 0};
    bool sample_at_center = false;
    const SobolTables* T = nullptr;
    int32_t bmin[2] = {0, 0}, bmax[2] = {0, 0};
    uint32_t resolution = 0, log2_resolution = 0, spp = 0;
    int32_t pixel[2] = {0, 0};
    uint32_t current_sample = 0;
    uint32_t dimension = 0;
    int64_t interval_sample_index = 0;
    static const uint32_t array_start_dim = 5;
    uint32_t array_end_dim = 5;
    uint32_t array2d_n = 0;          // size of the one requested 2-D sample array (AOIntegrator::new, ao.rs:24-31); 0 = none
    // DirectLightingIntegrator::preprocess ("all" strategy, directlighting.rs:50-65): 2 * lights * maxdepth arrays, the two of light j
    // of size n_j = round_count(light.get_sample_count()) -- round_count is the trait's identity for both global samplers
    // (core/sampler/sampler.rs:41-43; pbrt-v3's Sobol' rounds up to a power of two), get_sample_count the area light's "nsamples".
    // get_2d_array hands the arrays out in request order and returns None once they are used up (then the caller falls back to get_2d).
    uint32_t n_arrays1 = 0, array_next = 0;

    void init_halton(uint32_t samples_per_pixel, const int32_t sb[4], bool at_center) {   // halton.rs:57-112
        kind = 1;
        bmin[0] = sb[0]; bmin[1] = sb[1]; bmax[0] = sb[2]; bmax[1] = sb[3];
        spp = samples_per_pixel;
        sample_at_center = at_center;
        const int32_t res[2] = {bmax[0] - bmin[0], bmax[1] - bmin[1]};
        const int32_t bases[2] = {2, 3};
        for (int i = 0; i < 2; i++) {
            int32_t scale = 1, exp = 0;
            while (scale < std::min(res[i], 128)) { scale *= bases[i]; exp += 1; }
            base_scales[i] = scale;
            base_exponents[i] = exp;
        }
        sample_stride = base_scales[0] * base_scales[1];
        mult_inverse[0] = (int32_t)multiplicative_inverse(base_scales[1], base_scales[0]);
        mult_inverse[1] = (int32_t)multiplicative_inverse(base_scales[0], base_scales[1]);
    }
    void init(const SobolTables* t, uint32_t samples_per_pixel, const int32_t sb[4]) {
        kind = 0;
        T = t;
        bmin[0] = sb[0]; bmin[1] = sb[1]; bmax[0] = sb[2]; bmax[1] = sb[3];
        uint32_t dx = (uint32_t)(bmax[0] - bmin[0]), dy = (uint32_t)(bmax[1] - bmin[1]);
        resolution = round_up_pow2(dx > dy ? dx : dy);
        log2_resolution = log2int(resolution);
        spp = round_up_pow2(samples_per_pixel);
    }
    int64_t get_index_for_sample(int64_t sample_num) const {
        if (kind == 1) {                              // halton.rs:115-147
            int64_t offset = 0;
            if (sample_stride > 1) {
                const uint64_t pm[2] = {math_mod(pixel[0], 128), math_mod(pixel[1], 128)};
                const uint64_t bases[2] = {2, 3};
                for (int i = 0; i < 2; i++) {
                    uint64_t dim_offset = inverse_radical_inverse(bases[i], pm[i], (size_t)base_exponents[i]);
                    offset += (int64_t)(dim_offset * (uint64_t)((sample_stride / base_scales[i]) * mult_inverse[i]));
                }
                offset %= sample_stride;
            }
            return offset + sample_num * sample_stride;
        }
        return (int64_t)sobol_interval_to_index(*T, log2_resolution, (uint64_t)sample_num, pixel[0] - bmin[0], pixel[1] - bmin[1]);
    }
    Float sample_dimension(int64_t index, uint32_t dim) const {
        if (kind == 1) {                              // halton.rs:149-162
            if (sample_at_center && (dim == 0 || dim == 1)) return 0.5f;
            if (dim == 0) return radical_inverse(0, (uint64_t)(index >> base_exponents[0]));
            if (dim == 1) return radical_inverse(1, (uint64_t)(index / base_scales[1]));
            const HaltonTables& H = HaltonTables::get();
            if (dim >= (uint32_t)H.primes.size()) {          // the reference panics here (PRIME_SUMS holds 1000 entries; halton.rs:103-108):
                reference_panics().fetch_or(1u);               // noted (orc_reference_panics), and the render goes on with the last dimension
                dim = (uint32_t)H.primes.size() - 1u;
            }
            return scrambled_radical_inverse(H.primes[dim], &H.perms[H.prime_sums[dim]], (uint64_t)index);
        }
        Float s = sobol_sample_float(*T, index, dim, 0);
        if (dim == 0 || dim == 1) {
            s = s * (Float)resolution + (Float)bmin[dim];
            s = clampf(s - (Float)pixel[dim], 0.0f, kOneMinusEpsilon);
            s = s - std::trunc(s);  // Float::fract
        }
        return s;
    }
    void start_pixel(int32_t x, int32_t y) {
        pixel[0] = x; pixel[1] = y;
        current_sample = 0;
        dimension = 0;
        interval_sample_index = get_index_for_sample(0);
        array_end_dim = array_start_dim + (array2d_n ? 2u : 0u) + 2u * n_arrays1;      // sobol.rs:43-45, halton.rs:176-178
        array_next = 0;
    }
    // get_2d_array(n) for one of the n_arrays1 requested arrays: false = None (all handed out).  Element k for the current pixel sample is
    // sample_array2d[i][n * current + k] (base_sampler.rs:59-70), which start_pixel filled from sample number n * current + k -- and
    // every 2-D array is filled from the SAME pair of dimensions: start_pixel computes `dim` once, before the loop over the arrays, and
    // never advances it (sobol.rs:60-75, halton.rs:193-208; pbrt-v3 has `dim += 2` there).  With "strategy all" every light's u_light and
    // u_scattering arrays therefore hold the same points -- restated as is (quirk Q23).
    bool get_2d_array(uint32_t n, std::vector<V2>* out) {
        if (array_next >= n_arrays1) return false;
        array_next++;
        out->resize(n);
        for (uint32_t k = 0; k < n; k++) {
            const int64_t index = get_index_for_sample((int64_t)current_sample * (int64_t)n + (int64_t)k);
            (*out)[k] = V2(sample_dimension(index, array_start_dim), sample_dimension(index, array_start_dim + 1));
        }
        return true;
    }
    // Element k of get_2d_array(n) for the current pixel sample: start_pixel fills sample_array2d[0][j] from sample number j at
    // dimensions (5, 6) for j < n * spp (sobol.rs:60-75), get_2d_array slices [n * current, n * current + n) (base_sampler.rs:59-70).
    V2 array2d(uint32_t k) const {
        const int64_t index = get_index_for_sample((int64_t)current_sample * (int64_t)array2d_n + (int64_t)k);
        return V2(sample_dimension(index, array_start_dim), sample_dimension(index, array_start_dim + 1));
    }
    bool start_next_sample() {
        dimension = 0;
        array_next = 0;
        interval_sample_index = get_index_for_sample((int64_t)current_sample + 1);
        current_sample += 1;
        return current_sample < spp;
    }
    bool set_sample_number(uint32_t n) {
        dimension = 0;
        array_next = 0;
        interval_sample_index = get_index_for_sample(n);
        current_sample = n;
        return current_sample < spp;
    }
    Float get_1d() {
        if (dimension >= array_start_dim && dimension < array_end_dim) dimension = array_end_dim;
        Float x = sample_dimension(interval_sample_index, dimension);
        dimension += 1;
        return x;
    }
    V2 get_2d() {
        if (dimension + 1 >= array_start_dim && dimension < array_end_dim) dimension = array_end_dim;
        Float x = sample_dimension(interval_sample_index, dimension);
        Float y = sample_dimension(interval_sample_index, dimension + 1);
        dimension += 2;
        return V2(x, y);
    }
};

// ---- core/sampling/sampling.rs:109-172
inline V2 uniform_sample_triangle(V2 u) {
    Float su0 = std::sqrt(u.x);
    return V2(1.0f - su0, u.y * su0);
}
inline V2 concentric_sample_disk(V2 u) {
    V2 uo = u * 2.0f - V2(1.0f, 1.0f);
    if (uo.x == 0.0f && uo.y == 0.0f) return V2(0.0f, 0.0f);
    if (std::fabs(uo.x) > std::fabs(uo.y)) {
        Float r = uo.x;
        Float theta = kPiOver4 * (uo.y / uo.x);
        return V2(r * std::cos(theta), r * std::sin(theta));
    }
    Float r = uo.y;
    Float theta = kPiOver2 - kPiOver4 * (uo.x / uo.y);
    return V2(r * std::cos(theta), r * std::sin(theta));
}
inline V3 cosine_sample_hemisphere(V2 u) {
    V2 d = concentric_sample_disk(u);
    Float z = std::sqrt(fmax_(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
inline V3 uniform_sample_hemisphere(V2 u) {             // sampling.rs:85-90
    Float z = u.x;
    Float r = std::sqrt(fmax_(0.0f, 1.0f - z * z));
    Float phi = 2.0f * kPi * u.y;
    return V3(r * std::cos(phi), r * std::sin(phi), z);
}
inline V3 uniform_sample_sphere(V2 u) {
    Float z = 1.0f - 2.0f * u.x;
    Float r = std::sqrt(fmax_(0.0f, 1.0f - z * z));
    Float phi = 2.0f * kPi * u.y;
    return V3(r * std::cos(phi), r * std::sin(phi), z);
}
inline Float power_heuristic(int nf, Float f_pdf, int ng, Float g_pdf) {
    Float f = (Float)nf * f_pdf, g = (Float)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

// ---- core/sampling/distribution.rs:3-107
struct Distribution1D {
    std::vector<Float> func, cdf;
    Float func_int = 0.0f, inv_count = 0.0f;
    Distribution1D() {}
    explicit Distribution1D(const std::vector<Float>& f) {
        size_t n = f.size();
        func = f;
        cdf.assign(n + 1, 0.0f);
        for (size_t i = 1; i < n + 1; i++) cdf[i] = cdf[i - 1] + func[i - 1] / (Float)n;
        func_int = cdf[n];
        if (func_int == 0.0f) {
            for (size_t i = 1; i < n + 1; i++) cdf[i] = (Float)i / (Float)n;
        } else {
            for (size_t i = 1; i < n + 1; i++) cdf[i] /= func_int;
        }
        inv_count = 1.0f / (Float)n;
    }
    static size_t find_interval_cdf(const std::vector<Float>& cdf, Float u) {
        size_t first = 0, len = cdf.size();
        while (len > 0) {
            size_t half = len >> 1, middle = first + half;
            if (cdf[middle] <= u) {
                first = middle + 1;
                len -= half + 1;
            } else {
                len = half;
            }
        }
        size_t idx = first == 0 ? 0 : first - 1;
        if (idx > cdf.size() - 2) return cdf.size() - 2;
        return idx;
    }
    // (offset, pdf, remapped)
    size_t sample_discrete(Float u, Float* pdf, Float* remapped) const {
        size_t offset = find_interval_cdf(cdf, u);
        Float cdf0 = cdf[offset], cdf1 = cdf[offset + 1];
        *pdf = func_int > 0.0f ? func[offset] * inv_count / func_int : 0.0f;
        if (remapped) *remapped = (u - cdf0) / (cdf1 - cdf0);
        return offset;
    }
    // (value, pdf, offset)
    Float sample_continuous(Float u, Float* pdf, size_t* off) const {
        size_t offset = find_interval_cdf(cdf, u);
        Float cdf0 = cdf[offset], cdf1 = cdf[offset + 1];
        Float du = u - cdf0;
        Float span = cdf1 - cdf0;
        if (span > 0.0f) du /= span;
        *pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
        if (off) *off = offset;
        return ((Float)offset + du) * inv_count;
    }
    Float discrete_pdf(size_t index) const { return func[index] / (func_int * (Float)func.size()); }
};

}  // namespace orc
