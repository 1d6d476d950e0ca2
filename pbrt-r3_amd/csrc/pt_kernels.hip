// pt_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the wavefront
// path tracer.  HBM-bound pointer chasing: no MFMA anywhere.  Layout rules:
//   * one 4-wide BVH node = one 128-byte line, fetched as 8 x dwordx4 per lane
//   * triangle records are 48 bytes (3 x dwordx4) in leaf order
//   * per-lane traversal stack lives in LDS ([entry][lane] => conflict-free at any
//     depth mix, since 64 lanes x 4 B = two full bank rows), spilling to HBM past
//     PT_LDS_STACK entries
//   * queues are SoA index lists; compaction uses wave ballot + one atomic per wave
//   * persistent grids pull 64-item tickets from a device counter
// Arithmetic follows the reference operation by operation (-ffp-contract=off):
//   traversal  src/accelerators/bvh/accel/qbvh/qbvh_x86.rs:26-67, :230-343
//   triangle   src/shapes/triangle.rs:226-577
//   shading    src/integrators/path.rs:61-241, src/core/integrator/sample_lights.rs:129-176,:330-453
//   sampler    src/samplers/sobol.rs, src/core/lowdiscrepancy/sobol/sobol.rs:5-56,
//              src/samplers/halton.rs, src/core/lowdiscrepancy/radical_inverse.rs
//   camera     src/cameras/perspective.rs:121-183, src/core/transform/transform.rs:184-282
//   film       src/core/film/film_tile.rs:84-183, film.rs:219-241, :440-484
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "pt_device.h"
#include "pt_device_math.h"
#include "pt_bxdf.h"
#include "pt_sphere.h"
#include "pt_texture.h"
#include <type_traits>
#include "pt_lobes.h"
#include "pt_kernels.h"
#include "../../include/pbrtgpu.h"

#ifndef PT_SHADE_PF_DEEP
#define PT_SHADE_PF_DEEP 0       // 1: k_shade's one-iteration-ahead prefetch also covers the ray, the throughput and the leaf record
#endif
#ifndef PT_LOBES_IN_LDS
#define PT_LOBES_IN_LDS 1       // the lobe-list kernels keep the lobes of the material a wave's lanes share in LDS (shade_body): mixed materials 927 -> 942 Mrays/s
#endif
#ifndef PT_TEX_PROGS_GLOBAL
#define PT_TEX_PROGS_GLOBAL 1       // the texture programs' offsets are read from the material table, not from the per-lane copy (indexed by the job loop: scratch): crown-class 795 -> 813 Mrays/s
#endif
#ifndef PT_LIGHTS_IN_LDS
#define PT_LIGHTS_IN_LDS 8       // the shading kernels keep the light records of scenes with at most this many lights in LDS (0: never); RT1M 1 075 -> 1 081 Mrays/s
#endif
#ifndef PT_TEX_NOUNROLL
#define PT_TEX_NOUNROLL 0
#endif
#ifndef PT_TOUCH_VARIANT
#define PT_TOUCH_VARIANT 3       // k_trace_far built with PT_NODE_COOP 0 (its first form): which pushed children node_step_lean touches (see there)
#endif

// ============================================================ Sobol' sampler
PT_DEV uint64_t sobol_interval_to_index(const PtSobol& sb, uint64_t frame, int32_t px, int32_t py) {
    const uint32_t m = sb.log2_resolution;
    if (m == 0) return 0;
    uint64_t index = frame << (m << 1);
    uint64_t delta = 0;
    for (int c = 0; frame != 0; frame >>= 1, c++)
        if (frame & 1) delta ^= sb.vdc[c];
    uint64_t b = ((((uint64_t)(uint32_t)px) << m) | (uint64_t)(int64_t)py) ^ delta;
    for (int c = 0; b != 0; b >>= 1, c++)
        if (b & 1) index ^= sb.vdc_inv[c];
    return index;
}
// sobol_sample_float (sobol.rs:39-56): v = XOR of the generator-matrix columns selected by the set
// bits of the sample index.  The XOR is linear, so the host folds the 52 columns of each dimension
// into seven 256-entry tables indexed by one byte of the index: 4 independent loads (issued
// together) replace a ~15-iteration dependent-load loop.  Integer arithmetic: identical result.
PT_DEV uint32_t sobol_bits(const PtSobol& sb, uint64_t a, uint32_t dim) {
    uint32_t v = 0;
    if (dim < sb.n_tab_dims && (a >> 56) == 0) {
        const uint32_t* T = sb.bytetab + (size_t)dim * (7u * 256u);
        uint32_t lo = (uint32_t)a;
        v = T[lo & 255u] ^ T[256u + ((lo >> 8) & 255u)] ^ T[512u + ((lo >> 16) & 255u)] ^ T[768u + (lo >> 24)];
        uint32_t hi = (uint32_t)(a >> 32);
        if (hi != 0) v ^= T[1024u + (hi & 255u)] ^ T[1280u + ((hi >> 8) & 255u)] ^ T[1536u + ((hi >> 16) & 255u)];
        return v;
    }
    uint32_t base = dim * 52u;
    if (base > sb.m32_len - 1u) base = sb.m32_len - 1u;
    while (a != 0) {
        uint32_t bit = (uint32_t)__builtin_ctzll(a);
        uint32_t i = base + bit;
        if (i >= sb.m32_len) i %= sb.m32_len;
        v ^= sb.m32[i];
        a &= a - 1;
    }
    return v;
}
PT_DEV float sobol_sample_float(const PtSobol& sb, uint64_t a, uint32_t dim) {
    uint32_t v = sobol_bits(sb, a, dim);
    float fv = (float)((double)v * 2.3283064365386963e-10);
    return fminf(fv, PT_ONE_MINUS_EPS);
}
// ============================================================ Halton sampler (samplers/halton.rs)
// a / base for a < 2^32 is mulhi(a, floor(2^64/base)+1) exactly (Lemire & Kaser, "Faster remainder by
// direct computation", theorem 1 with N = 32); larger indices take the 64-bit division.
PT_DEV uint64_t div_small(uint64_t a, uint32_t base, uint64_t magic) { return (a >> 32) == 0 ? __umul64hi(a, magic) : a / base; }
// scrambled_radical_inverse (radical_inverse.rs:83-109); perm == nullptr is radical_inverse_specialized (:36-48)
PT_DEV float halton_radical_inverse(uint32_t base, uint64_t magic, const uint16_t* perm, uint64_t a) {
    const float inv_base = 1.0f / (float)base;
    uint64_t rev = 0;
    float inv_base_n = 1.0f;
    if ((a >> 32) == 0) {
        // every index a frame of up to 2^32 / stride samples produces: the same digits with 32-bit arithmetic.  mulhi64(a, magic) for a
        // 32-bit a is the top word of a * magic_hi + (a * magic_lo >> 32) -- three 32-bit multiplies and a carry instead of the four
        // multiply pairs of the general 64 x 64 product; the digit and the next quotient stay in one register each.
        uint32_t a32 = (uint32_t)a;
        const uint32_t mh = (uint32_t)(magic >> 32), ml = (uint32_t)magic;
        // (Round 4 tried the digits first and all their permutation look-ups in one batch -- up to 14 loads in flight per dimension instead of
        // one per trip of this loop: bit-identical, and 0.7 % SLOWER on RT1M under Halton, 1 036 -> 1 029 Mrays/s; the fixed 14-digit unroll costs
        // more than the waits it removes.  Not kept.)
        while (a32 != 0) {
            const uint32_t lo = a32 * mh, carry_in = __umulhi(a32, ml);
            const uint32_t next = __umulhi(a32, mh) + ((lo + carry_in) < lo ? 1u : 0u);
            const uint32_t digit = a32 - next * base;
            rev = rev * base + (perm ? (uint32_t)perm[digit] : digit);
            inv_base_n *= inv_base;
            a32 = next;
        }
        a = 0;
    }
    while (a != 0) {
        uint64_t next = div_small(a, base, magic);
        uint32_t digit = (uint32_t)(a - next * base);
        rev = rev * base + (perm ? (uint32_t)perm[digit] : digit);
        inv_base_n *= inv_base;
        a = next;
    }
    if (!perm) return fminf((float)rev * inv_base_n, PT_ONE_MINUS_EPS);
    return fminf(inv_base_n * ((float)rev + inv_base * (float)perm[0] / (1.0f - inv_base)), PT_ONE_MINUS_EPS);
}
// HaltonSampler::get_index_for_sample (halton.rs:115-147), evaluated per path instead of cached per pixel
PT_DEV uint64_t halton_index_for_sample(const PtSobol& sb, uint64_t sample_num, int32_t px, int32_t py) {
    uint64_t offset = 0;
    if (sb.h_stride > 1) {
        const int32_t pix[2] = {px, py};
        const uint32_t bases[2] = {2u, 3u};
        for (int i = 0; i < 2; i++) {
            int32_t r = pix[i] % 128;
            uint32_t inv = (uint32_t)(r < 0 ? r + 128 : r);       // math_mod (halton.rs:22-29)
            uint32_t dim_offset = 0;                                // inverse_radical_inverse (radical_inverse.rs:69-81)
            for (uint32_t k = 0; k < sb.h_exp[i]; k++) {
                uint32_t digit = inv % bases[i];
                inv /= bases[i];
                dim_offset = dim_offset * bases[i] + digit;
            }
            offset += (uint64_t)dim_offset * sb.h_mul[i];
        }
        offset %= sb.h_stride;
    }
    return offset + sample_num * sb.h_stride;
}
PT_DEV float halton_sample_dimension(const PtSobol& sb, uint64_t index, uint32_t dim) {   // halton.rs:149-162
    if (dim < 2) {
        if (sb.h_center) return 0.5f;
        if (dim == 0) return (float)__brevll(index >> sb.h_exp[0]) * 5.4210108624275222e-20f;
        return halton_radical_inverse(3u, 0x5555555555555556ull, nullptr, index / sb.h_scale1);
    }
    // The reference panics here (PRIME_SUMS holds 1000 entries, halton.rs:103-108).  The launch finishes on the last dimension; the condition
    // itself is caught where a sampler's dimension count is written back (k_rec_enter / k_rec_next: PtRec::panic) and, for the path
    // integrator -- 5 + 8 dimensions per vertex --, by the upload refusing maxdepth > 124 under this sampler (5 + 8 x 124 - 4 = 993 < 1000; 125 reaches 1001): a check in here costs every
    // shading kernel two or three registers at its tightest point (k_shade_general went from 256 to 258 and lost its second wave).
    if (dim >= sb.h_n_dims) dim = sb.h_n_dims - 1;
    uint4 e = sb.h_dims[dim];
    return halton_radical_inverse(e.x, (uint64_t)e.z | ((uint64_t)e.w << 32), sb.h_perms + e.y, index);
}
// GlobalSampler::get_index_for_sample for the scene's sampler; (px, py) is the absolute pixel
PT_DEV uint64_t sampler_index(const PtScene& sc, uint64_t sample_num, int32_t px, int32_t py) {
    if (sc.sobol.kind == PT_SAMPLER_HALTON) return halton_index_for_sample(sc.sobol, sample_num, px, py);
    return sobol_interval_to_index(sc.sobol, sample_num, px - sc.film.sample_bounds[0], py - sc.film.sample_bounds[1]);
}
// SobolSampler::sample_dimension (samplers/sobol.rs:167-185)
#ifdef PT_SAMPLER_NOINLINE
__device__ __noinline__ float sample_dimension(
#else
PT_DEV float sample_dimension(
#endif
const PtScene& sc, uint64_t index, uint32_t dim, int32_t px, int32_t py) {
    if (sc.sobol.kind == PT_SAMPLER_HALTON) return halton_sample_dimension(sc.sobol, index, dim);
    float s = sobol_sample_float(sc.sobol, index, dim);
    if (dim == 0 || dim == 1) {
        int32_t bmin = sc.film.sample_bounds[dim];
        int32_t pix = dim == 0 ? px : py;
        s = s * (float)sc.sobol.resolution + (float)bmin;
        s = clampf(s - (float)pix, 0.0f, PT_ONE_MINUS_EPS);
        s = s - truncf(s);
    }
    return s;
}
struct Sampler {
    uint64_t index;
    uint32_t dim;
    int32_t px, py;
    // no sample arrays are requested by PathIntegrator => array_end_dim == array_start_dim == 5,
    // and the skip tests of get_1d / get_2d (sobol.rs:95-141) never fire except get_2d at dim 4.
    PT_DEV float get_1d(const PtScene& sc) { float x = sample_dimension(sc, index, dim, px, py); dim += 1; return x; }
    PT_DEV V2 get_2d(const PtScene& sc) {
        float x = sample_dimension(sc, index, dim, px, py);
        float y = sample_dimension(sc, index, dim + 1, px, py);
        dim += 2;
        return mk2(x, y);
    }
};
// A path vertex draws at most eight dimensions (light choice, u_light, u_scattering, the continuation's u, Russian roulette), each a
// pure function of (index, dimension).  k_shade evaluates all eight as soon as it knows the path's index -- their table loads then
// overlap the reconstruction of the hit instead of stalling the wave five times -- and the sampler hands them out by position.
#define PT_PRE_DIMS 8
struct SamplerPre {
    Sampler s;
    float q0, q1, q2, q3, q4, q5, q6, q7;      // scalars, not an array: a select chain over an array becomes an indexed scratch access
    uint32_t dim0;
    bool have;               // false: every draw is evaluated where it is made (Halton: a digit loop per dimension, no loads to overlap)
    PT_DEV void begin(const PtScene& sc) {
        dim0 = s.dim;
        q0 = q1 = q2 = q3 = q4 = q5 = q6 = q7 = 0.0f;
        const PtSobol& sb = sc.sobol;
        // the byte-table form of sobol_bits for eight consecutive dimensions at once: every load is issued before the first is used
        have = sb.kind != PT_SAMPLER_HALTON && dim0 >= 2u && dim0 + PT_PRE_DIMS <= sb.n_tab_dims && (s.index >> 56) == 0;
        if (have) {
            const uint32_t lo = (uint32_t)s.index, hi = (uint32_t)(s.index >> 32);
            const uint32_t* T = sb.bytetab + (size_t)dim0 * (7u * 256u);
            const uint32_t i0 = lo & 255u, i1 = 256u + ((lo >> 8) & 255u), i2 = 512u + ((lo >> 16) & 255u), i3 = 768u + (lo >> 24);
#define PT_PRE_LOAD(k) const uint32_t a##k = T[k * 1792u + i0], b##k = T[k * 1792u + i1], c##k = T[k * 1792u + i2], d##k = T[k * 1792u + i3];
            PT_PRE_LOAD(0) PT_PRE_LOAD(1) PT_PRE_LOAD(2) PT_PRE_LOAD(3) PT_PRE_LOAD(4) PT_PRE_LOAD(5) PT_PRE_LOAD(6) PT_PRE_LOAD(7)
#undef PT_PRE_LOAD
            uint32_t v0 = a0 ^ b0 ^ c0 ^ d0, v1 = a1 ^ b1 ^ c1 ^ d1, v2 = a2 ^ b2 ^ c2 ^ d2, v3 = a3 ^ b3 ^ c3 ^ d3, v4 = a4 ^ b4 ^ c4 ^ d4, v5 = a5 ^ b5 ^ c5 ^ d5,
                     v6 = a6 ^ b6 ^ c6 ^ d6, v7 = a7 ^ b7 ^ c7 ^ d7;
            if (__ballot(hi != 0u)) {        // indices of 2^32 and more (a byte of zero selects no column: its table entry is 0)
                const uint32_t j0 = 1024u + (hi & 255u), j1 = 1280u + ((hi >> 8) & 255u), j2 = 1536u + ((hi >> 16) & 255u);
#define PT_PRE_HI(k) v##k ^= T[k * 1792u + j0] ^ T[k * 1792u + j1] ^ T[k * 1792u + j2];
                PT_PRE_HI(0) PT_PRE_HI(1) PT_PRE_HI(2) PT_PRE_HI(3) PT_PRE_HI(4) PT_PRE_HI(5) PT_PRE_HI(6) PT_PRE_HI(7)
#undef PT_PRE_HI
            }
#define PT_PRE_F(v) fminf((float)((double)(v) * 2.3283064365386963e-10), PT_ONE_MINUS_EPS)
            q0 = PT_PRE_F(v0); q1 = PT_PRE_F(v1); q2 = PT_PRE_F(v2); q3 = PT_PRE_F(v3); q4 = PT_PRE_F(v4); q5 = PT_PRE_F(v5); q6 = PT_PRE_F(v6); q7 = PT_PRE_F(v7);
#undef PT_PRE_F
        }
    }
    static PT_DEV float pick8(uint32_t k, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {     // by value: nothing to index
        float r = a0;
        r = k == 1u ? a1 : r; r = k == 2u ? a2 : r; r = k == 3u ? a3 : r; r = k == 4u ? a4 : r; r = k == 5u ? a5 : r; r = k == 6u ? a6 : r; r = k == 7u ? a7 : r;
        return r;
    }
    PT_DEV float pick(uint32_t k) const { return pick8(k, q0, q1, q2, q3, q4, q5, q6, q7); }
    PT_DEV float get_1d(const PtScene& sc) {
        if (!have || s.dim - dim0 >= PT_PRE_DIMS) return s.get_1d(sc);
        float x = pick(s.dim - dim0);
        s.dim += 1;
        return x;
    }
    PT_DEV V2 get_2d(const PtScene& sc) {
        if (!have || s.dim - dim0 + 1u >= PT_PRE_DIMS) return s.get_2d(sc);
        V2 r = mk2(pick(s.dim - dim0), pick(s.dim - dim0 + 1u));
        s.dim += 2;
        return r;
    }
};

// ============================================================ camera
PT_DEV V3 xform_point(const float* m, V3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.0f) return mk3(xp, yp, zp);
    return mk3(xp / wp, yp / wp, zp / wp);
}
PT_DEV V3 xform_vector(const float* m, V3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
PT_DEV void generate_camera_ray(const PtScene& sc, V2 p_film, V2 u_lens, V3* o_out, V3* d_out) {
    V3 p_camera = xform_point(sc.cam.raster_to_camera, mk3(p_film.x, p_film.y, 0.0f));
    V3 o = mk3(0.0f, 0.0f, 0.0f);
    V3 d = normalize(p_camera);
    if (sc.cam.lens_radius > 0.0f) {
        V2 dl = concentric_sample_disk(u_lens);
        V2 p_lens = mk2(dl.x * sc.cam.lens_radius, dl.y * sc.cam.lens_radius);
        float ft = sc.cam.focal_distance / d.z;
        V3 p_focus = o + d * ft;
        o = mk3(p_lens.x, p_lens.y, 0.0f);
        d = normalize(p_focus - o);
    }
    // Transform::transform_ray: nudge the origin along d by the transform's rounding error bound
    const float* m = sc.cam.camera_to_world;
    V3 op = xform_point(m, o);
    float xa = fabsf(m[0] * o.x) + fabsf(m[1] * o.y) + fabsf(m[2] * o.z) + fabsf(m[3]);
    float ya = fabsf(m[4] * o.x) + fabsf(m[5] * o.y) + fabsf(m[6] * o.z) + fabsf(m[7]);
    float za = fabsf(m[8] * o.x) + fabsf(m[9] * o.y) + fabsf(m[10] * o.z) + fabsf(m[11]);
    V3 o_err = PT_GAMMA(3.0f) * mk3(xa, ya, za);
    V3 dd = xform_vector(m, d);
    float ls = length_squared(dd);
    if (ls > 0.0f) {
        float dt = dot(vabs(dd), o_err) / ls;
        op = op + dd * dt;
    }
    *o_out = op;
    *d_out = dd;
}

// The camera ray's offset rays (PerspectiveCamera::generate_ray_differential, perspective.rs:146-171, transformed as
// transform_ray_differential does, transform.rs:284-297) after render_tile's scale_differentials(1 / sqrt(spp))
// (sampler.rs:218,233; ray_differential.rs:26-35).  ro / rd: the main ray as k_gen made it.
PT_DEV void camera_differentials(const PtScene& sc, V2 p_film, V2 u_lens, V3 ro, V3 rd, RayDiffs& out) {
    const float* r2c = sc.cam.raster_to_camera;
    V3 p_camera = xform_point(r2c, mk3(p_film.x, p_film.y, 0.0f));
    V3 c0 = xform_point(r2c, mk3(0.0f, 0.0f, 0.0f));
    V3 dx_camera = xform_point(r2c, mk3(1.0f, 0.0f, 0.0f)) - c0;
    V3 dy_camera = xform_point(r2c, mk3(0.0f, 1.0f, 0.0f)) - c0;
    V3 rxo = mk3(0.0f, 0.0f, 0.0f), ryo = rxo, rxd, ryd;
    if (sc.cam.lens_radius > 0.0f) {
        V2 dl = concentric_sample_disk(u_lens);
        V2 p_lens = mk2(dl.x * sc.cam.lens_radius, dl.y * sc.cam.lens_radius);
        V3 dx = normalize(p_camera + dx_camera);
        float ftx = sc.cam.focal_distance / dx.z;
        V3 pfx = mk3(0.0f, 0.0f, 0.0f) + (ftx * dx);
        rxo = mk3(p_lens.x, p_lens.y, 0.0f);
        rxd = normalize(pfx - rxo);
        V3 dy = normalize(p_camera + dy_camera);
        float fty = sc.cam.focal_distance / dy.z;
        V3 pfy = mk3(0.0f, 0.0f, 0.0f) + (fty * dy);
        ryo = mk3(p_lens.x, p_lens.y, 0.0f);
        ryd = normalize(pfy - ryo);
    } else {
        rxd = normalize(p_camera + dx_camera);
        ryd = normalize(p_camera + dy_camera);
    }
    const float* m = sc.cam.camera_to_world;
    rxo = xform_point(m, rxo); ryo = xform_point(m, ryo);
    rxd = xform_vector(m, rxd); ryd = xform_vector(m, ryd);
    const float scale = sqrtf(1.0f / (float)sc.sobol.spp);
    out.rx_o = ro + (rxo - ro) * scale;
    out.ry_o = ro + (ryo - ro) * scale;
    out.rx_d = rd + (rxd - rd) * scale;
    out.ry_d = rd + (ryd - rd) * scale;
}

// ============================================================ ray / triangle
struct RayPre {          // per-ray constants of the watertight test, hoisted out of the per-triangle code
    V3 o, d;
    int kx, ky, kz;
    float sx, sy, sz;
    V3 dperm;
};
PT_DEV float sel3(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
PT_DEV void ray_precompute(RayPre& r, V3 o, V3 d) {
    r.o = o; r.d = d;
    V3 a = vabs(d);
    int kz = (a.x > a.y) ? ((a.x > a.z) ? 0 : 2) : ((a.y > a.z) ? 1 : 2);   // max_dimension (misc.rs:35-50)
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    r.kx = kx; r.ky = ky; r.kz = kz;
    r.dperm = mk3(sel3(d, kx), sel3(d, ky), sel3(d, kz));
    r.sx = -r.dperm.x / r.dperm.z;
    r.sy = -r.dperm.y / r.dperm.z;
    r.sz = 1.0f / r.dperm.z;
}
struct TriHit { float t, b0, b1, b2; };
// triangle.rs:240-347 / :466-571 (intersect and intersect_p share this front half).
// Written as straight-line predicated code: in a wave some lane nearly always survives each of
// the reference's early-outs, so branching on them only adds SALU work; the values computed are
// the reference's, the early-outs become terms of one boolean.  Only the f64 re-evaluation of
// the edge functions (needed when one is exactly zero) stays behind a branch.
// tri_core evaluates everything that does not depend on the ray's current t_max; tri_accept is the one
// comparison that does (t_scaled against t_max * det, triangle.rs:297-303).  Sequential leaf loops call
// them back to back (= tri_test); the distributed leaf phase runs tri_core for all triangles of a leaf in
// parallel and applies tri_accept in leaf order with the t_max each triangle would have seen.
struct TriCore { float t_scaled, det, t, b0, b1, b2; };
PT_DEV bool tri_core(const RayPre& r, V3 p0, V3 p1, V3 p2, uint32_t flags, TriCore& h) {
    bool ok = true;
    if (flags & PT_TRI_ONE_SIDED) {
        V3 n = cross(p0 - p2, p1 - p2);
        if (flags & PT_TRI_FLIP) n = n * -1.0f;
        ok = !(dot(n, r.d) >= 0.0f);
    }
    V3 q0 = p0 - r.o, q1 = p1 - r.o, q2 = p2 - r.o;
    float p0x = sel3(q0, r.kx), p0y = sel3(q0, r.ky), p0z = sel3(q0, r.kz);
    float p1x = sel3(q1, r.kx), p1y = sel3(q1, r.ky), p1z = sel3(q1, r.kz);
    float p2x = sel3(q2, r.kx), p2y = sel3(q2, r.ky), p2z = sel3(q2, r.kz);
    p0x += r.sx * p0z; p0y += r.sy * p0z;
    p1x += r.sx * p1z; p1y += r.sy * p1z;
    p2x += r.sx * p2z; p2y += r.sy * p2z;
    float e0 = p1x * p2y - p1y * p2x;
    float e1 = p2x * p0y - p2y * p0x;
    float e2 = p0x * p1y - p0y * p1x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double a0 = (double)p2x * (double)p1y, b0 = (double)p2y * (double)p1x;
        e0 = (float)(b0 - a0);
        double a1 = (double)p0x * (double)p2y, b1 = (double)p0y * (double)p2x;
        e1 = (float)(b1 - a1);
        double a2 = (double)p1x * (double)p0y, b2 = (double)p1y * (double)p0x;
        e2 = (float)(b2 - a2);
    }
    bool any_neg = (e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f);
    bool any_pos = (e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f);
    ok &= !(any_neg & any_pos);
    float det = e0 + e1 + e2;
    ok &= !(det == 0.0f);
    p0z *= r.sz; p1z *= r.sz; p2z *= r.sz;
    float t_scaled = e0 * p0z + e1 * p1z + e2 * p2z;
    float inv_det = 1.0f / det;
    float t = t_scaled * inv_det;
    float max_zt = max3(fabsf(p0z), fabsf(p1z), fabsf(p2z));
    float delta_z = PT_GAMMA(3.0f) * max_zt;
    float max_xt = max3(fabsf(p0x), fabsf(p1x), fabsf(p2x));
    float max_yt = max3(fabsf(p0y), fabsf(p1y), fabsf(p2y));
    float delta_x = PT_GAMMA(5.0f) * (max_xt + max_zt);
    float delta_y = PT_GAMMA(5.0f) * (max_yt + max_zt);
    float delta_e = 2.0f * (PT_GAMMA(2.0f) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max3(fabsf(e0), fabsf(e1), fabsf(e2));
    float delta_t = 3.0f * (PT_GAMMA(3.0f) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    ok &= !(t <= delta_t);
    h.t_scaled = t_scaled; h.det = det; h.t = t;
    h.b0 = e0 * inv_det; h.b1 = e1 * inv_det; h.b2 = e2 * inv_det;
    return ok;
}
PT_DEV bool tri_accept(float t_scaled, float det, float t_max) {
    float tmd = t_max * det;
    bool rej_neg = (det < 0.0f) & ((t_scaled >= 0.0f) | (t_scaled < tmd));
    bool rej_pos = (det > 0.0f) & ((t_scaled <= 0.0f) | (t_scaled > tmd));
    return !(rej_neg | rej_pos);
}
PT_DEV bool tri_test(const RayPre& r, V3 p0, V3 p1, V3 p2, uint32_t flags, float t_max, TriHit& h) {
    TriCore c;
    bool ok = tri_core(r, p0, p1, p2, flags, c);
    ok &= tri_accept(c.t_scaled, c.det, t_max);
    h.t = c.t; h.b0 = c.b0; h.b1 = c.b1; h.b2 = c.b2;
    return ok;
}

struct TriVerts { V3 p0, p1, p2; uint32_t prim, flags; };
PT_DEV TriVerts load_tri(const PtTri* tris, uint32_t rec) {
    const float4* q = reinterpret_cast<const float4*>(tris) + (size_t)rec * 3;
    float4 a = q[0], b = q[1], c = q[2];
    TriVerts t;
    t.p0 = mk3(a.x, a.y, a.z); t.prim = __float_as_uint(a.w);
    t.p1 = mk3(b.x, b.y, b.z); t.flags = __float_as_uint(b.w);
    t.p2 = mk3(c.x, c.y, c.z);
    return t;
}

// ============================================================ BVH traversal
// _mm_max_ps / _mm_min_ps semantics: second operand when either is NaN
PT_DEV float sse_max(float a, float b) { return a > b ? a : b; }
PT_DEV float sse_min(float a, float b) { return a < b ? a : b; }

struct TravCtx {
    uint32_t* lds;           // &stack[0][tid]; entry e at lds[e * PT_BLOCK]
    uint32_t* spill;         // &spill[gtid]; entry e (>= PT_LDS_STACK) at spill[(e - PT_LDS_STACK) * spill_stride]
    uint32_t spill_stride;
    uint32_t spill_depth;
    uint32_t n_nodes, n_tris;
    uint32_t overflow;
    uint32_t lane_base;      // pooled-leaf kernels: LDS byte address of this lane's slot 0 (see FsStack)
    uint32_t top_lds, top_bytes;   // pooled-leaf kernels: LDS byte address of the cached top of the tree; 128 x the number of nodes cached
    uint32_t n_nodes_lds;          // node visits served from that copy
    // pooled-leaf kernels, staged node fetch (node_round_staged): LDS addresses
    uint32_t stage_wave;     // this wave's 8 KB staging area (wave-uniform)
    uint32_t stage_own;      // this lane's staged node: stage_wave + (lane >> 3) * 1024 + (lane & 7) * 128
    uint32_t idx_pub, idx_ld;// where the lane publishes its node index / reads the eight indices it loads for
    PT_DEV uint32_t stage_chunk(uint32_t q) const { return stage_own + ((q ^ (threadIdx.x & 7u)) << 4); }
};
// The stack lives in LDS (entries 0..PT_LDS_STACK-1, then the HBM spill area); `sp` is its depth and `top` a register
// copy of entry sp-1 (PT_EMPTY_REF when empty), so a pop has its reference at once and the read that refreshes `top`
// overlaps the node fetch.  The root is never written: nothing can lie above it when it is popped.
PT_DEV void stk_push(TravCtx& c, uint32_t& top, uint32_t& sp, uint32_t v) {
    if (sp < PT_LDS_STACK) c.lds[sp * PT_BLOCK] = v;
    else if (sp - PT_LDS_STACK < c.spill_depth) c.spill[(size_t)(sp - PT_LDS_STACK) * c.spill_stride] = v;
    else { c.overflow = 1; return; }
    sp++;
    top = v;
}
// Same effect as `if (en) stk_push(...)` while sp stays inside the LDS part, without branches: the value is always
// written to the slot above the stack and only counts if enabled (5 VALU instructions per push; k_trace is VALU-bound).
PT_DEV void stk_push_lds(TravCtx& c, uint32_t& top, uint32_t& sp, uint32_t v, bool en) {
    c.lds[sp * PT_BLOCK] = v;
    sp += en ? 1u : 0u;
    top = en ? v : top;
}
PT_DEV uint32_t stk_pop(TravCtx& c, uint32_t& top, uint32_t& sp) {      // precondition: top != PT_EMPTY_REF (sp >= 1)
    uint32_t v = top;
    sp--;
    if (sp > PT_LDS_STACK) {                       // rare: the new top lives in the HBM spill area
        top = c.spill[(size_t)(sp - 1u - PT_LDS_STACK) * c.spill_stride];
    } else {
        uint32_t below = sp > 0 ? sp - 1 : 0;
        uint32_t e = c.lds[below * PT_BLOCK];
        top = sp > 0 ? e : PT_EMPTY_REF;
    }
    return v;
}

// Bounds3f::intersect_p on the root (bounds3.rs:154-163, intersect.rs:46-65): NaN-ignoring max/min.
PT_DEV bool box_root_test(const float* bmin, const float* bmax, V3 o, V3 idir, uint32_t sbits, float t_max, float& tmin, float& tmax) {
    float t0 = 0.0f, t1 = t_max;
    float lo, hi;
    lo = (sbits & 1) ? bmax[0] : bmin[0]; hi = (sbits & 1) ? bmin[0] : bmax[0];
    t0 = fmaxf(t0, (lo - o.x) * idir.x); t1 = fminf(t1, (hi - o.x) * idir.x);
    lo = (sbits & 2) ? bmax[1] : bmin[1]; hi = (sbits & 2) ? bmin[1] : bmax[1];
    t0 = fmaxf(t0, (lo - o.y) * idir.y); t1 = fminf(t1, (hi - o.y) * idir.y);
    lo = (sbits & 4) ? bmax[2] : bmin[2]; hi = (sbits & 4) ? bmin[2] : bmax[2];
    t0 = fmaxf(t0, (lo - o.z) * idir.z); t1 = fminf(t1, (hi - o.z) * idir.z);
    if (t0 <= t1) { tmin = t0; tmax = t1; return true; }
    return false;
}
PT_DEV bool root_test(const PtScene& sc, V3 o, V3 idir, uint32_t sbits, float t_max, float& tmin, float& tmax) {
    return box_root_test(sc.wb_min, sc.wb_max, o, idir, sbits, t_max, tmin, tmax);
}

// One node: 4 slab tests (test_aabb) + ORDER_TABLE in closed form:
// children are pushed so that pops visit {0,1} before {2,3} iff the ray is non-negative
// along axis_top, 0 before 1 iff non-negative along axis_left, 2 before 3 iff along axis_right.
//
// EXACT selects the slab arithmetic.  _mm_max_ps(a,b) differs from IEEE maxNum only when an
// operand is NaN, and a NaN can only come from 0*inf, i.e. from a ray with an infinite (or NaN)
// reciprocal direction component.  Such rays (flagged once per ray) take the compare/select form
// that reproduces SSE exactly; all others use v_max_f32 / v_min_f32, which then give the same values.
template <bool EXACT>
PT_DEV void visit_node(const PtNode* nodes, uint32_t ni, V3 o, V3 idir, uint32_t sbits, float tmin, float tmax, TravCtx& c, uint32_t& top, uint32_t& sp) {
    // near / far planes per axis are picked by address (bmin row at +16*axis, bmax row 48 bytes later)
    // instead of loading both and selecting 24 registers
    const char* nb = reinterpret_cast<const char*>(nodes) + (size_t)(ni & PT_REF_INDEX_MASK) * 128u;
    const uint32_t ox = (sbits & 1u) ? 48u : 0u, oy = (sbits & 2u) ? 48u : 0u, oz = (sbits & 4u) ? 48u : 0u;
    float4 nx = *reinterpret_cast<const float4*>(nb + ox), fx = *reinterpret_cast<const float4*>(nb + (48u - ox));
    float4 ny = *reinterpret_cast<const float4*>(nb + 16u + oy), fy = *reinterpret_cast<const float4*>(nb + 16u + (48u - oy));
    float4 nz = *reinterpret_cast<const float4*>(nb + 32u + oz), fz = *reinterpret_cast<const float4*>(nb + 32u + (48u - oz));
    uint4 ch = *reinterpret_cast<const uint4*>(nb + 96u);
    uint32_t axes = *reinterpret_cast<const uint32_t*>(nb + 112u);
    uint32_t mask = 0;
#define PT_SLAB(k, C)                                                        \
    {                                                                        \
        float a = tmin, b = tmax;                                            \
        if (EXACT) {                                                         \
            a = sse_max(a, (nx.C - o.x) * idir.x); b = sse_min(b, (fx.C - o.x) * idir.x); \
            a = sse_max(a, (ny.C - o.y) * idir.y); b = sse_min(b, (fy.C - o.y) * idir.y); \
            a = sse_max(a, (nz.C - o.z) * idir.z); b = sse_min(b, (fz.C - o.z) * idir.z); \
        } else {                                                             \
            a = fmaxf(a, (nx.C - o.x) * idir.x); b = fminf(b, (fx.C - o.x) * idir.x); \
            a = fmaxf(a, (ny.C - o.y) * idir.y); b = fminf(b, (fy.C - o.y) * idir.y); \
            a = fmaxf(a, (nz.C - o.z) * idir.z); b = fminf(b, (fz.C - o.z) * idir.z); \
        }                                                                    \
        if (b >= a) mask |= (1u << k);                                       \
    }
    PT_SLAB(0, x) PT_SLAB(1, y) PT_SLAB(2, z) PT_SLAB(3, w)
#undef PT_SLAB
    mask &= (axes >> 8) & 15u;          // drop empty slots (their boxes are all-zero as in the reference)
    uint32_t s_top = (sbits >> (axes & 3)) & 1, s_left = (sbits >> ((axes >> 2) & 3)) & 1, s_right = (sbits >> ((axes >> 4) & 3)) & 1;
    // push order = reverse visit order: the far pair first, inside a pair the far child first
    uint32_t l0 = s_left ? ch.x : ch.y, l1 = s_left ? ch.y : ch.x;
    uint32_t l0b = s_left ? 1u : 2u, l1b = s_left ? 2u : 1u;
    uint32_t r0 = s_right ? ch.z : ch.w, r1 = s_right ? ch.w : ch.z;
    uint32_t r0b = s_right ? 4u : 8u, r1b = s_right ? 8u : 4u;
    uint32_t c0 = s_top ? l0 : r0, c1 = s_top ? l1 : r1, c2 = s_top ? r0 : l0, c3 = s_top ? r1 : l1;
    uint32_t b0 = s_top ? l0b : r0b, b1 = s_top ? l1b : r1b, b2 = s_top ? r0b : l0b, b3 = s_top ? r1b : l1b;
    if (__builtin_expect(__ballot(sp + 4u > PT_LDS_STACK) == 0ull, 1)) {   // whole wave stays inside the LDS part
        stk_push_lds(c, top, sp, c0, (mask & b0) != 0);
        stk_push_lds(c, top, sp, c1, (mask & b1) != 0);
        stk_push_lds(c, top, sp, c2, (mask & b2) != 0);
        stk_push_lds(c, top, sp, c3, (mask & b3) != 0);
    } else {
        if (mask & b0) stk_push(c, top, sp, c0);
        if (mask & b1) stk_push(c, top, sp, c1);
        if (mask & b2) stk_push(c, top, sp, c2);
        if (mask & b3) stk_push(c, top, sp, c3);
    }
}

// Per-lane traversal state.  A ray is a small state machine advanced one unit of work at a time:
// a node visit or a leaf visit.
struct LaneRay {
    V3 o, idir;
    V3 d;                        // read by the sphere test only: dead in the triangle-only kernels
    RayPre rp;
    float tmin, tmax, ray_tmax;
    uint32_t sbits;              // bits 0-2: direction signs, bit 3: needs the EXACT slab form
    uint32_t sp, top;            // top == PT_EMPTY_REF: traversal finished
    // k_trace / k_trace_sph_dist (the pooled-leaf kernels) keep the stack as an LDS byte address instead of `sp` (see FsStack) and
    // carry the per-ray constants of the lean node visit
    uint32_t sa, sa_limit;       // LDS address of the top entry's slot; the lean visit is allowed while sa < sa_limit
    uint32_t o_nx, o_fx, o_ny, o_fy, o_nz, o_fz;   // byte offsets of the near / far plane rows inside a node (by the direction signs)
    uint32_t m_t, m_l, m_r;      // this ray's bit in the node's three order tables (PtNode::order_lut)
    int32_t best;
    uint32_t best_inst;          // INST kernels: instance index + 1 of `best` (0 = a world primitive)
    uint32_t cur_inst;           // k_trace_inst: instance index + 1 of the object tree this ray is walking right now (0 = the world's)
};
PT_DEV void ray_begin(const PtScene& sc, LaneRay& r, V3 o, V3 d, float t_max) {
    r.o = o;
    r.d = d;
    r.idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.sbits = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    if (!(fabsf(r.idir.x) < PT_INF) || !(fabsf(r.idir.y) < PT_INF) || !(fabsf(r.idir.z) < PT_INF)) r.sbits |= 8u;
    r.best = -1;
    r.best_inst = 0;
    r.cur_inst = 0;
    r.ray_tmax = t_max;
    r.sp = 0;
    r.top = PT_EMPTY_REF;
    ray_precompute(r.rp, o, d);
    if (root_test(sc, o, r.idir, r.sbits, t_max, r.tmin, r.tmax)) { r.top = sc.root_ref; r.sp = 1; }
}
// The same for the pooled-leaf kernels: `sa` stack (root kept in `top` only: nothing lies above it when it is popped) and the
// per-ray constants of node_step_lean.
PT_DEV void ray_begin_fs(const PtScene& sc, LaneRay& r, const TravCtx& c, V3 o, V3 d, float t_max) {
    ray_begin(sc, r, o, d, t_max);
    r.sa = c.lane_base + (r.top != PT_EMPTY_REF ? PT_SLOT : 0u);
    if (r.idir.x == 0.0f || r.idir.y == 0.0f || r.idir.z == 0.0f) r.sbits |= 8u;      // infinite direction component: 0 * inf in the slabs
    // rows of a node: chunks 0..2 = bmin x / y / z, 3..5 = bmax x / y / z; near / far by the direction signs; as LDS addresses of
    // this lane's staged node (chunk q sits at q ^ own_i, see node_round_staged)
    const uint32_t ox = (r.sbits & 1u) ? 3u : 0u, oy = (r.sbits & 2u) ? 3u : 0u, oz = (r.sbits & 4u) ? 3u : 0u;
#if PT_NODE_STAGED
    r.o_nx = c.stage_chunk(ox); r.o_fx = c.stage_chunk(3u - ox);
    r.o_ny = c.stage_chunk(1u + oy); r.o_fy = c.stage_chunk(4u - oy);
    r.o_nz = c.stage_chunk(2u + oz); r.o_fz = c.stage_chunk(5u - oz);
#else       // byte offsets of the rows inside the node in HBM
    r.o_nx = ox << 4; r.o_fx = (3u - ox) << 4;
    r.o_ny = (1u + oy) << 4; r.o_fy = (4u - oy) << 4;
    r.o_nz = (2u + oz) << 4; r.o_fz = (5u - oz) << 4;
#endif
    const uint32_t oct = r.sbits & 7u;
    r.m_t = 1u << oct; r.m_l = 256u << oct; r.m_r = 65536u << oct;
    r.sa_limit = (r.sbits & 8u) ? 0u : c.lane_base + (PT_FS_SLOTS - 3u) * PT_SLOT;
}
PT_DEV bool ray_done(const LaneRay& r) { return r.top == PT_EMPTY_REF; }
PT_DEV bool ray_wants_tri(const LaneRay& r) { return r.top != PT_EMPTY_REF && (r.top & PT_LEAF_BIT) != 0; }
PT_DEV bool ray_wants_node(const LaneRay& r) { return (r.top & PT_LEAF_BIT) == 0; }   // PT_EMPTY_REF has the leaf bit
// Pop a node: 4 slab tests, ordered pushes.
PT_DEV void ray_step_node(const PtScene& sc, LaneRay& r, TravCtx& c) {
    uint32_t ref = stk_pop(c, r.top, r.sp);
    c.n_nodes++;
    if (r.sbits & 8u) visit_node<true>(sc.nodes, ref, r.o, r.idir, r.sbits, r.tmin, r.tmax, c, r.top, r.sp);
    else visit_node<false>(sc.nodes, ref, r.o, r.idir, r.sbits, r.tmin, r.tmax, c, r.top, r.sp);
}
// One leaf (entering it from the top of the stack): its 1..max_node_prims triangle records are
// fetched two per round trip and tested in order.  any_hit: stop at the first accepted triangle
// (sets best, empties the stack).
// A leaf record that stands for a sphere (SPH kernels only): Sphere::intersect / intersect_p, t from its EFloat root.
PT_DEV bool sphere_rec_test(const PtScene& sc, const TriVerts& tv, const LaneRay& r, bool any_hit, float* t) {
    SphHit sh;
    if (!sph_hit_test(sc.spheres[__float_as_uint(tv.p0.x)], r.o, r.d, r.ray_tmax, any_hit ? 2.0f * PT_PI : PT_PI, &sh)) return false;
    *t = sh.t;
    return true;
}
template <bool SPH>
PT_DEV bool prim_test(const PtScene& sc, const TriVerts& tv, const LaneRay& r, bool any_hit, TriHit& h) {
    if constexpr (SPH) {
        if (tv.flags & PT_TRI_SPHERE) return sphere_rec_test(sc, tv, r, any_hit, &h.t);
    }
    return tri_test(r.rp, tv.p0, tv.p1, tv.p2, tv.flags, r.ray_tmax, h);
}
// Transform::transform_ray with the instance's inverse (transformed_primitive.rs:27-29, transform.rs:184-203, :245-282)
PT_DEV void instance_ray(const PtInstance& in, V3 ro, V3 rd, V3* o_out, V3* d_out) {
    const float* m = in.minv;
    V3 o = sph_point(m, ro);
    V3 oe = PT_GAMMA(3.0f) * mk3(fabsf(m[0] * ro.x) + fabsf(m[1] * ro.y) + fabsf(m[2] * ro.z) + fabsf(m[3]),
                                  fabsf(m[4] * ro.x) + fabsf(m[5] * ro.y) + fabsf(m[6] * ro.z) + fabsf(m[7]),
                                  fabsf(m[8] * ro.x) + fabsf(m[9] * ro.y) + fabsf(m[10] * ro.z) + fabsf(m[11]));
    V3 d = sph_vector(m, rd);
    float ls = length_squared(d);
    if (ls > 0.0f) {
        float dt = dot(vabs(d), oe) / ls;
        o = o + d * dt;
    }
    *o_out = o; *d_out = d;
}
template <bool SPH, bool INST>
PT_DEV void ray_step_tri(const PtScene& sc, LaneRay& r, bool any_hit, TravCtx& c, uint32_t base = 0);
PT_DEV void ray_step_node(const PtScene& sc, LaneRay& r, TravCtx& c);
// TransformedPrimitive::intersect / intersect_p (transformed_primitive.rs:26-48): the ray goes to instance space, the object's
// accelerator is walked to the end on the stack above the caller's entries (the other lanes of the wave wait: instances are a
// breadth feature, not the hot path), t carries over because the direction is not renormalised.
template <bool SPH>
__device__ __noinline__ bool instance_rec_test(const PtScene& sc, uint32_t inst, const LaneRay& r, bool any_hit, TravCtx& c, float* t_out, int32_t* rec_out) {
    const PtInstance& in = sc.instances[inst];
    V3 o, d;
    instance_ray(in, r.o, r.d, &o, &d);
    LaneRay ri;
    ri.o = o; ri.d = d;
    ri.idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    ri.sbits = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    if (!(fabsf(ri.idir.x) < PT_INF) || !(fabsf(ri.idir.y) < PT_INF) || !(fabsf(ri.idir.z) < PT_INF)) ri.sbits |= 8u;
    ri.best = -1; ri.best_inst = 0;
    ri.ray_tmax = r.ray_tmax;
    ray_precompute(ri.rp, o, d);
    const uint32_t base = r.sp;
    ri.sp = base; ri.top = PT_EMPTY_REF;
    if (in.direct) {                     // a single primitive, wrapped without an accelerator: no root box, not a counted leaf test
        TriVerts tv = load_tri(sc.tris, in.root_ref);
        TriHit h;
        if (prim_test<SPH>(sc, tv, ri, any_hit, h)) { ri.best = (int32_t)in.root_ref; ri.ray_tmax = h.t; }
    } else if (box_root_test(in.root_lo, in.root_hi, o, ri.idir, ri.sbits, ri.ray_tmax, ri.tmin, ri.tmax)) {
        ri.top = in.root_ref; ri.sp = base + 1;
        while (ri.sp > base) {
            if (ri.top & PT_LEAF_BIT) ray_step_tri<SPH, false>(sc, ri, any_hit, c, base);
            else ray_step_node(sc, ri, c);
        }
    }
    if (ri.best < 0) return false;
    *t_out = ri.ray_tmax; *rec_out = ri.best;
    return true;
}
template <bool SPH, bool INST>
PT_DEV void ray_step_tri(const PtScene& sc, LaneRay& r, bool any_hit, TravCtx& c, uint32_t base) {
    uint32_t rec = stk_pop(c, r.top, r.sp) & PT_LEAF_FIRST_MASK;
    bool leaf_hit = false;
    if constexpr (!INST) {
        for (;;) {
            TriVerts t0 = load_tri(sc.tris, rec), t1 = load_tri(sc.tris, rec + 1);   // array is padded by one record
            TriHit h;
            c.n_tris++;
            if (prim_test<SPH>(sc, t0, r, any_hit, h)) {
                r.best = (int32_t)rec; leaf_hit = true;
                if (any_hit) { r.sp = base; r.top = PT_EMPTY_REF; return; }
                r.ray_tmax = h.t;                           // GeometricPrimitive::intersect: r.t_max = t_hit
            }
            if (t0.flags & PT_TRI_LAST) break;
            c.n_tris++;
            if (prim_test<SPH>(sc, t1, r, any_hit, h)) {
                r.best = (int32_t)(rec + 1); leaf_hit = true;
                if (any_hit) { r.sp = base; r.top = PT_EMPTY_REF; return; }
                r.ray_tmax = h.t;
            }
            if (t1.flags & PT_TRI_LAST) break;
            rec += 2;
        }
        if (leaf_hit) r.tmax = r.ray_tmax;                  // intersect_simd: tmax shrinks after the whole leaf
        return;
    }
    for (;;) {
        TriVerts t0 = load_tri(sc.tris, rec);
        TriHit h;
        c.n_tris++;
        bool hit;
        int32_t hit_rec = (int32_t)rec;
        uint32_t hit_inst = 0;
        if (INST && (t0.flags & PT_TRI_INSTANCE)) {
            hit_inst = __float_as_uint(t0.p0.x) + 1u;
            hit = instance_rec_test<SPH>(sc, hit_inst - 1u, r, any_hit, c, &h.t, &hit_rec);
        } else hit = prim_test<SPH>(sc, t0, r, any_hit, h);
        if (hit) {
            r.best = hit_rec; leaf_hit = true;
            if (INST) r.best_inst = hit_inst;
            if (any_hit) { r.sp = base; r.top = PT_EMPTY_REF; return; }
            r.ray_tmax = h.t;                           // GeometricPrimitive::intersect: r.t_max = t_hit
        }
        if (t0.flags & PT_TRI_LAST) break;
        rec++;
    }
    if (leaf_hit) r.tmax = r.ray_tmax;                  // intersect_simd: tmax shrinks after the whole leaf
}
template <bool SPH, bool INST>
PT_DEV void ray_step(const PtScene& sc, LaneRay& r, bool any_hit, TravCtx& c) {
    if (ray_wants_tri(r)) ray_step_tri<SPH, INST>(sc, r, any_hit, c);
    else ray_step_node(sc, r, c);
}

// k_trace_inst's leaf step: an instance is ENTERED, not walked.  The blocking form above runs an object's whole tree inside one lane's leaf step while the
// other 63 lanes of the wave wait (fine for a batch of probe rays, 6x too slow for a frame: 50 instances of a 500 k-triangle object traced at 174 Mrays/s).
// Here the lane that meets an instance record whose root box it hits pushes what it needs to come back -- the rest of its leaf as a leaf reference, its
// world-space slab interval, a marker -- turns its ray into the instance-space ray (TransformedPrimitive::intersect, transformed_primitive.rs:26-48: t
// carries over, the direction is not renormalised) and returns to the wave's phases with the object's root on top of its stack: node visits and leaf walks
// of object trees then run side by side with everybody else's.  Popping the marker (trace_body) brings the world ray back from the path arrays.  Order of
// tests, t_max updates and counters are the blocking form's: a leaf cut in two behaves like the whole one because nothing reads the world-space slab
// interval while the ray is inside the object (a hit among the leaf's earlier records shrinks it before it is saved, a hit inside the object after it is
// restored, the leaf's later records at their own end).  Objects hold no instances (scene_context.rs:1352-1357), so one level is all there is.
#ifndef PT_INST_LEAF_PREFETCH
#define PT_INST_LEAF_PREFETCH 0     // the next record asked for before this one is tested: 172 registers (two waves per SIMD) or, held to 168, no faster (637 / 1 060 / 432 against 628 / 1 184 / 440 Mrays/s on three instanced scenes)
#endif
#define PT_INST_EXIT_REF (PT_LEAF_BIT | PT_LEAF_FIRST_MASK)          // the marker: a leaf reference no scene can hold (fewer than 2^26 - 16 primitives)
PT_DEV void ray_set_direction_state(LaneRay& r, V3 o, V3 d) {        // what ray_begin derives from (o, d), without touching hit, t_max or stack
    r.o = o; r.d = d;
    r.idir = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    r.sbits = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    if (!(fabsf(r.idir.x) < PT_INF) || !(fabsf(r.idir.y) < PT_INF) || !(fabsf(r.idir.z) < PT_INF)) r.sbits |= 8u;
    ray_precompute(r.rp, o, d);
}
template <bool SPH>
PT_DEV void ray_step_tri_enter(const PtScene& sc, LaneRay& r, bool any_hit, TravCtx& c) {
    uint32_t rec = stk_pop(c, r.top, r.sp) & PT_LEAF_FIRST_MASK;
    bool leaf_hit = false;
    TriVerts t0 = load_tri(sc.tris, rec);
    for (;; rec++) {
#if PT_INST_LEAF_PREFETCH
        const TriVerts t_next = load_tri(sc.tris, rec + 1u);          // on its way while this record is tested (the array is padded by one record)
#endif
        TriHit h;
        c.n_tris++;
        bool hit = false;
        int32_t hit_rec = (int32_t)rec;
        uint32_t hit_inst = r.cur_inst;
        const bool last = (t0.flags & PT_TRI_LAST) != 0;
        if (t0.flags & PT_TRI_INSTANCE) {            // (world leaves only)
            const uint32_t inst = __float_as_uint(t0.p0.x);
            const PtInstance& in = sc.instances[inst];
            V3 o, d;
            instance_ray(in, r.o, r.d, &o, &d);
            LaneRay ri;
            ray_set_direction_state(ri, o, d);
            ri.ray_tmax = r.ray_tmax;
            if (in.direct) {                         // a single primitive, wrapped without an accelerator: no root box, not a counted leaf test
                TriVerts tv = load_tri(sc.tris, in.root_ref);
                hit = prim_test<SPH>(sc, tv, ri, any_hit, h);
                hit_rec = (int32_t)in.root_ref; hit_inst = inst + 1u;
            } else if (box_root_test(in.root_lo, in.root_hi, o, ri.idir, ri.sbits, ri.ray_tmax, ri.tmin, ri.tmax)) {
                if (leaf_hit) r.tmax = r.ray_tmax;
                if (!last) stk_push(c, r.top, r.sp, PT_LEAF_BIT | (rec + 1u));
                stk_push(c, r.top, r.sp, __float_as_uint(r.tmin));
                stk_push(c, r.top, r.sp, __float_as_uint(r.tmax));
                stk_push(c, r.top, r.sp, PT_INST_EXIT_REF);
                r.o = ri.o; r.d = ri.d; r.idir = ri.idir; r.sbits = ri.sbits; r.rp = ri.rp; r.tmin = ri.tmin; r.tmax = ri.tmax;
                r.cur_inst = inst + 1u;
                stk_push(c, r.top, r.sp, in.root_ref);
                return;
            }
        } else hit = prim_test<SPH>(sc, t0, r, any_hit, h);
        if (hit) {
            r.best = hit_rec; leaf_hit = true;
            r.best_inst = hit_inst;
            if (any_hit) { r.sp = 0; r.top = PT_EMPTY_REF; return; }
            r.ray_tmax = h.t;
        }
        if (last) break;
#if PT_INST_LEAF_PREFETCH
        t0 = t_next;
#else
        t0 = load_tri(sc.tris, rec + 1u);
#endif
    }
    if (leaf_hit) r.tmax = r.ray_tmax;
}

// ============================================================ lean node visit (k_trace, k_trace_sph_dist)
// The VALU-issue budget decides this kernel (tools/ubench/valu_issue.hip: a two-operand add / sub / mul / and / mov issues in
// 2.7 SIMD cycles at four waves per SIMD, every three-operand, compare or select form in 4.4-4.8, packed f32 in 4.8 per pair).
// The first version of the visit spent 177 instructions, two thirds of them on addresses, the order selects and four branch-free
// pushes.  This one keeps the 48 sub / mul and 16 min / max of the four slab tests -- the parity contract -- and rebuilds the rest:
//   * stack: `sa` is the LDS byte address of the top entry's slot, slot 0 of every lane holds PT_EMPTY_REF for good, so a pop is
//     one add and one ds_read (the read that falls on slot 0 empties `top` by itself) and a push one add, one ds_write, one mov;
//   * pushes are predicated by EXEC, not by selects: the four hit conditions stay in SGPR lane masks as v_cmp leaves them, the
//     front-to-back order (ORDER_TABLE in closed form, qbvh_x86.rs:187-204) is applied to those masks by the scalar unit, and each
//     push runs with EXEC = its condition;
//   * the three order signs of a (node, ray octant) pair come from three 8-bit tables in the node (one AND + compare each);
//   * the six plane rows are addressed as SGPR base + 32-bit offset with the sign-dependent part hoisted per ray;
//   * empty child slots hold an inverted box (+inf / -inf) in HBM, so they fail the slab test by themselves.
// Rays whose reciprocal direction has an infinite, NaN or zero component, and lanes within four slots of the LDS part's end,
// take the general form (node_step_general) for the whole wave: same results, the first version's cost.
typedef __attribute__((address_space(3))) uint32_t* pt_lds_u32;
PT_DEV uint32_t lds_addr_of(const uint32_t* p) { return (uint32_t)(size_t)(pt_lds_u32)p; }
PT_DEV uint32_t lds_load(uint32_t a) { return *(pt_lds_u32)(size_t)a; }
PT_DEV void lds_store(uint32_t a, uint32_t v) { *(pt_lds_u32)(size_t)a = v; }
PT_DEV uint32_t fs_slot(const TravCtx& c, uint32_t sa) { return (sa - c.lane_base) / PT_SLOT; }     // slot of the top entry; 0 = empty stack
PT_DEV void fs_push(TravCtx& c, uint32_t& sa, uint32_t& top, uint32_t v) {
    const uint32_t s = fs_slot(c, sa) + 1u;
    if (s < PT_FS_SLOTS) lds_store(sa + PT_SLOT, v);
    else if (s - PT_FS_SLOTS < c.spill_depth) c.spill[(size_t)(s - PT_FS_SLOTS) * c.spill_stride] = v;
    else { c.overflow = 1; return; }
    sa += PT_SLOT;
    top = v;
}
PT_DEV uint32_t fs_pop(TravCtx& c, uint32_t& sa, uint32_t& top) {      // precondition: top != PT_EMPTY_REF
    const uint32_t v = top;
    sa -= PT_SLOT;
    const uint32_t s = fs_slot(c, sa);
    if (s >= PT_FS_SLOTS) top = c.spill[(size_t)(s - PT_FS_SLOTS) * c.spill_stride];
    else top = lds_load(sa);                  // slot 0 hands back PT_EMPTY_REF
    return v;
}
// single-instruction forms: fmaxf / fminf would get a canonicalising v_max in front of every operand the compiler cannot prove quiet
PT_DEV float v_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_DEV float v_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PT_DEV float v_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PT_DEV float v_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Old-form visit on the `sa` stack (any mix of lanes: NaN-exact slabs, HBM spill).
template <bool EXACT>
PT_DEV void visit_node_general(const PtNode* nodes, uint32_t ni, V3 o, V3 idir, uint32_t sbits, float tmin, float tmax, TravCtx& c, uint32_t& top, uint32_t& sa) {
    const char* nb = reinterpret_cast<const char*>(nodes) + (size_t)(ni & PT_REF_INDEX_MASK) * 128u;
    const uint32_t ox = (sbits & 1u) ? 48u : 0u, oy = (sbits & 2u) ? 48u : 0u, oz = (sbits & 4u) ? 48u : 0u;
    float4 nx = *reinterpret_cast<const float4*>(nb + ox), fx = *reinterpret_cast<const float4*>(nb + (48u - ox));
    float4 ny = *reinterpret_cast<const float4*>(nb + 16u + oy), fy = *reinterpret_cast<const float4*>(nb + 16u + (48u - oy));
    float4 nz = *reinterpret_cast<const float4*>(nb + 32u + oz), fz = *reinterpret_cast<const float4*>(nb + 32u + (48u - oz));
    uint4 ch = *reinterpret_cast<const uint4*>(nb + 96u);
    uint32_t axes = *reinterpret_cast<const uint32_t*>(nb + 112u);
    uint32_t mask = 0;
#define PT_SLAB(k, C)                                                        \
    {                                                                        \
        float a = tmin, b = tmax;                                            \
        if (EXACT) {                                                         \
            a = sse_max(a, (nx.C - o.x) * idir.x); b = sse_min(b, (fx.C - o.x) * idir.x); \
            a = sse_max(a, (ny.C - o.y) * idir.y); b = sse_min(b, (fy.C - o.y) * idir.y); \
            a = sse_max(a, (nz.C - o.z) * idir.z); b = sse_min(b, (fz.C - o.z) * idir.z); \
        } else {                                                             \
            a = fmaxf(a, (nx.C - o.x) * idir.x); b = fminf(b, (fx.C - o.x) * idir.x); \
            a = fmaxf(a, (ny.C - o.y) * idir.y); b = fminf(b, (fy.C - o.y) * idir.y); \
            a = fmaxf(a, (nz.C - o.z) * idir.z); b = fminf(b, (fz.C - o.z) * idir.z); \
        }                                                                    \
        if (b >= a) mask |= (1u << k);                                       \
    }
    PT_SLAB(0, x) PT_SLAB(1, y) PT_SLAB(2, z) PT_SLAB(3, w)
#undef PT_SLAB
    mask &= (axes >> 8) & 15u;          // empty slots (inverted boxes here, all-zero in the reference) never count
    uint32_t s_top = (sbits >> (axes & 3)) & 1, s_left = (sbits >> ((axes >> 2) & 3)) & 1, s_right = (sbits >> ((axes >> 4) & 3)) & 1;
    uint32_t l0 = s_left ? ch.x : ch.y, l1 = s_left ? ch.y : ch.x;
    uint32_t l0b = s_left ? 1u : 2u, l1b = s_left ? 2u : 1u;
    uint32_t r0 = s_right ? ch.z : ch.w, r1 = s_right ? ch.w : ch.z;
    uint32_t r0b = s_right ? 4u : 8u, r1b = s_right ? 8u : 4u;
    uint32_t c0 = s_top ? l0 : r0, c1 = s_top ? l1 : r1, c2 = s_top ? r0 : l0, c3 = s_top ? r1 : l1;
    uint32_t b0 = s_top ? l0b : r0b, b1 = s_top ? l1b : r1b, b2 = s_top ? r0b : l0b, b3 = s_top ? r1b : l1b;
    if (mask & b0) fs_push(c, sa, top, c0);
    if (mask & b1) fs_push(c, sa, top, c1);
    if (mask & b2) fs_push(c, sa, top, c2);
    if (mask & b3) fs_push(c, sa, top, c3);
}
PT_DEV void node_step_general(const PtScene& sc, LaneRay& r, TravCtx& c) {
    const uint32_t ref = fs_pop(c, r.sa, r.top);
    c.n_nodes++;
    if (r.sbits & 8u) visit_node_general<true>(sc.nodes, ref, r.o, r.idir, r.sbits, r.tmin, r.tmax, c, r.top, r.sa);
    else visit_node_general<false>(sc.nodes, ref, r.o, r.idir, r.sbits, r.tmin, r.tmax, c, r.top, r.sa);
}
// Four pushes, each with EXEC = its own condition (lane masks inside the current EXEC).
PT_DEV void fs_push4_exec(uint32_t& sa, uint32_t& top, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, unsigned long long e0, unsigned long long e1,
                          unsigned long long e2, unsigned long long e3) {
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[e0]\n\t"
        "v_add_u32 %[sa], %[slot], %[sa]\n\t"
        "v_mov_b32 %[top], %[c0]\n\t"
        "ds_write_b32 %[sa], %[c0]\n\t"
        "s_mov_b64 exec, %[e1]\n\t"
        "v_add_u32 %[sa], %[slot], %[sa]\n\t"
        "v_mov_b32 %[top], %[c1]\n\t"
        "ds_write_b32 %[sa], %[c1]\n\t"
        "s_mov_b64 exec, %[e2]\n\t"
        "v_add_u32 %[sa], %[slot], %[sa]\n\t"
        "v_mov_b32 %[top], %[c2]\n\t"
        "ds_write_b32 %[sa], %[c2]\n\t"
        "s_mov_b64 exec, %[e3]\n\t"
        "v_add_u32 %[sa], %[slot], %[sa]\n\t"
        "v_mov_b32 %[top], %[c3]\n\t"
        "ds_write_b32 %[sa], %[c3]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sa] "+v"(sa), [top] "+v"(top), [sv] "=&s"(sv)
        : [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [e3] "s"(e3), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [slot] "s"(PT_SLOT)
        : "memory");
}
// One node round of the whole wave.  The kernel is bound by the vector L1's request rate, not by arithmetic: with one ray per lane
// every lane's 128-byte node is a line of its own, and the eight loads of the first version were 8 x 64 tag look-ups per round
// (tools/ubench/node_fetch.hip: 77 G node visits/s is all the chip can do that way, k_trace sat at 61 G beside its triangle
// loads).  Here the wave fetches the up-to-64 nodes TOGETHER: in load j, lane l fetches 16-byte chunk of the node of lane
// 8 j + (l >> 3) -- eight neighbouring lanes cover one whole line -- straight into LDS (global_load_lds_dwordx4, no VGPR round
// trip; one instruction lands 1 KiB = 8 nodes), and each owner then reads its rows from LDS.  164-171 G node visits/s in the
// micro-benchmark.  Chunk c of sub-node i of a piece is stored at position c ^ i (the swizzle is applied to the SOURCE address,
// the LDS image of an LDS-DMA is lane-linear), so the eight owners of a piece read any one row from eight different bank groups.
//   w_node  this lane has a node to visit (its reference is r.top)
typedef __attribute__((address_space(3))) void* pt_lds_void;
typedef const __attribute__((address_space(1))) void* pt_global_cvoid;
typedef float pt_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t pt_v4u __attribute__((ext_vector_type(4)));
PT_DEV float4 lds_load4(uint32_t a) { const pt_v4f v = *(__attribute__((address_space(3))) const pt_v4f*)(size_t)a; return make_float4(v.x, v.y, v.z, v.w); }
PT_DEV uint4 lds_load4u(uint32_t a) { const pt_v4u v = *(__attribute__((address_space(3))) const pt_v4u*)(size_t)a; return make_uint4(v.x, v.y, v.z, v.w); }
// issue half: pops, publishes and starts the LDS-DMA fetches; returns the entry below the popped one (the fall-back `top`)
PT_DEV uint32_t node_round_issue(const PtScene& sc, LaneRay& r, TravCtx& c, bool w_node) {
    // owners: pop, publish the node index (transposed: the eight indices a loader lane needs are contiguous)
    uint32_t ref = PT_EMPTY_REF, top = PT_EMPTY_REF;
    if (w_node) {
#ifdef PT_STACK_HIST
        atomicAdd(c.spill - ((size_t)blockIdx.x * PT_BLOCK + threadIdx.x) - PT_DIAG_WORDS + 1024u + min(fs_slot(c, r.sa), 63u), 1u);   // diagnostic build: stack depth at each node visit
#endif
        ref = r.top;
        r.sa -= PT_SLOT;
        top = lds_load(r.sa);                 // the entry below (slot 0: PT_EMPTY_REF)
        c.n_nodes++;
    }
    lds_store(c.idx_pub, ref);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {   // loaders
        const uint4 ia = lds_load4u(c.idx_ld), ib = lds_load4u(c.idx_ld + 16u);
        const uint32_t ids[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t coff = ((lane & 7u) ^ (lane >> 3)) << 4;
        const char* nb = reinterpret_cast<const char*>(sc.nodes);
        const uint32_t piece0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.stage_wave);
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++)
            if (ids[j] != PT_EMPTY_REF)
                __builtin_amdgcn_global_load_lds((pt_global_cvoid)(nb + ((ids[j] << 7) + coff)), (pt_lds_void)(size_t)(piece0 + j * 1024u), 16, 0, 0);
    }
    return top;
}
// finish half, after `s_waitcnt vmcnt(0)`: rows from LDS, four slab tests, ordered EXEC-predicated pushes
PT_DEV void node_round_finish(LaneRay& r, TravCtx& c, bool w_node, uint32_t top) {
    if (w_node) {
        const float4 nx = lds_load4(r.o_nx), fx = lds_load4(r.o_fx), ny = lds_load4(r.o_ny), fy = lds_load4(r.o_fy), nz = lds_load4(r.o_nz), fz = lds_load4(r.o_fz);
        const uint4 ch = lds_load4u(c.stage_chunk(6u));
        const uint32_t lut = lds_load(c.stage_chunk(7u) + 4u);
        const float ox = r.o.x, oy = r.o.y, oz = r.o.z, ix = r.idir.x, iy = r.idir.y, iz = r.idir.z;
#define PT_SLAB(C) (v_min3(v_min(r.tmax, (fx.C - ox) * ix), (fy.C - oy) * iy, (fz.C - oz) * iz) >= v_max3(v_max(r.tmin, (nx.C - ox) * ix), (ny.C - oy) * iy, (nz.C - oz) * iz))
        const bool h0 = PT_SLAB(x), h1 = PT_SLAB(y), h2 = PT_SLAB(z), h3 = PT_SLAB(w);
#undef PT_SLAB
        // ORDER_TABLE in closed form: pops visit {0,1} before {2,3} iff the ray is non-negative along axis_top, 0 before 1 iff along
        // axis_left, 2 before 3 iff along axis_right; pushes run in the reverse of the visit order
        const bool T = (lut & r.m_t) != 0u, L = (lut & r.m_l) != 0u, R = (lut & r.m_r) != 0u;
        const uint32_t l0 = L ? ch.x : ch.y, l1 = L ? ch.y : ch.x, r0 = R ? ch.z : ch.w, r1 = R ? ch.w : ch.z;
        const uint32_t c0 = T ? l0 : r0, c1 = T ? l1 : r1, c2 = T ? r0 : l0, c3 = T ? r1 : l1;
        // the same selects on the hit conditions, as 64-bit lane masks on the scalar unit (a ? x : y == y ^ (a & (x ^ y))); lanes
        // outside EXEC are zero in every v_cmp result, so each push condition lies inside EXEC
        const unsigned long long H0 = __ballot(h0), H1 = __ballot(h1), H2 = __ballot(h2), H3 = __ballot(h3), Tm = __ballot(T), Lm = __ballot(L), Rm = __ballot(R);
        const unsigned long long yl = Lm & (H0 ^ H1), el0 = H1 ^ yl, el1 = H0 ^ yl;
        const unsigned long long yr = Rm & (H2 ^ H3), er0 = H3 ^ yr, er1 = H2 ^ yr;
        const unsigned long long y0 = Tm & (el0 ^ er0), e0 = er0 ^ y0, e2 = el0 ^ y0;
        const unsigned long long y1 = Tm & (el1 ^ er1), e1 = er1 ^ y1, e3 = el1 ^ y1;
        fs_push4_exec(r.sa, top, c0, c1, c2, c3, e0, e1, e2, e3);
        r.top = top;
    }
}

// The lean visit with per-lane loads (PT_NODE_STAGED == 0): every lane fetches its own node, SGPR base + 32-bit offset with the
// sign-dependent row offsets hoisted per ray.  Keeps the 32-slot LDS stack and four blocks per CU; bound by the vector L1's
// request rate (eight requests per lane and visit).
template <int TOUCH>
PT_DEV void node_step_lean(const PtScene& sc, LaneRay& r, TravCtx& c) {
    const uint32_t ref = r.top;
    r.sa -= PT_SLOT;
    uint32_t top = lds_load(r.sa);            // the entry below (slot 0: PT_EMPTY_REF); lands while the node is on its way
    c.n_nodes++;
    const char* nb = reinterpret_cast<const char*>(sc.nodes);
    const uint32_t no = ref << 7;             // the shift drops the axis bits a child reference carries above its node index
    float4 nx, fx, ny, fy, nz, fz;
    uint4 ch;
#if PT_TOP_NODES > 0
    // two ifs, LDS first: the two paths fill the same registers, and the wait the compiler puts between them (it tracks registers, not
    // lanes) is then the short LDS one, not the global loads' round trip
    const bool in_lds = no < c.top_bytes;
    c.n_nodes_lds += in_lds ? 1u : 0u;
    if (in_lds) {                             // one of the top nodes: its seven rows sit in LDS, 112 bytes apart (no - no / 8 = 112 x index)
        const uint32_t la = c.top_lds + no - (no >> 3);
        nx = lds_load4(la + r.o_nx); fx = lds_load4(la + r.o_fx); ny = lds_load4(la + r.o_ny); fy = lds_load4(la + r.o_fy);
        nz = lds_load4(la + r.o_nz); fz = lds_load4(la + r.o_fz);
        ch = lds_load4u(la + 96u);
    }
    if (!in_lds)
#endif
    {
        nx = *reinterpret_cast<const float4*>(nb + (no + r.o_nx)); fx = *reinterpret_cast<const float4*>(nb + (no + r.o_fx));
        ny = *reinterpret_cast<const float4*>(nb + (no + r.o_ny)); fy = *reinterpret_cast<const float4*>(nb + (no + r.o_fy));
        nz = *reinterpret_cast<const float4*>(nb + (no + r.o_nz)); fz = *reinterpret_cast<const float4*>(nb + (no + r.o_fz));
        ch = *reinterpret_cast<const uint4*>(nb + (no + 96u));
    }
    const float ox = r.o.x, oy = r.o.y, oz = r.o.z, ix = r.idir.x, iy = r.idir.y, iz = r.idir.z;
#define PT_SLAB(C) (v_min3(v_min(r.tmax, (fx.C - ox) * ix), (fy.C - oy) * iy, (fz.C - oz) * iz) >= v_max3(v_max(r.tmin, (nx.C - ox) * ix), (ny.C - oy) * iy, (nz.C - oz) * iz))
    const bool h0 = PT_SLAB(x), h1 = PT_SLAB(y), h2 = PT_SLAB(z), h3 = PT_SLAB(w);
#undef PT_SLAB
    // the node's split axes ride in bits 26..27 of child references 0 / 1 / 3 (seven L1 requests per visit instead of eight)
    const bool T = __builtin_amdgcn_ubfe(r.sbits, __builtin_amdgcn_ubfe(ch.x, PT_REF_AXIS_SHIFT, 2), 1) != 0u;
    const bool L = __builtin_amdgcn_ubfe(r.sbits, __builtin_amdgcn_ubfe(ch.y, PT_REF_AXIS_SHIFT, 2), 1) != 0u;
    const bool R = __builtin_amdgcn_ubfe(r.sbits, __builtin_amdgcn_ubfe(ch.w, PT_REF_AXIS_SHIFT, 2), 1) != 0u;
    const uint32_t l0 = L ? ch.x : ch.y, l1 = L ? ch.y : ch.x, r0 = R ? ch.z : ch.w, r1 = R ? ch.w : ch.z;
    const uint32_t c0 = T ? l0 : r0, c1 = T ? l1 : r1, c2 = T ? r0 : l0, c3 = T ? r1 : l1;
    const unsigned long long H0 = __ballot(h0), H1 = __ballot(h1), H2 = __ballot(h2), H3 = __ballot(h3), Tm = __ballot(T), Lm = __ballot(L), Rm = __ballot(R);
    const unsigned long long yl = Lm & (H0 ^ H1), el0 = H1 ^ yl, el1 = H0 ^ yl;
    const unsigned long long yr = Rm & (H2 ^ H3), er0 = H3 ^ yr, er1 = H2 ^ yr;
    const unsigned long long y0 = Tm & (el0 ^ er0), e0 = er0 ^ y0, e2 = el0 ^ y0;
    const unsigned long long y1 = Tm & (el1 ^ er1), e1 = er1 ^ y1, e3 = el1 ^ y1;
    fs_push4_exec(r.sa, top, c0, c1, c2, c3, e0, e1, e2, e3);
    r.top = top;
    if constexpr (TOUCH != 0) {
        // k_trace_far's first form (PT_NODE_COOP 0; superseded by node_step_coop below, which gains more on the same scenes).  Every reference pushed here IS
        // visited later -- the reference's stack holds no distances --, so a pushed LEAF that is not the next visit has its first record's line
        // touched now: one 4-byte load into a register nobody reads.  A leaf round waits for the slowest of up to 64 record fetches, each the
        // first access to a line of its own; touched at push time the line is on its way (or in L2) when the round comes.  Loads return in
        // order, so the touch is paid for by this lane's next fetch waiting behind it -- which is why it loses wherever the records hit the
        // caches anyway (RT1M -20 %, a 3.5 M-triangle scene -7 %) and is not the default.  Measured variants (16 M sparse triangles, 130 node visits
        // per ray, Mrays/s at 64 spp; TOUCH = the variant number): none 361; 3 not-next leaves 421; 2 not-next interior nodes 356; 1 both 353;
        // 4 every pushed leaf, the next visit too 352; 5 = 4 + the second line of leaves of three records and more 284.
        unsigned long long p0, p1, p2, p3 = 0ull;
        if (TOUCH >= 4) { p0 = e0; p1 = e1; p2 = e2; p3 = e3; }
        else { p0 = e0 & (e1 | e2 | e3); p1 = e1 & (e2 | e3); p2 = e2 & e3; }
        const unsigned long long me = 1ull << (threadIdx.x & 63u);
        const char* tb = reinterpret_cast<const char*>(sc.tris);
        auto touch = [&](uint32_t cref) {
            if (TOUCH == 2 && (cref & PT_LEAF_BIT)) return;
            if (TOUCH >= 3 && !(cref & PT_LEAF_BIT)) return;
            const char* a = (cref & PT_LEAF_BIT) ? tb + (size_t)(cref & PT_LEAF_FIRST_MASK) * 48u : nb + (size_t)(cref << 7);
            uint32_t v = *reinterpret_cast<const uint32_t*>(a);
            asm volatile("" :: "v"(v));
            if (TOUCH == 5 && ((cref >> PT_LEAF_COUNT_SHIFT) & 7u) >= 2u) {
                uint32_t w = *reinterpret_cast<const uint32_t*>(a + 128);
                asm volatile("" :: "v"(w));
            }
        };
        if (p0 & me) touch(c0);
        if (p1 & me) touch(c1);
        if (p2 & me) touch(c2);
        if (p3 & me) touch(c3);
    }
}

#ifndef PT_NODE_COOP
#define PT_NODE_COOP 1           // k_trace_far fetches nodes pairwise (node_step_coop); 2 = k_trace too (experiment)
#endif
#ifndef PT_COOP_TOUCH
#define PT_COOP_TOUCH 0          // ... and touches pushed leaves as node_step_lean<TOUCH> does (slower with the pairwise fetch: 16 M sparse triangles 440 vs 485 Mrays/s)
#endif
#if PT_NODE_COOP
// k_trace_far's visit, for scenes whose rays miss the caches (chosen per scene by a timed trial, pt_context.cpp): the lean visit with the two lanes of a
// PAIR fetching each other's node together (tools/ubench/visit_quad.hip `coop`).  Node A belongs to the
// even lane, node B to the odd one.  In loads 1-3 both lanes read node A (its owner the three near rows, the partner A's three far rows, at the addresses
// the OWNER computed and handed over by DPP), in loads 4-6 both read node B; the child references each lane fetches for itself.  Six of the seven loads of
// a visit then see two lanes per 128-byte line -- four tag look-ups per visit instead of seven, still 64 rays per wave.  The odd lanes swap the address
// registers before and the data registers after the loads (v_swap_b32 under EXEC = odd lanes), so that every lane finds its near rows in x1-3 and its far
// rows in the PARTNER's x4-6, read through DPP as a source modifier of the slab arithmetic.  A lane whose partner visits a node while it does not helps
// with the partner's far rows and does nothing else.  m_node = __ballot(w_node).  Measured (Mrays/s at 64 spp, k_trace / this): 16 M sparse triangles
// 421 / 485, 8 M sparse 525 / 603; where nodes hit the caches the 19 extra instructions per visit cost more than the look-ups save (RT1M 1083 / 1040,
// 4 M triangles 1288 / 1241), hence the trial.
#define PT_PAIR_U(x) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0xB1, 0xF, 0xF, false))
#define PT_PAIR_F(x) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0xB1, 0xF, 0xF, false))
template <int TOUCH>
PT_DEV void node_step_coop(const PtScene& sc, LaneRay& r, TravCtx& c, bool w_node, unsigned long long m_node) {
    constexpr unsigned long long kEven = 0x5555555555555555ull, kOdd = 0xAAAAAAAAAAAAAAAAull;
    const unsigned long long ev = m_node & kEven, od = m_node & kOdd;
    const unsigned long long m_y = ev | (ev << 1), m_z = od | (od >> 1);          // the lanes that load from node A / from node B
    if (__builtin_amdgcn_inverse_ballot_w64(m_y | m_z)) {                           // (closed under lane ^ 1: every DPP source below is an active lane)
        const uint32_t ref = r.top;
        const uint32_t no = ref << 7;
        const char* nb = reinterpret_cast<const char*>(sc.nodes);
        uint32_t top = PT_EMPTY_REF;
        const bool in_lds = w_node && no < c.top_bytes;
        if (w_node) { r.sa -= PT_SLOT; top = lds_load(r.sa); c.n_nodes++; c.n_nodes_lds += in_lds ? 1u : 0u; }
        const unsigned long long m_lds = __ballot(in_lds), le = m_lds & kEven, lo = m_lds & kOdd;
        const unsigned long long y_lds = le | (le << 1), z_lds = lo | (lo >> 1);  // node A / node B is one of the top nodes in LDS
        const uint32_t base = in_lds ? c.top_lds + no - (no >> 3) : no;           // LDS address or byte offset in HBM, the owner knows which
        uint32_t a1 = base + r.o_nx, a2 = base + r.o_ny, a3 = base + r.o_nz;      // my near rows
        uint32_t b1 = PT_PAIR_U(base + r.o_fx), b2 = PT_PAIR_U(base + r.o_fy), b3 = PT_PAIR_U(base + r.o_fz);     // my partner's far rows
        unsigned long long sv;
        asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, exec, %[om]\n\tv_swap_b32 %0, %3\n\tv_swap_b32 %1, %4\n\tv_swap_b32 %2, %5\n\ts_mov_b64 exec, %[sv]"
                     : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b1), "+v"(b2), "+v"(b3), [sv] "=&s"(sv) : [om] "s"(kOdd));
        float4 x1, x2, x3, x4, x5, x6;
        uint4 ch;
        if (__builtin_amdgcn_inverse_ballot_w64(m_y & y_lds)) { x1 = lds_load4(a1); x2 = lds_load4(a2); x3 = lds_load4(a3); }
        if (__builtin_amdgcn_inverse_ballot_w64(m_z & z_lds)) { x4 = lds_load4(b1); x5 = lds_load4(b2); x6 = lds_load4(b3); }
        if (in_lds) ch = lds_load4u(base + 96u);
        if (__builtin_amdgcn_inverse_ballot_w64(m_y & ~y_lds)) {
            x1 = *reinterpret_cast<const float4*>(nb + a1); x2 = *reinterpret_cast<const float4*>(nb + a2); x3 = *reinterpret_cast<const float4*>(nb + a3);
        }
        if (__builtin_amdgcn_inverse_ballot_w64(m_z & ~z_lds)) {
            x4 = *reinterpret_cast<const float4*>(nb + b1); x5 = *reinterpret_cast<const float4*>(nb + b2); x6 = *reinterpret_cast<const float4*>(nb + b3);
        }
        if (__builtin_amdgcn_inverse_ballot_w64(m_node & ~m_lds)) ch = *reinterpret_cast<const uint4*>(nb + (no + 96u));
        asm volatile("s_mov_b64 %[sv], exec\n\ts_and_b64 exec, exec, %[om]\n\t"
                     "v_swap_b32 %0, %12\n\tv_swap_b32 %1, %13\n\tv_swap_b32 %2, %14\n\tv_swap_b32 %3, %15\n\t"
                     "v_swap_b32 %4, %16\n\tv_swap_b32 %5, %17\n\tv_swap_b32 %6, %18\n\tv_swap_b32 %7, %19\n\t"
                     "v_swap_b32 %8, %20\n\tv_swap_b32 %9, %21\n\tv_swap_b32 %10, %22\n\tv_swap_b32 %11, %23\n\t"
                     "s_mov_b64 exec, %[sv]\n\ts_nop 1"          // (a DPP read of a register a VALU instruction has just written needs two wait states; the compiler does not see into this block)
                     : "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x1.w), "+v"(x2.x), "+v"(x2.y), "+v"(x2.z), "+v"(x2.w), "+v"(x3.x), "+v"(x3.y), "+v"(x3.z), "+v"(x3.w),
                       "+v"(x4.x), "+v"(x4.y), "+v"(x4.z), "+v"(x4.w), "+v"(x5.x), "+v"(x5.y), "+v"(x5.z), "+v"(x5.w), "+v"(x6.x), "+v"(x6.y), "+v"(x6.z), "+v"(x6.w), [sv] "=&s"(sv)
                     : [om] "s"(kOdd));
        const float ox = r.o.x, oy = r.o.y, oz = r.o.z, ix = r.idir.x, iy = r.idir.y, iz = r.idir.z;
#define PT_SLAB(C) (v_min3(v_min(r.tmax, (PT_PAIR_F(x4.C) - ox) * ix), (PT_PAIR_F(x5.C) - oy) * iy, (PT_PAIR_F(x6.C) - oz) * iz) >= \
                    v_max3(v_max(r.tmin, (x1.C - ox) * ix), (x2.C - oy) * iy, (x3.C - oz) * iz))
        const bool h0 = PT_SLAB(x), h1 = PT_SLAB(y), h2 = PT_SLAB(z), h3 = PT_SLAB(w);
#undef PT_SLAB
        const bool T = __builtin_amdgcn_ubfe(r.sbits, __builtin_amdgcn_ubfe(ch.x, PT_REF_AXIS_SHIFT, 2), 1) != 0u;
        const bool L = __builtin_amdgcn_ubfe(r.sbits, __builtin_amdgcn_ubfe(ch.y, PT_REF_AXIS_SHIFT, 2), 1) != 0u;
        const bool R = __builtin_amdgcn_ubfe(r.sbits, __builtin_amdgcn_ubfe(ch.w, PT_REF_AXIS_SHIFT, 2), 1) != 0u;
        const uint32_t l0 = L ? ch.x : ch.y, l1 = L ? ch.y : ch.x, r0 = R ? ch.z : ch.w, r1 = R ? ch.w : ch.z;
        const uint32_t c0 = T ? l0 : r0, c1 = T ? l1 : r1, c2 = T ? r0 : l0, c3 = T ? r1 : l1;
        // the helping lanes' tests are noise: only the visiting lanes' results count (every push mask below is a select among H0..H3)
        const unsigned long long H0 = __ballot(h0) & m_node, H1 = __ballot(h1) & m_node, H2 = __ballot(h2) & m_node, H3 = __ballot(h3) & m_node;
        const unsigned long long Tm = __ballot(T), Lm = __ballot(L), Rm = __ballot(R);
        const unsigned long long yl = Lm & (H0 ^ H1), el0 = H1 ^ yl, el1 = H0 ^ yl;
        const unsigned long long yr = Rm & (H2 ^ H3), er0 = H3 ^ yr, er1 = H2 ^ yr;
        const unsigned long long y0 = Tm & (el0 ^ er0), e0 = er0 ^ y0, e2 = el0 ^ y0;
        const unsigned long long y1 = Tm & (el1 ^ er1), e1 = er1 ^ y1, e3 = el1 ^ y1;
        fs_push4_exec(r.sa, top, c0, c1, c2, c3, e0, e1, e2, e3);
        if (w_node) r.top = top;
        if constexpr (TOUCH != 0 && PT_COOP_TOUCH) {         // (as in node_step_lean)
            const unsigned long long p0 = e0 & (e1 | e2 | e3), p1 = e1 & (e2 | e3), p2 = e2 & e3;
            const char* tb = reinterpret_cast<const char*>(sc.tris);
            auto touch = [&](uint32_t cref) {
                if (!(cref & PT_LEAF_BIT)) return;
                uint32_t v = *reinterpret_cast<const uint32_t*>(tb + (size_t)(cref & PT_LEAF_FIRST_MASK) * 48u);
                asm volatile("" :: "v"(v));
            };
            if (__builtin_amdgcn_inverse_ballot_w64(p0)) touch(c0);
            if (__builtin_amdgcn_inverse_ballot_w64(p1)) touch(c1);
            if (__builtin_amdgcn_inverse_ballot_w64(p2)) touch(c2);
        }
    }
}
#endif

// intersect_simd (qbvh_x86.rs:230-287): closest hit.  Returns record index or -1.
template <bool SPH, bool INST>
PT_DEV int32_t trace_closest(const PtScene& sc, V3 o, V3 d, float t_max, TravCtx& c, float* t_out, uint32_t* inst_out) {
    LaneRay r;
    ray_begin(sc, r, o, d, t_max);
    while (!ray_done(r)) ray_step<SPH, INST>(sc, r, false, c);
    *t_out = r.ray_tmax;
    *inst_out = r.best_inst;
    return r.best;
}
// intersect_simd_p (qbvh_x86.rs:289-343): any hit
template <bool SPH, bool INST>
PT_DEV bool trace_any(const PtScene& sc, V3 o, V3 d, float t_max, TravCtx& c) {
    LaneRay r;
    ray_begin(sc, r, o, d, t_max);
    while (!ray_done(r)) ray_step<SPH, INST>(sc, r, true, c);
    return r.best >= 0;
}

// Block-level counter flush: LDS accumulate, one global atomic per counter per block.
PT_DEV void flush_counters(PtCounters* g, unsigned long long* s_cnt, unsigned long long regular, unsigned long long shadow,
                           unsigned long long nodes, unsigned long long tris) {
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    if (regular) atomicAdd(&s_cnt[0], regular);
    if (shadow) atomicAdd(&s_cnt[1], shadow);
    if (nodes) atomicAdd(&s_cnt[2], nodes);
    if (tris) atomicAdd(&s_cnt[3], tris);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cnt[0]) atomicAdd(&g->regular_rays, s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&g->shadow_rays, s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&g->nodes, s_cnt[2]);
        if (s_cnt[3]) atomicAdd(&g->tris, s_cnt[3]);
    }
}

// ------------------------------------------------------------------ wave-level ticket
PT_DEV uint32_t wave_ticket(uint32_t* ticket) {
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(ticket, 64u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base);      // lane 0 holds it; uniform for the compiler too (scalar loop state in the callers)
}

// ============================================================ K_TRACE (wavefront)
// Items [0, n_cur) are continuation rays (closest hit -> hit_t / hit_rec), then the previous bounce's
// next-event rays as independent items: n_shadow shadow rays (any hit -> occluded flag) and n_probe MIS probe
// rays (closest hit -> probe_rec).  k_nee_resolve, launched right behind, turns the two results into
// L += beta * ((A + B) / pdf_light)   (path.rs:122-136).
//
// Persistent lanes: ray lengths are heavy-tailed (a wave of 64 fresh rays would idle most of
// its lanes waiting for the slowest one), so every lane runs a small state machine and a wave
// re-fills its idle lanes from a register-resident reservation of prefetched rays as soon as
// PT_REFILL_MIN of them are free.  Each traversal step handles one reference per ray: a 128-byte
// node (4 slab tests, ordered pushes) or a leaf, whose triangle tests are pooled over the wave
// (DIST) or walked by the owning lane (leaves of more than 8 triangles).  The order of pops,
// tests and t_max updates per ray is exactly the reference's, whatever the interleaving across lanes.
#ifdef PT_PROFILE_PHASES
#define PT_PROF_T(name) const unsigned long long name = (unsigned long long)__builtin_readcyclecounter()
#define PT_PROF_SET(name) name = (unsigned long long)__builtin_readcyclecounter()
#else
#define PT_PROF_T(name) do { } while (0)
#define PT_PROF_SET(name) do { } while (0)
#endif
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 8
#endif
#ifndef PT_LEAF_MIN
#define PT_LEAF_MIN 24          // k_trace_seq: leaf phase once this many lanes are parked on a leaf
#endif
#ifndef PT_LEAF_MIN_INST
#define PT_LEAF_MIN_INST 40     // k_trace_inst: a leaf step there may be an instance entry (~300 instructions): wait for more lanes
#endif
#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 4        // waves per SIMD the register allocator must leave room for (k_trace_seq)
#endif
#ifndef PT_TRACE_DIST_WAVES
#if PT_NODE_STAGED
#define PT_TRACE_DIST_WAVES 3   // pooled-leaf kernels with staged node fetch: 49 KB of LDS per block => three blocks per CU
#else
#define PT_TRACE_DIST_WAVES 4
#endif
#endif

#ifndef PT_LEAF_TRIS_MIN
#define PT_LEAF_TRIS_MIN 56          // distributed leaf phase on its own (general visit in the wave): once this many triangle tests are parked
#endif
#ifndef PT_LEAF_LANES_MIN
#define PT_LEAF_LANES_MIN 18         // > 0: a leaf round once this many LANES are parked on a leaf, instead of PT_LEAF_TRIS_MIN parked tests (0).  The exact
                                     // test count costs four ballots + popcounts in EVERY iteration's phase decision; RT1M at 64 spp, k_trace ms per launch:
                                     // tests >= 56: 39.66; lanes >= 12 / 16 / 18 / 20 / 22 / 24 / 28 / 32: 40.88 / 39.39 / 39.34 / 39.43 / 40.00 / 40.52 / 41.79 / 43.50
#endif
#ifndef PT_WALK_BATCH
#define PT_WALK_BATCH 1              // 1: a leaf's owner reads its results from LDS four at a time instead of one per loop iteration
#endif
#ifndef PT_SPH_LANES_MIN
#define PT_SPH_LANES_MIN 8           // scenes with spheres: a sphere round once this many lanes are parked on a leaf that holds a sphere (2 / 4 / 6 / 8 / 10 / 16 / 24 / 32:
                                     // 46.9 / 45.2 / 44.6 / 44.4 / 44.5 / 44.9 / 48.6 / 54.2 ms per launch, RT1M lit by a sphere, 64 spp)
#endif
#ifndef PT_LEAF_TRIS_FUSED
#define PT_LEAF_TRIS_FUSED 0         // > 0: a leaf round rides along a staged node round once this many tests are parked (experiment)
#endif
// DIST: every leaf holds at most 8 triangles and its reference carries the count, so a leaf phase can pool the
// triangles of all parked lanes and hand one (ray, triangle) test to each lane of the wave.
template <bool DIST, bool SPH, bool INST = false, int TOUCH = 0>
PT_DEV void trace_body(const PtScene& sc, const PtPaths& P, const PtQueues& Q, PtCounters* cnt, uint32_t* spill, uint32_t spill_depth, uint32_t* err) {
    // LDS per block.  Pooled-leaf kernels: 16 stack slots x 1 KB + 8 KB node staging per wave + node-index exchange = 49 KB, three
    // blocks per CU (RT1M: 99.98 % of node visits find the stack at 13 entries or fewer; deeper lanes spill to HBM); a leaf round's
    // 64 test results and owner map live in the wave's staging area, which a node round of the same wave never uses at the same time.
    constexpr uint32_t TB = DIST ? PT_TBLOCK : PT_BLOCK;          // threads per block (PT_TRACE_WIDE: the pooled-leaf kernels run 1024)
    __shared__ uint32_t s_stack[(DIST ? PT_FS_SLOTS : PT_LDS_STACK) * TB];
    __shared__ unsigned long long s_cnt[4];
#if PT_NODE_STAGED
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[DIST ? (PT_BLOCK / 64) * 8192 : 16];
    __shared__ uint32_t s_idx[DIST ? PT_BLOCK + PT_BLOCK / 4 : 1];       // per wave: 64 node indices, then the leaf round's 64-byte owner map
    float4* const s_res = reinterpret_cast<float4*>(s_stage + (threadIdx.x >> 6) * 8192u) - (threadIdx.x & ~63u);            // indexed [wbase + k]
    unsigned char* const s_map = reinterpret_cast<unsigned char*>(&s_idx[(threadIdx.x >> 6) * 80u + 64u]) - (threadIdx.x & ~63u);
#else
    __shared__ float4 s_res[DIST ? TB : 1];          // per wave: 64 test results (ok, t_scaled, det, t)
    __shared__ unsigned char s_map[DIST ? TB : 1];   // per wave: work item -> owner lane
#endif
#if PT_TOP_NODES > 0 && !PT_NODE_STAGED
    __shared__ float4 s_top[DIST ? PT_TOP_NODES * 7 : 1];      // rows 0..6 (six plane rows, child references) of the first nodes of the tree
#endif
    TravCtx c;
    c.lds = &s_stack[threadIdx.x];
    c.top_lds = 0; c.top_bytes = 0; c.n_nodes_lds = 0;
    c.spill_stride = gridDim.x * TB;
    c.spill = spill + PT_DIAG_WORDS + (size_t)blockIdx.x * TB + threadIdx.x;      // the buffer's first PT_DIAG_WORDS words belong to the diagnostic builds
    c.spill_depth = spill_depth;
    c.n_nodes = 0; c.n_tris = 0; c.overflow = 0;
    c.lane_base = 0; c.stage_wave = 0; c.stage_own = 0; c.idx_pub = 0; c.idx_ld = 0;
    if constexpr (DIST) {        // slot 0 of the lane's stack: the sentinel a pop of the last entry reads back (see node_step_lean)
        s_stack[threadIdx.x] = PT_EMPTY_REF;
        c.lane_base = lds_addr_of(&s_stack[threadIdx.x]);
#if PT_NODE_STAGED
        const uint32_t ln = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        c.stage_wave = lds_addr_of(reinterpret_cast<const uint32_t*>(s_stage + wv * 8192u));
        c.stage_own = c.stage_wave + (ln >> 3) * 1024u + (ln & 7u) * 128u;
        c.idx_pub = lds_addr_of(&s_idx[wv * 80u + (ln & 7u) * 8u + (ln >> 3)]);
        c.idx_ld = lds_addr_of(&s_idx[wv * 80u + (ln >> 3) * 8u]);
#endif
    }
#if PT_TOP_NODES > 0 && !PT_NODE_STAGED
    if constexpr (DIST) {
        const uint32_t n_top = min(sc.n_top, (uint32_t)PT_TOP_NODES);
        const float4* src = reinterpret_cast<const float4*>(sc.nodes);
        for (uint32_t k = threadIdx.x; k < n_top * 7u; k += TB) { const uint32_t nd = k / 7u; s_top[k] = src[nd * 8u + (k - nd * 7u)]; }
        c.top_lds = lds_addr_of(reinterpret_cast<const uint32_t*>(s_top));
        c.top_bytes = n_top << 7;
        __syncthreads();
    }
#endif
    const uint32_t n_cur = Q.counts[PT_Q_CUR], n_sh = Q.counts[PT_Q_SHADOW], n_pr = Q.counts[PT_Q_PROBE];
    const uint32_t total = n_cur + n_sh + n_pr;
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned long long regular = 0, shadow = 0;

    // lane job: 0 idle, 1 continuation (closest), 2 shadow (any), 3 MIS probe (closest).  The three kinds are
    // independent work items; a finished ray only stores its result (hit record / occlusion flag / probe record),
    // k_nee_resolve combines the two NEE results afterwards, so retiring never waits on a load.
    int kind = 0;
    uint32_t p = 0;
    LaneRay r;
    r.sp = 0;
    r.sa = c.lane_base; r.sa_limit = 0;
    r.top = PT_EMPTY_REF;
    bool more = total > 0;
    bool sph_wait = false;      // SPH pooled-leaf kernels: this lane's parked leaf holds a sphere and waits for a sphere round (see the phase decision)
    // prefetch reservation (see the loop): stage, ticket result (lane 0), rays reserved / handed out, one ray per lane
    int pf_stage = 0, pf_kind = 0;
    uint32_t pf_raw = 0, pf_count = 0, pf_used = 0, pf_p = 0;
    uint32_t seg = blockIdx.x & 7u, seg_dry = 0;
    const uint32_t seg_len = (((total + 7u) >> 3) + 63u) & ~63u;
    float4 pf_o = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pf_d = pf_o;
#ifdef PT_PROFILE_PHASES
    // pooled-leaf kernels: 0 node issue clk, 1 node wait clk, 2 node finish clk, 3 node rounds, 4 node lanes, 5 leaf issue clk, 6 leaf wait clk,
    // 7 leaf finish clk, 8 leaf rounds, 9 leaf items, 10 service clk, 11 wave total clk, 12 general-visit clk, 13 general-visit rounds
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long prof_t0 = __builtin_readcyclecounter();
#endif
    for (;;) {
#ifdef PT_PROFILE_PHASES
        long long pt_iter = __builtin_readcyclecounter();
#endif
        // ---- ray prefetch pipeline.  A refill used to be three dependent round trips (ticket atomic, path id, ray
        // record) with the whole wave waiting; now the wave keeps the next 64 rays in registers, one per lane, and
        // fetches them one stage per iteration so that every wait is covered by a traversal step:
        //   0 -> 1  reserve 64 items (atomic)        1 -> 2  load the path ids        2 -> 3  load origin / direction
        // Idle lanes then take their ray from the reservation lane by ds_bpermute, so refills are cheap enough to
        // happen as soon as PT_REFILL_MIN lanes are free.
        if (pf_stage == 0) {
            if (more) {
                if (lane == 0) pf_raw = atomicAdd(&Q.counts[PT_Q_SEG_TICKET0 + 32u * seg], 64u);
                pf_stage = 1;
            }
        } else if (pf_stage == 1) {
            // The queue is cut into 8 segments and a wave starts on the segment of its XCD (workgroups are dealt to
            // the XCDs round-robin), so each L2 sees one contiguous, pixel-ordered slice of the rays; a wave whose
            // segment has run dry moves on to the next one.
            const uint32_t seg_lo = seg * seg_len, seg_hi = min(total, seg_lo + seg_len);
            // readfirstlane, not a shuffle: the compiler then KNOWS the ticket is wave-uniform, and with it the whole prefetch state machine
            // (stage, segment, counts) lives in scalar registers and branches on the scalar unit instead of v_cmp + EXEC masking every iteration
            const uint32_t base = seg_lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)pf_raw);
            if (seg_lo >= total || base >= seg_hi) {
                seg = (seg + 1u) & 7u;
                if (++seg_dry == 8u) more = false;
                pf_stage = 0;
            } else {
                seg_dry = 0;
                pf_count = min(64u, seg_hi - base);
                pf_used = 0;
                const uint32_t item = base + lane;
                if (lane < pf_count) {
                    if (item < n_cur) { pf_p = Q.cur[item]; pf_kind = 1; }
                    else if (item < n_cur + n_sh) { pf_p = Q.shadow[item - n_cur]; pf_kind = 2; }
                    else { pf_p = Q.probe[item - n_cur - n_sh]; pf_kind = 3; }
                }
                pf_stage = 2;
            }
        } else if (pf_stage == 2) {
            if (lane < pf_count) {
                const float4* so = pf_kind == 1 ? P.ray_o : (pf_kind == 2 ? P.sh_o : P.pr_o);
                const float4* sd = pf_kind == 1 ? P.ray_d : (pf_kind == 2 ? P.sh_d : P.pr_d);
                pf_o = so[pf_p]; pf_d = sd[pf_p];
            }
            pf_stage = 3;
        }
#ifdef PT_PROFILE_PHASES
        const long long pt_pf = __builtin_readcyclecounter();
        prof[14] += (unsigned long long)(pt_pf - pt_iter);       // the prefetch state machine
#endif
        // ---- retire finished rays: stores only
        if (kind != 0 && ray_done(r)) {
            if (kind == 1) { P.hit_t[p] = r.ray_tmax; P.hit_rec[p] = r.best; if (INST) P.hit_inst[p] = r.best_inst; }
            else if (kind == 2) P.occluded[p] = r.best >= 0 ? 1 : 0;
            else P.probe_rec[p] = r.best;
            kind = 0;
        }
        // ---- hand prefetched rays to idle lanes (one ray_begin site: it is long)
        {
            const unsigned long long idle = __ballot(kind == 0);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (pf_stage == 3 && (n_idle >= PT_REFILL_MIN || idle == ~0ull || (!more && n_idle != 0))) {
                const uint32_t take = min(n_idle, pf_count - pf_used);
                const uint32_t rank = (uint32_t)__popcll(idle & below);
                const int src = (int)((pf_used + rank) & 63u);
                const uint32_t np = (uint32_t)__shfl((int)pf_p, src, 64);
                const int nk = __shfl(pf_kind, src, 64);
                float4 ro, rd;
                ro.x = __shfl(pf_o.x, src, 64); ro.y = __shfl(pf_o.y, src, 64); ro.z = __shfl(pf_o.z, src, 64); ro.w = __shfl(pf_o.w, src, 64);
                rd.x = __shfl(pf_d.x, src, 64); rd.y = __shfl(pf_d.y, src, 64); rd.z = __shfl(pf_d.z, src, 64);
                if (kind == 0 && rank < take) {
                    p = np; kind = nk;
                    if (nk == 2) shadow++; else regular++;
                    if constexpr (DIST) ray_begin_fs(sc, r, c, f4_3(ro), mk3(rd.x, rd.y, rd.z), ro.w);
                    else ray_begin(sc, r, f4_3(ro), mk3(rd.x, rd.y, rd.z), ro.w);
                }
                pf_used += take;
                if (pf_used == pf_count) pf_stage = 0;
            }
        }
#ifdef PT_PROFILE_PHASES
        prof[15] += (unsigned long long)(__builtin_readcyclecounter() - pt_pf);       // retiring and handing out rays
#endif
        if (__ballot(kind != 0) == 0) {
            if (!more && pf_stage == 0) break;
            continue;
        }
        // ---- one traversal phase for the whole wave.  A node visit (one 128-byte line, 4 slab tests)
        // and a leaf visit (1-4 watertight triangle tests) are long, different code paths; running both
        // every iteration would execute each for a fraction of the lanes.  The wave therefore does ONE
        // kind per iteration: leaves once PT_LEAF_MIN lanes are parked on one (or nobody has a node to
        // visit), node visits otherwise.
        bool w_tri = kind != 0 && ray_wants_tri(r);
        bool w_node = kind != 0 && ray_wants_node(r);
        unsigned long long m_tri = __ballot(w_tri), m_node = __ballot(w_node);
#ifdef PT_PROFILE_PHASES
        long long pt0 = __builtin_readcyclecounter();
        prof[10] += (unsigned long long)(pt0 - pt_iter);
#endif
        if constexpr (DIST) {
            // triangles parked: the count rides in bits 28..30 of the leaf reference
            // Scenes with spheres: a leaf that holds a sphere waits for a round of its own.  The sphere test is ~600 instructions of interval
            // arithmetic against ~80 for a triangle, and in a pooled round the whole wave walks through it for the one or two lanes that
            // have a sphere (with a sphere LIGHT every shadow ray ends in its leaf: k_trace +23 %).  A normal round therefore only finds
            // out that a leaf holds a sphere (its records are loaded anyway) and leaves it parked with `sph_wait` set; once PT_SPH_LANES_MIN
            // lanes wait like that -- or nothing else can run -- a sphere round serves exactly those leaves, spheres and triangles alike.
            // Every leaf is still tested whole, by its owner, in leaf order: results and counters do not change.
            constexpr bool SPHDEF = SPH && !PT_NODE_STAGED && PT_LEAF_LANES_MIN > 0;      // (the experiment builds test spheres in every round, as before)
            bool sround = SPH && !SPHDEF;              // this iteration's leaf round is a sphere round
            unsigned long long m_tri_s = 0;
            if constexpr (SPHDEF) {
                m_tri_s = __ballot(w_tri && sph_wait);
                m_tri &= ~m_tri_s;
            }
            const bool sph_go = SPHDEF && (uint32_t)__popcll(m_tri_s) >= (uint32_t)PT_SPH_LANES_MIN;
            const uint32_t tcnt = w_tri ? ((r.top >> PT_LEAF_COUNT_SHIFT) & 7u) + 1u : 0u;
            uint32_t tcnt_r = tcnt;                    // ... of the lanes the coming round serves (0 for the others)
            bool w_serve = w_tri;
#if PT_LEAF_LANES_MIN > 0
            // the leaf-round trigger from the number of parked LANES (experiment: the exact count of parked tests costs four ballots and
            // popcounts per iteration, needed by every iteration's phase decision; leaf_issue computes its own prefix sums when a round runs)
            unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            const uint32_t n_parked = (sph_go || (uint32_t)__popcll(m_tri) >= (uint32_t)PT_LEAF_LANES_MIN) ? (uint32_t)PT_LEAF_TRIS_MIN : 0u;
#else
            const unsigned long long c0 = __ballot((tcnt & 1u) != 0), c1 = __ballot((tcnt & 2u) != 0), c2 = __ballot((tcnt & 4u) != 0),
                                     c3 = __ballot((tcnt & 8u) != 0);
            const uint32_t n_parked = (uint32_t)(__popcll(c0) + 2 * __popcll(c1) + 4 * __popcll(c2) + 8 * __popcll(c3));
#endif
            // ---- distributed leaf round, in two halves so that its triangle loads can fly together with a node round's fetches.
            // issue: owners are served in lane order while their whole leaf fits (the others stay parked); lane w becomes the helper
            // of triangle k of owner o, takes the owner's ray constants by ds_bpermute and starts the 48-byte record's load.
            const uint32_t wbase = threadIdx.x & ~63u;
            uint32_t lf_pre = 0, lf_items = 0;
            bool lf_served = false, lf_valid = false;
            int lf_kk = 0;
            float lf_tmax = PT_INF;             // sphere rounds: the owner's t_max when the round starts (an upper bound of what its leaf walk will compare with)
            RayPre lf_rp;
            TriVerts lf_tv;
            auto leaf_issue = [&](auto sr_tag) {
                constexpr bool SR = decltype(sr_tag)::value;      // compile-time: this copy is the sphere round (two copies: the normal round carries no sphere code)
                (void)SR;
#if PT_LEAF_LANES_MIN > 0
                c0 = __ballot((tcnt_r & 1u) != 0); c1 = __ballot((tcnt_r & 2u) != 0); c2 = __ballot((tcnt_r & 4u) != 0); c3 = __ballot((tcnt_r & 8u) != 0);
#endif
                lf_pre = (uint32_t)(__popcll(c0 & below) + 2 * __popcll(c1 & below) + 4 * __popcll(c2 & below) + 8 * __popcll(c3 & below));
                lf_served = w_serve && lf_pre + tcnt_r <= 64u;
                const unsigned long long m_served = __ballot(lf_served);
                const int last = 63 - __clzll(m_served);                      // m_served != 0: the first parked lane always fits
                lf_items = (uint32_t)__shfl((int)(lf_pre + tcnt_r), last, 64);
                if (lf_served)
                    for (uint32_t k = 0; k < tcnt_r; k++) s_map[wbase + lf_pre + k] = (unsigned char)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                lf_valid = lane < lf_items;
                const int o = lf_valid ? (int)s_map[wbase + lane] : 0;
                const uint32_t k = lane - (uint32_t)__shfl((int)lf_pre, o, 64);
                const uint32_t first = (uint32_t)__shfl((int)(r.top & PT_LEAF_FIRST_MASK), o, 64);
                lf_rp.o = mk3(__shfl(r.rp.o.x, o, 64), __shfl(r.rp.o.y, o, 64), __shfl(r.rp.o.z, o, 64));
                lf_rp.d = mk3(0.0f, 0.0f, 0.0f);        // only the one-sided test (and a sphere) reads the direction
                if ((SPH && SR) || sc.any_one_sided) lf_rp.d = mk3(__shfl(r.rp.d.x, o, 64), __shfl(r.rp.d.y, o, 64), __shfl(r.rp.d.z, o, 64));     // (no sphere is tested outside a sphere round)
                if constexpr (SPH && SR) lf_tmax = __shfl(r.ray_tmax, o, 64);
                lf_kk = __shfl(r.rp.kx | (r.rp.ky << 2) | (r.rp.kz << 4) | ((kind == 2 ? 1 : 0) << 6), o, 64);
                lf_rp.kx = lf_kk & 3; lf_rp.ky = (lf_kk >> 2) & 3; lf_rp.kz = (lf_kk >> 4) & 3;
                lf_rp.sx = __shfl(r.rp.sx, o, 64); lf_rp.sy = __shfl(r.rp.sy, o, 64); lf_rp.sz = __shfl(r.rp.sz, o, 64);
                lf_rp.dperm = lf_rp.d;        // not read by tri_core
                lf_tv.p0 = lf_rp.d; lf_tv.p1 = lf_rp.d; lf_tv.p2 = lf_rp.d; lf_tv.prim = 0; lf_tv.flags = 0;
                if (lf_valid) lf_tv = load_tri(sc.tris, first + k);
            };
            // finish: every helper runs its (ray, triangle) test, results go to LDS, the owner walks its leaf's results in leaf order
            // and applies the one comparison that depends on the ray's shrinking t_max with the t_max each triangle would have seen
            auto leaf_finish = [&](auto sr_tag) {
                constexpr bool SR = decltype(sr_tag)::value;
                (void)SR;
                bool sphere_rec = false;
                if constexpr (SPH) sphere_rec = lf_valid && (lf_tv.flags & PT_TRI_SPHERE) != 0;
                unsigned long long m_sitems = 0;            // normal round: the items that are spheres (item i is helper lane i)
                if constexpr (SPHDEF && !SR) { m_sitems = __ballot(sphere_rec); }
                if (lf_valid) {
                    // The triangle test runs for EVERY item, a sphere's record included (its fields are not vertices: the result is thrown
                    // away).  With the test behind `if (!sphere)` the compiler moved the record's loads into the two branches, behind the
                    // load of the flags they depend on: a second memory round trip in every leaf round of the sphere-capable kernel
                    // (5 760 against 4 340 clocks per round on a scene WITHOUT spheres, -DPT_PROFILE_PHASES).
                    TriCore tc;
                    const bool tri_ok = tri_core(lf_rp, lf_tv.p0, lf_tv.p1, lf_tv.p2, lf_tv.flags, tc);
                    float4 res = make_float4(tri_ok ? 1.0f : 0.0f, tc.t_scaled, tc.det, tc.t);
                    if (SR && sphere_rec) {       // Sphere::intersect(_p) against the owner's t_max as the round starts; the owner applies the t_max tests again, in
                        // leaf order, with the t_max each item would have seen (never larger: a test that fails here fails there, and the value-lane
                        // reject inside sph_hit_test_inl ends nearly every shadow ray's test before the interval arithmetic starts)
                        SphHit sh;
                        sh.t = 0.0f; sh.a_hi = 0.0f; sh.b_hi = 0.0f;
                        const bool ok = sph_hit_test_inl(sc.spheres[__float_as_uint(lf_tv.p0.x)], lf_rp.o, lf_rp.d, lf_tmax, (lf_kk & 64) ? 2.0f * PT_PI : PT_PI, &sh);
                        res = make_float4(ok ? 2.0f : 0.0f, sh.a_hi, sh.b_hi, sh.t);
                    }
                    // (a sphere in a normal round: the owner sees the item in m_sitems and leaves its leaf parked for a sphere round)
                    s_res[wbase + lane] = res;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                bool walk = lf_served;
                if constexpr (SPHDEF) {
                    if (!SR && lf_served && ((m_sitems >> lf_pre) & ((1ull << tcnt_r) - 1ull)) != 0ull) { sph_wait = true; walk = false; }     // a sphere among this leaf's items
                    if (SR && lf_served) sph_wait = false;
                }
                if (walk) {
                    const uint32_t rec0 = fs_pop(c, r.sa, r.top) & PT_LEAF_FIRST_MASK;
                    const bool any_hit = kind == 2;
                    bool leaf_hit = false;
                    uint32_t tested = tcnt;
#if PT_WALK_BATCH
                    // The results come four at a time: the loop's one LDS read per item was a round trip per item for the owner with the
                    // fullest leaf, and the whole wave waits for that owner.  (Slots past the leaf's own are read and ignored: they lie in
                    // the wave's 64 and hold other owners' results.)
                    bool stop = false;
                    for (uint32_t k0 = 0; k0 < tcnt && !stop; k0 += 4u) {
                        const uint32_t b = wbase + lf_pre + k0;
                        const float4 va = s_res[b], vb = s_res[wbase + min(lf_pre + k0 + 1u, 63u)], vc = s_res[wbase + min(lf_pre + k0 + 2u, 63u)],
                                     vd = s_res[wbase + min(lf_pre + k0 + 3u, 63u)];
                        auto step = [&](const float4 v, uint32_t k) {
                            if (stop || k >= tcnt) return;
                            bool acc;
                            if (SPH && SR && v.x == 2.0f) acc = !(v.y > r.ray_tmax) && !(v.z > r.ray_tmax);      // (a sphere's result: only a sphere round makes one)
                            else acc = v.x != 0.0f && tri_accept(v.y, v.z, r.ray_tmax);
                            if (acc) {
                                r.best = (int32_t)(rec0 + k); leaf_hit = true;
                                if (any_hit) { r.sa = c.lane_base; r.top = PT_EMPTY_REF; tested = k + 1; stop = true; return; }
                                r.ray_tmax = v.w;
                            }
                        };
                        step(va, k0); step(vb, k0 + 1u); step(vc, k0 + 2u); step(vd, k0 + 3u);
                    }
#else
                    for (uint32_t k = 0; k < tcnt; k++) {
                        const float4 v = s_res[wbase + lf_pre + k];
                        bool acc;
                        if (SPH && SR && v.x == 2.0f) acc = !(v.y > r.ray_tmax) && !(v.z > r.ray_tmax);      // (a sphere's result: only a sphere round makes one)
                        else acc = v.x != 0.0f && tri_accept(v.y, v.z, r.ray_tmax);
                        if (acc) {
                            r.best = (int32_t)(rec0 + k); leaf_hit = true;
                            if (any_hit) { r.sa = c.lane_base; r.top = PT_EMPTY_REF; tested = k + 1; break; }
                            r.ray_tmax = v.w;
                        }
                    }
#endif
                    c.n_tris += tested;
                    if (leaf_hit && !any_hit) r.tmax = r.ray_tmax;
                }
            };
#ifdef PT_FORCE_GENERAL_VISIT
            const bool lean = false;                                                // experiment: the first version's visit for everyone
#elif defined(PT_EXP_NO_GENERAL_VISIT)
            const bool lean = true;                                                 // timing experiment (wrong for NaN-exact lanes): the general visit compiled out
#else
            const bool lean = __ballot(w_node && r.sa >= r.sa_limit) == 0ull;      // nobody NaN-exact, nobody near the LDS part's end
#endif
#if !PT_NODE_STAGED
            if (lean && m_node != 0 && n_parked < PT_LEAF_TRIS_MIN) {
                PT_PROF_T(t0);
#if PT_NODE_COOP
                if constexpr (TOUCH != 0 || PT_NODE_COOP == 2) node_step_coop<TOUCH>(sc, r, c, w_node, m_node);
                else if (w_node) node_step_lean<TOUCH>(sc, r, c);
#else
                if (w_node) node_step_lean<TOUCH>(sc, r, c);
#endif
                PT_PROF_T(t1);
#ifdef PT_PROFILE_PHASES
                prof[2] += t1 - t0; prof[3] += 1; prof[4] += (unsigned long long)__popcll(m_node);
#endif
#else
            if (lean) {
                // Staged node round and / or leaf round.  With PT_LEAF_TRIS_FUSED > 0 a leaf round rides along a node round once that
                // many tests are parked, so that the node round's LDS-DMA fetches and the leaf round's record loads fly together
                // (measured: slower -- the smaller leaf rounds cost more than the shared round trip saves; kept as an experiment switch).
                const bool do_node = m_node != 0 && (PT_LEAF_TRIS_FUSED > 0 || n_parked < PT_LEAF_TRIS_MIN);
                const bool do_leaf = m_tri != 0 && (!do_node || (PT_LEAF_TRIS_FUSED > 0 && n_parked >= PT_LEAF_TRIS_FUSED));
                (void)do_leaf;
#if PT_LEAF_TRIS_FUSED > 0
                uint32_t below_top = PT_EMPTY_REF;
                PT_PROF_T(t0);
                if (do_node) below_top = node_round_issue(sc, r, c, w_node);
                PT_PROF_T(t1);
                if (do_leaf) leaf_issue(std::integral_constant<bool, SPH>{});
                PT_PROF_T(t2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA writes have landed (nothing else orders a ds_read behind them)
                __builtin_amdgcn_wave_barrier();
                PT_PROF_T(t3);
                if (do_node) node_round_finish(r, c, w_node, below_top);
                PT_PROF_T(t4);
                if (do_leaf) leaf_finish(std::integral_constant<bool, SPH>{});
                PT_PROF_T(t5);
#else
                unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
                (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5;
                PT_PROF_SET(t0);
                if (do_node) {
                    const uint32_t below_top = node_round_issue(sc, r, c, w_node);
                    PT_PROF_SET(t1); PT_PROF_SET(t2);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA writes have landed (nothing else orders a ds_read behind them)
                    __builtin_amdgcn_wave_barrier();
                    PT_PROF_SET(t3);
                    node_round_finish(r, c, w_node, below_top);
                    PT_PROF_SET(t4); PT_PROF_SET(t5);
                } else {
                    PT_PROF_SET(t1);
                    leaf_issue(std::integral_constant<bool, SPH>{});
                    PT_PROF_SET(t2);
#ifdef PT_PROFILE_PHASES
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    PT_PROF_SET(t3); PT_PROF_SET(t4);
                    leaf_finish(std::integral_constant<bool, SPH>{});
                    PT_PROF_SET(t5);
                }
#endif
#ifdef PT_PROFILE_PHASES
                if (do_node) { prof[0] += t1 - t0; prof[2] += t4 - t3; prof[3] += 1; prof[4] += (unsigned long long)__popcll(m_node); }
                if (do_leaf) { prof[5] += t2 - t1; prof[7] += t5 - t4; prof[8] += 1; prof[9] += (unsigned long long)lf_items; }
                prof[do_node ? 1 : 6] += t3 - t2;
#endif
#endif
            } else if (m_node != 0 && n_parked < PT_LEAF_TRIS_MIN) {
                PT_PROF_T(t0);
                if (w_node) node_step_general(sc, r, c);
                PT_PROF_T(t1);
#ifdef PT_PROFILE_PHASES
                prof[12] += t1 - t0; prof[13] += 1; prof[4] += (unsigned long long)__popcll(m_node);
#endif
            } else if ((m_tri | m_tri_s) != 0) {
                if constexpr (SPHDEF) {
                    sround = sph_go || m_tri == 0;          // the sphere leaves' turn: enough of them wait, or nothing else can run
                    w_serve = w_tri && (sph_wait == sround);
                    tcnt_r = w_serve ? tcnt : 0u;
                }
                PT_PROF_T(t0);
                if (SPHDEF && sround) leaf_issue(std::true_type{}); else leaf_issue(std::integral_constant<bool, SPH && !SPHDEF>{});
#ifdef PT_PROFILE_PHASES
                PT_PROF_T(t0b);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PT_PROF_T(t0c);
                prof[5] += t0b - t0; prof[6] += t0c - t0b;
#endif
                if (SPHDEF && sround) leaf_finish(std::true_type{}); else leaf_finish(std::integral_constant<bool, SPH && !SPHDEF>{});
                PT_PROF_T(t1);
#ifdef PT_PROFILE_PHASES
                prof[7] += t1 - t0c; prof[8] += 1; prof[9] += (unsigned long long)lf_items;
#endif
            }
        } else {
        if (m_node != 0 && __popcll(m_tri) < (INST ? PT_LEAF_MIN_INST : PT_LEAF_MIN)) {
            if (w_node) ray_step_node(sc, r, c);
#ifdef PT_PROFILE_PHASES
            prof[12] += (unsigned long long)(__builtin_readcyclecounter() - pt0); prof[13] += 1; prof[4] += (unsigned long long)__popcll(m_node);
#endif
        } else {
            if constexpr (INST) {
                if (w_tri) {
                    if (r.top == PT_INST_EXIT_REF) {          // back from an object's tree (ray_step_tri_enter): the world ray again, from where it was handed out
                        (void)stk_pop(c, r.top, r.sp);
                        const float w_tmax = __uint_as_float(stk_pop(c, r.top, r.sp)), w_tmin = __uint_as_float(stk_pop(c, r.top, r.sp));
                        const float4 ro = (kind == 1 ? P.ray_o : (kind == 2 ? P.sh_o : P.pr_o))[p], rd = (kind == 1 ? P.ray_d : (kind == 2 ? P.sh_d : P.pr_d))[p];
                        const bool hit_inside = r.best >= 0 && r.best_inst == r.cur_inst;      // (an instance is one primitive of the world's tree: entered once per ray)
                        ray_set_direction_state(r, f4_3(ro), mk3(rd.x, rd.y, rd.z));
                        r.tmin = w_tmin; r.tmax = hit_inside ? r.ray_tmax : w_tmax;
                        r.cur_inst = 0;
                    } else ray_step_tri_enter<SPH>(sc, r, kind == 2, c);
                }
            } else {
                if (w_tri) ray_step_tri<SPH, INST>(sc, r, kind == 2, c);
            }
#ifdef PT_PROFILE_PHASES
            prof[7] += (unsigned long long)(__builtin_readcyclecounter() - pt0); prof[8] += 1; prof[9] += (unsigned long long)__popcll(m_tri);
#endif
        }
        }
    }
#ifdef PT_PROFILE_PHASES
    prof[11] = (unsigned long long)(__builtin_readcyclecounter() - prof_t0);
    if (lane == 0) for (int i = 0; i < 16; i++) atomicAdd(reinterpret_cast<unsigned long long*>(spill) + i, prof[i]);   // diagnostic build only: the buffer's diagnostic words
#endif
    if (c.overflow) atomicOr(err, 1u);
    flush_counters(cnt, s_cnt, regular, shadow, c.n_nodes, c.n_tris);
    {   // LDS-served node visits: one add per wave
        unsigned long long v = c.n_nodes_lds;
        for (int o = 32; o > 0; o >>= 1) v += (unsigned long long)__shfl_xor((int)(v >> 32), o, 64) << 32 | (unsigned int)__shfl_xor((int)(unsigned int)v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&cnt->nodes_lds, v);
    }
}
extern "C" __global__ void __launch_bounds__(PT_TBLOCK, PT_TRACE_DIST_WAVES) k_trace(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt, uint32_t* spill,
                                                              uint32_t spill_depth, uint32_t* err) {
    trace_body<true, false>(sc, P, Q, cnt, spill, spill_depth, err);
}
// the same with nodes fetched pairwise (node_step_coop): scenes whose rays miss the caches, chosen per scene by a timed trial (pt_context.cpp)
extern "C" __global__ void __launch_bounds__(PT_TBLOCK, PT_TRACE_DIST_WAVES) k_trace_far(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt, uint32_t* spill,
                                                                   uint32_t spill_depth, uint32_t* err) {
    trace_body<true, false, false, PT_TOUCH_VARIANT>(sc, P, Q, cnt, spill, spill_depth, err);
}
// leaves of more than 8 triangles ("maxnodeprims" > 8): every lane walks its own leaf
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_TRACE_WAVES) k_trace_seq(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt, uint32_t* spill,
                                                                  uint32_t spill_depth, uint32_t* err) {
    trace_body<false, false>(sc, P, Q, cnt, spill, spill_depth, err);
}
// scenes with object instances (and possibly spheres): a leaf record may stand for a TransformedPrimitive
#ifndef PT_TRACE_INST_WAVES
#define PT_TRACE_INST_WAVES 3      // 164 registers, 14 spilled; at two (154 / 41 as the compiler chose) 615 / 1 130 / 433 Mrays/s on three instanced scenes, here 628 / 1 184 / 440;
#endif                            // a triangle-only instantiation for scenes without spheres (162, nothing spilled) is no faster: 637 / 1 165 / 444

extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_TRACE_INST_WAVES) k_trace_inst(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt, uint32_t* spill,
                                                                       uint32_t spill_depth, uint32_t* err) {
    trace_body<false, true, true>(sc, P, Q, cnt, spill, spill_depth, err);
}
// scenes with spheres: a leaf record may stand for a sphere
extern "C" __global__ void __launch_bounds__(PT_TBLOCK, PT_TRACE_DIST_WAVES) k_trace_sph_dist(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt, uint32_t* spill,
                                                                           uint32_t spill_depth, uint32_t* err) {
    trace_body<true, true>(sc, P, Q, cnt, spill, spill_depth, err);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, 2) k_trace_sph(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt, uint32_t* spill,
                                                                      uint32_t spill_depth, uint32_t* err) {
    trace_body<false, true>(sc, P, Q, cnt, spill, spill_depth, err);
}

// ============================================================ hooks: plain ray batches
template <bool SPH, bool INST = false>
PT_DEV void trace_batch_body(const PtScene& sc, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out, uint8_t* occ_out, int any_hit,
                             uint32_t* ticket, PtCounters* cnt, uint32_t* spill, uint32_t spill_depth, uint32_t* err) {
    __shared__ uint32_t s_stack[PT_LDS_STACK * PT_BLOCK];
    __shared__ unsigned long long s_cnt[4];
    TravCtx c;
    c.lds = &s_stack[threadIdx.x];
    c.spill_stride = gridDim.x * PT_BLOCK;
    c.spill = spill + PT_DIAG_WORDS + (size_t)blockIdx.x * PT_BLOCK + threadIdx.x;      // the buffer's first PT_DIAG_WORDS words belong to the diagnostic builds
    c.spill_depth = spill_depth;
    c.n_nodes = 0; c.n_tris = 0; c.overflow = 0;
    unsigned long long regular = 0, shadow = 0;
    for (;;) {
        uint32_t base = wave_ticket(ticket);
        if (base >= n) break;
        uint32_t i = base + (threadIdx.x & 63);
        if (i >= n) continue;
        V3 ro = ld3(o + 3 * (size_t)i), rd = ld3(d + 3 * (size_t)i);
        if (any_hit) {
            shadow++;
            occ_out[i] = trace_any<SPH, INST>(sc, ro, rd, tmax[i], c) ? 1 : 0;
        } else {
            regular++;
            float t;
            uint32_t inst = 0;
            int32_t rec = trace_closest<SPH, INST>(sc, ro, rd, tmax[i], c, &t, &inst);
            pt_hit h;
            h.t = 0.0f; h.prim = -1; h.b0 = 0.0f; h.b1 = 0.0f;
            if (rec >= 0 && INST && inst) {          // the hit is reported as the instance's own world primitive
                h.t = t; h.prim = (int32_t)sc.instances[inst - 1u].world_prim;
            } else if (rec >= 0) {
                TriVerts tv = load_tri(sc.tris, (uint32_t)rec);
                h.t = t; h.prim = (int32_t)tv.prim;
                if (!(SPH && (tv.flags & PT_TRI_SPHERE))) {
                    RayPre rp;
                    ray_precompute(rp, ro, rd);
                    TriHit th;
                    tri_test(rp, tv.p0, tv.p1, tv.p2, tv.flags, PT_INF, th);   // same arithmetic => same t, b
                    h.b0 = th.b0; h.b1 = th.b1;
                }
            }
            out[i] = h;
        }
    }
    if (c.overflow) atomicOr(err, 1u);
    flush_counters(cnt, s_cnt, regular, shadow, c.n_nodes, c.n_tris);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_trace_batch(PtScene sc, uint32_t n, const float* o, const float* d, const float* tmax,
                                                                    pt_hit* out, uint8_t* occ_out, int any_hit, uint32_t* ticket, PtCounters* cnt,
                                                                    uint32_t* spill, uint32_t spill_depth, uint32_t* err) {
    trace_batch_body<false>(sc, n, o, d, tmax, out, occ_out, any_hit, ticket, cnt, spill, spill_depth, err);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_trace_batch_sph(PtScene sc, uint32_t n, const float* o, const float* d, const float* tmax,
                                                                        pt_hit* out, uint8_t* occ_out, int any_hit, uint32_t* ticket, PtCounters* cnt,
                                                                        uint32_t* spill, uint32_t spill_depth, uint32_t* err) {
    trace_batch_body<true>(sc, n, o, d, tmax, out, occ_out, any_hit, ticket, cnt, spill, spill_depth, err);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_trace_batch_inst(PtScene sc, uint32_t n, const float* o, const float* d, const float* tmax,
                                                                         pt_hit* out, uint8_t* occ_out, int any_hit, uint32_t* ticket, PtCounters* cnt,
                                                                         uint32_t* spill, uint32_t spill_depth, uint32_t* err) {
    trace_batch_body<true, true>(sc, n, o, d, tmax, out, occ_out, any_hit, ticket, cnt, spill, spill_depth, err);
}

// ============================================================ hook: caller rays through the wavefront's own traversal kernel
// pt_trace_wavefront loads caller rays into the path pool as the three kinds of work items a bounce mixes in one launch and
// runs ptk_trace (k_trace / k_trace_seq / k_trace_sph_dist / k_trace_sph / k_trace_inst -- whatever the scene renders with).
// This kernel turns what the traversal stored (hit_t / hit_rec / hit_inst, occluded, probe_rec) into the hook's outputs.
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_wavefront_results(PtScene sc, PtPaths P, uint32_t n, const uint8_t* kind, pt_hit* out, uint8_t* occ) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t k = kind[i];
        pt_hit h;
        h.t = 0.0f; h.prim = -1; h.b0 = 0.0f; h.b1 = 0.0f;
        uint8_t oc = 0;
        if (k == 2u) oc = P.occluded[i];
        else {
            const int32_t rec = k == 1u ? P.hit_rec[i] : P.probe_rec[i];
            if (rec >= 0) {
                const uint32_t inst = (k == 1u && sc.n_instances) ? P.hit_inst[i] : 0u;
                TriVerts tv = load_tri(sc.tris, (uint32_t)rec);
                h.prim = inst ? (int32_t)sc.instances[inst - 1u].world_prim : (int32_t)tv.prim;
                if (k == 1u) {
                    h.t = P.hit_t[i];
                    if (!inst && !(tv.flags & (PT_TRI_SPHERE | PT_TRI_INSTANCE))) {
                        RayPre rp;
                        ray_precompute(rp, f4_3(P.ray_o[i]), f4_3(P.ray_d[i]));
                        TriHit th;
                        tri_test(rp, tv.p0, tv.p1, tv.p2, tv.flags, PT_INF, th);   // same arithmetic => same b
                        h.b0 = th.b0; h.b1 = th.b1;
                    }
                }
            }
        }
        out[i] = h;
        occ[i] = oc;
    }
}
hipError_t ptk_wavefront_results(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, uint32_t n, const uint8_t* kind, pt_hit* out, uint8_t* occ) {
    hipLaunchKernelGGL(k_wavefront_results, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, n, kind, out, occ);
    return hipGetLastError();
}

// ============================================================ tiles -> pixel list
// One block per tile: writes the tile's pixels (row-major, x | y << 16 relative to the sample bounds) at its offset of the
// pass's pixel list, and marks them in a bitmap over the sample bounds -- a pixel marked twice means overlapping tiles
// (k_film folds a pixel's samples with a plain read-modify-write, so the tiles of one call must be disjoint).
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_expand_tiles(const int4* tiles, const uint32_t* tile_off, uint32_t n_tiles, int32_t sb_x0, int32_t sb_y0,
                                                                     uint32_t sb_w, uint32_t* pixels, uint32_t* bitmap, uint32_t* err) {
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int4 tl = tiles[t];
        const uint32_t w = (uint32_t)(tl.z - tl.x), n = w * (uint32_t)(tl.w - tl.y), off = tile_off[t];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t x = (uint32_t)(tl.x - sb_x0) + i % w, y = (uint32_t)(tl.y - sb_y0) + i / w;
            pixels[off + i] = x | (y << 16);
            const uint32_t bit = y * sb_w + x;
            if (atomicOr(&bitmap[bit >> 5], 1u << (bit & 31u)) & (1u << (bit & 31u))) atomicOr(err, 2u);
        }
    }
}
hipError_t ptk_expand_tiles(hipStream_t st, const int4* tiles, const uint32_t* tile_off, uint32_t n_tiles, int32_t sb_x0, int32_t sb_y0, uint32_t sb_w,
                            uint32_t* pixels, uint32_t* bitmap, uint32_t* err) {
    hipLaunchKernelGGL(k_expand_tiles, dim3(n_tiles < 4096u ? n_tiles : 4096u), dim3(PT_BLOCK), 0, st, tiles, tile_off, n_tiles, sb_x0, sb_y0, sb_w, pixels, bitmap, err);
    return hipGetLastError();
}

// ============================================================ K_GEN: camera samples
// path i of the pass: pixel = pixels[i % n_pix], sample = s0 + i / n_pix
// (render_tile, sampler.rs:221-251: start_pixel / get_camera_sample / generate_ray_differential)
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_gen(PtScene sc, PtPaths P, PtQueues Q, const uint32_t* pixels, uint32_t n_pix,
                                                            uint32_t s0, uint32_t n_samples, PtCounters* cnt) {
    uint32_t n = n_pix * n_samples;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t pk = pixels[i % n_pix];
        uint32_t s = s0 + i / n_pix;
        int32_t rx = (int32_t)(pk & 0xffffu), ry = (int32_t)(pk >> 16);
        int32_t px = rx + sc.film.sample_bounds[0], py = ry + sc.film.sample_bounds[1];
        Sampler sm;
        sm.index = sampler_index(sc, s, px, py);
        sm.dim = 0; sm.px = px; sm.py = py;
        V2 uf = sm.get_2d(sc);
        V2 p_film = mk2((float)px + uf.x, (float)py + uf.y);
        V2 u_lens = sm.get_2d(sc);
        (void)sm.get_1d(sc);   // time
        V3 o, d;
        generate_camera_ray(sc, p_film, u_lens, &o, &d);
        P.ray_o[i] = make_float4(o.x, o.y, o.z, PT_INF);
        P.ray_d[i] = make_float4(d.x, d.y, d.z, 0.0f);
        P.beta[i] = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        P.L[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        P.p_film[i] = make_float2(p_film.x, p_film.y);
        P.sobol_index[i] = sm.index;
        P.pixel[i] = pk;
        P.state[i] = sm.dim | (PT_ST_CAMERA << 24);          // dim = 5, bounces = 0, flags = camera ray
        P.nee[i] = 0;
        Q.cur[i] = i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        Q.counts[PT_Q_CUR] = n; Q.counts[PT_Q_NEXT] = 0; Q.counts[PT_Q_NEE] = 0; Q.counts[PT_Q_TICKET] = 0; Q.counts[PT_Q_SHADOW] = 0; Q.counts[PT_Q_PROBE] = 0;
        for (uint32_t k = 0; k < 8u; k++) Q.counts[PT_Q_SEG_TICKET0 + 32u * k] = 0;
        atomicAdd(&cnt->camera_rays, (unsigned long long)n);
    }
}

// queue bookkeeping between stages (single thread)
extern "C" __global__ void k_prep(PtQueues Q, int mode) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (mode == 0) {            // before SHADE: nee and next start empty
            Q.counts[PT_Q_NEXT] = 0; Q.counts[PT_Q_NEE] = 0; Q.counts[PT_Q_TICKET] = 0; Q.counts[PT_Q_TICKET2] = 0; Q.counts[PT_Q_TICKET3] = 0; Q.counts[PT_Q_SHADOW] = 0; Q.counts[PT_Q_PROBE] = 0;
            Q.counts[PT_Q_TICKET_N1] = 0; Q.counts[PT_Q_TICKET_N2] = 0; Q.counts[PT_Q_TICKET_N3] = 0;
            for (uint32_t k = 0; k < 8u; k++) Q.counts[PT_Q_SEG_TICKET0 + 32u * k] = 0;
        } else {                    // after SHADE: next becomes cur (host swaps the pointers)
            Q.counts[PT_Q_CUR] = Q.counts[PT_Q_NEXT]; Q.counts[PT_Q_NEXT] = 0; Q.counts[PT_Q_TICKET] = 0;
            for (uint32_t k = 0; k < 8u; k++) Q.counts[PT_Q_SEG_TICKET0 + 32u * k] = 0;
        }
    }
}

// ============================================================ shading helpers
struct Surf {            // what SurfaceInteraction carries for this path (surface_interaction.rs:25-57)
    V3 p, p_error, n, wo;
    V3 sh_n, sh_dpdu;
    uint32_t prim;
    int32_t material, light;     // from the triangle record (-1 = none)
    V2 uv;                       // read by textures only (dead in the kernels without them)
    V3 dpdu, dpdv;               // geometric partials (compute_differentials)
    V3 sh_dpdv, sh_dndu, sh_dndv;    // bump mapping
};

// Triangle::get_dpdu_dpdv (triangle.rs:132-186)
PT_DEV void tri_dpdu(const PtScene& sc, bool has_attr, const PtTriInfo& ti, V3 p0, V3 p1, V3 p2, V3* dpdu, V3* dpdv, V2* uvs) {
    V2 uv0 = mk2(0.0f, 0.0f), uv1 = mk2(1.0f, 0.0f), uv2 = mk2(1.0f, 1.0f);
    if (has_attr && (ti.mesh_flags & PT_MESH_HAS_UV) && sc.UV) {
        uv0 = mk2(sc.UV[2 * ti.v[0]], sc.UV[2 * ti.v[0] + 1]);
        uv1 = mk2(sc.UV[2 * ti.v[1]], sc.UV[2 * ti.v[1] + 1]);
        uv2 = mk2(sc.UV[2 * ti.v[2]], sc.UV[2 * ti.v[2] + 1]);
    }
    uvs[0] = uv0; uvs[1] = uv1; uvs[2] = uv2;
    float du02x = uv0.x - uv2.x, du02y = uv0.y - uv2.y, du12x = uv1.x - uv2.x, du12y = uv1.y - uv2.y;
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = du02x * du12y - du02y * du12x;
    if (!(fabsf(determinant) < 1e-8f)) {
        float invdet = 1.0f / determinant;
        V3 du = (du12y * dp02 - du02y * dp12) * invdet;
        V3 dv = (-du12x * dp02 + du02x * dp12) * invdet;
        if (!(length_squared(cross(du, dv)) <= 0.0f)) { *dpdu = du; *dpdv = dv; return; }
    }
    V3 ng = cross(p2 - p0, p1 - p0);
    coordinate_system(normalize(ng), dpdu, dpdv);
}

// Triangle::intersect's back half (triangle.rs:349-449): rebuild the interaction from the ray and
// the triangle's vertices.  Everything shading needs for an attribute-less mesh (material, light,
// orientation) rides in the 48-byte record, so the common case costs one dependent fetch.
PT_DEV bool make_surf_tv(const PtScene& sc, V3 ro, V3 rd, const TriVerts& tv, uint32_t light1, Surf& s, float* t_out, uint32_t rec = 0) {
    RayPre rp;
    ray_precompute(rp, ro, rd);
    TriHit h;
    if (!tri_test(rp, tv.p0, tv.p1, tv.p2, tv.flags, PT_INF, h)) return false;
    const bool has_attr = (tv.flags & PT_TRI_HAS_ATTR) != 0;
    PtTriInfo ti;
    ti.mesh_flags = 0; ti.v[0] = ti.v[1] = ti.v[2] = 0;
    if (has_attr) ti = sc.tri_info[rec];        // per leaf record
    V3 n = cross(tv.p0 - tv.p2, tv.p1 - tv.p2);
    if (tv.flags & PT_TRI_FLIP) n = n * -1.0f;
    n = normalize(n);
    V3 dpdu, dpdv;
    V2 uvs[3];
    tri_dpdu(sc, has_attr, ti, tv.p0, tv.p1, tv.p2, &dpdu, &dpdv, uvs);
    s.uv = mk2(h.b0 * uvs[0].x + h.b1 * uvs[1].x + h.b2 * uvs[2].x, h.b0 * uvs[0].y + h.b1 * uvs[1].y + h.b2 * uvs[2].y);   // triangle.rs:352
    s.dpdu = dpdu; s.dpdv = dpdv;
    s.sh_dpdv = dpdv; s.sh_dndu = mk3(0.0f, 0.0f, 0.0f); s.sh_dndv = s.sh_dndu;
    float xa = fabsf(h.b0 * tv.p0.x) + fabsf(h.b1 * tv.p1.x) + fabsf(h.b2 * tv.p2.x);
    float ya = fabsf(h.b0 * tv.p0.y) + fabsf(h.b1 * tv.p1.y) + fabsf(h.b2 * tv.p2.y);
    float za = fabsf(h.b0 * tv.p0.z) + fabsf(h.b1 * tv.p1.z) + fabsf(h.b2 * tv.p2.z);
    s.p_error = PT_GAMMA(7.0f) * mk3(xa, ya, za);
    s.p = h.b0 * tv.p0 + h.b1 * tv.p1 + h.b2 * tv.p2;
    s.wo = -rd;
    s.n = n;
    s.sh_n = n;
    s.sh_dpdu = dpdu;
    s.prim = tv.prim;
    s.material = (int32_t)(tv.flags >> PT_TRI_MATERIAL_SHIFT) - 1;
    s.light = (int32_t)light1 - 1;
    bool has_n = has_attr && (ti.mesh_flags & PT_MESH_HAS_N) && sc.N, has_s = has_attr && (ti.mesh_flags & PT_MESH_HAS_S) && sc.S;
    if (has_n || has_s) {
        V3 ns = s.n;
        if (has_n) {
            V3 nns = h.b0 * ld3(sc.N + 3 * (size_t)ti.v[0]) + h.b1 * ld3(sc.N + 3 * (size_t)ti.v[1]) + h.b2 * ld3(sc.N + 3 * (size_t)ti.v[2]);
            if (length_squared(nns) > 0.0f) ns = normalize(nns);
        }
        V3 ss = normalize(dpdu);
        if (has_s) {
            V3 nns = h.b0 * ld3(sc.S + 3 * (size_t)ti.v[0]) + h.b1 * ld3(sc.S + 3 * (size_t)ti.v[1]) + h.b2 * ld3(sc.S + 3 * (size_t)ti.v[2]);
            if (length_squared(nns) > 0.0f) ss = normalize(nns);
        }
        V3 ts = cross(ns, ss);
        if (length_squared(ts) > 0.0f) {
            ts = normalize(ts);
            ss = normalize(cross(ts, ns));
        } else {
            coordinate_system(ns, &ss, &ts);
        }
        if (has_n) {             // shading dndu / dndv from the vertex normals (triangle.rs:405-437); bump mapping reads them
            V3 n0 = ld3(sc.N + 3 * (size_t)ti.v[0]), n1 = ld3(sc.N + 3 * (size_t)ti.v[1]), n2 = ld3(sc.N + 3 * (size_t)ti.v[2]);
            float du02x = uvs[0].x - uvs[2].x, du02y = uvs[0].y - uvs[2].y, du12x = uvs[1].x - uvs[2].x, du12y = uvs[1].y - uvs[2].y;
            V3 dn1 = n0 - n2, dn2 = n1 - n2;
            float determinant = du02x * du12y - du02y * du12x;
            if (fabsf(determinant) < 1e-8f) {
                V3 dn = cross(n2 - n0, n1 - n0);
                if (length_squared(dn) != 0.0f) coordinate_system(dn, &s.sh_dndu, &s.sh_dndv);
            } else {
                float inv_det = 1.0f / determinant;
                s.sh_dndu = (du12y * dn1 - du02y * dn2) * inv_det;
                s.sh_dndv = (-du12x * dn1 + du02x * dn2) * inv_det;
            }
        }
        if (ti.mesh_flags & PT_MESH_REVERSE_ORIENTATION) ts = ts * -1.0f;
        s.sh_dpdv = ts;
        s.sh_n = normalize(cross(ss, ts));          // set_shading_geometry, orientation authoritative
        s.n = face_forward(s.n, s.sh_n);
        s.sh_dpdu = ss;
    }
    *t_out = h.t;
    return true;
}
// (a, b, c): the three 16-byte rows of leaf record `rec`, loaded by the caller (k_shade starts that load before anything else)
PT_DEV bool make_surf_rec(const PtScene& sc, V3 ro, V3 rd, uint32_t rec, float4 a, float4 b, float4 c, Surf& s, float* t_out) {
    TriVerts tv;
    tv.p0 = mk3(a.x, a.y, a.z); tv.prim = __float_as_uint(a.w);
    tv.p1 = mk3(b.x, b.y, b.z); tv.flags = __float_as_uint(b.w);
    tv.p2 = mk3(c.x, c.y, c.z);
    return make_surf_tv(sc, ro, rd, tv, __float_as_uint(c.w), s, t_out, rec);
}
PT_DEV bool make_surf(const PtScene& sc, V3 ro, V3 rd, uint32_t rec, Surf& s, float* t_out) {
    const float4* q = reinterpret_cast<const float4*>(sc.tris) + (size_t)rec * 3;
    return make_surf_rec(sc, ro, rd, rec, q[0], q[1], q[2], s, t_out);
}

// A record of a sphere-capable scene: Sphere::intersect again on the same ray (deterministic, and t_max only ever
// rejected candidates, so the unbounded re-test finds the hit the traversal found) and its world-space interaction.
template <bool SPH>
PT_DEV bool make_surf_any_rec(const PtScene& sc, V3 ro, V3 rd, uint32_t rec, float4 a, float4 b, float4 c, Surf& s, float* t_out) {
    if constexpr (SPH) {
        const uint32_t flags = __float_as_uint(b.w);
        if (flags & PT_TRI_SPHERE) {
            SphHit sh;
            const PtSphere& sp = sc.spheres[__float_as_uint(a.x)];
            if (!sph_hit_test(sp, ro, rd, PT_INF, PT_PI, &sh)) return false;
            sph_interaction(sp, sh, &s.p, &s.p_error, &s.n, &s.wo, &s.sh_n, &s.sh_dpdu, &s.dpdv, &s.uv, &s.sh_dndu, &s.sh_dndv);
            s.dpdu = s.sh_dpdu; s.sh_dpdv = s.dpdv;
            s.prim = __float_as_uint(a.w);
            s.material = (int32_t)(flags >> PT_TRI_MATERIAL_SHIFT) - 1;
            s.light = (int32_t)__float_as_uint(c.w) - 1;
            *t_out = sh.t;
            return true;
        }
    }
    return make_surf_rec(sc, ro, rd, rec, a, b, c, s, t_out);
}
template <bool SPH>
PT_DEV bool make_surf_any(const PtScene& sc, V3 ro, V3 rd, uint32_t rec, Surf& s, float* t_out) {
    const float4* q = reinterpret_cast<const float4*>(sc.tris) + (size_t)rec * 3;
    return make_surf_any_rec<SPH>(sc, ro, rd, rec, q[0], q[1], q[2], s, t_out);
}

// A hit inside an object instance: Shape::intersect again in instance space, then Transform::transform_surface_interaction
// back to the world (transformed_primitive.rs:30-38, transform.rs:299-323).
template <bool SPH>
PT_DEV bool make_surf_inst(const PtScene& sc, V3 ro, V3 rd, uint32_t rec, uint32_t inst1, Surf& s, float* t_out) {
    if (inst1 == 0) return make_surf_any<SPH>(sc, ro, rd, rec, s, t_out);
    const PtInstance& in = sc.instances[inst1 - 1u];
    V3 o, d;
    instance_ray(in, ro, rd, &o, &d);
    if (!make_surf_any<SPH>(sc, o, d, rec, s, t_out)) return false;
    V3 p = s.p, pe = s.p_error;
    s.p = sph_point(in.m, p);
    s.p_error = sph_point_abs_error(in.m, p, pe);
    s.n = normalize(sph_normal(in.minv, s.n));
    s.wo = normalize(sph_vector(in.m, s.wo));
    s.dpdu = sph_vector(in.m, s.dpdu); s.dpdv = sph_vector(in.m, s.dpdv);
    s.sh_n = normalize(sph_normal(in.minv, s.sh_n));
    s.sh_dpdu = sph_vector(in.m, s.sh_dpdu); s.sh_dpdv = sph_vector(in.m, s.sh_dpdv);
    s.sh_dndu = sph_normal(in.minv, s.sh_dndu); s.sh_dndv = sph_normal(in.minv, s.sh_dndv);
    s.sh_n = face_forward(s.sh_n, s.n);
    s.light = -1;                                 // TransformedPrimitive::get_area_light is None
    return true;
}

// ---- BSDF with at most one diffuse-reflection lobe (Matte: Lambertian or OrenNayar)
struct Bsdf {
    V3 ns, ng, ss, ts;
    V3 r;
    float oa, ob;
    int n_lobes;     // 0 when Kd is black (matte.rs:41)
    int oren;
};
PT_DEV float sin2_theta(V3 w) { return fmaxf(0.0f, 1.0f - w.z * w.z); }
PT_DEV float sin_theta(V3 w) { return sqrtf(sin2_theta(w)); }
PT_DEV float cos_phi(V3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
PT_DEV float sin_phi(V3 w) { float s = sin_theta(w); return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
PT_DEV V3 lobe_f(const Bsdf& b, V3 wo, V3 wi) {
    if (!b.oren) return b.r * PT_INV_PI;
    float sti = sin_theta(wi), sto = sin_theta(wo);
    float max_cos = 0.0f;
    if (sti > 1e-4f && sto > 1e-4f) {
        float d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
        max_cos = fmaxf(d_cos, 0.0f);
    }
    float sin_alpha, tan_beta;
    if (fabsf(wi.z) > fabsf(wo.z)) { sin_alpha = sto; tan_beta = sti / fabsf(wi.z); }
    else { sin_alpha = sti; tan_beta = sto / fabsf(wo.z); }
    return (b.r * PT_INV_PI) * (b.oa + b.ob * max_cos * sin_alpha * tan_beta);
}
PT_DEV float lobe_pdf(V3 wo, V3 wi) { return (wo.z * wi.z > 0.0f) ? fabsf(wi.z) * PT_INV_PI : 0.0f; }
PT_DEV V3 w2l(const Bsdf& b, V3 v) { return mk3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
PT_DEV V3 l2w(const Bsdf& b, V3 v) {
    return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
PT_DEV bool finite3(V3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
// BSDF::sample_f (bsdf.rs:92-206) for a single matching lobe (matching_comps == 1: comp = 0, u remap = min(u0, 1-eps))
PT_DEV bool bsdf_sample_f(const Bsdf& b, V3 wo_w, V2 u, V3* f, V3* wi_w, float* pdf) {
    if (b.n_lobes == 0) return false;
    V2 ur = mk2(fminf((u.x * 1.0f) - 0.0f, PT_ONE_MINUS_EPS), u.y);
    V3 wo = w2l(b, wo_w);
    if (wo.z == 0.0f || !finite3(wo)) return false;
    V3 wi = cosine_sample_hemisphere(ur);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    float p = lobe_pdf(wo, wi);
    if (p <= 0.0f) return false;
    V3 wiw = l2w(b, wi);
    bool reflect = (dot(wiw, b.ng) * dot(wo_w, b.ng)) > 0.0f;
    V3 ff = mk3(0.0f, 0.0f, 0.0f);
    if (reflect) ff = ff + lobe_f(b, wo, wi);
    *f = ff; *wi_w = wiw; *pdf = p;
    return true;
}
PT_DEV V3 bsdf_f(const Bsdf& b, V3 wo_w, V3 wi_w) {
    V3 wi = w2l(b, wi_w), wo = w2l(b, wo_w);
    V3 r = mk3(0.0f, 0.0f, 0.0f);
    if (wo.z == 0.0f || !finite3(wo)) return r;
    bool reflect = (dot(wi_w, b.ng) * dot(wo_w, b.ng)) > 0.0f;
    if (b.n_lobes && reflect) r = r + lobe_f(b, wo, wi);
    return r;
}
PT_DEV float bsdf_pdf(const Bsdf& b, V3 wo_w, V3 wi_w) {
    V3 wi = w2l(b, wi_w), wo = w2l(b, wo_w);
    if (wo.z == 0.0f || !finite3(wo)) return 0.0f;
    if (b.n_lobes == 0) return 0.0f;
    float p = 0.0f + lobe_pdf(wo, wi);
    return p / 1.0f;
}

// ---- lights
PT_DEV V3 light_L(const PtLight& l, V3 n, V3 w) {     // diffuse.rs:155-163
    if (l.two_sided || dot(n, w) > 0.0f) return ld3(l.L);
    return mk3(0.0f, 0.0f, 0.0f);
}
// Triangle::sample + sample_from (triangle.rs:590-651) + DiffuseAreaLight::sample_li (diffuse.rs:70-87)
PT_DEV bool light_sample_li(const PtLight& l, V3 ref_p, V2 u, V3* li, V3* wi, float* pdf, V3* lp, V3* lperr, V3* ln) {
    V3 p0 = ld3(l.p0), p1 = ld3(l.p1), p2 = ld3(l.p2);
    V2 b = uniform_sample_triangle(u);
    float b2 = 1.0f - b.x - b.y;
    V3 p = b.x * p0 + b.y * p1 + b2 * p2;
    V3 n = normalize(cross(p1 - p0, p2 - p0));
    if (l.mesh_flags & PT_MESH_HAS_N) {
        V3 ns = b.x * ld3(l.n0) + b.y * ld3(l.n1) + b2 * ld3(l.n2);
        n = face_forward(n, ns);
    } else if (((l.mesh_flags & PT_MESH_REVERSE_ORIENTATION) != 0) ^ ((l.mesh_flags & PT_MESH_SWAPS_HANDEDNESS) != 0)) {
        n = n * -1.0f;
    }
    V3 pas = vabs(b.x * p0) + vabs(b.y * p1) + vabs(b2 * p2);
    *lperr = PT_GAMMA6_REF * pas;
    float pd = 1.0f / l.area;
    V3 w = p - ref_p;
    if (length_squared(w) <= 0.0f) return false;
    w = normalize(w);
    if (!(l.mesh_flags & PT_MESH_TWO_SIDED)) {
        if (dot(n, -w) <= 0.0f) return false;
    }
    pd = pd * distance_squared(ref_p, p) / abs_dot(n, -w);
    if (pd <= 0.0f || isinf(pd)) return false;
    // sample_li
    if (length_squared(p - ref_p) <= 0.0f) return false;
    *wi = normalize(p - ref_p);
    *li = light_L(l, n, -*wi);
    *pdf = pd; *lp = p; *ln = n;
    return true;
}

// DiffuseAreaLight::sample_li over either shape.  The reference interaction's p_error and n only matter to
// Sphere::sample_from (its inside test starts from the offset origin, sphere.rs:309-315).
template <bool SPH>
PT_DEV bool light_sample_any(const PtScene& sc, const PtLight& l, V3 ref_p, V3 ref_pe, V3 ref_n, V2 u, V3* li, V3* wi, float* pdf, V3* lp, V3* lperr,
                             V3* ln) {
    if constexpr (SPH) {
        if (l.mesh_flags & PT_LIGHT_SPHERE) {
            V3 p, n, pe;
            float pd;
            if (!sph_sample_from(sc.spheres[__float_as_uint(l.p0[0])], ref_p, ref_pe, ref_n, u, &p, &n, &pe, &pd)) return false;
            if (pd <= 0.0f || length_squared(p - ref_p) <= 0.0f) return false;      // diffuse.rs:79-81
            *wi = normalize(p - ref_p);
            *li = light_L(l, n, -*wi);
            *pdf = pd; *lp = p; *lperr = pe; *ln = n;
            return true;
        }
    }
    return light_sample_li(l, ref_p, u, li, wi, pdf, lp, lperr, ln);
}

// ---- light distribution lookup (spatial.rs:84-111 + distribution.rs:12-31, :88-106)
PT_DEV const float* grid_lookup(const PtLightGrid& g, V3 p) {
    if (g.single) return g.data;   // uniform / power: one table for every point
    uint32_t pi[3];
    float pc[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; i++) {
        float o = pc[i] - g.wb_min[i];
        if (g.wb_max[i] > g.wb_min[i]) o = o / (g.wb_max[i] - g.wb_min[i]);
        o = clampf(o, 0.0f, 1.0f);
        float f = o * (float)g.voxels[i];
        uint32_t v = f > 0.0f ? (uint32_t)f : 0u;
        if (v > g.voxels[i] - 1) v = g.voxels[i] - 1;
        pi[i] = v;
    }
    size_t vox = ((size_t)pi[2] * g.voxels[1] + pi[1]) * g.voxels[0] + pi[0];
    if (g.row_of) vox = (size_t)(uint32_t)g.row_of[vox];          // lazily filled grid: k_grid_mark + k_light_grid_rows have made the row before this bounce is shaded
    return g.data + vox * g.stride;
}
PT_DEV uint32_t sample_discrete(const float* tab, uint32_t n, float u, float* pdf) {
    uint32_t idx;
    float f_idx, func_int;
    if (n <= 3) {
        // whole row (func[n], cdf[n+1], func_int <= 8 floats, 16-byte aligned) in one round trip;
        // find_interval_cdf over at most 4 cdf entries unrolls to compares
        const float4* q = reinterpret_cast<const float4*>(tab);
        float4 a = q[0], b = q[1];
        auto get = [&](uint32_t i) -> float {       // register selects, no private-memory array
            float lo = i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w));
            float hi = i == 4 ? b.x : (i == 5 ? b.y : (i == 6 ? b.z : b.w));
            return i < 4 ? lo : hi;
        };
        uint32_t first = 0, len = n + 1;
        while (len > 0) {
            uint32_t half = len >> 1, middle = first + half;
            if (get(n + middle) <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        idx = first == 0 ? 0 : first - 1;
        if (idx > n - 1) idx = n - 1;
        f_idx = get(idx);
        func_int = get(2 * n + 1);
    } else {
        const float* cdf = tab + n;
        uint32_t first = 0, len = n + 1;
        while (len > 0) {
            uint32_t half = len >> 1, middle = first + half;
            if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        idx = first == 0 ? 0 : first - 1;
        if (idx > n - 1) idx = n - 1;
        f_idx = tab[idx];
        func_int = tab[2 * n + 1];
    }
    *pdf = func_int > 0.0f ? f_idx * (1.0f / (float)n) / func_int : 0.0f;
    return idx;
}

// ============================================================ NEE resolve
// After TRACE: L += beta * ((A + B) / pdf_light) for every path with a pending next-event estimate, where the
// light-sample term A counts if its shadow ray was unoccluded and the BSDF-sample term B if its probe ray's
// closest hit is the sampled light's own triangle (sample_lights.rs:372-385, :419-447; path.rs:122-136).
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_nee_resolve(PtScene sc, PtPaths P, PtQueues Q) {
    const uint32_t n = Q.counts[PT_Q_NEE];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = Q.nee[i];
        const uint32_t nee = P.nee[p];
        const float4 A = P.pendA[p];
        V3 ld = mk3(0.0f, 0.0f, 0.0f);
        if ((nee & PT_NEE_SHADOW) && P.occluded[p] == 0) ld = ld + mk3(A.x, A.y, A.z);
        if (nee & PT_NEE_PROBE) {
            const int32_t best = P.probe_rec[p];
            if (best >= 0 && (uint32_t)best == sc.lights[nee >> 8].tri_rec) { float4 B = P.pendB[p]; ld = ld + mk3(B.x, B.y, B.z); }
        }
        V3 ldn = ld / A.w;
        float4 pb = P.pbeta[p];
        float4 L = P.L[p];
        V3 add = mk3(pb.x, pb.y, pb.z) * ldn;
        L.x += add.x; L.y += add.y; L.z += add.z;
        P.L[p] = L;
    }
}

// ============================================================ material-sorted shade queue
// Scenes with non-Matte materials: after TRACE the shade queue is counting-sorted by the material bin of
// each path's hit (Matte bins first), misses are dropped.  k_shade then runs over the Matte segment and
// k_shade_general over the rest with one material per run of lanes, so a wave executes one lobe list.
PT_DEV uint32_t path_sort_bin(const PtScene& sc, const PtPaths& P, uint32_t p) {
    int32_t rec = P.hit_rec[p];
    if (rec < 0) return 0xffffffffu;                                   // miss: nothing to shade
    uint32_t m1 = sc.tris[rec].flags >> PT_TRI_MATERIAL_SHIFT;         // material index + 1, 0 = none
    // material-less surfaces ride with bin 0 (with object instances: the first general bin -- one kernel shades everything there)
    return m1 == 0 ? (sc.n_instances ? PT_SORT_GENERAL0 : 0u) : sc.materials[m1 - 1].sort_bin;
}
// One atomic per (wave, distinct bin): returns this lane's slot among the lanes of its bin.
PT_DEV uint32_t wave_bin_reserve(uint32_t* counters, uint32_t bin, bool valid) {
    uint32_t slot = 0;
    unsigned long long todo = __ballot(valid);
    const uint32_t lane = threadIdx.x & 63;
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        uint32_t b0 = (uint32_t)__shfl((int)bin, leader, 64);
        unsigned long long same = __ballot(valid && bin == b0) & todo;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&counters[b0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (valid && bin == b0) slot = base + (uint32_t)__popcll(same & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
        todo &= ~same;
    }
    return slot;
}
// Block b owns the contiguous chunk [b*chunk, (b+1)*chunk) of the queue.  Bins are counted in LDS first
// (wave-aggregated), so the device-wide counters see one atomic per (block, bin), not one per item:
// with a handful of materials, per-item atomics on the same few addresses serialise in L2.
PT_DEV void sort_chunk(uint32_t n, uint32_t* lo, uint32_t* hi) {
    uint32_t chunk = (n + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 63u) & ~63u;
    *lo = min(n, blockIdx.x * chunk);
    *hi = min(n, *lo + chunk);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_sort_count(PtScene sc, PtPaths P, PtQueues Q) {
    __shared__ uint32_t s_cnt[PT_SORT_BINS];
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t lo, hi;
    sort_chunk(Q.counts[PT_Q_CUR], &lo, &hi);
    for (uint32_t i = lo + threadIdx.x; i < ((hi + 63u) & ~63u); i += blockDim.x) {
        bool valid = i < hi;
        uint32_t bin = valid ? path_sort_bin(sc, P, Q.cur[i]) : 0xffffffffu;
        if (valid && Q.bin) Q.bin[i] = (uint16_t)bin;          // the hit record -> triangle flags -> material chain is walked once per entry, here
        valid = valid && bin != 0xffffffffu;
        (void)wave_bin_reserve(s_cnt, valid ? bin : 0u, valid);
    }
    __syncthreads();
    uint32_t c = s_cnt[threadIdx.x];
    if (c) atomicAdd(&Q.counts[PT_SORT_COUNT0 + threadIdx.x], c);
}
extern "C" __global__ void k_sort_scan(PtQueues Q) {          // one block of PT_SORT_BINS threads
    __shared__ uint32_t s[PT_SORT_BINS];
    const uint32_t t = threadIdx.x;
    uint32_t c = Q.counts[PT_SORT_COUNT0 + t];
    s[t] = c;
    __syncthreads();
    for (uint32_t d = 1; d < PT_SORT_BINS; d <<= 1) {           // Hillis-Steele inclusive scan
        uint32_t v = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    uint32_t excl = s[t] - c;
    Q.counts[PT_SORT_CURSOR0 + t] = excl;
    Q.counts[PT_SORT_COUNT0 + t] = 0;                           // ready for the next bounce
    if (t == PT_SORT_GENERAL0) Q.counts[PT_Q_MATTE_END] = excl;
    if (t == PT_SORT_TEX0) Q.counts[PT_Q_TEX_BEGIN] = excl;
    if (t == PT_SORT_BINS - 1) Q.counts[PT_Q_GENERAL_END] = s[t];
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_sort_scatter(PtScene sc, PtPaths P, PtQueues Q) {
    __shared__ uint32_t s_cnt[PT_SORT_BINS];      // phase A: this block's count per bin; phase C: running cursor
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t lo, hi;
    sort_chunk(Q.counts[PT_Q_CUR], &lo, &hi);
    const uint32_t hi_round = (hi + 63u) & ~63u;
    auto bin_of = [&](uint32_t i, uint32_t p) -> uint32_t {
        if (Q.bin) { const uint32_t b = Q.bin[i]; return b == 0xffffu ? 0xffffffffu : b; }
        return path_sort_bin(sc, P, p);
    };
    for (uint32_t i = lo + threadIdx.x; i < hi_round; i += blockDim.x) {
        bool valid = i < hi;
        uint32_t bin = valid ? bin_of(i, Q.bin ? 0u : Q.cur[i]) : 0xffffffffu;
        valid = valid && bin != 0xffffffffu;
        (void)wave_bin_reserve(s_cnt, valid ? bin : 0u, valid);
    }
    __syncthreads();
    {   // phase B: reserve this block's range in every bin it uses
        uint32_t c = s_cnt[threadIdx.x];
        s_cnt[threadIdx.x] = c ? atomicAdd(&Q.counts[PT_SORT_CURSOR0 + threadIdx.x], c) : 0u;
    }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi_round; i += blockDim.x) {
        bool valid = i < hi;
        uint32_t p = valid ? Q.cur[i] : 0u;
        uint32_t bin = valid ? bin_of(i, p) : 0xffffffffu;
        valid = valid && bin != 0xffffffffu;
        uint32_t pos = wave_bin_reserve(s_cnt, valid ? bin : 0u, valid);
        if (valid) Q.sorted[pos] = p;
    }
}

// The material sort kept LOCAL (round 4; scenes with constant non-Matte materials): every block orders one run of PT_LOCAL_CHUNK consecutive
// entries of the shade queue by material bin, in place -- bins ascending, misses last -- and ONE lobe-list kernel (k_shade_all) walks the
// result.  A wave still sees one material (a run holds 4 096 paths: hundreds per bin), but consecutive waves now work on neighbouring path
// slots: the global sort hands each kernel a list that skips the other kernels' paths, so every 128-byte line of the eleven path-state arrays
// is fetched for half its paths, once by each kernel (k_shade_matte_sorted took 33 % longer over 55 % of the vertices than k_shade takes over
// all of them).
#ifndef PT_LOCAL_CHUNK
#define PT_LOCAL_CHUNK 4096u
#endif
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_sort_local(PtScene sc, PtPaths P, PtQueues Q) {
    __shared__ uint32_t s_cnt[PT_SORT_BINS + 1];          // per bin (the last one: misses): count, then the bin's cursor inside the run
    __shared__ uint32_t s_p[PT_LOCAL_CHUNK];
    __shared__ uint16_t s_b[PT_LOCAL_CHUNK];
    const uint32_t n = Q.counts[PT_Q_CUR];
    const uint32_t n_runs = (n + PT_LOCAL_CHUNK - 1u) / PT_LOCAL_CHUNK;
    for (uint32_t run = blockIdx.x; run < n_runs; run += gridDim.x) {
        const uint32_t lo = run * PT_LOCAL_CHUNK, len = min(PT_LOCAL_CHUNK, n - lo);
        for (uint32_t k = threadIdx.x; k <= PT_SORT_BINS; k += blockDim.x) s_cnt[k] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
            const uint32_t p = Q.cur[lo + i];
            const uint32_t bin = path_sort_bin(sc, P, p);
            const uint32_t b = bin == 0xffffffffu ? PT_SORT_BINS : bin;
            s_p[i] = p; s_b[i] = (uint16_t)b;
            atomicAdd(&s_cnt[b], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {                              // exclusive scan over 257 counters (a handful of them non-zero)
            uint32_t acc = 0;
            for (uint32_t k = 0; k <= PT_SORT_BINS; k++) { const uint32_t c = s_cnt[k]; s_cnt[k] = acc; acc += c; }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
            const uint32_t pos = atomicAdd(&s_cnt[s_b[i]], 1u);
            Q.sorted[lo + pos] = s_p[i];
        }
        __syncthreads();
    }
}

// ============================================================ K_SHADE
// One bounce of PathIntegrator::li for every path in Q.cur (path.rs:85-234).
#ifndef PT_SHADE_TICKET
#define PT_SHADE_TICKET 4        // k_shade: iterations (of 64 paths) per work ticket
#endif
#ifndef PT_SHADE_FLUSH
#define PT_SHADE_FLUSH 4         // k_shade: iterations (of 64 paths) per queue reservation
#endif
#ifndef PT_WIDE_KERNEL_WAVES
#define PT_WIDE_KERNEL_WAVES 2   // the sphere / instance shading kernels and the recursive integrators' two kernels: compiled for this many waves per SIMD
#endif
#ifndef PT_SHADE_GEN_WAVES
#define PT_SHADE_GEN_WAVES 2     // k_shade_general: 256 registers = two waves per SIMD (with the LDS lobe copy it wants 260 when left alone: bound to two waves it
                                 // fits without spilling); compiled for 3 (168 registers, 149 spilled): mixed bench 856 -> 835
#endif
#ifndef PT_SHADE_TEX_WAVES
#define PT_SHADE_TEX_WAVES 2     // waves per SIMD the textured shading kernel is compiled for: 256 registers and 112 spilled, against 300 and none
                                 // at one wave per SIMD -- 700 -> 729 Mrays/s on the textured bench (needs the texture code out of line, pt_texture.h)
#endif
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 2
#endif
// GENERAL = false: every material is Matte (at most one diffuse lobe; the RT1M / Cornell fast path).
// GENERAL = true: BSDFs are the per-material lobe lists of pt_bxdf.h (specular bounces, eta_scale, glass without a BSDF).
// Textured material at a hit: evaluate the parameter textures (Texture::evaluate(si) inside each
// compute_scattering_functions) into a copy of the parameter block, then build the BxDF list the host builds for constant
// materials (pt_lobes.h).
// bump: material_bump (core/material.rs:31-72) first -- it bends the shading frame (sh_n, sh_dpdu) in place.
// build_lobes (pt_lobes.h, shared with the host) as a call: inlined into textured_lobes it made that function the register peak of
// the textured shading kernel
__device__ __noinline__ void build_lobes_call(const pt_material& in, float a_r, float a_u, float a_v, PtMaterial& m) { build_lobes(in, a_r, a_u, a_v, m); }
// textured_params: the per-hit half -- the bump map bends the frame, every programmed parameter is evaluated into `mp`, the roughness
// values are remapped into a3 = {a_r, a_u, a_v}.  k_tex_resolve stops here and hands the values to k_shade_general_res (PtTexRes).
template <bool INL>
PT_DEV void textured_params_t(const PtScene& sc, int32_t material, const TexHit& th, PtMatParams& mp, float* a3, V3 n, V2 uv, V3* sh_n, V3* sh_dpdu,
                              V3 sh_dpdv, V3 sh_dndu, V3 sh_dndv, float* vbuf = nullptr) {
    mp = sc.mat_params[material];
    // k_tex_resolve reads the program offsets where they lie: indexed by the job loop's variable, the copy in `mp` lives in scratch (the out-of-line
    // form, inside the one-kernel textured shading, spills more with it and keeps the copy)
    const uint32_t* progs = (PT_TEX_PROGS_GLOBAL && INL) ? sc.mat_params[material].prog : mp.prog;
    // One call site for every texture program of the hit (the evaluations are independent of one another, so their order is free): jobs 0-2
    // are the bump map's displacement at p + du * dpdu, p + dv * dpdv and p (core/material.rs:31-72), jobs 3-10 the parameters Kd Ks Kr Kt
    // opacity sigma metal-eta metal-k, jobs 11-14 roughness uroughness vroughness eta.
    float du = 0.0f, dv = 0.0f, disp0 = 0.0f, disp1 = 0.0f;          // (scalars, not an array indexed by the loop variable)
    if (progs[8]) {
        du = 0.5f * (fabsf(th.dudx) + fabsf(th.dudy));
        if (du == 0.0f) du = 0.0005f;
        dv = 0.5f * (fabsf(th.dvdx) + fabsf(th.dvdy));
        if (dv == 0.0f) dv = 0.0005f;
    }
    float a_r = mp.a_r, a_u = mp.a_u, a_v = mp.a_v;
#if PT_TEX_NOUNROLL
#pragma nounroll
#endif
    for (int j = 0; j < 15; j++) {
        const int k = j < 3 ? 8 : (j < 11 ? j - 3 : j - 2);          // index into PtMatParams::prog
        const uint32_t pr = progs[k];
        if (!pr) continue;
        TexHit ev = th;
        if (j == 0) { ev.p = th.p + du * *sh_dpdu; ev.uv = mk2(uv.x + du, uv.y + 0.0f); }
        else if (j == 1) { ev.p = th.p + dv * sh_dpdv; ev.uv = mk2(uv.x + 0.0f, uv.y + dv); }
        V3 v;
        if constexpr (INL) v = tex_eval_inl(sc.textures, sc.tex_prog + pr, ev, sc.images, vbuf);       // k_tex_resolve: the whole texture code inline (pt_texture_calls.inc)
        else v = tex_eval(sc.textures, sc.tex_prog + pr, ev, sc.images);
        switch (j) {
            case 0: disp0 = v.x; break;
            case 1: disp1 = v.x; break;
            case 2: {
                const float displace = v.x;
                V3 dpdu = *sh_dpdu + (disp0 - displace) / du * *sh_n + displace * sh_dndu;
                V3 dpdv = sh_dpdv + (disp1 - displace) / dv * *sh_n + displace * sh_dndv;
                *sh_n = face_forward(normalize(cross(dpdu, dpdv)), n);       // set_shading_geometry(.., false) (surface_interaction.rs:140-161)
                *sh_dpdu = dpdu;
                break;
            }
            case 3: mp.m.kd[0] = v.x; mp.m.kd[1] = v.y; mp.m.kd[2] = v.z; break;
            case 4: mp.m.ks[0] = v.x; mp.m.ks[1] = v.y; mp.m.ks[2] = v.z; break;
            case 5: mp.m.kr[0] = v.x; mp.m.kr[1] = v.y; mp.m.kr[2] = v.z; break;
            case 6: mp.m.kt[0] = v.x; mp.m.kt[1] = v.y; mp.m.kt[2] = v.z; break;
            case 7: mp.m.opacity[0] = v.x; mp.m.opacity[1] = v.y; mp.m.opacity[2] = v.z; break;
            case 8: mp.m.sigma = v.x; break;
            case 9: mp.m.metal_eta[0] = v.x; mp.m.metal_eta[1] = v.y; mp.m.metal_eta[2] = v.z; break;
            case 10: mp.m.metal_k[0] = v.x; mp.m.metal_k[1] = v.y; mp.m.metal_k[2] = v.z; break;
            // float parameters behind textures: "roughness" / "uroughness" / "vroughness" (then roughness_to_alpha, unless "remaproughness"
            // is off) and "eta"; the constant ones keep the values the host remapped at upload
            case 11: mp.m.roughness = v.x; a_r = mp.m.remap_roughness ? pt_roughness_to_alpha(v.x) : v.x; break;
            case 12: mp.m.uroughness = v.x; a_u = mp.m.remap_roughness ? pt_roughness_to_alpha(v.x) : v.x; break;
            case 13: mp.m.vroughness = v.x; a_v = mp.m.remap_roughness ? pt_roughness_to_alpha(v.x) : v.x; break;
            default: mp.m.eta = v.x; break;
        }
    }
    a3[0] = a_r; a3[1] = a_u; a3[2] = a_v;
}
__device__ __noinline__ void textured_params(const PtScene& sc, int32_t material, const TexHit& th, PtMatParams& mp, float* a3, V3 n, V2 uv, V3* sh_n, V3* sh_dpdu,
                                             V3 sh_dpdv, V3 sh_dndu, V3 sh_dndv) {
    textured_params_t<false>(sc, material, th, mp, a3, n, uv, sh_n, sh_dpdu, sh_dpdv, sh_dndu, sh_dndv);
}
__device__ __noinline__ void textured_lobes(const PtScene& sc, int32_t material, const TexHit& th, PtMaterial* out, V3 n, V2 uv, V3* sh_n, V3* sh_dpdu,
                                            V3 sh_dpdv, V3 sh_dndu, V3 sh_dndv) {
    PtMatParams mp;
    float a3[3];
    textured_params(sc, material, th, mp, a3, n, uv, sh_n, sh_dpdu, sh_dpdv, sh_dndu, sh_dndv);
    build_lobes_call(mp.m, a3[0], a3[1], a3[2], *out);
}
// Sort key of a ray for pt_raysort.hip: Morton code of the cell of its origin inside the world bound (2^PT_SORT_CELL_BITS cells per axis), direction octant on top
PT_DEV uint32_t ray_sort_key(const PtScene& sc, V3 o, V3 d) {
    const float fx = (o.x - sc.wb_min[0]) * sc.cell_scale[0], fy = (o.y - sc.wb_min[1]) * sc.cell_scale[1], fz = (o.z - sc.wb_min[2]) * sc.cell_scale[2];
    const float top = (float)((1u << PT_SORT_CELL_BITS) - 1u);
    uint32_t cx = (uint32_t)fminf(fmaxf(fx, 0.0f), top), cy = (uint32_t)fminf(fmaxf(fy, 0.0f), top), cz = (uint32_t)fminf(fmaxf(fz, 0.0f), top);
    uint32_t m = 0;
#pragma unroll
    for (uint32_t b = 0; b < (uint32_t)PT_SORT_CELL_BITS; b++) m |= (((cx >> b) & 1u) | (((cy >> b) & 1u) << 1) | (((cz >> b) & 1u) << 2)) << (3u * b);
    return m | ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)) << (3u * PT_SORT_CELL_BITS);
}
// Diagnostic build -DPT_PROFILE_SHADE: wave clocks of shade_body by section (tools/tune_shade.sh "prof:-DPT_PROFILE_SHADE")
#ifdef PT_PROFILE_SHADE
__device__ unsigned long long g_shade_prof[16];
#define PT_SHP(k) do { const unsigned long long t_now_ = (unsigned long long)__builtin_readcyclecounter(); shp_acc[k] += t_now_ - shp_last; shp_last = t_now_; } while (0)
#define PT_SHP_SYNC(k) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PT_SHP(k); } while (0)      // section ends once every load issued so far is back
#else
#define PT_SHP(k) do { } while (0)
#define PT_SHP_SYNC(k) do { } while (0)
#endif
// 0.0f the optimiser cannot see through (one v_mov): see the continuation store in shade_body
PT_DEV float opaque_zero() { float z = 0.0f; asm volatile("" : "+v"(z)); return z; }
// PART: 0 = the whole vertex in one kernel.  1 = next-event estimation only (light pick, light sample, BSDF sample for MIS; writes the
// shadow / probe rays, the pending terms and -- for every vertex with a BSDF -- the nee word, whose PT_NEE_DIMS bits tell part 2 how many
// sample dimensions were drawn); 2 = everything else (emission, pass-through, continuation, Russian roulette), run AFTER part 1 on the same
// list because it overwrites the ray.  Both halves rebuild the interaction from (ray, record): the split trades that for register room.
template <bool GENERAL, bool SPH, bool TEX = false, bool INST = false, bool RES = false, int PART = 0>
PT_DEV void shade_body(const PtScene& sc, const PtPaths& P, const PtQueues& Q, PtCounters* cnt, const uint32_t* list, uint32_t begin, uint32_t end,
                       uint32_t* ticket) {
    __shared__ unsigned long long s_vert;
    // A vertex's next-event outputs wait here (28 KB per block) until every load of the iteration has come back: on this ISA loads and
    // stores share one in-order counter (vmcnt), so a store issued in the middle of the dependent chain grid -> light table -> light
    // makes each later wait sit out the store's round trip as well.
    __shared__ float4 s_stage[PART == 2 ? 1 : 7][PT_BLOCK];
    __shared__ uint32_t s_pend[PT_SHADE_FLUSH][PT_BLOCK];      // per lane: path ids of the iterations not yet queued (their queue bits: `pend_want`)
#if PT_LOBES_IN_LDS
    __shared__ uint32_t s_lobes[GENERAL ? PT_BLOCK / 64 : 1][PT_MAX_LOBES * (sizeof(PtLobe) / 4)];      // per wave: the lobe list of the material its lanes share
#endif
    __shared__ uint32_t s_pkey[PT_SHADE_FLUSH][PT_BLOCK];      // per lane: the sort key of those iterations' shadow rays
    if (threadIdx.x == 0) s_vert = 0;
#if PT_LIGHTS_IN_LDS
    // Scenes with a handful of lights (BASELINE's have two): the light records sit in LDS for the kernel's life, which takes one round trip out of
    // every vertex's dependent chain grid row -> light number -> light record -> sample.
    __shared__ PtLight s_lights[PT_LIGHTS_IN_LDS];
    const bool lights_lds = sc.n_lights <= (uint32_t)PT_LIGHTS_IN_LDS && PART != 2;
    if (lights_lds) {
        const uint32_t nd = sc.n_lights * (uint32_t)(sizeof(PtLight) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.lights);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_lights);
        for (uint32_t k = threadIdx.x; k < nd; k += PT_BLOCK) dst[k] = src[k];
    }
#endif
    __syncthreads();
    uint32_t n_vert = 0;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t chunk_left = 0, next_base = 0;
#ifdef PT_PROFILE_SHADE
    unsigned long long shp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long shp_last = (unsigned long long)__builtin_readcyclecounter();
#endif
    // Queue entries are handed out in batches of PT_SHADE_FLUSH iterations.  A returning atomic on one address is served every 11.3 ns
    // (tools/ubench/atomic_rate.hip: 88 M/s however many waves ask), so one reservation per queue and 64 paths -- 540 k of them
    // for a 34 M-path bounce -- made the queue counter, not the shading, set this kernel's duration.
    uint32_t n_batch = 0, pend_want = 0;      // pend_want: four queue bits (next, nee, shadow, probe) per pending iteration
    uint32_t pf_p = 0;                  // prefetched for the next iteration: path id, its hit record, its state word, sampler index and pixel
    int32_t pf_rec = -1;
    uint32_t pf_st = 0, pf_pk = 0;
    uint64_t pf_idx = 0;
    bool pf_valid = false;
    // PT_SHADE_PF_DEEP: the next iteration's ray, throughput and leaf record come one iteration ahead as well, so that an iteration starts with
    // everything its surface needs and its dependent chain is grid -> light instead of path state -> record -> grid -> light
    constexpr bool DEEP = PT_SHADE_PF_DEEP && !GENERAL && !SPH && !INST && PART == 0;      // the Matte kernels have the 21 registers it takes (232 -> 253); the lobe-list kernels sit at 256
    float4 pf_ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pf_rd = pf_ro, pf_beta = pf_ro, pf_ra = pf_ro, pf_rb = pf_ro, pf_rc = pf_ro;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    auto flush_batch = [&]() {
        if (n_batch == 0) return;
        uint32_t tc = 0, tn = 0, ts = 0, tp = 0;
#pragma unroll
        for (uint32_t j = 0; j < PT_SHADE_FLUSH; j++) {
            if (j < n_batch) {
                const uint32_t e = pend_want >> (4u * j);
                tc += (uint32_t)__popcll(__ballot((e & 1u) != 0)); tn += (uint32_t)__popcll(__ballot((e & 2u) != 0));
                ts += (uint32_t)__popcll(__ballot((e & 4u) != 0)); tp += (uint32_t)__popcll(__ballot((e & 8u) != 0));
            }
        }
        uint32_t bc = 0, bn = 0, bs = 0, bp = 0;
        if (lane == 0) {
            if (tc) bc = atomicAdd(&Q.counts[PT_Q_NEXT], tc);
            if (tn) bn = atomicAdd(&Q.counts[PT_Q_NEE], tn);
            if (ts) bs = atomicAdd(&Q.counts[PT_Q_SHADOW], ts);
            if (tp) bp = atomicAdd(&Q.counts[PT_Q_PROBE], tp);
        }
        bc = (uint32_t)__builtin_amdgcn_readfirstlane((int)bc); bn = (uint32_t)__builtin_amdgcn_readfirstlane((int)bn);       // lane 0 made the reservations; uniform for the compiler too
        bs = (uint32_t)__builtin_amdgcn_readfirstlane((int)bs); bp = (uint32_t)__builtin_amdgcn_readfirstlane((int)bp);
#pragma unroll
        for (uint32_t j = 0; j < PT_SHADE_FLUSH; j++) {
            if (j < n_batch) {
                const uint32_t e = pend_want >> (4u * j), ep = s_pend[j][threadIdx.x];
                const unsigned long long mc = __ballot((e & 1u) != 0), mn = __ballot((e & 2u) != 0), ms = __ballot((e & 4u) != 0), mp = __ballot((e & 8u) != 0);
                if (e & 1u) Q.next[bc + (uint32_t)__popcll(mc & below)] = ep;
                if (e & 2u) Q.nee[bn + (uint32_t)__popcll(mn & below)] = ep;
                if (e & 4u) {
                    Q.shadow[bs + (uint32_t)__popcll(ms & below)] = ep;
                    if (Q.shadow_key) Q.shadow_key[bs + (uint32_t)__popcll(ms & below)] = s_pkey[j][threadIdx.x];
                }
                if (e & 8u) Q.probe[bp + (uint32_t)__popcll(mp & below)] = ep;
                bc += (uint32_t)__popcll(mc); bn += (uint32_t)__popcll(mn); bs += (uint32_t)__popcll(ms); bp += (uint32_t)__popcll(mp);
            }
        }
        n_batch = 0; pend_want = 0;
    };
    for (;;) {
        PT_SHP(11);
        if (chunk_left == 0) {                 // one ticket atomic per PT_SHADE_TICKET x 64 items
            uint32_t t0 = 0;
            if (lane == 0) t0 = atomicAdd(ticket, 64u * PT_SHADE_TICKET);
            next_base = begin + (uint32_t)__builtin_amdgcn_readfirstlane((int)t0);
            chunk_left = PT_SHADE_TICKET;
        }
        uint32_t base = next_base;
        next_base += 64u; chunk_left--;
        if (base >= end) break;
#ifdef PT_PROFILE_SHADE
        shp_acc[12]++;           // iterations
#endif
        uint32_t item = base + lane;
        bool active = item < end;
        bool cont = false, want_nee = false, want_sh = false, want_pr = false;
        uint32_t p = 0;
        uint32_t wr = 0, o_nee = 0, o_state = 0;     // outputs held back to the end of the iteration: 1 ray_o, 2 ray_d + beta, 4 shadow ray, 8 probe ray, 16 pending NEE, 32 state
        float4 o_ray_o = make_float4(0.0f, 0.0f, 0.0f, 0.0f), o_ray_d = o_ray_o, o_beta = o_ray_o;
        auto commit = [&]() {
            if constexpr (PART != 2) {
                if (wr & 4u) { P.sh_o[p] = s_stage[0][threadIdx.x]; P.sh_d[p] = s_stage[1][threadIdx.x]; }
                if (wr & 8u) { P.pr_o[p] = s_stage[2][threadIdx.x]; P.pr_d[p] = s_stage[3][threadIdx.x]; }
                if (wr & 16u) { P.pendA[p] = s_stage[4][threadIdx.x]; P.pendB[p] = s_stage[5][threadIdx.x]; P.pbeta[p] = s_stage[6][threadIdx.x]; P.nee[p] = o_nee; }
            }
            if (wr & 1u) P.ray_o[p] = o_ray_o;
            if (wr & 2u) { P.ray_d[p] = o_ray_d; P.beta[p] = o_beta; }
            if (wr & 32u) P.state[p] = o_state;
            wr = 0;
        };
        // The textured / instanced kernels (425 registers and still spilling) store where the value is made, as they always did: held back
        // to the end of the iteration, the continuation ray came out wrong for about 1 % of the samples of seven feature scenes (bisected
        // to exactly that deferral; the kernels that do not spill are bit-exact either way).
#ifndef PT_DEFER_WIDE
#define PT_DEFER_WIDE 0          // 1: the textured / instanced kernels hold their stores back as well (the build tools/r03_defer_probe.sh examines)
#endif
#ifndef PT_DEFER_SITES
#define PT_DEFER_SITES (PT_DEFER_WIDE ? 63 : 0)      // which store sites of those kernels are held back: 1 pass-through ray, 2 shadow ray, 4 probe ray, 8 pending terms, 16 continuation, 32 state
#endif
#define PT_COMMIT_NOW(site) do { if constexpr ((TEX || INST) && !((PT_DEFER_SITES >> (site)) & 1)) commit(); } while (0)
        // the next iteration's path id and hit record are asked for one iteration ahead (within a work ticket): two of the three
        // dependent round trips at the head of an iteration -- list -> path state -> leaf record -- then overlap this iteration's work
        const bool had_pf = pf_valid;
        const uint32_t pf_p_now = pf_p;
        const int32_t pf_rec_now = pf_rec;
        const uint32_t pf_st_now = pf_st, pf_pk_now = pf_pk;
        const uint64_t pf_idx_now = pf_idx;
        const float4 pf_ro_now = pf_ro, pf_rd_now = pf_rd, pf_beta_now = pf_beta, pf_ra_now = pf_ra, pf_rb_now = pf_rb, pf_rc_now = pf_rc;
        pf_valid = chunk_left > 0;
        const bool pf_lane = pf_valid && next_base + lane < end;
        if (pf_lane) pf_p = list[next_base + lane];
        if (active) {
            p = had_pf ? pf_p_now : list[item];
            PT_SHP_SYNC(13);
            float4 ro4, rd4, beta4;
            if (DEEP && had_pf) { ro4 = pf_ro_now; rd4 = pf_rd_now; beta4 = pf_beta_now; }
            else { ro4 = P.ray_o[p]; rd4 = P.ray_d[p]; beta4 = P.beta[p]; }
            V3 ro = f4_3(ro4), rd = f4_3(rd4);
            int32_t rec = had_pf ? pf_rec_now : P.hit_rec[p];
            uint32_t st = had_pf ? pf_st_now : P.state[p];
            uint32_t dim = st & 0xffffu, bounces = (st >> 16) & 0xffu, flags = st >> 24;
            V3 beta = f4_3(beta4);
            float eta_scale = beta4.w;
            PT_SHP_SYNC(14);
            // the hit's leaf record: asked for first, so that it travels while the sample tables are read
            float4 rec_a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rec_b = rec_a, rec_c = rec_a;
            if (DEEP && had_pf) { rec_a = pf_ra_now; rec_b = pf_rb_now; rec_c = pf_rc_now; }
            else if (rec >= 0) {
                const float4* q = reinterpret_cast<const float4*>(sc.tris) + (size_t)(uint32_t)rec * 3;
                rec_a = q[0]; rec_b = q[1]; rec_c = q[2];
            }
            // this vertex's sample dimensions, started before the hit is reconstructed (see SamplerPre)
            SamplerPre sm;
            sm.s.index = 0; sm.s.dim = dim; sm.s.px = 0; sm.s.py = 0; sm.dim0 = dim; sm.have = false;
            if (rec >= 0 && (int32_t)bounces < sc.max_depth) {
                sm.s.index = had_pf ? pf_idx_now : P.sobol_index[p];
                const uint32_t pk = had_pf ? pf_pk_now : P.pixel[p];
                sm.s.px = (int32_t)(pk & 0xffffu) + sc.film.sample_bounds[0];
                sm.s.py = (int32_t)(pk >> 16) + sc.film.sample_bounds[1];
                PT_SHP_SYNC(15);
                sm.begin(sc);
                PT_SHP_SYNC(10);
            }
            Surf s;
            float thit;
            bool found;
            if constexpr (INST) found = rec >= 0 && make_surf_inst<SPH>(sc, ro, rd, (uint32_t)rec, P.hit_inst[p], s, &thit);
            else found = rec >= 0 && make_surf_any_rec<SPH>(sc, ro, rd, (uint32_t)rec, rec_a, rec_b, rec_c, s, &thit);
            PT_SHP(0);
            if (pf_lane) {                                   // pf_p has long arrived: the head of the next iteration's dependent chain, one iteration early
                pf_rec = P.hit_rec[pf_p];
                pf_st = P.state[pf_p]; pf_idx = P.sobol_index[pf_p]; pf_pk = P.pixel[pf_p];
                if constexpr (DEEP) { pf_ro = P.ray_o[pf_p]; pf_rd = P.ray_d[pf_p]; pf_beta = P.beta[pf_p]; }
            }
            // emitted radiance at the first vertex / after a specular bounce (path.rs:87-98)
            if (PART != 1 && found && (bounces == 0 || (flags & PT_ST_SPECULAR))) {
                int32_t li = s.light;
                if (li >= 0) {
                    V3 le = light_L(sc.lights[li], s.n, -rd);
                    float4 L = P.L[p];
                    V3 add = beta * le;
                    L.x += add.x; L.y += add.y; L.z += add.z;
                    P.L[p] = L;
                }
            }
            if (found && (int32_t)bounces < sc.max_depth) {
                bool no_bsdf = s.material < 0;
                PtMaterial tm;                     // TEX / RES: this hit's lobes
                bool use_tm = false;
                if constexpr (TEX) {
#if PT_TEX_EXP == 4          // timing experiment: what the second half of a split kernel would cost -- lobes built from the unevaluated parameter block
                    if (!no_bsdf && sc.materials[s.material].textured) {
                        PtMatParams mp = sc.mat_params[s.material];
                        build_lobes_call(mp.m, mp.a_r, mp.a_u, mp.a_v, tm);
                        use_tm = true;
                        no_bsdf = tm.has_bsdf == 0;
                    }
#else
                    if (!no_bsdf && sc.materials[s.material].textured) {
                        TexHit th;
                        th.p = s.p; th.uv = s.uv;
                        RayDiffs rdf;
                        const bool has_diff = (flags & PT_ST_CAMERA) != 0;
                        if (has_diff) {             // rebuild the camera ray's offset rays from its camera sample
                            Sampler sl;
                            sl.index = P.sobol_index[p];
                            sl.dim = 2;             // get_camera_sample: film (0,1), lens (2,3), time (4)
                            uint32_t pk2 = P.pixel[p];
                            sl.px = (int32_t)(pk2 & 0xffffu) + sc.film.sample_bounds[0];
                            sl.py = (int32_t)(pk2 >> 16) + sc.film.sample_bounds[1];
                            V2 u_lens = mk2(0.0f, 0.0f);
                            if (sc.cam.lens_radius > 0.0f) u_lens = sl.get_2d(sc);
                            float2 pf = P.p_film[p];
                            camera_differentials(sc, mk2(pf.x, pf.y), u_lens, ro, rd, rdf);
                        }
                        compute_differentials(th, s.p, s.n, s.dpdu, s.dpdv, has_diff, rdf);
                        textured_lobes(sc, s.material, th, &tm, s.n, s.uv, &s.sh_n, &s.sh_dpdu, s.sh_dpdv, s.sh_dndu, s.sh_dndv);
                        use_tm = true;
                        no_bsdf = tm.has_bsdf == 0;
                    }
#endif
                }
                if constexpr (RES) {          // k_tex_resolve has evaluated this hit's textures: the values, the alphas and the bump-mapped frame come from P.tex_res
                    if (!no_bsdf && sc.materials[s.material].textured) {
                        const float4* r = P.tex_res + (size_t)p * PT_TEX_RES_F4;
                        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
                        PtMatParams mp = sc.mat_params[s.material];
                        mp.m.kd[0] = r0.x; mp.m.kd[1] = r0.y; mp.m.kd[2] = r0.z; mp.m.ks[0] = r0.w; mp.m.ks[1] = r1.x; mp.m.ks[2] = r1.y;
                        mp.m.kr[0] = r1.z; mp.m.kr[1] = r1.w; mp.m.kr[2] = r2.x; mp.m.kt[0] = r2.y; mp.m.kt[1] = r2.z; mp.m.kt[2] = r2.w;
                        mp.m.opacity[0] = r3.x; mp.m.opacity[1] = r3.y; mp.m.opacity[2] = r3.z; mp.m.sigma = r3.w;
                        mp.m.metal_eta[0] = r4.x; mp.m.metal_eta[1] = r4.y; mp.m.metal_eta[2] = r4.z; mp.m.metal_k[0] = r4.w; mp.m.metal_k[1] = r5.x; mp.m.metal_k[2] = r5.y;
                        mp.m.uroughness = r5.z; mp.m.vroughness = r5.w; mp.m.eta = r6.x;
                        build_lobes_call(mp.m, r6.y, r6.z, r6.w, tm);
                        s.sh_n = mk3(r7.x, r7.y, r7.z); s.sh_dpdu = mk3(r7.w, r8.x, r8.y);
                        use_tm = true;
                        no_bsdf = tm.has_bsdf == 0;
                    }
                }
                if constexpr (GENERAL) { if (!no_bsdf && !use_tm) no_bsdf = sc.materials[s.material].has_bsdf == 0; }
                if (no_bsdf) {
                    // no BSDF: continue through the surface, same bounce count (path.rs:108-111)
                    if constexpr (PART != 1) {
                        V3 no = offset_ray_origin(s.p, s.p_error, s.n, rd);
                        o_ray_o = make_float4(no.x, no.y, no.z, PT_INF); wr |= 1u;
                        cont = true; PT_COMMIT_NOW(0);
                    }
                } else {
                    if constexpr (PART != 1) n_vert++;
                    Bsdf b;
                    GBsdf gb;
                    bool nonspecular;
                    float bsdf_eta = 1.0f;
                    if constexpr (GENERAL) {
                        const PtMaterial& m = use_tm ? tm : sc.materials[s.material];
                        gb.ns = s.sh_n; gb.ng = s.n;
                        gb.ss = normalize(s.sh_dpdu);
                        gb.ts = normalize(cross(gb.ns, gb.ss));
                        gb.lobes = m.lobes; gb.n_lobes = m.n_lobes;
#if PT_LOBES_IN_LDS
                        if constexpr (!INST && !TEX) {          // (the one-kernel textured form spills as it is: 109 registers, 171 with this)
                            // The lobe evaluators walk the material's lobe list four times per vertex (f, pdf, two sample_f), a dependent chain of
                            // small loads each time.  The queue is ordered by material, so the lanes that got here nearly always share one: then
                            // the wave copies that material's lobes (<= 400 bytes) into LDS once and every walk reads them there.
                            const int32_t mat0 = __builtin_amdgcn_readfirstlane(s.material);
                            const unsigned long long here = __ballot(1);
                            if (__ballot(use_tm || s.material != mat0) == 0ull) {
                                const uint32_t nd = (uint32_t)__builtin_amdgcn_readfirstlane((int)m.n_lobes) * (uint32_t)(sizeof(PtLobe) / 4);
                                const uint32_t nact = (uint32_t)__popcll(here), rank = (uint32_t)__popcll(here & below);
                                const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.materials[mat0].lobes);
                                uint32_t* dst = &s_lobes[threadIdx.x >> 6][0];
                                for (uint32_t k = rank; k < nd; k += nact) dst[k] = src[k];
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                                gb.lobes = reinterpret_cast<const PtLobe*>(dst);
                            }
                        }
#endif
                        nonspecular = m.nonspecular > 0;
                        bsdf_eta = m.bsdf_eta;
                    } else {
                        const PtMaterial& m = sc.materials[s.material];
                        b.ns = s.sh_n; b.ng = s.n;
                        b.ss = normalize(s.sh_dpdu);
                        b.ts = normalize(cross(b.ns, b.ss));
                        b.r = mk3(m.kd[0], m.kd[1], m.kd[2]);
                        b.n_lobes = is_black(b.r) ? 0 : 1;
                        b.oren = clampf(m.sigma, 0.0f, 90.0f) != 0.0f;
                        b.oa = m.oren_a; b.ob = m.oren_b;
                        nonspecular = b.n_lobes > 0;
                    }
                    const uint32_t kNoSpec = PT_BSDF_ALL & ~PT_BSDF_SPECULAR;
                    auto eval_f = [&](V3 wo_w, V3 wi_w) -> V3 {
                        if constexpr (GENERAL) return gbsdf_f(gb, wo_w, wi_w, kNoSpec); else return bsdf_f(b, wo_w, wi_w);
                    };
                    auto eval_pdf = [&](V3 wo_w, V3 wi_w) -> float {
                        if constexpr (GENERAL) return gbsdf_pdf(gb, wo_w, wi_w, kNoSpec); else return bsdf_pdf(b, wo_w, wi_w);
                    };
                    auto sample_bsdf = [&](V3 wo_w, V2 uu, uint32_t fl, V3* f_o, V3* wi_o, float* pdf_o, uint32_t* type_o) -> bool {
                        if constexpr (GENERAL) return gbsdf_sample_f(gb, wo_w, uu, fl, f_o, wi_o, pdf_o, type_o);
                        else { *type_o = PT_BSDF_REFLECTION | PT_BSDF_DIFFUSE; return bsdf_sample_f(b, wo_w, uu, f_o, wi_o, pdf_o); }
                    };
                    PT_SHP(1);
                    // ---- next-event estimation (uniform_sample_one_light_surface, sample_lights.rs:129-176)
                    if constexpr (PART == 2) {           // part 1 drew this vertex's next-event dimensions and noted how many
                        if (nonspecular && sc.n_lights > 0) sm.s.dim += (P.nee[p] & PT_NEE_DIMS5) ? 5u : 1u;
                    }
                    if (PART != 2 && nonspecular && sc.n_lights > 0) {
                        const float* tab = grid_lookup(sc.grid, s.p);
                        float light_pdf;
                        uint32_t light_num = sample_discrete(tab, sc.n_lights, sm.get_1d(sc), &light_pdf);
                        PT_SHP(2);
                        if (light_pdf > 0.0f) {
                            V2 u_light = sm.get_2d(sc);
                            V2 u_scat = sm.get_2d(sc);
                            PT_SHP(3);
#if PT_LIGHTS_IN_LDS
                            const PtLight& lt = lights_lds ? s_lights[light_num] : sc.lights[light_num];
#else
                            const PtLight& lt = sc.lights[light_num];
#endif
                            uint32_t nee = (light_num << 8) | (PART == 1 ? PT_NEE_DIMS5 : 0u);
                            V3 A = mk3(0.0f, 0.0f, 0.0f), B = mk3(0.0f, 0.0f, 0.0f);
                            V3 li, wi, lp, lperr, ln;
                            float lpdf;
                            if (light_sample_any<SPH>(sc, lt, s.p, s.p_error, s.n, u_light, &li, &wi, &lpdf, &lp, &lperr, &ln)) {
                                if (lpdf > 0.0f && !is_black(li)) {
                                    V3 f = eval_f(s.wo, wi) * abs_dot(wi, s.sh_n);
                                    float spdf = eval_pdf(s.wo, wi);
                                    if (!is_black(f)) {
                                        // VisibilityTester: Interaction::spawn_ray_to (interaction.rs:118-127)
                                        V3 origin = offset_ray_origin(s.p, s.p_error, s.n, lp - s.p);
                                        V3 target = offset_ray_origin(lp, lperr, ln, origin - lp);
                                        V3 sd = target - origin;
                                        s_stage[0][threadIdx.x] = make_float4(origin.x, origin.y, origin.z, 1.0f - PT_SHADOW_EPS);
                                        s_stage[1][threadIdx.x] = make_float4(sd.x, sd.y, sd.z, 0.0f);
                                        if (Q.shadow_key) s_pkey[n_batch][threadIdx.x] = ray_sort_key(sc, origin, sd);
                                        wr |= 4u; PT_COMMIT_NOW(1);
                                        float weight = power_heuristic(lpdf, spdf);
                                        A = f * li * (weight / lpdf);
                                        nee |= PT_NEE_SHADOW;
                                    }
                                }
                            }
                            PT_SHP(4);
                            // BSDF sampling half of MIS (sample_lights.rs:393-451)
                            V3 f2, wi2;
                            float spdf2;
                            uint32_t type2;      // never specular: the flags exclude BSDF_SPECULAR (sample_lights.rs:343-347)
                            if (sample_bsdf(s.wo, u_scat, kNoSpec, &f2, &wi2, &spdf2, &type2)) {
                                V3 f = f2 * abs_dot(wi2, s.sh_n);
                                if (!is_black(f) && spdf2 > 0.0f) {
                                    // light.pdf_li -> Shape::pdf_from (shape.rs:40-54): one test against the light's own triangle
                                    V3 po = offset_ray_origin(s.p, s.p_error, s.n, wi2);
                                    Surf ls;
                                    float lt_t;
                                    bool lhit;
                                    if (lt.mesh_flags & (PT_MESH_HAS_N | PT_MESH_HAS_S | PT_MESH_HAS_UV | PT_LIGHT_SPHERE)) {
                                        lhit = make_surf_any<SPH>(sc, po, wi2, lt.tri_rec, ls, &lt_t);
                                    } else {           // the light record already holds the triangle
                                        TriVerts ltv;
                                        ltv.p0 = ld3(lt.p0); ltv.p1 = ld3(lt.p1); ltv.p2 = ld3(lt.p2);
                                        ltv.prim = lt.prim;
                                        ltv.flags = ((lt.mesh_flags & PT_MESH_TWO_SIDED) ? 0u : PT_TRI_ONE_SIDED) |
                                                    ((((lt.mesh_flags & PT_MESH_REVERSE_ORIENTATION) != 0) ^ ((lt.mesh_flags & PT_MESH_SWAPS_HANDEDNESS) != 0)) ? PT_TRI_FLIP : 0u);
                                        lhit = make_surf_tv(sc, po, wi2, ltv, 0u, ls, &lt_t);
                                    }
                                    if (lhit) {
                                        float lp2 = distance_squared(s.p, ls.p) / (abs_dot(ls.n, -wi2) * lt.area);
                                        if (isinf(lp2)) lp2 = 0.0f;
                                        if (lp2 != 0.0f) {
                                            float weight = power_heuristic(spdf2, lp2);
                                            // if the probe lands on this triangle its interaction is `ls` again
                                            V3 le = light_L(lt, ls.n, -wi2);
                                            B = f * le * 1.0f * (weight / spdf2);
                                            s_stage[2][threadIdx.x] = make_float4(po.x, po.y, po.z, PT_INF);
                                            s_stage[3][threadIdx.x] = make_float4(wi2.x, wi2.y, wi2.z, 0.0f);
                                            wr |= 8u; PT_COMMIT_NOW(2);
                                            nee |= PT_NEE_PROBE;
                                        }
                                    }
                                }
                            }
                            PT_SHP(5);
                            if (nee & (PT_NEE_SHADOW | PT_NEE_PROBE)) {
                                s_stage[4][threadIdx.x] = make_float4(A.x, A.y, A.z, light_pdf);
                                s_stage[5][threadIdx.x] = make_float4(B.x, B.y, B.z, 0.0f);
                                s_stage[6][threadIdx.x] = make_float4(beta.x, beta.y, beta.z, 0.0f);
                                o_nee = nee; wr |= 16u; PT_COMMIT_NOW(3);
                                want_nee = true;
                                want_sh = (nee & PT_NEE_SHADOW) != 0;
                                want_pr = (nee & PT_NEE_PROBE) != 0;
                            } else if constexpr (PART == 1) { P.nee[p] = nee; }        // nothing pending, but part 2 reads the dimension count
                        } else if constexpr (PART == 1) { P.nee[p] = 0u; }              // light_pdf == 0: one dimension drawn
                    }
                    PT_SHP(6);
                    // ---- continuation (path.rs:139-233)
                    if constexpr (PART != 1) {
                    V2 u = sm.get_2d(sc);
                    PT_SHP(7);
                    V3 f, wi;
                    float pdf;
                    uint32_t stype;
                    const V3 wo_ray = -rd;        // path.rs:140 samples with -ray.d; isect.wo differs from it on a sphere (renormalised)
                    if (sample_bsdf(wo_ray, u, PT_BSDF_ALL, &f, &wi, &pdf, &stype) && !(is_black(f) || pdf == 0.0f)) {
                        beta = beta * (f * (abs_dot(wi, s.sh_n) / pdf));
                        if (stype & PT_BSDF_SPECULAR) flags |= PT_ST_SPECULAR; else flags &= ~PT_ST_SPECULAR;
                        if ((stype & PT_BSDF_SPECULAR) && (stype & PT_BSDF_TRANSMISSION))     // path.rs:157-168
                            eta_scale *= dot(wo_ray, s.n) > 0.0f ? bsdf_eta * bsdf_eta : 1.0f / (bsdf_eta * bsdf_eta);
                        V3 no = offset_ray_origin(s.p, s.p_error, s.n, wi);
                        bool alive = true;
                        V3 rr = beta * eta_scale;
                        float mx = max3(rr.x, rr.y, rr.z);
                        if (mx < sc.rr_threshold && bounces > 3) {
                            float q = fmaxf(0.05f, 1.0f - mx);
                            if (sm.get_1d(sc) < q) alive = false;
                            else beta = beta / (1.0f - q);
                        }
                        if (alive) {
                            o_ray_o = make_float4(no.x, no.y, no.z, PT_INF);
                            // .w is an OPAQUE zero, not the literal: with a literal the compiler keeps (wi.x, wi.y, wi.z, 0) in a register
                            // tuple whose last member is its shared constant-zero register, pairs that zero with the register holding
                            // wi.z as the high / low halves of zero-extended 64-bit table offsets, and then uses the low half as scratch
                            // inside the Russian-roulette branch (the sampler's table addressing) while wi.z is still waiting to be stored:
                            // hipcc 7.2 -O2 / -O3, the textured and instanced kernels with the stores held back -- survivors of the
                            // roulette continued with direction z = 0 (profiles/r03_deferred_store_miscompile.md)
                            o_ray_d = make_float4(wi.x, wi.y, wi.z, opaque_zero());
                            o_beta = make_float4(beta.x, beta.y, beta.z, eta_scale);
                            wr |= 3u; PT_COMMIT_NOW(4);
                            bounces++;
                            cont = true;
                        }
                    }
                    }
                    dim = sm.s.dim;
                }
                if (cont) { o_state = (dim & 0xffffu) | ((bounces & 0xffu) << 16) | ((flags & ~PT_ST_CAMERA) << 24); wr |= 32u; } PT_COMMIT_NOW(5);     // later rays are plain Rays
            }
        }
        if constexpr (DEEP) {          // the next iteration's leaf record: its index has arrived by now (asked for at the top of this iteration's work)
            if (pf_lane && pf_rec >= 0) {
                const float4* q = reinterpret_cast<const float4*>(sc.tris) + (size_t)(uint32_t)pf_rec * 3;
                pf_ra = q[0]; pf_rb = q[1]; pf_rc = q[2];
            }
        }
        PT_SHP(8);
        // ---- order-preserving wave compaction into the next / nee queues, one iteration behind: the four reservations of this
        // iteration are asked for here and used after the next iteration's work, so that nobody waits for the atomics' round trip
        // ---- everything this vertex writes in one burst, after the last load of the iteration
        commit();
        // ---- where it goes next: remembered (path | queue bits), queued with the batch
        s_pend[n_batch][threadIdx.x] = p;
        pend_want |= ((cont ? 1u : 0u) | (want_nee ? 2u : 0u) | (want_sh ? 4u : 0u) | (want_pr ? 8u : 0u)) << (4u * n_batch);
        if (++n_batch == PT_SHADE_FLUSH) flush_batch();
        PT_SHP(9);
    }
    flush_batch();
#ifdef PT_PROFILE_SHADE
    if (lane == 0) { for (int k = 0; k < 16; k++) atomicAdd(&g_shade_prof[k], shp_acc[k]); }
#endif
    if (n_vert) atomicAdd(&s_vert, (unsigned long long)n_vert);
    __syncthreads();
    if (threadIdx.x == 0 && s_vert) atomicAdd(&cnt->vertices, s_vert);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_WAVES) k_shade(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<false, false>(sc, P, Q, cnt, Q.cur, 0u, Q.counts[PT_Q_CUR], &Q.counts[PT_Q_TICKET]);
}
// the two halves of a material-sorted queue
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_WAVES) k_shade_matte_sorted(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<false, false>(sc, P, Q, cnt, Q.sorted, 0u, Q.counts[PT_Q_MATTE_END], &Q.counts[PT_Q_TICKET]);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_GEN_WAVES) k_shade_general(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, false>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_MATTE_END], Q.counts[PT_Q_TEX_BEGIN], &Q.counts[PT_Q_TICKET2]);
}
// scenes with spheres: hits and lights may be spheres
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_shade_matte_sorted_sph(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<false, true>(sc, P, Q, cnt, Q.sorted, 0u, Q.counts[PT_Q_MATTE_END], &Q.counts[PT_Q_TICKET]);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_shade_general_sph(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, true>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_MATTE_END], Q.counts[PT_Q_TEX_BEGIN], &Q.counts[PT_Q_TICKET2]);
}
// scenes with object instances: hits inside an instance are rebuilt in instance space and transformed back; every material
// rides in the general half of the sorted queue, so this one kernel shades them all
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_shade_general_inst(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, true, true, true>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_MATTE_END], Q.counts[PT_Q_GENERAL_END], &Q.counts[PT_Q_TICKET2]);
}
// ... without textures and without spheres: the lobe-list kernel with the instance-space reconstruction, nothing else
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_shade_general_inst_plain(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, false, false, true>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_MATTE_END], Q.counts[PT_Q_GENERAL_END], &Q.counts[PT_Q_TICKET2]);
}
// scenes with textured materials (and possibly spheres): lobes are built per hit for the textured ones
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_TEX_WAVES) k_shade_general_tex(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, true, true>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_TEX_BEGIN], Q.counts[PT_Q_GENERAL_END], &Q.counts[PT_Q_TICKET3]);
}

// The textured segment in two kernels (PBRTGPU_TEX_SPLIT, default on; scenes without instances).  k_shade_general_tex carries the texture
// interpreter, the MIP lookups and the lobe builder next to next-event estimation: 256 registers, 130 spilled, 3 KB of scratch per lane, and
// everything AFTER the textures runs 1.6x slower than in k_shade_general for it (profiles/r03_q_tex_kernel_experiments.txt).  k_tex_resolve
// does the per-hit half of Material::compute_scattering_functions alone -- surface, ray differentials, bump map, every programmed
// parameter -- and leaves 36 floats per path (PtPaths::tex_res); k_shade_general_res is k_shade_general with the lobe list built from them.
// ALL: the locally sorted queue (k_sort_local) -- every entry of Q.cur is looked at, the ones whose material is not textured are passed over
// after one look at the hit record's material field.
template <bool SPH, bool ALL = false>
PT_DEV void tex_resolve_body(const PtScene& sc, const PtPaths& P, const PtQueues& Q) {
    const uint32_t begin = ALL ? 0u : Q.counts[PT_Q_TEX_BEGIN], end = ALL ? Q.counts[PT_Q_CUR] : Q.counts[PT_Q_GENERAL_END];
    const uint32_t* list = ALL ? Q.cur : Q.sorted;
    __shared__ float s_texval[3 * PT_TEX_PROG_MAX][PT_BLOCK];          // the texture interpreter's node values (tex_eval_inl), a column per lane
    for (uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        const uint32_t p = list[i];
        const int32_t rec = P.hit_rec[p];
        if constexpr (ALL) {
            if (rec < 0) continue;
            const uint32_t m1 = sc.tris[rec].flags >> PT_TRI_MATERIAL_SHIFT;
            if (m1 == 0u || !sc.materials[m1 - 1u].textured) continue;
        }
        const uint32_t st = P.state[p];
        const uint32_t bounces = (st >> 16) & 0xffu, flags = st >> 24;
        if (rec < 0 || (int32_t)bounces >= sc.max_depth) continue;         // shade_body builds no BSDF there either
        const float4 ro4 = P.ray_o[p], rd4 = P.ray_d[p];
        const V3 ro = f4_3(ro4), rd = f4_3(rd4);
        const float4* q = reinterpret_cast<const float4*>(sc.tris) + (size_t)(uint32_t)rec * 3;
        const float4 rec_a = q[0], rec_b = q[1], rec_c = q[2];
        Surf s;
        float thit;
        if (!make_surf_any_rec<SPH>(sc, ro, rd, (uint32_t)rec, rec_a, rec_b, rec_c, s, &thit)) continue;
        if (s.material < 0 || !sc.materials[s.material].textured) continue;
        TexHit th;
        th.p = s.p; th.uv = s.uv;
        RayDiffs rdf;
        const bool has_diff = (flags & PT_ST_CAMERA) != 0;
        if (has_diff) {             // rebuild the camera ray's offset rays from its camera sample (as shade_body does)
            Sampler sl;
            sl.index = P.sobol_index[p];
            sl.dim = 2;
            const uint32_t pk2 = P.pixel[p];
            sl.px = (int32_t)(pk2 & 0xffffu) + sc.film.sample_bounds[0];
            sl.py = (int32_t)(pk2 >> 16) + sc.film.sample_bounds[1];
            V2 u_lens = mk2(0.0f, 0.0f);
            if (sc.cam.lens_radius > 0.0f) u_lens = sl.get_2d(sc);
            const float2 pf = P.p_film[p];
            camera_differentials(sc, mk2(pf.x, pf.y), u_lens, ro, rd, rdf);
        }
        compute_differentials(th, s.p, s.n, s.dpdu, s.dpdv, has_diff, rdf);
        PtMatParams mp;
        float a3[3];
        textured_params_t<true>(sc, s.material, th, mp, a3, s.n, s.uv, &s.sh_n, &s.sh_dpdu, s.sh_dpdv, s.sh_dndu, s.sh_dndv, &s_texval[0][threadIdx.x]);
        float4* o = P.tex_res + (size_t)p * PT_TEX_RES_F4;
        o[0] = make_float4(mp.m.kd[0], mp.m.kd[1], mp.m.kd[2], mp.m.ks[0]);
        o[1] = make_float4(mp.m.ks[1], mp.m.ks[2], mp.m.kr[0], mp.m.kr[1]);
        o[2] = make_float4(mp.m.kr[2], mp.m.kt[0], mp.m.kt[1], mp.m.kt[2]);
        o[3] = make_float4(mp.m.opacity[0], mp.m.opacity[1], mp.m.opacity[2], mp.m.sigma);
        o[4] = make_float4(mp.m.metal_eta[0], mp.m.metal_eta[1], mp.m.metal_eta[2], mp.m.metal_k[0]);
        o[5] = make_float4(mp.m.metal_k[1], mp.m.metal_k[2], mp.m.uroughness, mp.m.vroughness);
        o[6] = make_float4(mp.m.eta, a3[0], a3[1], a3[2]);
        o[7] = make_float4(s.sh_n.x, s.sh_n.y, s.sh_n.z, s.sh_dpdu.x);
        o[8] = make_float4(s.sh_dpdu.y, s.sh_dpdu.z, 0.0f, 0.0f);
    }
}
#ifndef PT_TEX_RESOLVE_WAVES
#define PT_TEX_RESOLVE_WAVES 2
#endif
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_TEX_RESOLVE_WAVES) k_tex_resolve(PtScene sc, PtPaths P, PtQueues Q) { tex_resolve_body<false>(sc, P, Q); }
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_TEX_RESOLVE_WAVES) k_tex_resolve_sph(PtScene sc, PtPaths P, PtQueues Q) { tex_resolve_body<true>(sc, P, Q); }
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_TEX_RESOLVE_WAVES) k_tex_resolve_all(PtScene sc, PtPaths P, PtQueues Q) { tex_resolve_body<false, true>(sc, P, Q); }
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_GEN_WAVES) k_shade_all_res(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, false, false, false, true>(sc, P, Q, cnt, Q.cur, 0u, Q.counts[PT_Q_CUR], &Q.counts[PT_Q_TICKET]);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_GEN_WAVES) k_shade_general_res(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, false, false, false, true>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_TEX_BEGIN], Q.counts[PT_Q_GENERAL_END], &Q.counts[PT_Q_TICKET3]);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_shade_general_res_sph(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, true, false, false, true>(sc, P, Q, cnt, Q.sorted, Q.counts[PT_Q_TEX_BEGIN], Q.counts[PT_Q_GENERAL_END], &Q.counts[PT_Q_TICKET3]);
}

// ---- the vertex in two kernels (shade_body PART 1 / PART 2; PBRTGPU_NEE_SPLIT selects the kernel families that run this way).  Next-event
// estimation is the register peak of every shading kernel and, in scenes with sphere lights, the only place where Sphere::sample_from and the
// EFloat quadratic of Shape::pdf_from are needed: on its own it leaves the continuation half lean, and the sphere code out of it.
#ifndef PT_NEE_WAVES
#define PT_NEE_WAVES 2
#endif
#ifndef PT_CONT_WAVES
#define PT_CONT_WAVES 3
#endif
#ifndef PT_NEE_GEN_WAVES
#define PT_NEE_GEN_WAVES 2
#endif
#ifndef PT_CONT_GEN_WAVES
#define PT_CONT_GEN_WAVES 3
#endif
#ifndef PT_CONT_SPH_WAVES
#define PT_CONT_SPH_WAVES 2      // the sphere-capable continuation halves rebuild hits on spheres (Sphere::intersect + its interaction): 185-200 registers
#endif
#ifndef PT_NEE_SPLIT_DEFAULT
#define PT_NEE_SPLIT_DEFAULT 0
#endif
#define PT_SPLIT_KERNELS(NAME, WN, WC, LIST, BEGIN, END, TN, TC, ...)                                                                              \
    extern "C" __global__ void __launch_bounds__(PT_BLOCK, WN) NAME##_nee(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {                     \
        shade_body<__VA_ARGS__, 1>(sc, P, Q, cnt, LIST, BEGIN, END, &Q.counts[TN]);                                                                \
    }                                                                                                                                              \
    extern "C" __global__ void __launch_bounds__(PT_BLOCK, WC) NAME##_cont(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {                    \
        shade_body<__VA_ARGS__, 2>(sc, P, Q, cnt, LIST, BEGIN, END, &Q.counts[TC]);                                                                \
    }
PT_SPLIT_KERNELS(k_shade, PT_NEE_WAVES, PT_CONT_WAVES, Q.cur, 0u, Q.counts[PT_Q_CUR], PT_Q_TICKET_N1, PT_Q_TICKET, false, false, false, false, false)
PT_SPLIT_KERNELS(k_shade_matte_sorted, PT_NEE_WAVES, PT_CONT_WAVES, Q.sorted, 0u, Q.counts[PT_Q_MATTE_END], PT_Q_TICKET_N1, PT_Q_TICKET, false, false, false, false, false)
PT_SPLIT_KERNELS(k_shade_general, PT_NEE_GEN_WAVES, PT_CONT_GEN_WAVES, Q.sorted, Q.counts[PT_Q_MATTE_END], Q.counts[PT_Q_TEX_BEGIN], PT_Q_TICKET_N2, PT_Q_TICKET2, true, false, false, false, false)
PT_SPLIT_KERNELS(k_shade_matte_sorted_sph, PT_NEE_WAVES, PT_CONT_SPH_WAVES, Q.sorted, 0u, Q.counts[PT_Q_MATTE_END], PT_Q_TICKET_N1, PT_Q_TICKET, false, true, false, false, false)
PT_SPLIT_KERNELS(k_shade_general_sph, PT_NEE_GEN_WAVES, PT_CONT_SPH_WAVES, Q.sorted, Q.counts[PT_Q_MATTE_END], Q.counts[PT_Q_TEX_BEGIN], PT_Q_TICKET_N2, PT_Q_TICKET2, true, true, false, false, false)
PT_SPLIT_KERNELS(k_shade_general_res, PT_NEE_GEN_WAVES, PT_CONT_GEN_WAVES, Q.sorted, Q.counts[PT_Q_TEX_BEGIN], Q.counts[PT_Q_GENERAL_END], PT_Q_TICKET_N3, PT_Q_TICKET3, true, false, false, false, true)
PT_SPLIT_KERNELS(k_shade_general_res_sph, PT_NEE_GEN_WAVES, PT_CONT_SPH_WAVES, Q.sorted, Q.counts[PT_Q_TEX_BEGIN], Q.counts[PT_Q_GENERAL_END], PT_Q_TICKET_N3, PT_Q_TICKET3, true, true, false, false, true)

// Sort keys of a bounce's continuation rays, for scenes larger than the Infinity Cache (pt_context.cpp sort_cont): a pass of its own over the
// list k_shade has just written -- inside the shading kernels the key arithmetic cost the general kernel registers it does not have
// (mixed materials 857 -> 832 Mrays/s with the sort off), here it is 32 bytes read per ray where traversal is 96 % of the frame
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_cont_keys(PtScene sc, PtPaths P, const uint32_t* list, uint32_t n, uint32_t* keys) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = list[i];
        const float4 o = P.ray_o[p], d = P.ray_d[p];
        keys[i] = ray_sort_key(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z));
    }
}

// Every material through the lobe-list path in PATH order, no material sort (experiment switch PBRTGPU_SHADE_UNSORTED=1: sorted queues make
// a wave see one material but walk the path pool with gaps; this form keeps the pool accesses dense and lets the lobe dispatch diverge)
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_GEN_WAVES) k_shade_all(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, false>(sc, P, Q, cnt, Q.cur, 0u, Q.counts[PT_Q_CUR], &Q.counts[PT_Q_TICKET]);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_shade_all_sph(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, true>(sc, P, Q, cnt, Q.cur, 0u, Q.counts[PT_Q_CUR], &Q.counts[PT_Q_TICKET]);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_SHADE_TEX_WAVES) k_shade_all_tex(PtScene sc, PtPaths P, PtQueues Q, PtCounters* cnt) {
    shade_body<true, true, true>(sc, P, Q, cnt, Q.cur, 0u, Q.counts[PT_Q_CUR], &Q.counts[PT_Q_TICKET]);
}

// ============================================================ film
PT_DEV V3 validate_radiance(V3 l) {    // sampler.rs:151-176
    if (!finite3(l)) return mk3(0.0f, 0.0f, 0.0f);
    float y = lum_y(l);
    if (y < -1e-5f) return mk3(0.0f, 0.0f, 0.0f);
    if (isinf(y)) return mk3(0.0f, 0.0f, 0.0f);
    return l;
}
// FilmTile::add_sample_filter (film_tile.rs:84-183) for one sample.  `own` accumulates the
// common case (the sample's own pixel gets weight exactly 1) without atomics so that the f32
// sum runs in sample order exactly like the reference's tile loop; every other footprint goes
// through float atomics into `spill`.
PT_DEV float film_weight(const PtFilm& fm, int32_t x, int32_t y, V2 pf) {
    float dx = fabsf((float)x + 0.5f - pf.x), dy = fabsf((float)y + 0.5f - pf.y);
    if (!(dx <= fm.filter_radius[0]) || !(dy <= fm.filter_radius[1])) return 0.0f;
    int ix = min((int)floorf(dx * (fm.inv_filter_radius[0] * 15.0f)), 15);
    int iy = min((int)floorf(dy * (fm.inv_filter_radius[1] * 15.0f)), 15);
    return fm.filter_table[iy * 16 + ix];
}
PT_DEV void film_add(const PtFilm& fm, float4* spill, int32_t px, int32_t py, V2 pf, V3 l, float4& own_acc) {
    float ly = lum_y(l);
    if (ly > fm.max_sample_luminance) l = l * (fm.max_sample_luminance / ly);
    float rx = fm.filter_radius[0], ry = fm.filter_radius[1];
    int32_t p0x = (int32_t)floorf(pf.x - rx), p0y = (int32_t)floorf(pf.y - ry);
    int32_t p1x = (int32_t)ceilf(pf.x + rx), p1y = (int32_t)ceilf(pf.y + ry);
    p0x = max(p0x, fm.crop[0]); p0y = max(p0y, fm.crop[1]);
    p1x = min(p1x, fm.crop[2]); p1y = min(p1y, fm.crop[3]);
    if (p1x - p0x <= 0 || p1y - p0y <= 0) return;
    // normalise the footprint's weights to sum 1 (reference quirk Q1)
    float sum = 0.0f;
    int nz = 0;
    for (int32_t y = p0y; y < p1y; y++)
        for (int32_t x = p0x; x < p1x; x++) {
            float w = film_weight(fm, x, y, pf);
            sum += w;
            if (w != 0.0f) nz++;
        }
    if (sum <= 0.0f) return;
    float isum = 1.0f / sum;
    int32_t width = fm.crop[2] - fm.crop[0];
    for (int32_t y = p0y; y < p1y; y++)
        for (int32_t x = p0x; x < p1x; x++) {
            float w = film_weight(fm, x, y, pf);
            if (w == 0.0f) continue;      // contributes +0 in the reference
            w *= isum;
            V3 cadd = (l * 1.0f) * w;     // l * sample_weight * filter_weight
            if (nz == 1 && x == px && y == py) {
                own_acc.x += cadd.x; own_acc.y += cadd.y; own_acc.z += cadd.z; own_acc.w += w;
            } else {
                float* dst = reinterpret_cast<float*>(spill + (size_t)(y - fm.crop[1]) * width + (x - fm.crop[0]));
                atomicAdd(dst + 0, cadd.x); atomicAdd(dst + 1, cadd.y); atomicAdd(dst + 2, cadd.z); atomicAdd(dst + 3, w);
            }
        }
}
// One thread per pixel of the pass; samples are folded in sample order.
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_film(PtScene sc, PtPaths P, const uint32_t* pixels, uint32_t n_pix, uint32_t n_samples,
                                                             float4* own, float4* spill, float* radiance_out, uint32_t s0, uint32_t spp_total) {
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_pix; j += gridDim.x * blockDim.x) {
        uint32_t pk = pixels[j];
        int32_t px = (int32_t)(pk & 0xffffu) + sc.film.sample_bounds[0], py = (int32_t)(pk >> 16) + sc.film.sample_bounds[1];
        bool inside = px >= sc.film.crop[0] && px < sc.film.crop[2] && py >= sc.film.crop[1] && py < sc.film.crop[3];
        size_t fidx = inside ? (size_t)(py - sc.film.crop[1]) * (sc.film.crop[2] - sc.film.crop[0]) + (px - sc.film.crop[0]) : 0;
        float4 acc = inside ? own[fidx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t k = 0; k < n_samples; k++) {
            size_t path = (size_t)k * n_pix + j;
            float4 L4 = P.L[path];
            V3 l = validate_radiance(mk3(L4.x, L4.y, L4.z));
            if (radiance_out) {
                float* ro = radiance_out + ((size_t)j * spp_total + (s0 + k)) * 3;
                ro[0] = l.x; ro[1] = l.y; ro[2] = l.z;
            }
            float2 pf = P.p_film[path];
            film_add(sc.film, spill, inside ? px : -0x7fffffff, py, mk2(pf.x, pf.y), l, acc);
        }
        if (inside) own[fidx] = acc;
    }
}
// Film::merge_film_tile's RGB -> XYZ per contribution buffer (film.rs:219-241)
extern "C" __global__ void k_film_xyzw(const float4* own, const float4* spill, float4* xyzw, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 a = own[i], b = spill[i];
        V3 xa = rgb_to_xyz(mk3(a.x, a.y, a.z)), xb = rgb_to_xyz(mk3(b.x, b.y, b.z));
        xyzw[i] = make_float4(xa.x + xb.x, xa.y + xb.y, xa.z + xb.z, a.w + b.w);
    }
}
// Film::merge_film_tile across hosts without a collective (film.rs:219-241): another rank's {X,Y,Z,weight} film, staged through the host, is added
extern "C" __global__ void k_film_add(float4* xyzw, const float4* other, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 a = xyzw[i], b = other[i];
        xyzw[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
// Film::write_image (film.rs:440-484)
extern "C" __global__ void k_film_rgb(const float4* xyzw, float* rgb, uint32_t n, float scale) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = xyzw[i];
        V3 c = xyz_to_rgb(mk3(v.x, v.y, v.z));
        if (v.w > 0.0f) {
            float inv = 1.0f / v.w;
            c = mk3(fmaxf(0.0f, c.x * inv), fmaxf(0.0f, c.y * inv), fmaxf(0.0f, c.z * inv));
        }
        rgb[3 * (size_t)i + 0] = (c.x + 0.0f) * scale;
        rgb[3 * (size_t)i + 1] = (c.y + 0.0f) * scale;
        rgb[3 * (size_t)i + 2] = (c.z + 0.0f) * scale;
    }
}

// ============================================================ light grid
// SpatialLightDistribution::compute_distribution for every voxel (spatial.rs:113-196) followed
// by Distribution1D::new (distribution.rs:34-58).  One thread per voxel.
PT_DEV float radical_inverse_dev(uint32_t base_index, uint64_t a) {   // radical_inverse.rs:38-58
    if (base_index == 0) {
        uint64_t r = __brevll(a);
        return (float)r * 5.4210108624275222e-20f;
    }
    const uint64_t primes[5] = {2, 3, 5, 7, 11};
    uint64_t base = primes[base_index];
    float inv_base = 1.0f / (float)base;
    uint64_t rev = 0;
    float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        rev = rev * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return fminf((float)rev * inv_base_n, PT_ONE_MINUS_EPS);
}
// Two kernels.  The 128 probe points of a voxel are independent, so a workgroup of 128 threads takes (voxel, light) pairs by the batch,
// every thread one probe of each; then one thread per pair adds its 128 terms in probe order, which is the order the reference's loop
// adds them in.  (One thread per voxel walking 128 probes x all lights, its sums in memory and five radical inverses per probe, took
// 0.73 ms of every upload for two lights.)  The second kernel turns each voxel's sums into its distribution.
PT_DEV void light_grid_voxel(const PtLightGrid& g, uint32_t v, V3* vmin, V3* vmax) {
    uint32_t pi0 = v % g.voxels[0], pi1 = (v / g.voxels[0]) % g.voxels[1], pi2 = v / (g.voxels[0] * g.voxels[1]);
    V3 wmin = ld3(g.wb_min), wmax = ld3(g.wb_max);
    V3 p0 = mk3((float)pi0 / (float)g.voxels[0], (float)pi1 / (float)g.voxels[1], (float)pi2 / (float)g.voxels[2]);
    V3 p1 = mk3((float)(pi0 + 1) / (float)g.voxels[0], (float)(pi1 + 1) / (float)g.voxels[1], (float)(pi2 + 1) / (float)g.voxels[2]);
    V3 a = mk3(lerpf(p0.x, wmin.x, wmax.x), lerpf(p0.y, wmin.y, wmax.y), lerpf(p0.z, wmin.z, wmax.z));
    V3 b = mk3(lerpf(p1.x, wmin.x, wmax.x), lerpf(p1.y, wmin.y, wmax.y), lerpf(p1.z, wmin.z, wmax.z));
    *vmin = mk3(a.x <= b.x ? a.x : b.x, a.y <= b.y ? a.y : b.y, a.z <= b.z ? a.z : b.z);
    *vmax = mk3(a.x >= b.x ? a.x : b.x, a.y >= b.y ? a.y : b.y, a.z >= b.z ? a.z : b.z);
}
constexpr int kGridBatch = 64;            // (voxel, light) pairs a workgroup takes at a time
// vox_list: the voxels whose tables rows 0 .. n_vox-1 of `data` receive (the lazily filled grid: the voxels a bounce has touched for the first time);
// nullptr: row k is voxel k (the dense grid, filled at upload).
template <bool SPH>
PT_DEV void light_grid_sums(const PtScene& sc, float* data, uint32_t n_vox, const uint32_t* vox_list) {
    __shared__ float s_term[kGridBatch][129];
    __shared__ float s_box[kGridBatch][6];
    const PtLightGrid& g = sc.grid;
    const uint32_t nl = g.n_lights;
    const uint32_t i = threadIdx.x;
    const V3 t = mk3(radical_inverse_dev(0, i), radical_inverse_dev(1, i), radical_inverse_dev(2, i));
    const V2 u = mk2(radical_inverse_dev(3, i), radical_inverse_dev(4, i));
    const uint64_t n_jobs = (uint64_t)n_vox * nl, n_batches = (n_jobs + kGridBatch - 1) / kGridBatch;
    for (uint64_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        // thread i: probe i of every pair of the batch.  A probe that yields nothing contributes +0, which leaves the sum as the reference's
        // skipped addition does (the sum starts at +0 and no addition of these terms can make it -0).
        if (i < (uint32_t)kGridBatch && batch * kGridBatch + i < n_jobs) {          // the voxel of pair i, once
            V3 vmin, vmax;
            const uint32_t row = (uint32_t)((batch * kGridBatch + i) / nl);
            light_grid_voxel(g, vox_list ? vox_list[row] : row, &vmin, &vmax);
            s_box[i][0] = vmin.x; s_box[i][1] = vmin.y; s_box[i][2] = vmin.z; s_box[i][3] = vmax.x; s_box[i][4] = vmax.y; s_box[i][5] = vmax.z;
        }
        __syncthreads();
        for (uint32_t b = 0; b < (uint32_t)kGridBatch; b++) {
            const uint64_t job = batch * kGridBatch + b;
            if (job >= n_jobs) break;
            const uint32_t j = (uint32_t)(job % nl);
            const V3 vmin = mk3(s_box[b][0], s_box[b][1], s_box[b][2]), vmax = mk3(s_box[b][3], s_box[b][4], s_box[b][5]);
            V3 po = mk3(lerpf(t.x, vmin.x, vmax.x), lerpf(t.y, vmin.y, vmax.y), lerpf(t.z, vmin.z, vmax.z));
            V3 li, wi, lp, le, ln;
            float pdf, term = 0.0f;
            // the reference point is a bare Interaction: zero normal and error (spatial.rs:152-159)
            if (light_sample_any<SPH>(sc, sc.lights[j], po, mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f), u, &li, &wi, &pdf, &lp, &le, &ln))
                if (pdf > 0.0f) term = lum_y(li) / pdf;
            s_term[b][i] = term;
        }
        __syncthreads();
        // thread b: the 128 terms of pair b, added in probe order
        const uint64_t job = batch * kGridBatch + i;
        if (i < (uint32_t)kGridBatch && job < n_jobs) {
            float acc = 0.0f;
            for (uint32_t k = 0; k < 128; k++) acc += s_term[i][k];
            data[(size_t)(job / nl) * g.stride + (uint32_t)(job % nl)] = acc;
        }
        __syncthreads();
    }
}
PT_DEV void light_grid_cdf(const PtScene& sc, float* data, uint32_t n_vox) {
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n_vox; v += gridDim.x * blockDim.x) {
        const PtLightGrid& g = sc.grid;
        const uint32_t nl = g.n_lights;
        float* func = data + (size_t)v * g.stride;
        float* cdf = func + nl;
        float sum = 0.0f;
        for (uint32_t j = 0; j < nl; j++) sum += func[j];
        float avg = sum / (float)(128u * nl);
        float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
        for (uint32_t j = 0; j < nl; j++) func[j] = fmaxf(min_contrib, func[j]);
        cdf[0] = 0.0f;
        for (uint32_t j = 1; j < nl + 1; j++) cdf[j] = cdf[j - 1] + func[j - 1] / (float)nl;
        float func_int = cdf[nl];
        if (func_int == 0.0f) for (uint32_t j = 1; j < nl + 1; j++) cdf[j] = (float)j / (float)nl;
        else for (uint32_t j = 1; j < nl + 1; j++) cdf[j] /= func_int;
        cdf[nl + 1] = func_int;
    }
}
extern "C" __global__ __launch_bounds__(128) void k_light_grid(PtScene sc, float* data, uint32_t n_vox, const uint32_t* vox_list) { light_grid_sums<false>(sc, data, n_vox, vox_list); }
extern "C" __global__ __launch_bounds__(128) void k_light_grid_sph(PtScene sc, float* data, uint32_t n_vox, const uint32_t* vox_list) { light_grid_sums<true>(sc, data, n_vox, vox_list); }
// ---- the lazily filled grid (PtLightGrid::row_of; SpatialLightDistribution::lookup, spatial.rs:199-260, fills a voxel on its first touch).
// Before a bounce is shaded, k_grid_mark rebuilds every hit of the bounce's queue exactly as the shading kernels will (same interaction point,
// same voxel arithmetic) and lists the voxels nobody has asked for yet; the host gives them rows, k_light_grid fills those, k_grid_assign
// publishes them.  A voxel's tables depend on the voxel alone, so filling more voxels than the shading ends up reading changes nothing.
PT_DEV uint32_t grid_voxel_of(const PtLightGrid& g, V3 p) {
    uint32_t pi[3];
    float pc[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; i++) {
        float o = pc[i] - g.wb_min[i];
        if (g.wb_max[i] > g.wb_min[i]) o = o / (g.wb_max[i] - g.wb_min[i]);
        o = clampf(o, 0.0f, 1.0f);
        float f = o * (float)g.voxels[i];
        uint32_t v = f > 0.0f ? (uint32_t)f : 0u;
        if (v > g.voxels[i] - 1) v = g.voxels[i] - 1;
        pi[i] = v;
    }
    return (pi[2] * g.voxels[1] + pi[1]) * g.voxels[0] + pi[0];
}
template <bool SPH, bool INST>
PT_DEV void grid_mark_body(const PtScene& sc, const PtPaths& P, const PtQueues& Q, int32_t* row_of, uint32_t* todo, uint32_t* todo_count) {
    const uint32_t n = Q.counts[PT_Q_CUR];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = Q.cur[i];
        const int32_t rec = P.hit_rec[p];
        if (rec < 0) continue;
        const V3 ro = f4_3(P.ray_o[p]), rd = f4_3(P.ray_d[p]);
        Surf s;
        float thit;
        bool found;
        if constexpr (INST) found = make_surf_inst<SPH>(sc, ro, rd, (uint32_t)rec, P.hit_inst[p], s, &thit);
        else found = make_surf_any<SPH>(sc, ro, rd, (uint32_t)rec, s, &thit);
        if (!found) continue;
        const uint32_t vox = grid_voxel_of(sc.grid, s.p);
        if (row_of[vox] == -1 && atomicCAS(&row_of[vox], -1, -2) == -1) todo[atomicAdd(todo_count, 1u)] = vox;      // -2: listed, row pending
    }
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_grid_mark(PtScene sc, PtPaths P, PtQueues Q, int32_t* row_of, uint32_t* todo, uint32_t* todo_count) {
    grid_mark_body<false, false>(sc, P, Q, row_of, todo, todo_count);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_grid_mark_sph(PtScene sc, PtPaths P, PtQueues Q, int32_t* row_of, uint32_t* todo, uint32_t* todo_count) {
    grid_mark_body<true, false>(sc, P, Q, row_of, todo, todo_count);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_grid_mark_inst(PtScene sc, PtPaths P, PtQueues Q, int32_t* row_of, uint32_t* todo, uint32_t* todo_count) {
    grid_mark_body<true, true>(sc, P, Q, row_of, todo, todo_count);
}
extern "C" __global__ void k_grid_assign(int32_t* row_of, const uint32_t* todo, uint32_t n, uint32_t row0, uint32_t* todo_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) row_of[todo[i]] = (int32_t)(row0 + i);
    if (i == 0) *todo_count = 0;
}
extern "C" __global__ void k_light_grid_cdf(PtScene sc, float* data, uint32_t n_vox) { light_grid_cdf(sc, data, n_vox); }

// ============================================================ hooks: sampler / camera
extern "C" __global__ void k_camera_rays(PtScene sc, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, float* out_o, float* out_d,
                                         float* out_pf) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int32_t px = pixel_xy[2 * i], py = pixel_xy[2 * i + 1];
        Sampler sm;
        sm.index = sampler_index(sc, sample_index[i], px, py);
        sm.dim = 0; sm.px = px; sm.py = py;
        V2 uf = sm.get_2d(sc);
        V2 pf = mk2((float)px + uf.x, (float)py + uf.y);
        V2 ul = sm.get_2d(sc);
        V3 o, d;
        generate_camera_ray(sc, pf, ul, &o, &d);
        out_o[3 * i] = o.x; out_o[3 * i + 1] = o.y; out_o[3 * i + 2] = o.z;
        out_d[3 * i] = d.x; out_d[3 * i + 1] = d.y; out_d[3 * i + 2] = d.z;
        out_pf[2 * i] = pf.x; out_pf[2 * i + 1] = pf.y;
    }
}
extern "C" __global__ void k_sobol_samples(PtScene sc, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim, float* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int32_t px = pixel_xy[2 * i], py = pixel_xy[2 * i + 1];
        uint64_t idx = sampler_index(sc, sample_index[i], px, py);
        out[i] = sample_dimension(sc, idx, dim[i], px, py);
    }
}

// ---- hooks: BSDF on the canonical frame
PT_DEV GBsdf canonical_bsdf(const PtScene& sc, uint32_t material) {
    GBsdf gb;
    gb.ns = mk3(0.0f, 0.0f, 1.0f); gb.ng = gb.ns;
    gb.ss = normalize(mk3(1.0f, 0.0f, 0.0f));
    gb.ts = normalize(cross(gb.ns, gb.ss));
    gb.lobes = sc.materials[material].lobes;
    gb.n_lobes = sc.materials[material].n_lobes;
    return gb;
}
extern "C" __global__ void k_bsdf_eval(PtScene sc, uint32_t material, uint32_t n, const float* wo, const float* wi, uint32_t flags, float* f, float* pdf) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        GBsdf gb = canonical_bsdf(sc, material);
        V3 o = ld3(wo + 3 * i), w = ld3(wi + 3 * i);
        V3 r = gbsdf_f(gb, o, w, flags);
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
        pdf[i] = gbsdf_pdf(gb, o, w, flags);
    }
}
extern "C" __global__ void k_bsdf_sample(PtScene sc, uint32_t material, uint32_t n, const float* wo, const float* u, uint32_t flags, float* f, float* wi,
                                         float* pdf, uint32_t* type) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        GBsdf gb = canonical_bsdf(sc, material);
        V3 r = mk3(0.0f, 0.0f, 0.0f), w = r;
        float p = 0.0f;
        uint32_t t = 0;
        if (!gbsdf_sample_f(gb, ld3(wo + 3 * i), mk2(u[2 * i], u[2 * i + 1]), flags, &r, &w, &p, &t)) { r = mk3(0.0f, 0.0f, 0.0f); w = r; p = 0.0f; t = 0; }
        f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z;
        wi[3 * i] = w.x; wi[3 * i + 1] = w.y; wi[3 * i + 2] = w.z;
        pdf[i] = p;
        type[i] = t;
    }
}

// ============================================================ recursive integrators (directlighting, whitted)
// DirectLightingIntegrator::li (integrators/directlighting.rs:66-135) and WhittedIntegrator::li (integrators/whitted.rs:38-110) with
// SamplerIntegrator::specular_reflect / specular_transmit (core/integrator/sampler.rs:37-143).  The reference recurses, and its sampler
// is consumed in that order (the reflect subtree first, then the transmit branch), so every camera sample walks its tree depth first:
// a stack of frames in HBM (PtRec::frames), one k_trace launch per tree level for the child rays and one for the node's next-event
// rays.  Per iteration of the host loop:
//   k_trace        the current ray of every live camera sample (continuation items)
//   k_rec_enter    the node at its hit: emitted light, BSDF, the next-event rays of the node (entries in PtRec), a new frame
//   k_trace        those shadow / MIS probe rays
//   k_rec_next     resolves the node's direct light in the reference's order of additions, then runs the walk -- sample specular_reflect,
//                  else specular_transmit, else return to the parent frame and continue there -- until a child ray has to be traced or
//                  the root has returned
// The interaction of a frame is rebuilt from its ray and hit record whenever it is needed again (after the reflect subtree, for the
// transmit sample): the same arithmetic, so the same values.  One kernel variant (spheres, textures, instances all compiled in).
struct RecNode {
    Surf s;
    GBsdf gb;
    PtMaterial tm;
    TexHit th;
    V3 n_before;
    float bsdf_eta;
    uint32_t spec_mask;          // PtMaterial::spec_mask of the node's BSDF
    bool found, has_bsdf;
};
// FULL = false: the instantiation for scenes without spheres, instances and textured materials (the triangle-only test, no texture programs, no
// per-hit lobe list in scratch): what `directlighting` / `whitted` run on BASELINE's scenes.  FULL = true: everything compiled in.
template <bool FULL>
PT_DEV void rec_build(const PtScene& sc, V3 ro, V3 rd, int32_t rec, uint32_t inst, bool has_diff, const RayDiffs& rdf, RecNode& nd) {
    float thit;
    if constexpr (FULL) nd.found = rec >= 0 && make_surf_inst<true>(sc, ro, rd, (uint32_t)rec, inst, nd.s, &thit);
    else nd.found = rec >= 0 && make_surf_any<false>(sc, ro, rd, (uint32_t)rec, nd.s, &thit);
    nd.has_bsdf = false;
    nd.bsdf_eta = 1.0f;
    nd.spec_mask = 0;
    if (!nd.found) return;
    nd.n_before = nd.s.sh_n;
    nd.th.p = nd.s.p; nd.th.uv = nd.s.uv;
    compute_differentials(nd.th, nd.s.p, nd.s.n, nd.s.dpdu, nd.s.dpdv, has_diff, rdf);
    if (nd.s.material < 0) return;
    const PtMaterial* m = &sc.materials[nd.s.material];
    if constexpr (FULL) {
        if (m->textured) {
            textured_lobes(sc, nd.s.material, nd.th, &nd.tm, nd.s.n, nd.s.uv, &nd.s.sh_n, &nd.s.sh_dpdu, nd.s.sh_dpdv, nd.s.sh_dndu, nd.s.sh_dndv);
            m = &nd.tm;
        }
    }
    if (!m->has_bsdf) return;
    nd.has_bsdf = true;
    nd.gb.ns = nd.s.sh_n; nd.gb.ng = nd.s.n;
    nd.gb.ss = normalize(nd.s.sh_dpdu);
    nd.gb.ts = normalize(cross(nd.gb.ns, nd.gb.ss));
    nd.gb.lobes = m->lobes; nd.gb.n_lobes = m->n_lobes;
    nd.bsdf_eta = m->bsdf_eta;
    nd.spec_mask = m->spec_mask;
}
PT_DEV float4* rec_frame(const PtRec& R, uint32_t depth, uint32_t k, uint32_t p) { return R.frames + ((size_t)depth * PT_REC_FRAME_F4 + k) * R.n_paths + p; }
// estimate_direct (sample_lights.rs:178-328) for light `light_num`: the two MIS terms and their rays go to entry e, the rays' results
// are combined by k_rec_next.  Returns the PT_NEE_* flags of the entry.
template <bool FULL>
PT_DEV uint32_t rec_estimate_direct(const PtScene& sc, const PtRec& R, const RecNode& nd, uint32_t light_num, V2 u_light, V2 u_scat, uint32_t e, float divisor = 1.0f) {
    const Surf& s = nd.s;
    const PtLight& lt = sc.lights[light_num];
    const uint32_t kNoSpec = PT_BSDF_ALL & ~PT_BSDF_SPECULAR;
    uint32_t nee = light_num << 8;
    V3 A = mk3(0.0f, 0.0f, 0.0f), B = A;
    V3 li, wi, lp, lperr, ln;
    float lpdf;
    if (light_sample_any<FULL>(sc, lt, s.p, s.p_error, s.n, u_light, &li, &wi, &lpdf, &lp, &lperr, &ln)) {
        if (lpdf > 0.0f && !is_black(li)) {
            V3 f = gbsdf_f(nd.gb, s.wo, wi, kNoSpec) * abs_dot(wi, s.sh_n);
            float spdf = gbsdf_pdf(nd.gb, s.wo, wi, kNoSpec);
            if (!is_black(f)) {
                V3 origin = offset_ray_origin(s.p, s.p_error, s.n, lp - s.p);
                V3 target = offset_ray_origin(lp, lperr, ln, origin - lp);
                V3 sd = target - origin;
                R.sh_o[e] = make_float4(origin.x, origin.y, origin.z, 1.0f - PT_SHADOW_EPS);
                R.sh_d[e] = make_float4(sd.x, sd.y, sd.z, 0.0f);
                float weight = power_heuristic(lpdf, spdf);
                A = f * li * (weight / lpdf);
                nee |= PT_NEE_SHADOW;
            }
        }
    }
    V3 f2, wi2;
    float spdf2;
    uint32_t type2;
    if (gbsdf_sample_f(nd.gb, s.wo, u_scat, kNoSpec, &f2, &wi2, &spdf2, &type2)) {
        V3 f = f2 * abs_dot(wi2, s.sh_n);
        if (!is_black(f) && spdf2 > 0.0f) {
            V3 po = offset_ray_origin(s.p, s.p_error, s.n, wi2);
            Surf ls;
            float lt_t;
            if (make_surf_any<FULL>(sc, po, wi2, lt.tri_rec, ls, &lt_t)) {          // light.pdf_li -> Shape::pdf_from (shape.rs:40-54)
                float lp2 = distance_squared(s.p, ls.p) / (abs_dot(ls.n, -wi2) * lt.area);
                if (isinf(lp2)) lp2 = 0.0f;
                if (lp2 != 0.0f) {
                    float weight = power_heuristic(spdf2, lp2);
                    V3 le = light_L(lt, ls.n, -wi2);
                    B = f * le * 1.0f * (weight / spdf2);
                    R.pr_o[e] = make_float4(po.x, po.y, po.z, PT_INF);
                    R.pr_d[e] = make_float4(wi2.x, wi2.y, wi2.z, 0.0f);
                    nee |= PT_NEE_PROBE;
                }
            }
        }
    }
    R.A[e] = make_float4(A.x, A.y, A.z, divisor);
    R.B[e] = make_float4(B.x, B.y, B.z, 0.0f);
    R.flags[e] = nee;
    return nee;
}
// camera rays carry differentials: the offset rays of every camera sample (k_gen made the main ray), the sampler past its arrays
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_rec_init(PtScene sc, PtPaths P, PtRec R, uint32_t n) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        Sampler sl;
        sl.index = P.sobol_index[p];
        sl.dim = 2;
        const uint32_t pk = P.pixel[p];
        sl.px = (int32_t)(pk & 0xffffu) + sc.film.sample_bounds[0];
        sl.py = (int32_t)(pk >> 16) + sc.film.sample_bounds[1];
        V2 u_lens = mk2(0.0f, 0.0f);
        if (sc.cam.lens_radius > 0.0f) u_lens = sl.get_2d(sc);
        const float2 pf = P.p_film[p];
        RayDiffs rdf;
        camera_differentials(sc, mk2(pf.x, pf.y), u_lens, f4_3(P.ray_o[p]), f4_3(P.ray_d[p]), rdf);
        R.diff[p] = make_float4(rdf.rx_o.x, rdf.rx_o.y, rdf.rx_o.z, 0.0f);
        R.diff[(size_t)R.n_paths + p] = make_float4(rdf.ry_o.x, rdf.ry_o.y, rdf.ry_o.z, 0.0f);
        R.diff[2 * (size_t)R.n_paths + p] = make_float4(rdf.rx_d.x, rdf.rx_d.y, rdf.rx_d.z, 0.0f);
        R.diff[3 * (size_t)R.n_paths + p] = make_float4(rdf.ry_d.x, rdf.ry_d.y, rdf.ry_d.z, 0.0f);
        // get_2d right after the camera sample jumps over the array dimensions (sobol.rs:118-141): dimension 5 + 2 * arrays
        P.state[p] = (5u + 2u * R.n_arrays) | (0u << 16) | (PT_ST_DIFF << 24);
        P.nee[p] = 0;                        // arrays handed out so far
    }
}
PT_DEV RayDiffs rec_load_diff(const PtRec& R, uint32_t p) {
    RayDiffs d;
    d.rx_o = f4_3(R.diff[p]); d.ry_o = f4_3(R.diff[(size_t)R.n_paths + p]);
    d.rx_d = f4_3(R.diff[2 * (size_t)R.n_paths + p]); d.ry_d = f4_3(R.diff[3 * (size_t)R.n_paths + p]);
    return d;
}
template <bool FULL>
PT_DEV void rec_enter_body(const PtScene& sc, const PtPaths& P, const PtQueues& Q, const PtQueues& Qn, const PtRec& R, PtCounters* cnt) {
    const uint32_t n = Q.counts[PT_Q_CUR];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool whitted = sc.integrator == PT_INTEGRATOR_WHITTED;
    uint32_t n_vert = 0;
    __shared__ uint32_t s_ws[PT_BLOCK / 64], s_wp[PT_BLOCK / 64], s_qbase[2];
    // block-uniform trips: the node's next-event entries are compacted into the shadow / probe work lists at the end of every trip, one
    // reservation per list and BLOCK (a reservation per wave would queue on the two counters: 88 M atomics/s each)
    for (uint32_t bbase = blockIdx.x * blockDim.x; bbase < n; bbase += gridDim.x * blockDim.x) {
        const uint32_t i = bbase + threadIdx.x;
        uint32_t ns = 0, np = 0, e_first = 0;             // this lane's live shadow / probe rays; its first entry
        if (i < n) {
        const uint32_t p = Q.cur[i];
        const V3 ro = f4_3(P.ray_o[p]), rd = f4_3(P.ray_d[p]);
        const int32_t rec = P.hit_rec[p];
        const uint32_t inst = sc.n_instances ? P.hit_inst[p] : 0u;
        uint32_t st = P.state[p];
        uint32_t dim = st & 0xffffu, depth = (st >> 16) & 0xffu, flags = (st >> 24) & 3u;
        const bool has_diff = (flags & PT_ST_DIFF) != 0;
        RayDiffs rdf;
        rdf.rx_o = rdf.ry_o = rdf.rx_d = rdf.ry_d = mk3(0.0f, 0.0f, 0.0f);
        if (has_diff) rdf = rec_load_diff(R, p);
        RecNode nd;
        rec_build<FULL>(sc, ro, rd, rec, inst, has_diff, rdf, nd);
        uint32_t outcome = PT_REC_OUT_FRAME;
        if (!nd.found) outcome = PT_REC_OUT_RETURN0;              // the lights' le(ray) sum: zero for area lights
        else if (!nd.has_bsdf) {
            if (whitted) outcome = PT_REC_OUT_RETURN0;
            else {                                                  // directlighting.rs:113-116: a plain ray through the surface, same depth
                const V3 no = offset_ray_origin(nd.s.p, nd.s.p_error, nd.s.n, rd);
                P.ray_o[p] = make_float4(no.x, no.y, no.z, PT_INF);
                flags &= ~PT_ST_DIFF;
                outcome = PT_REC_OUT_RETRACE;
            }
        }
        if (outcome == PT_REC_OUT_FRAME) {
            n_vert++;
            Sampler sm;
            sm.index = P.sobol_index[p];
            sm.dim = dim;
            const uint32_t pk = P.pixel[p];
            sm.px = (int32_t)(pk & 0xffffu) + sc.film.sample_bounds[0];
            sm.py = (int32_t)(pk >> 16) + sc.film.sample_bounds[1];
            V3 l = mk3(0.0f, 0.0f, 0.0f);
            const uint32_t e0 = p * R.epp;
            if (whitted) {
                // one light sample per light, no MIS, weighted by the shading normal from before the bump map (whitted.rs:52-78)
                for (uint32_t j = 0; j < sc.n_lights; j++) {
                    const V2 u = sm.get_2d(sc);
                    const PtLight& lt = sc.lights[j];
                    uint32_t nee = j << 8;
                    V3 A = mk3(0.0f, 0.0f, 0.0f);
                    V3 li, wi, lp, lperr, ln;
                    float lpdf;
                    if (light_sample_any<FULL>(sc, lt, nd.s.p, nd.s.p_error, nd.s.n, u, &li, &wi, &lpdf, &lp, &lperr, &ln) && !(lpdf <= 0.0f || is_black(li))) {
                        V3 f = gbsdf_f(nd.gb, nd.s.wo, wi, PT_BSDF_ALL);
                        if (!is_black(f)) {
                            V3 origin = offset_ray_origin(nd.s.p, nd.s.p_error, nd.s.n, lp - nd.s.p);
                            V3 target = offset_ray_origin(lp, lperr, ln, origin - lp);
                            V3 sd = target - origin;
                            R.sh_o[e0 + j] = make_float4(origin.x, origin.y, origin.z, 1.0f - PT_SHADOW_EPS);
                            R.sh_d[e0 + j] = make_float4(sd.x, sd.y, sd.z, 0.0f);
                            A = f * li * (abs_dot(wi, nd.n_before) / lpdf);
                            nee |= PT_NEE_SHADOW;
                        }
                    }
                    R.A[e0 + j] = make_float4(A.x, A.y, A.z, 1.0f);
                    R.flags[e0 + j] = nee;
                }
            } else {
                if (nd.s.light >= 0) l = light_L(sc.lights[nd.s.light], nd.s.n, nd.s.wo);          // isect.le(wo)
                if (sc.n_lights > 0) {
                    if (sc.direct_strategy == PT_DIRECT_ALL) {        // uniform_sample_all_lights (sample_lights.rs:24-78): n_j estimates per light from its two
                        // sample arrays while arrays are left, else one estimate from two plain 2-D draws.  Element k of an array of n, for pixel
                        // sample s, is drawn at sample number s n + k (base_sampler.rs:59-70), and EVERY array comes from dimensions (5, 6): the
                        // reference computes the array dimension once, before its loop over the arrays (sobol.rs:60-75, halton.rs:193-208;
                        // quirk Q23) -- u_light and u_scattering coincide.
                        uint32_t arr = P.nee[p], off = 0;
                        const uint64_t sample_num = (uint64_t)R.s0 + p / R.n_pix;
                        for (uint32_t j = 0; j < sc.n_lights; j++) {
                            const uint32_t n = sc.lights[j].n_samples;
                            if (arr + 2u <= R.n_arrays) {
                                arr += 2u;
                                for (uint32_t k = 0; k < n; k++) {
                                    const uint64_t idx = n == 1u ? sm.index : sampler_index(sc, sample_num * n + k, sm.px, sm.py);
                                    const V2 u = mk2(sample_dimension(sc, idx, 5u, sm.px, sm.py), sample_dimension(sc, idx, 6u, sm.px, sm.py));
                                    rec_estimate_direct<FULL>(sc, R, nd, j, u, u, e0 + off + k, (float)n);
                                }
                            } else {
                                arr = R.n_arrays;
                                const V2 u_light = sm.get_2d(sc);
                                const V2 u_scat = sm.get_2d(sc);
                                rec_estimate_direct<FULL>(sc, R, nd, j, u_light, u_scat, e0 + off, 1.0f);
                                for (uint32_t k = 1; k < n; k++) R.flags[e0 + off + k] = 0u;
                            }
                            off += n;
                        }
                        P.nee[p] = arr;
                    } else {                                          // uniform_sample_one_light without a distribution (sample_lights.rs:105-127)
                        const float fl = sm.get_1d(sc) * (float)sc.n_lights;
                        uint32_t light_num = fl > 0.0f ? (fl >= 4294967296.0f ? 0xffffffffu : (uint32_t)fl) : 0u;      // `as usize` saturates
                        if (light_num > sc.n_lights - 1u) light_num = sc.n_lights - 1u;
                        const float light_pdf = 1.0f / (float)sc.n_lights;
                        const V2 u_light = sm.get_2d(sc);
                        const V2 u_scat = sm.get_2d(sc);
                        rec_estimate_direct<FULL>(sc, R, nd, light_num, u_light, u_scat, e0);
                        float4 a = R.A[e0];
                        a.w = light_pdf;
                        R.A[e0] = a;
                    }
                }
            }
            dim = sm.dim;
            if (sc.sobol.kind == PT_SAMPLER_HALTON && dim > sc.sobol.h_n_dims) atomicOr(R.panic, 4u);       // a dimension past the table was asked for
            // the frame: what a later visit needs to rebuild this interaction, and the node's radiance so far
            *rec_frame(R, depth, 0, p) = make_float4(ro.x, ro.y, ro.z, __uint_as_float((uint32_t)rec));
            *rec_frame(R, depth, 1, p) = make_float4(rd.x, rd.y, rd.z, __uint_as_float(inst | (has_diff ? 0x80000000u : 0u)));
            *rec_frame(R, depth, 2, p) = make_float4(rdf.rx_o.x, rdf.rx_o.y, rdf.rx_o.z, 0.0f);
            *rec_frame(R, depth, 3, p) = make_float4(rdf.ry_o.x, rdf.ry_o.y, rdf.ry_o.z, __uint_as_float(nd.spec_mask));       // .w: which specular children this BSDF can have at all
            *rec_frame(R, depth, 4, p) = make_float4(rdf.rx_d.x, rdf.rx_d.y, rdf.rx_d.z, 0.0f);
            *rec_frame(R, depth, 5, p) = make_float4(rdf.ry_d.x, rdf.ry_d.y, rdf.ry_d.z, 0.0f);
            *rec_frame(R, depth, 6, p) = make_float4(l.x, l.y, l.z, 0.0f);
        }
        P.state[p] = (dim & 0xffffu) | (depth << 16) | ((flags | (outcome << 2)) << 24);
        if (outcome == PT_REC_OUT_FRAME) {
            e_first = p * R.epp;
            for (uint32_t j = 0; j < R.epp; j++) { const uint32_t fl = R.flags[e_first + j]; ns += (fl & PT_NEE_SHADOW) ? 1u : 0u; np += (fl & PT_NEE_PROBE) ? 1u : 0u; }
        }
        }
        // ---- this trip's entries with a live shadow / probe ray, in path and entry order, appended to the work lists of the next traversal launch
        uint32_t is = ns, ip = np;                       // inclusive scan over the wave
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = (uint32_t)__shfl_up((int)is, o, 64), b = (uint32_t)__shfl_up((int)ip, o, 64);
            if ((int)lane >= o) { is += a; ip += b; }
        }
        if (lane == 63u) { s_ws[wave] = is; s_wp[wave] = ip; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t ts = 0, tp = 0;
            for (uint32_t w = 0; w < PT_BLOCK / 64; w++) { ts += s_ws[w]; tp += s_wp[w]; }
            s_qbase[0] = ts ? atomicAdd(&Qn.counts[PT_Q_SHADOW], ts) : 0u;
            s_qbase[1] = tp ? atomicAdd(&Qn.counts[PT_Q_PROBE], tp) : 0u;
        }
        __syncthreads();
        if (ns | np) {
            uint32_t os = s_qbase[0] + (is - ns), op = s_qbase[1] + (ip - np);
            for (uint32_t w = 0; w < wave; w++) { os += s_ws[w]; op += s_wp[w]; }
            for (uint32_t j = 0; j < R.epp; j++) {
                const uint32_t e = e_first + j, fl = R.flags[e];
                if (fl & PT_NEE_SHADOW) {
                    Qn.shadow[os] = e;
                    if (Qn.shadow_key) { const float4 o4 = R.sh_o[e], d4 = R.sh_d[e]; Qn.shadow_key[os] = ray_sort_key(sc, mk3(o4.x, o4.y, o4.z), mk3(d4.x, d4.y, d4.z)); }
                    os++;
                }
                if (fl & PT_NEE_PROBE) Qn.probe[op++] = e;
            }
        }
        __syncthreads();                                 // s_ws / s_wp / s_qbase are rewritten by the next trip
    }
    __shared__ unsigned long long s_vert;
    if (threadIdx.x == 0) s_vert = 0;
    __syncthreads();
    if (n_vert) atomicAdd(&s_vert, (unsigned long long)n_vert);
    __syncthreads();
    if (threadIdx.x == 0 && s_vert) atomicAdd(&cnt->vertices, s_vert);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_rec_enter(PtScene sc, PtPaths P, PtQueues Q, PtQueues Qn, PtRec R, PtCounters* cnt) {
    rec_enter_body<true>(sc, P, Q, Qn, R, cnt);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_rec_enter_plain(PtScene sc, PtPaths P, PtQueues Q, PtQueues Qn, PtRec R, PtCounters* cnt) {
    rec_enter_body<false>(sc, P, Q, Qn, R, cnt);
}
// specular_reflect / specular_transmit at the frame `depth` (sampler.rs:37-143): true = a child ray was set up (cur ray, differentials, pending f / scale)
template <bool FULL>
PT_DEV bool rec_sample_child(const PtScene& sc, const PtPaths& P, const PtRec& R, uint32_t p, uint32_t depth, bool transmit, Sampler& sm, V3* pend_f, float* pend_scale, uint32_t* child_flags) {
    // A BSDF without a lobe that matches (REFLECTION | SPECULAR) / (TRANSMISSION | SPECULAR): the reference draws its 2-D sample
    // (sampler.rs:45, :92) and BSDF::sample_f returns None at `matching_comps == 0` (bsdf.rs:104-107) -- every Matte, Plastic, Metal or Substrate
    // node.  The draw is a dimension count here, and nothing of the frame needs rebuilding (rec_build was 2/3 of k_rec_next on a diffuse scene).
    if (!(__float_as_uint(rec_frame(R, depth, 3, p)->w) & (transmit ? 2u : 1u))) { sm.dim += 2; return false; }
    const float4 f0 = *rec_frame(R, depth, 0, p), f1 = *rec_frame(R, depth, 1, p);
    const V3 ro = f4_3(f0), rd = f4_3(f1);
    const int32_t rec = (int32_t)__float_as_uint(f0.w);
    const uint32_t iw = __float_as_uint(f1.w);
    const bool has_diff = (iw & 0x80000000u) != 0;
    RayDiffs rdf;
    rdf.rx_o = f4_3(*rec_frame(R, depth, 2, p)); rdf.ry_o = f4_3(*rec_frame(R, depth, 3, p));
    rdf.rx_d = f4_3(*rec_frame(R, depth, 4, p)); rdf.ry_d = f4_3(*rec_frame(R, depth, 5, p));
    RecNode nd;
    rec_build<FULL>(sc, ro, rd, rec, iw & 0x7fffffffu, has_diff, rdf, nd);
    const V2 u = sm.get_2d(sc);
    V3 f, wi;
    float pdf;
    uint32_t ty;
    if (!gbsdf_sample_f(nd.gb, nd.s.wo, u, (transmit ? PT_BSDF_TRANSMISSION : PT_BSDF_REFLECTION) | PT_BSDF_SPECULAR, &f, &wi, &pdf, &ty)) return false;
    const Surf& s = nd.s;
    const V3 wo = s.wo;
    V3 ns = s.sh_n;
    float wi_ns = abs_dot(wi, ns);
    float wo_ns = dot(wo, ns);
    if (!(pdf > 0.0f && !is_black(f) && wi_ns != 0.0f)) return false;
    const V3 no = offset_ray_origin(s.p, s.p_error, s.n, wi);
    P.ray_o[p] = make_float4(no.x, no.y, no.z, PT_INF);
    P.ray_d[p] = make_float4(wi.x, wi.y, wi.z, 0.0f);
    *child_flags = 0;
    if (has_diff) {
        const V3 rx_o = s.p + nd.th.dpdx, ry_o = s.p + nd.th.dpdy;
        V3 dndx = s.sh_dndu * nd.th.dudx + s.sh_dndv * nd.th.dvdx;
        V3 dndy = s.sh_dndu * nd.th.dudy + s.sh_dndv * nd.th.dvdy;
        V3 rx_d, ry_d;
        if (!transmit) {
            const V3 dwodx = -rdf.rx_d - wo, dwody = -rdf.ry_d - wo;
            const float d_dndx = dot(dwodx, ns) + dot(wo, dndx);
            const float d_dndy = dot(dwody, ns) + dot(wo, dndy);
            rx_d = wi - dwodx + 2.0f * (wo_ns * dndx + d_dndx * ns);
            ry_d = wi - dwody + 2.0f * (wo_ns * dndy + d_dndy * ns);
        } else {
            float eta = 1.0f / nd.bsdf_eta;
            if (dot(wo, ns) < 0.0f) {
                eta = 1.0f / eta;
                ns = -ns; dndx = -dndx; dndy = -dndy;
                wi_ns = abs_dot(wi, ns);
                wo_ns = dot(wo, ns);
            }
            const V3 dwodx = -rdf.rx_d - wo, dwody = -rdf.ry_d - wo;
            const float d_dndx = dot(dwodx, ns) + dot(wo, dndx);
            const float d_dndy = dot(dwody, ns) + dot(wo, dndy);
            const float mu = eta * wo_ns - wi_ns;
            const float dmudx = (eta - (eta * eta * wo_ns) / wi_ns) * d_dndx;
            const float dmudy = (eta - (eta * eta * wo_ns) / wi_ns) * d_dndy;
            rx_d = wi - eta * dwodx + (mu * dndx + dmudx * ns);
            ry_d = wi - eta * dwody + (mu * dndy + dmudy * ns);
        }
        R.diff[p] = make_float4(rx_o.x, rx_o.y, rx_o.z, 0.0f);
        R.diff[(size_t)R.n_paths + p] = make_float4(ry_o.x, ry_o.y, ry_o.z, 0.0f);
        R.diff[2 * (size_t)R.n_paths + p] = make_float4(rx_d.x, rx_d.y, rx_d.z, 0.0f);
        R.diff[3 * (size_t)R.n_paths + p] = make_float4(ry_d.x, ry_d.y, ry_d.z, 0.0f);
        *child_flags = PT_ST_DIFF;
    }
    *pend_f = f;
    *pend_scale = wi_ns / pdf;
    return true;
}
template <bool FULL>
PT_DEV void rec_next_body(const PtScene& sc, const PtPaths& P, const PtQueues& Q, const PtRec& R) {
    const uint32_t n = Q.counts[PT_Q_CUR];
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool whitted = sc.integrator == PT_INTEGRATOR_WHITTED;
    __shared__ uint32_t s_e[PT_SHADE_FLUSH][PT_BLOCK];       // paths of the iterations not yet queued (one reservation per PT_SHADE_FLUSH iterations, see shade_body)
    uint32_t n_batch = 0, want = 0;
    auto flush = [&]() {
        if (n_batch == 0) return;
        uint32_t tc = 0;
#pragma unroll
        for (uint32_t j = 0; j < PT_SHADE_FLUSH; j++)
            if (j < n_batch) tc += (uint32_t)__popcll(__ballot(((want >> j) & 1u) != 0));
        uint32_t bc = 0;
        if (lane == 0 && tc) bc = atomicAdd(&Q.counts[PT_Q_NEXT], tc);
        bc = (uint32_t)__builtin_amdgcn_readfirstlane((int)bc);
#pragma unroll
        for (uint32_t j = 0; j < PT_SHADE_FLUSH; j++)
            if (j < n_batch) {
                const bool w = ((want >> j) & 1u) != 0;
                const unsigned long long mc = __ballot(w);
                if (w) Q.next[bc + (uint32_t)__popcll(mc & below)] = s_e[j][threadIdx.x];
                bc += (uint32_t)__popcll(mc);
            }
        n_batch = 0; want = 0;
    };
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + lane;
        bool cont = false;
        uint32_t p = 0;
        if (i < n) {
            p = Q.cur[i];
            uint32_t st = P.state[p];
            uint32_t dim = st & 0xffffu, depth = (st >> 16) & 0xffu, flags = (st >> 24) & 3u;
            const uint32_t outcome = (st >> 26) & 3u;
            if (outcome == PT_REC_OUT_RETRACE) cont = true;
            else {
                Sampler sm;
                sm.index = P.sobol_index[p];
                sm.dim = dim;
                const uint32_t pk = P.pixel[p];
                sm.px = (int32_t)(pk & 0xffffu) + sc.film.sample_bounds[0];
                sm.py = (int32_t)(pk >> 16) + sc.film.sample_bounds[1];
                V3 v = mk3(0.0f, 0.0f, 0.0f);       // the value a finished node hands to its parent
                bool returning = outcome == PT_REC_OUT_RETURN0;
                int32_t d = returning ? (int32_t)depth - 1 : (int32_t)depth;
                uint32_t phase = 0;                  // of the frame d when not returning: 0 = sample reflect next, 1 = sample transmit next
                V3 l = mk3(0.0f, 0.0f, 0.0f);
                if (!returning) {
                    // the node's direct light, in the reference's order of additions
                    l = f4_3(*rec_frame(R, depth, 6, p));
                    const uint32_t e0 = p * R.epp;
                    if (whitted) {
                        for (uint32_t j = 0; j < sc.n_lights; j++) {
                            const uint32_t fl = R.flags[e0 + j];
                            V3 term = mk3(0.0f, 0.0f, 0.0f);
                            if ((fl & PT_NEE_SHADOW) && R.occ[e0 + j] == 0) term = f4_3(R.A[e0 + j]);
                            l = l + term;
                        }
                    } else if (sc.n_lights > 0) {
                        // per light: ld = sum of its estimates in order (each estimate = light half + BSDF half, estimate_direct's own sum),
                        // then l += ld / n ("all": sample_lights.rs:43-57; the single-draw fallback has divisor 1) or ld / pdf ("one")
                        V3 ld_all = mk3(0.0f, 0.0f, 0.0f);
                        const bool all = sc.direct_strategy == PT_DIRECT_ALL;
                        const uint32_t n_groups = all ? sc.n_lights : 1u;
                        uint32_t off = 0;
                        for (uint32_t j = 0; j < n_groups; j++) {
                            const uint32_t n = all ? sc.lights[j].n_samples : 1u;
                            const float divisor = R.A[e0 + off].w;
                            V3 ld = mk3(0.0f, 0.0f, 0.0f);
                            for (uint32_t k = 0; k < n; k++) {
                                const uint32_t e = e0 + off + k;
                                const uint32_t fl = R.flags[e];
                                V3 est = mk3(0.0f, 0.0f, 0.0f);
                                if ((fl & PT_NEE_SHADOW) && R.occ[e] == 0) est = est + f4_3(R.A[e]);
                                if (fl & PT_NEE_PROBE) {
                                    const int32_t best = R.prec[e];
                                    if (best >= 0 && (uint32_t)best == sc.lights[fl >> 8].tri_rec) est = est + f4_3(R.B[e]);
                                }
                                ld = ld + est;
                            }
                            if (all) ld_all = ld_all + ld / divisor;
                            else ld_all = ld / divisor;
                            off += n;
                        }
                        l = l + ld_all;
                    }
                }
                for (;;) {
                    if (returning) {
                        if (d < 0) { P.L[p] = make_float4(v.x, v.y, v.z, 0.0f); break; }
                        // back in frame d: add the finished child's term, move on to the next phase
                        const float4 fr2 = *rec_frame(R, (uint32_t)d, 2, p);
                        const uint32_t ph = __float_as_uint(fr2.w);              // 1 = the reflect child was pending, 2 = the transmit child
                        const float4 pf = *rec_frame(R, (uint32_t)d, 7, p);
                        const float scale = rec_frame(R, (uint32_t)d, 4, p)->w;
                        l = f4_3(*rec_frame(R, (uint32_t)d, 6, p));
                        l = l + (f4_3(pf) * v) * scale;
                        returning = false;
                        if (ph == 2u) { v = l; returning = true; d--; continue; }     // both children done: this node returns
                        phase = 1;
                    }
                    if ((uint32_t)d + 1u < (uint32_t)sc.max_depth) {
                        V3 pend_f;
                        float pend_scale;
                        uint32_t cflags;
                        bool child = false;
                        if (phase == 0) {
                            child = rec_sample_child<FULL>(sc, P, R, p, (uint32_t)d, false, sm, &pend_f, &pend_scale, &cflags);
                            if (!child) { l = l + mk3(0.0f, 0.0f, 0.0f); phase = 1; }
                        }
                        if (!child && phase == 1) {
                            child = rec_sample_child<FULL>(sc, P, R, p, (uint32_t)d, true, sm, &pend_f, &pend_scale, &cflags);
                            if (child) phase = 2;
                            else l = l + mk3(0.0f, 0.0f, 0.0f);
                        } else if (child) phase = 1;
                        if (child) {
                            // park this frame: radiance so far, which child is pending, its factor; the child becomes the current ray
                            *rec_frame(R, (uint32_t)d, 6, p) = make_float4(l.x, l.y, l.z, 0.0f);
                            *rec_frame(R, (uint32_t)d, 7, p) = make_float4(pend_f.x, pend_f.y, pend_f.z, 0.0f);
                            float4 fr2 = *rec_frame(R, (uint32_t)d, 2, p);
                            fr2.w = __uint_as_float(phase);
                            *rec_frame(R, (uint32_t)d, 2, p) = fr2;
                            float4 fr4 = *rec_frame(R, (uint32_t)d, 4, p);
                            fr4.w = pend_scale;
                            *rec_frame(R, (uint32_t)d, 4, p) = fr4;
                            depth = (uint32_t)d + 1u;
                            flags = cflags;
                            cont = true;
                            break;
                        }
                    }
                    v = l; returning = true; d--;      // no (further) child: the node returns its radiance
                }
                dim = sm.dim;
                if (sc.sobol.kind == PT_SAMPLER_HALTON && dim > sc.sobol.h_n_dims) atomicOr(R.panic, 4u);
                P.state[p] = (dim & 0xffffu) | (depth << 16) | (flags << 24);
            }
            if (outcome == PT_REC_OUT_RETRACE) P.state[p] = st & ~(3u << 26);
        }
        s_e[n_batch][threadIdx.x] = p;
        want |= (cont ? 1u : 0u) << n_batch;
        if (++n_batch == PT_SHADE_FLUSH) flush();
    }
    flush();
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_rec_next(PtScene sc, PtPaths P, PtQueues Q, PtRec R) { rec_next_body<true>(sc, P, Q, R); }
extern "C" __global__ void __launch_bounds__(PT_BLOCK, PT_WIDE_KERNEL_WAVES) k_rec_next_plain(PtScene sc, PtPaths P, PtQueues Q, PtRec R) { rec_next_body<false>(sc, P, Q, R); }
hipError_t ptk_rec_init(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtRec& R, uint32_t n) {
    hipLaunchKernelGGL(k_rec_init, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, R, n);
    return hipGetLastError();
}
hipError_t ptk_rec_enter(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, const PtQueues& Qn, const PtRec& R, PtCounters* cnt,
                         uint32_t lights_per_node) {
    (void)lights_per_node;
    if (sc.n_spheres || sc.n_instances || sc.textured) hipLaunchKernelGGL(k_rec_enter, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, Qn, R, cnt);
    else hipLaunchKernelGGL(k_rec_enter_plain, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, Qn, R, cnt);
    return hipGetLastError();
}
hipError_t ptk_rec_next(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, const PtRec& R) {
    if (sc.n_spheres || sc.n_instances || sc.textured) hipLaunchKernelGGL(k_rec_next, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, R);
    else hipLaunchKernelGGL(k_rec_next_plain, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, R);
    return hipGetLastError();
}

// ============================================================ launch wrappers (host side of this TU)
#define PT_LAUNCH_CHECK() hipGetLastError()

int ptk_trace_dist_blocks_per_cu() { return PT_TRACE_WIDE ? 1 : PT_TRACE_DIST_WAVES; }      // PT_TRACE_WIDE: one 1024-thread block per CU
int ptk_trace_wide() { return PT_TRACE_WIDE; }
int ptk_shade_prof_read(unsigned long long* out16) {       // 1 when the library is the -DPT_PROFILE_SHADE diagnostic build (reads and clears)
#ifdef PT_PROFILE_SHADE
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_shade_prof), sizeof(z)) != hipSuccess) return 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_shade_prof), z, sizeof(z));
    return 1;
#else
    (void)out16;
    return 0;
#endif
}
bool ptk_trace_has_far(const PtScene& sc) { return sc.dist_leaves && !sc.n_instances && !sc.n_spheres; }
hipError_t ptk_trace(hipStream_t st, int grid, int grid_dist, const PtScene& sc, const PtPaths& P, const PtQueues& Q, PtCounters* cnt, uint32_t* spill,
                     uint32_t spill_depth, uint32_t* err, int far) {
    static const bool force_sph = std::getenv("PBRTGPU_FORCE_SPH_TRACE") != nullptr;      // diagnosis: what the sphere-capable kernel costs a scene without spheres
    if (sc.dist_leaves && !sc.n_instances) grid = grid_dist;      // the pooled-leaf kernels fit three blocks per CU, the others four
    if (sc.n_instances) hipLaunchKernelGGL(k_trace_inst, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt, spill, spill_depth, err);
    else if ((sc.n_spheres || force_sph) && sc.dist_leaves) hipLaunchKernelGGL(k_trace_sph_dist, dim3(grid), dim3(PT_TBLOCK), 0, st, sc, P, Q, cnt, spill, spill_depth, err);
    else if (sc.n_spheres) hipLaunchKernelGGL(k_trace_sph, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt, spill, spill_depth, err);
    else if (sc.dist_leaves && far) hipLaunchKernelGGL(k_trace_far, dim3(grid), dim3(PT_TBLOCK), 0, st, sc, P, Q, cnt, spill, spill_depth, err);
    else if (sc.dist_leaves) hipLaunchKernelGGL(k_trace, dim3(grid), dim3(PT_TBLOCK), 0, st, sc, P, Q, cnt, spill, spill_depth, err);
    else hipLaunchKernelGGL(k_trace_seq, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt, spill, spill_depth, err);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_nee_resolve(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q) {
    hipLaunchKernelGGL(k_nee_resolve, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_trace_batch(hipStream_t st, int grid, const PtScene& sc, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out,
                           uint8_t* occ, int any_hit, uint32_t* ticket, PtCounters* cnt, uint32_t* spill, uint32_t spill_depth, uint32_t* err) {
    if (sc.n_instances) hipLaunchKernelGGL(k_trace_batch_inst, dim3(grid), dim3(PT_BLOCK), 0, st, sc, n, o, d, tmax, out, occ, any_hit, ticket, cnt, spill, spill_depth, err);
    else if (sc.n_spheres) hipLaunchKernelGGL(k_trace_batch_sph, dim3(grid), dim3(PT_BLOCK), 0, st, sc, n, o, d, tmax, out, occ, any_hit, ticket, cnt, spill, spill_depth, err);
    else hipLaunchKernelGGL(k_trace_batch, dim3(grid), dim3(PT_BLOCK), 0, st, sc, n, o, d, tmax, out, occ, any_hit, ticket, cnt, spill, spill_depth, err);
    return PT_LAUNCH_CHECK();
}
// ============================================================ AOIntegrator (integrators/ao.rs:49-110)
// After the camera rays have been traced: every path that hit something writes its n_s occlusion rays into a compacted batch
// (one slot allocation per wave), the batch goes through the any-hit traversal, and k_ao_resolve adds the unoccluded terms in
// sample order -- the reference's summation order, so the radiance is bit-identical.
//   counter[0] = hit paths so far; P.nee[path] = batch slot + 1 (0 = the camera ray escaped)
template <bool SPH, bool INST>
PT_DEV void ao_rays_body(const PtScene& sc, const PtPaths& P, uint32_t n_paths, float4* ao_o, float4* ao_d, float* ao_w, uint32_t* counter,
                         PtCounters* cnt) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_s = (uint32_t)sc.ao_samples;
    for (uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) - lane; base < n_paths; base += gridDim.x * blockDim.x) {
        const uint32_t p = base + lane;
        bool found = false;
        Surf s;
        V3 rd = mk3(0.0f, 0.0f, 1.0f);
        if (p < n_paths) {
            const V3 ro = f4_3(P.ray_o[p]);
            rd = f4_3(P.ray_d[p]);
            const int32_t rec = P.hit_rec[p];
            float thit;
            if constexpr (INST) found = rec >= 0 && make_surf_inst<SPH>(sc, ro, rd, (uint32_t)rec, P.hit_inst[p], s, &thit);
            else found = rec >= 0 && make_surf_any<SPH>(sc, ro, rd, (uint32_t)rec, s, &thit);
        }
        const uint64_t hits = __ballot(found);
        uint32_t slot0 = 0;
        if (lane == 0 && hits) {
            slot0 = atomicAdd(counter, (uint32_t)__popcll(hits));
            atomicAdd(&cnt->vertices, (unsigned long long)__popcll(hits));
        }
        slot0 = (uint32_t)__shfl((int)slot0, 0, 64);
        if (p < n_paths) P.nee[p] = found ? slot0 + (uint32_t)__popcll(hits & ((1ull << lane) - 1ull)) + 1u : 0u;
        if (!found) continue;
        const size_t r0 = (size_t)(slot0 + (uint32_t)__popcll(hits & ((1ull << lane) - 1ull))) * n_s;
        const V3 n = face_forward(s.n, -rd);               // the true geometry's frame, not the shading frame (ao.rs:73-77)
        const V3 ss = normalize(s.dpdu);
        const V3 tt = cross(s.n, ss);
        const uint32_t pk = P.pixel[p];
        const int32_t px = (int32_t)(pk & 0xffffu) + sc.film.sample_bounds[0], py = (int32_t)(pk >> 16) + sc.film.sample_bounds[1];
        const uint32_t cam_sample = (uint32_t)P.probe_rec[p];      // k_ao_tag stored the pixel-sample number here
        for (uint32_t k = 0; k < n_s; k++) {
            const uint64_t index = sampler_index(sc, (uint64_t)cam_sample * n_s + k, px, py);
            const V2 u = mk2(sample_dimension(sc, index, 5u, px, py), sample_dimension(sc, index, 6u, px, py));
            V3 wi;
            float pdf;
            if (sc.ao_cos_sample) { wi = cosine_sample_hemisphere(u); pdf = fabsf(wi.z) * PT_INV_PI; }
            else {
                const float z = u.x, r = sqrtf(fmaxf(0.0f, 1.0f - z * z)), phi = 2.0f * PT_PI * u.y;
                float sn, cs;
                pt_sincosf(phi, &sn, &cs);
                wi = mk3(r * cs, r * sn, z);
                pdf = PT_INV_PI * 0.5f;
            }
            wi = mk3(wi.x * ss.x + wi.y * tt.x + wi.z * n.x, wi.x * ss.y + wi.y * tt.y + wi.z * n.y, wi.x * ss.z + wi.y * tt.z + wi.z * n.z);
            const V3 o = offset_ray_origin(s.p, s.p_error, s.n, wi);
            const size_t r = r0 + k;
            ao_o[r] = make_float4(o.x, o.y, o.z, PT_INF);          // the record layout of a shadow work item (origin, t_max | direction)
            ao_d[r] = make_float4(wi.x, wi.y, wi.z, 0.0f);
            ao_w[r] = dot(wi, n) / (pdf * (float)n_s);
        }
    }
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_ao_rays(PtScene sc, PtPaths P, uint32_t n_paths, float4* ao_o, float4* ao_d, float* ao_w,
                                                                uint32_t* counter, PtCounters* cnt) {
    ao_rays_body<false, false>(sc, P, n_paths, ao_o, ao_d, ao_w, counter, cnt);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_ao_rays_sph(PtScene sc, PtPaths P, uint32_t n_paths, float4* ao_o, float4* ao_d, float* ao_w,
                                                                    uint32_t* counter, PtCounters* cnt) {
    ao_rays_body<true, false>(sc, P, n_paths, ao_o, ao_d, ao_w, counter, cnt);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_ao_rays_inst(PtScene sc, PtPaths P, uint32_t n_paths, float4* ao_o, float4* ao_d, float* ao_w,
                                                                     uint32_t* counter, PtCounters* cnt) {
    ao_rays_body<true, true>(sc, P, n_paths, ao_o, ao_d, ao_w, counter, cnt);
}
// path i of the pass took pixel-sample number s0 + i / n_pix (k_gen); the array slice of get_2d_array starts at n_s times that
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_ao_tag(PtPaths P, uint32_t n_pix, uint32_t n_paths, uint32_t s0) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_paths; i += gridDim.x * blockDim.x) P.probe_rec[i] = (int32_t)(s0 + i / n_pix);
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_ao_resolve(PtScene sc, PtPaths P, uint32_t n_paths, const float* ao_w, const uint8_t* occ) {
    const uint32_t n_s = (uint32_t)sc.ao_samples;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < n_paths; p += gridDim.x * blockDim.x) {
        float l = 0.0f;
        const uint32_t slot1 = P.nee[p];
        if (slot1) {
            const size_t r0 = (size_t)(slot1 - 1u) * n_s;
            for (uint32_t k = 0; k < n_s; k++)
                if (!occ[r0 + k]) l += ao_w[r0 + k];
        }
        P.L[p] = make_float4(l, l, l, 0.0f);
        P.nee[p] = 0;
    }
}
hipError_t ptk_ao_tag(hipStream_t st, int grid, const PtPaths& P, uint32_t n_pix, uint32_t n_paths, uint32_t s0) {
    hipLaunchKernelGGL(k_ao_tag, dim3(grid), dim3(PT_BLOCK), 0, st, P, n_pix, n_paths, s0);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_ao_rays(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, uint32_t n_paths, float4* ao_o, float4* ao_d, float* ao_w,
                       uint32_t* counter, PtCounters* cnt) {
    if (sc.n_instances) hipLaunchKernelGGL(k_ao_rays_inst, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, n_paths, ao_o, ao_d, ao_w, counter, cnt);
    else if (sc.n_spheres) hipLaunchKernelGGL(k_ao_rays_sph, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, n_paths, ao_o, ao_d, ao_w, counter, cnt);
    else hipLaunchKernelGGL(k_ao_rays, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, n_paths, ao_o, ao_d, ao_w, counter, cnt);
    return PT_LAUNCH_CHECK();
}
// The occlusion rays become the shadow work items of one wavefront traversal launch: item i is ray i (identity list), the count
// comes from the device-side hit counter -- no host round trip between the two traversals of a pass.
extern "C" __global__ void k_ao_queue(PtQueues Q, const uint32_t* counter, uint32_t n_s) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        Q.counts[PT_Q_CUR] = 0; Q.counts[PT_Q_SHADOW] = counter[0] * n_s; Q.counts[PT_Q_PROBE] = 0; Q.counts[PT_Q_NEE] = 0;
        for (uint32_t k = 0; k < 8u; k++) Q.counts[PT_Q_SEG_TICKET0 + 32u * k] = 0;
    }
}
extern "C" __global__ void __launch_bounds__(PT_BLOCK) k_iota(uint32_t* out, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = i;
}
hipError_t ptk_ao_queue(hipStream_t st, const PtQueues& Q, const uint32_t* counter, uint32_t n_s) {
    hipLaunchKernelGGL(k_ao_queue, dim3(1), dim3(64), 0, st, Q, counter, n_s);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_iota(hipStream_t st, int grid, uint32_t* out, uint32_t n) {
    hipLaunchKernelGGL(k_iota, dim3(grid), dim3(PT_BLOCK), 0, st, out, n);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_ao_resolve(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, uint32_t n_paths, const float* ao_w, const uint8_t* occ) {
    hipLaunchKernelGGL(k_ao_resolve, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, n_paths, ao_w, occ);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_gen(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, const uint32_t* pixels, uint32_t n_pix,
                   uint32_t s0, uint32_t n_samples, PtCounters* cnt) {
    hipLaunchKernelGGL(k_gen, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, pixels, n_pix, s0, n_samples, cnt);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_prep(hipStream_t st, const PtQueues& Q, int mode) {
    hipLaunchKernelGGL(k_prep, dim3(1), dim3(64), 0, st, Q, mode);
    return PT_LAUNCH_CHECK();
}
int ptk_nee_split_default() { return PT_NEE_SPLIT_DEFAULT; }
hipError_t ptk_shade(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, PtCounters* cnt, int nsplit, int local_sort) {
    if (sc.general_materials && local_sort && !sc.n_instances && sc.textured && !sc.n_spheres && P.tex_res) {
        // ... with textured materials: the texture half over the same list (it passes over the untextured entries), then the lobe-list kernel that
        // takes a textured hit's parameters from what the texture half left
        hipLaunchKernelGGL(k_sort_local, dim3(grid * 4), dim3(PT_BLOCK), 0, st, sc, P, Q);
        PtQueues Ql = Q;
        Ql.cur = Q.sorted;
        hipLaunchKernelGGL(k_tex_resolve_all, dim3(4096), dim3(PT_BLOCK), 0, st, sc, P, Ql);
        hipLaunchKernelGGL(k_shade_all_res, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Ql, cnt);
        return PT_LAUNCH_CHECK();
    }
    if (sc.general_materials && local_sort && !sc.n_instances && !sc.textured) {
        // the shade queue ordered by material inside runs of 4 096 entries, one lobe-list kernel over all of it (see k_sort_local)
        hipLaunchKernelGGL(k_sort_local, dim3(grid * 4), dim3(PT_BLOCK), 0, st, sc, P, Q);
        PtQueues Ql = Q;
        Ql.cur = Q.sorted;
        if (sc.n_spheres) hipLaunchKernelGGL(k_shade_all_sph, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Ql, cnt);
        else hipLaunchKernelGGL(k_shade_all, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Ql, cnt);
        return PT_LAUNCH_CHECK();
    }
    static const int unsorted = [] { const char* e = std::getenv("PBRTGPU_SHADE_UNSORTED"); return e ? std::atoi(e) : 0; }();
    if (sc.general_materials && unsorted && !sc.n_instances && (!sc.n_spheres || sc.textured)) {
        if (sc.textured) hipLaunchKernelGGL(k_shade_all_tex, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt);
        else hipLaunchKernelGGL(k_shade_all, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt);
        return PT_LAUNCH_CHECK();
    }
    // nsplit (the context's PBRTGPU_NEE_SPLIT): which kernel families run a vertex as two kernels (next-event estimation, then everything else):
    // 1 the Matte kernels, 2 the lobe-list kernels, 4 the sphere-capable kernels (Matte and lobe-list), 8 the textured segment's second half
    const int per_cu = grid / 2;          // grid = two blocks per CU (the whole-vertex kernels' occupancy)
#define PT_RUN(K) hipLaunchKernelGGL(K, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt)
#define PT_RUN2(K, WN, WC) do { hipLaunchKernelGGL(K##_nee, dim3(per_cu * (WN)), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt); \
                                hipLaunchKernelGGL(K##_cont, dim3(per_cu * (WC)), dim3(PT_BLOCK), 0, st, sc, P, Q, cnt); } while (0)
    auto matte_sorted = [&]() {
        if (sc.n_spheres) { if (nsplit & 4) PT_RUN2(k_shade_matte_sorted_sph, PT_NEE_WAVES, PT_CONT_SPH_WAVES); else PT_RUN(k_shade_matte_sorted_sph); }
        else { if (nsplit & 1) PT_RUN2(k_shade_matte_sorted, PT_NEE_WAVES, PT_CONT_WAVES); else PT_RUN(k_shade_matte_sorted); }
    };
    auto general = [&]() {
        if (sc.n_spheres) { if (nsplit & 4) PT_RUN2(k_shade_general_sph, PT_NEE_GEN_WAVES, PT_CONT_SPH_WAVES); else PT_RUN(k_shade_general_sph); }
        else { if (nsplit & 2) PT_RUN2(k_shade_general, PT_NEE_GEN_WAVES, PT_CONT_GEN_WAVES); else PT_RUN(k_shade_general); }
    };
    if (sc.general_materials) {
        hipLaunchKernelGGL(k_sort_count, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(PT_SORT_BINS), 0, st, Q);
        hipLaunchKernelGGL(k_sort_scatter, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q);
        if (sc.n_instances) {
            if (!sc.textured && !sc.n_spheres) PT_RUN(k_shade_general_inst_plain); else PT_RUN(k_shade_general_inst);
        } else if (sc.textured) {       // three segments: Matte | other constant materials | textured materials
            matte_sorted();
            general();
            static const int split = [] { const char* e = std::getenv("PBRTGPU_TEX_SPLIT"); return e ? std::atoi(e) : 1; }();
            if (split && P.tex_res) {
                if (sc.n_spheres) {
                    hipLaunchKernelGGL(k_tex_resolve_sph, dim3(4096), dim3(PT_BLOCK), 0, st, sc, P, Q);
                    if (nsplit & 4) PT_RUN2(k_shade_general_res_sph, PT_NEE_GEN_WAVES, PT_CONT_SPH_WAVES); else PT_RUN(k_shade_general_res_sph);
                } else {
                    hipLaunchKernelGGL(k_tex_resolve, dim3(4096), dim3(PT_BLOCK), 0, st, sc, P, Q);
                    if (nsplit & 8) PT_RUN2(k_shade_general_res, PT_NEE_GEN_WAVES, PT_CONT_GEN_WAVES); else PT_RUN(k_shade_general_res);
                }
            } else {
                PT_RUN(k_shade_general_tex);
            }
        } else {
            matte_sorted();
            general();
        }
    } else {
        if (nsplit & 1) PT_RUN2(k_shade, PT_NEE_WAVES, PT_CONT_WAVES); else PT_RUN(k_shade);
    }
#undef PT_RUN
#undef PT_RUN2
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_cont_keys(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const uint32_t* list, uint32_t n, uint32_t* keys) {
    hipLaunchKernelGGL(k_cont_keys, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, list, n, keys);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_film(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const uint32_t* pixels, uint32_t n_pix, uint32_t n_samples,
                    float4* own, float4* spill, float* radiance_out, uint32_t s0, uint32_t spp_total) {
    hipLaunchKernelGGL(k_film, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, pixels, n_pix, n_samples, own, spill, radiance_out, s0, spp_total);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_film_xyzw(hipStream_t st, const float4* own, const float4* spill, float4* xyzw, uint32_t n) {
    hipLaunchKernelGGL(k_film_xyzw, dim3(1024), dim3(PT_BLOCK), 0, st, own, spill, xyzw, n);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_film_add(hipStream_t st, float4* xyzw, const float4* other, uint32_t n) {
    hipLaunchKernelGGL(k_film_add, dim3(1024), dim3(PT_BLOCK), 0, st, xyzw, other, n);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_film_rgb(hipStream_t st, const float4* xyzw, float* rgb, uint32_t n, float scale) {
    hipLaunchKernelGGL(k_film_rgb, dim3(1024), dim3(PT_BLOCK), 0, st, xyzw, rgb, n, scale);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_light_grid(hipStream_t st, const PtScene& sc, float* data, uint32_t n_vox, const uint32_t* vox_list) {
    const uint64_t jobs = (uint64_t)n_vox * sc.grid.n_lights;
    if (jobs == 0) return hipSuccess;
    const uint64_t batches = (jobs + 63) / 64;          // kGridBatch pairs each
    const uint32_t grid = (uint32_t)(batches < 2048u ? batches : 2048u);
    if (sc.n_spheres) hipLaunchKernelGGL(k_light_grid_sph, dim3(grid), dim3(128), 0, st, sc, data, n_vox, vox_list);
    else hipLaunchKernelGGL(k_light_grid, dim3(grid), dim3(128), 0, st, sc, data, n_vox, vox_list);
    hipLaunchKernelGGL(k_light_grid_cdf, dim3((n_vox + 63) / 64), dim3(64), 0, st, sc, data, n_vox);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_grid_mark(hipStream_t st, int grid, const PtScene& sc, const PtPaths& P, const PtQueues& Q, int32_t* row_of, uint32_t* todo, uint32_t* todo_count) {
    if (sc.n_instances) hipLaunchKernelGGL(k_grid_mark_inst, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, row_of, todo, todo_count);
    else if (sc.n_spheres) hipLaunchKernelGGL(k_grid_mark_sph, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, row_of, todo, todo_count);
    else hipLaunchKernelGGL(k_grid_mark, dim3(grid), dim3(PT_BLOCK), 0, st, sc, P, Q, row_of, todo, todo_count);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_grid_assign(hipStream_t st, int32_t* row_of, const uint32_t* todo, uint32_t n, uint32_t row0, uint32_t* todo_count) {
    hipLaunchKernelGGL(k_grid_assign, dim3((n + 255u) / 256u), dim3(256), 0, st, row_of, todo, n, row0, todo_count);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_camera_rays(hipStream_t st, const PtScene& sc, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, float* o, float* d,
                           float* pf) {
    hipLaunchKernelGGL(k_camera_rays, dim3((n + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, st, sc, n, pixel_xy, sample_index, o, d, pf);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_bsdf_eval(hipStream_t st, const PtScene& sc, uint32_t material, uint32_t n, const float* wo, const float* wi, uint32_t flags, float* f,
                         float* pdf) {
    hipLaunchKernelGGL(k_bsdf_eval, dim3((n + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, st, sc, material, n, wo, wi, flags, f, pdf);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_bsdf_sample(hipStream_t st, const PtScene& sc, uint32_t material, uint32_t n, const float* wo, const float* u, uint32_t flags, float* f,
                           float* wi, float* pdf, uint32_t* type) {
    hipLaunchKernelGGL(k_bsdf_sample, dim3((n + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, st, sc, material, n, wo, u, flags, f, wi, pdf, type);
    return PT_LAUNCH_CHECK();
}
hipError_t ptk_sobol_samples(hipStream_t st, const PtScene& sc, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim,
                             float* out) {
    hipLaunchKernelGGL(k_sobol_samples, dim3((n + PT_BLOCK - 1) / PT_BLOCK), dim3(PT_BLOCK), 0, st, sc, n, pixel_xy, sample_index, dim, out);
    return PT_LAUNCH_CHECK();
}
