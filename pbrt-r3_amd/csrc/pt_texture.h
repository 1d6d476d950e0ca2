// pt_texture.h -- procedural textures at a hit (src/textures/, src/core/texture/mapping2d.rs, mapping3d.rs) and the ray
// differentials they filter with (surface_interaction.rs:221-282).  Only a camera ray carries differentials in the path
// integrator (path.rs replaces the ray by a plain Ray after the first vertex), so they are rebuilt from the camera sample
// in the shade kernel instead of travelling with the path.
//
// A texture graph is evaluated bottom-up from a small program the host writes per (material, parameter): the nodes the
// parameter's root needs, children before parents (pt_scene_desc.textures is in definition order, so that is index order).
#pragma once
#include "pt_device.h"
#include "pt_device_math.h"
#include "../../include/pbrtgpu.h"
#include "../../include/pbrtgpu_noise_perm.h"
#include "../../include/pbrtgpu_ewa_lut.h"


struct TexHit {                     // what SurfaceInteraction offers a texture
    V3 p, dpdx, dpdy;
    V2 uv;
    float dudx, dvdx, dudy, dvdy;
};
struct RayDiffs { V3 rx_o, ry_o, rx_d, ry_d; };

PT_DEV bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float* x0, float* x1) {   // matrix4x4.rs:9-20
    float det = a00 * a11 - a01 * a10;
    if (fabsf(det) < 1e-10f) return false;
    float r0 = (a11 * b0 - a01 * b1) / det;
    float r1 = (a00 * b1 - a10 * b0) / det;
    if (r0 != r0 || r1 != r1) return false;
    *x0 = r0; *x1 = r1;
    return true;
}
// SurfaceInteraction::compute_differentials (surface_interaction.rs:221-282); has == false: everything zero
PT_DEV void compute_differentials(TexHit& t, V3 p, V3 n, V3 dpdu, V3 dpdv, bool has, const RayDiffs& rd) {
    t.dpdx = mk3(0.0f, 0.0f, 0.0f); t.dpdy = t.dpdx;
    t.dudx = t.dvdx = t.dudy = t.dvdy = 0.0f;
    if (!has) return;
    float d = dot(n, p);
    float tx = -(dot(n, rd.rx_o) - d) / dot(n, rd.rx_d);
    if (!isfinite(tx)) return;
    V3 px = rd.rx_o + tx * rd.rx_d;
    float ty = -(dot(n, rd.ry_o) - d) / dot(n, rd.ry_d);
    if (!isfinite(ty)) return;
    V3 py = rd.ry_o + ty * rd.ry_d;
    t.dpdx = px - p;
    t.dpdy = py - p;
    int d0, d1;
    if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
    else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    float a00 = comp(dpdu, d0), a01 = comp(dpdv, d0), a10 = comp(dpdu, d1), a11 = comp(dpdv, d1);
    float bx0 = comp(px, d0) - comp(p, d0), bx1 = comp(px, d1) - comp(p, d1);
    float by0 = comp(py, d0) - comp(p, d0), by1 = comp(py, d1) - comp(p, d1);
    if (!solve_2x2(a00, a01, a10, a11, bx0, bx1, &t.dudx, &t.dvdx)) { t.dudx = 0.0f; t.dvdx = 0.0f; }
    if (!solve_2x2(a00, a01, a10, a11, by0, by1, &t.dudy, &t.dvdy)) { t.dudy = 0.0f; t.dvdy = 0.0f; }
}

// matrix4x4.rs:284-297 with the homogeneous divide (texture transforms may be anything the CTM was)
PT_DEV V3 tex_point(const float* m, V3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.0f) return mk3(xp, yp, zp);
    return mk3(xp / wp, yp / wp, zp / wp);
}
PT_DEV V2 sphere_st(const float* w2t, V3 p) {               // SphericalMapping2D::sphere (mapping2d.rs:66-72)
    V3 vec = normalize(tex_point(w2t, p) - mk3(0.0f, 0.0f, 0.0f));
    float theta = pt_acosf(clampf(vec.z, -1.0f, 1.0f));
    float ph = pt_atan2f(vec.y, vec.x);
    float phi = ph < 0.0f ? ph + 2.0f * PT_PI : ph;
    return mk2(theta * PT_INV_PI, phi * (PT_INV_PI * 0.5f));
}
PT_DEV V2 cylinder_st(const float* w2t, V3 p) {             // CylindricalMapping2D::cylinder (mapping2d.rs:116-120)
    V3 vec = normalize(tex_point(w2t, p) - mk3(0.0f, 0.0f, 0.0f));
    return mk2((PT_PI + pt_atan2f(vec.y, vec.x)) * (PT_INV_PI * 0.5f), vec.z);
}
PT_DEV float fix_wrap(float d) { return d > 0.5f ? 1.0f - d : (d < -0.5f ? -(d + 1.0f) : d); }
PT_DEV float bump_int(float x) { return floorf(x / 2.0f) + 2.0f * fmaxf(x / 2.0f - floorf(x / 2.0f) - 0.5f, 0.0f); }
PT_DEV int32_t f2i_sat(float f) {           // Rust `as i32`: saturating, NaN -> 0
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 0x7fffffff;
    if (f <= -2147483648.0f) return (int32_t)0x80000000u;
    return (int32_t)f;
}
PT_DEV bool even_sum(int32_t a, int32_t b) { return ((int32_t)((uint32_t)a + (uint32_t)b) % 2) == 0; }

// ---- Perlin noise, fBm, turbulence (core/texture/noise.rs)
__device__ const unsigned char pt_noise_perm[512] = {PT_NOISE_PERM_256, PT_NOISE_PERM_256};
PT_DEV float noise_grad(uint32_t x, uint32_t y, uint32_t z, float dx, float dy, float dz) {
    uint32_t h = pt_noise_perm[pt_noise_perm[pt_noise_perm[x] + y] + z] & 15u;
    float u = (h < 8u || h == 12u || h == 13u) ? dx : dy;
    float v = (h < 4u || h == 12u || h == 13u) ? dy : dz;
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}
PT_DEV float noise_weight(float t) { float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
PT_DEV float smooth_step(float mn, float mx, float value) {
    float v = clampf((value - mn) / (mx - mn), 0.0f, 1.0f);
    return v * v * (-2.0f * v + 3.0f);
}
PT_DEV V3 tex_vector(const float* m, V3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// ---- MIPMap (core/texture/mipmap.rs) over a pyramid in HBM; float images read as RGB with equal channels
__device__ const float pt_ewa_lut[PT_EWA_LUT_SIZE] = {PT_EWA_LUT_VALUES};
struct MipRef { const PtImage* im; int swrap, twrap; };
PT_DEV void mip_dims(const PtImage& im, uint32_t l, int32_t* w, int32_t* h) {
    uint32_t ww = im.width >> l, hh = im.height >> l;
    *w = (int32_t)(ww ? ww : 1u); *h = (int32_t)(hh ? hh : 1u);
}
PT_DEV V3 mip_texel(const MipRef& m, uint32_t l, int32_t s, int32_t t) {           // texel_static (:503-544)
    int32_t w, h;
    mip_dims(*m.im, l, &w, &h);
    if (m.swrap == PT_WRAP_REPEAT) s &= w - 1;
    else if (m.swrap == PT_WRAP_CLAMP) s = s < 0 ? 0 : (s > w - 1 ? w - 1 : s);
    else if (s < 0 || w <= s) return mk3(0.0f, 0.0f, 0.0f);
    if (m.twrap == PT_WRAP_REPEAT) t &= h - 1;
    else if (m.twrap == PT_WRAP_CLAMP) t = t < 0 ? 0 : (t > h - 1 ? h - 1 : t);
    else if (t < 0 || h <= t) return mk3(0.0f, 0.0f, 0.0f);
    const float* d = m.im->texels + m.im->level_off[l];
    const size_t i = (size_t)t * (size_t)w + (size_t)s;
    if (m.im->channels == 1) { float v = d[i]; return mk3(v, v, v); }
    return mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
}
PT_DEV V3 mip_triangle(const MipRef& m, uint32_t l, V2 st) {                       // triangle (:711-765)
    if (l > m.im->n_levels - 1) l = m.im->n_levels - 1;
    int32_t w, h;
    mip_dims(*m.im, l, &w, &h);
    float s = st.x * (float)w - 0.5f, t = st.y * (float)h - 0.5f;
    int32_t s0 = f2i_sat(floorf(s)), t0 = f2i_sat(floorf(t));
    float ds = s - (float)s0, dt = t - (float)t0;
    return mip_texel(m, l, s0, t0) * ((1.0f - ds) * (1.0f - dt)) + mip_texel(m, l, s0, t0 + 1) * ((1.0f - ds) * dt) +
           mip_texel(m, l, s0 + 1, t0) * (ds * (1.0f - dt)) + mip_texel(m, l, s0 + 1, t0 + 1) * (ds * dt);
}

#ifndef PT_TEX_EXP
#define PT_TEX_EXP 0
#endif

#define PT_TEX_FN __device__ __noinline__
#define PT_TEXN(x) x
#define PT_TEX_VBUF_PARAM
#define PT_TEX_VBUF_DECL V3 val[PT_TEX_PROG_MAX];
#define PT_TEX_VBUF_GET(s_) val[s_]
#define PT_TEX_VBUF_SET(i_, v_) val[i_] = (v_)
#include "pt_texture_calls.inc"
#undef PT_TEX_FN
#undef PT_TEXN
#undef PT_TEX_VBUF_PARAM
#undef PT_TEX_VBUF_DECL
#undef PT_TEX_VBUF_GET
#undef PT_TEX_VBUF_SET
#define PT_TEX_FN __device__ __forceinline__
#define PT_TEXN(x) x##_inl
#define PT_TEX_VBUF_PARAM , float* vbuf
#define PT_TEX_VBUF_DECL
#define PT_TEX_VBUF_GET(s_) mk3(vbuf[(3u * (s_)) * PT_BLOCK], vbuf[(3u * (s_) + 1u) * PT_BLOCK], vbuf[(3u * (s_) + 2u) * PT_BLOCK])
#define PT_TEX_VBUF_SET(i_, v_) do { const V3 v__ = (v_); vbuf[(3u * (i_)) * PT_BLOCK] = v__.x; vbuf[(3u * (i_) + 1u) * PT_BLOCK] = v__.y; vbuf[(3u * (i_) + 2u) * PT_BLOCK] = v__.z; } while (0)
#include "pt_texture_calls.inc"
#undef PT_TEX_FN
#undef PT_TEXN
#undef PT_TEX_VBUF_PARAM
#undef PT_TEX_VBUF_DECL
#undef PT_TEX_VBUF_GET
#undef PT_TEX_VBUF_SET
