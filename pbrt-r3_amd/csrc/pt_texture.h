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
__device__ __noinline__ void map2d(const pt_texture& t, const TexHit& si, V2* st, V2* dstdx, V2* dstdy) {
    if (t.mapping == PT_MAPPING_SPHERICAL || t.mapping == PT_MAPPING_CYLINDRICAL) {
        const bool sph = t.mapping == PT_MAPPING_SPHERICAL;
        const float delta = 0.1f;
        V3 p0 = si.p, p1 = si.p + delta * si.dpdx, p2 = si.p + delta * si.dpdy;
        V2 s0 = sph ? sphere_st(t.world_to_texture, p0) : cylinder_st(t.world_to_texture, p0);
        V2 sx = sph ? sphere_st(t.world_to_texture, p1) : cylinder_st(t.world_to_texture, p1);
        V2 sy = sph ? sphere_st(t.world_to_texture, p2) : cylinder_st(t.world_to_texture, p2);
        V2 dx = mk2((sx.x - s0.x) * (1.0f / delta), (sx.y - s0.y) * (1.0f / delta));
        V2 dy = mk2((sy.x - s0.x) * (1.0f / delta), (sy.y - s0.y) * (1.0f / delta));
        dx.y = fix_wrap(dx.y); dy.y = fix_wrap(dy.y);
        *st = s0; *dstdx = dx; *dstdy = dy;
    } else if (t.mapping == PT_MAPPING_PLANAR) {
        V3 vs = ld3(t.v1), vt = ld3(t.v2);
        *st = mk2(t.du + dot(si.p, vs), t.dv + dot(si.p, vt));
        *dstdx = mk2(dot(si.dpdx, vs), dot(si.dpdx, vt));
        *dstdy = mk2(dot(si.dpdy, vs), dot(si.dpdy, vt));
    } else {
        *dstdx = mk2(t.su * si.dudx, t.sv * si.dvdx);
        *dstdy = mk2(t.su * si.dudy, t.sv * si.dvdy);
        *st = mk2(t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv);
    }
}
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
__device__ __noinline__ float noise3(float x, float y, float z) {
    int32_t ixi = f2i_sat(floorf(x)), iyi = f2i_sat(floorf(y)), izi = f2i_sat(floorf(z));
    float dx = x - (float)ixi, dy = y - (float)iyi, dz = z - (float)izi;
    uint32_t ix = (uint32_t)ixi & 255u, iy = (uint32_t)iyi & 255u, iz = (uint32_t)izi & 255u;
    float w000 = noise_grad(ix, iy, iz, dx, dy, dz);
    float w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz);
    float w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f);
    float w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f);
    float w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    float x00 = lerpf(wx, w000, w100), x10 = lerpf(wx, w010, w110), x01 = lerpf(wx, w001, w101), x11 = lerpf(wx, w011, w111);
    float y0 = lerpf(wy, x00, x10), y1 = lerpf(wy, x01, x11);
    return lerpf(wz, y0, y1);
}
PT_DEV float smooth_step(float mn, float mx, float value) {
    float v = clampf((value - mn) / (mx - mn), 0.0f, 1.0f);
    return v * v * (-2.0f * v + 3.0f);
}
__device__ __noinline__ float fbm_turb(V3 p, V3 dpdx, V3 dpdy, float omega, uint32_t max_octaves, bool turb) {     // noise.rs:98-150
    float len2 = fmaxf(length_squared(dpdx), length_squared(dpdy));
    float n = clampf(-1.0f - 0.5f * pt_log2f(len2), 0.0f, (float)max_octaves);
    float nf = floorf(n);
    uint32_t n_int = nf != nf ? 0u : (nf <= 0.0f ? 0u : (uint32_t)nf);
    float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint32_t i = 0; i < n_int; i++) {
        V3 lp = lambda * p;
        float nz = noise3(lp.x, lp.y, lp.z);
        sum += o * (turb ? fabsf(nz) : nz);
        lambda *= 1.99f;
        o *= omega;
    }
    float n_partial = n - (float)n_int;
    V3 lp = lambda * p;
    sum += o * smooth_step(0.3f, 0.7f, n_partial) * noise3(lp.x, lp.y, lp.z);
    if (turb)
        for (uint32_t i = 0; i < n_int; i++) { sum += o * 0.2f; o *= omega; }
    return sum;
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
__device__ __noinline__ V3 mip_lookup(const MipRef& m, V2 st, float width) {       // lookup (:620-637)
    float max_level = (float)(m.im->n_levels - 1);
    float lvl = max_level + pt_log2f(fmaxf(width, 1e-8f));
    if (lvl < 0.0f) return mip_triangle(m, 0, st);
    if (lvl >= max_level) return mip_texel(m, m.im->n_levels - 1, 0, 0);
    uint32_t il = (uint32_t)floorf(lvl);
    float delta = clampf(lvl - (float)il, 0.0f, 1.0f);
    V3 a = mip_triangle(m, il, st), b = mip_triangle(m, il + 1, st);
    return a * (1.0f - delta) + b * delta;
}
__device__ __noinline__ V3 mip_ewa(const MipRef& m, uint32_t l, V2 st0, V2 d0, V2 d1) {     // make_ewa_params + ewa_core (:141-217)
    if (l >= m.im->n_levels) return mip_texel(m, m.im->n_levels - 1, 0, 0);
    int32_t wi, hi;
    mip_dims(*m.im, l, &wi, &hi);
    float ww = (float)wi, hh = (float)hi;
    V2 st = mk2(st0.x * ww - 0.5f, st0.y * hh - 0.5f);
    V2 dst0 = mk2(d0.x * ww, d0.y * hh), dst1 = mk2(d1.x * ww, d1.y * hh);
    float a = dst0.y * dst0.y + dst1.y * dst1.y + 1.0f;
    float b = -2.0f * (dst0.x * dst0.y + dst1.x * dst1.y);
    float c = dst0.x * dst0.x + dst1.x * dst1.x + 1.0f;
    float inv_f = 1.0f / (a * c - b * b * 0.25f);
    a = a * inv_f; b = b * inv_f; c = c * inv_f;
    float det = -b * b + 4.0f * a * c;
    float inv_det = 1.0f / det;
    float u_sqrt = sqrtf(det * c), v_sqrt = sqrtf(det * a);
    int32_t s0 = f2i_sat(ceilf(st.x - 2.0f * inv_det * u_sqrt)), s1 = f2i_sat(floorf(st.x + 2.0f * inv_det * u_sqrt));
    int32_t t0 = f2i_sat(ceilf(st.y - 2.0f * inv_det * v_sqrt)), t1 = f2i_sat(floorf(st.y + 2.0f * inv_det * v_sqrt));
    V3 sum = mk3(0.0f, 0.0f, 0.0f);
    float sum_wts = 0.0f;
    for (long long it = t0; it <= t1; it++) {
        float tt = (float)(int32_t)it - st.y;
        for (long long is = s0; is <= s1; is++) {
            float ss = (float)(int32_t)is - st.x;
            float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
            if (r2 < 1.0f) {
                float fi = r2 * (float)PT_EWA_LUT_SIZE;
                uint32_t index = fi != fi ? 0u : (fi <= 0.0f ? 0u : (uint32_t)fi);
                if (index > PT_EWA_LUT_SIZE - 1) index = PT_EWA_LUT_SIZE - 1;
                float weight = pt_ewa_lut[index];
                V3 v = mip_texel(m, l, (int32_t)is, (int32_t)it);
                sum.x += v.x * weight; sum.y += v.y * weight; sum.z += v.z * weight;
                sum_wts += weight;
            }
        }
    }
    float inv_sum = 1.0f / sum_wts;
    return mk3(sum.x * inv_sum, sum.y * inv_sum, sum.z * inv_sum);
}
__device__ __noinline__ V3 mip_lookup_delta(const MipRef& m, V2 st, V2 dst0, V2 dst1, bool trilinear, float max_aniso) {     // :819-852, :913-946
    if (trilinear) {
        float width = fmaxf(fmaxf(fabsf(dst0.x), fabsf(dst0.y)), fmaxf(fabsf(dst1.x), fabsf(dst1.y)));
        return mip_lookup(m, st, width);
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) { V2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
    float major_length = sqrtf(dst0.x * dst0.x + dst0.y * dst0.y);
    float minor_length = sqrtf(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minor_length * max_aniso < major_length && minor_length > 0.0f) {
        float scale = major_length / (minor_length * max_aniso);
        dst1 = mk2(dst1.x * scale, dst1.y * scale);
        minor_length *= scale;
    }
    if (minor_length <= 0.0f) return mip_lookup(m, st, 0.0f);
    float lod = fmaxf(0.0f, (float)m.im->n_levels - 1.0f + pt_log2f(minor_length));
    uint32_t ilod = (uint32_t)floorf(lod);
    float t = lod - (float)ilod;
    V3 e0 = mip_ewa(m, ilod, st, dst0, dst1), e1 = mip_ewa(m, ilod + 1, st, dst0, dst1);
    return mk3(lerpf(t, e0.x, e1.x), lerpf(t, e0.y, e1.y), lerpf(t, e0.z, e1.z));
}

// One node, its children already evaluated (c0, c1, c2 = tex1, tex2, amount).
__device__ __noinline__ V3 tex_node(const pt_texture& t, const TexHit& si, V3 c0, V3 c1, V3 c2, const PtImage* images) {
    switch (t.type) {
        case PT_TEX_IMAGEMAP: {                                   // imagemap.rs:57-70
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            MipRef m;
            m.im = images + t.image; m.swrap = t.swrap; m.twrap = t.twrap;
            return mip_lookup_delta(m, st, dx, dy, t.trilinear != 0, t.max_anisotropy);
        }
        case PT_TEX_SCALE: return c0 * c1;
        case PT_TEX_MIX: { float amt = c2.x; return c0 * (1.0f - amt) + c1 * amt; }
        case PT_TEX_CHECKERBOARD_2D: {
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            const bool first = even_sum(f2i_sat(floorf(st.x)), f2i_sat(floorf(st.y)));
            if (t.aa_none) return first ? c0 : c1;
            float ds = fmaxf(fabsf(dx.x), fabsf(dy.x)), dt = fmaxf(fabsf(dx.y), fabsf(dy.y));
            float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1 = st.y + dt;
            if (floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1)) return first ? c0 : c1;
            float sint = (bump_int(s1) - bump_int(s0)) / (2.0f * ds);
            float tint = (bump_int(t1) - bump_int(t0)) / (2.0f * dt);
            float area2 = sint + tint - 2.0f * sint * tint;
            if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
            return c0 * (1.0f - area2) + c1 * area2;
        }
        case PT_TEX_CHECKERBOARD_3D: {
            V3 p = tex_point(t.world_to_texture, si.p);          // IdentityMapping3D: the matrix is tex2world (checkerboard.rs:159)
            int32_t s = (int32_t)((uint32_t)f2i_sat(floorf(p.x)) + (uint32_t)f2i_sat(floorf(p.y)) + (uint32_t)f2i_sat(floorf(p.z)));
            return (s % 2) == 0 ? c0 : c1;
        }
        case PT_TEX_UV: {
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            return mk3(st.x - floorf(st.x), st.y - floorf(st.y), 0.0f);
        }
        case PT_TEX_DOTS: {                                       // dots.rs:27-43
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            float s_cell = floorf(st.x + 0.5f), t_cell = floorf(st.y + 0.5f);
            if (noise3(s_cell + 0.5f, t_cell + 0.5f, 0.0f) > 0.0f) {
                float radius = 0.35f;
                float max_shift = 0.5f - radius;
                float s_center = s_cell + max_shift * noise3(s_cell + 1.5f, t_cell + 2.8f, 0.0f);
                float t_center = t_cell + max_shift * noise3(s_cell + 4.5f, t_cell + 9.8f, 0.0f);
                float ddx = st.x - s_center, ddy = st.y - t_center;
                if (ddx * ddx + ddy * ddy < radius * radius) return c1;
            }
            return c0;
        }
        case PT_TEX_FBM: case PT_TEX_WRINKLED: case PT_TEX_WINDY: case PT_TEX_MARBLE: {
            // IdentityMapping3D::map (mapping3d.rs:24-31): both differentials come back as dpdx (as written)
            V3 p = tex_point(t.world_to_texture, si.p);
            V3 dpdx = tex_vector(t.world_to_texture, si.dpdx), dpdy = dpdx;
            if (t.type == PT_TEX_WINDY) {                         // windy.rs:14-19
                float wind_strength = fbm_turb(0.1f * p, 0.1f * dpdx, 0.1f * dpdy, 0.5f, 3u, false);
                float wave_height = fbm_turb(p, dpdx, dpdy, 0.5f, 6u, false);
                float v = fabsf(wind_strength) * wave_height;
                return mk3(v, v, v);
            }
            if (t.type != PT_TEX_MARBLE) {
                float v = fbm_turb(p, dpdx, dpdy, t.omega, (uint32_t)t.octaves, t.type == PT_TEX_WRINKLED);
                return mk3(v, v, v);
            }
            // marble.rs:35-62
            p = t.scale * p;
            float marble = p.y + t.variation * fbm_turb(p, t.scale * dpdx, t.scale * dpdy, t.omega, (uint32_t)t.octaves, false);
            float sn, cs;
            pt_sincosf(marble, &sn, &cs);
            float tt = 0.5f + 0.5f * sn;
            const float nseg = 6.0f;
            float fl = floorf(tt * nseg);
            const bool second = !(fl != fl) && fl >= 1.0f;          // first = min(1, floor(t * nseg) as usize)
            tt = tt * nseg - (second ? 1.0f : 0.0f);
            const V3 ca = mk3(0.58f, 0.58f, 0.6f), cb = mk3(0.5f, 0.5f, 0.5f), cc = mk3(0.6f, 0.59f, 0.58f);
            V3 k0 = ca, k1 = ca, k2 = second ? cb : ca, k3 = second ? cc : cb;     // C[first .. first+3]
            V3 s0 = k0 * (1.0f - tt) + k1 * tt, s1 = k1 * (1.0f - tt) + k2 * tt, s2 = k2 * (1.0f - tt) + k3 * tt;
            s0 = s0 * (1.0f - tt) + s1 * tt; s1 = s1 * (1.0f - tt) + s2 * tt;
            return (s0 * (1.0f - tt) + s1 * tt) * 1.5f;
        }
        case PT_TEX_BILERP: {
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            float a = (1.0f - st.x) * (1.0f - st.y), b = (1.0f - st.x) * st.y, c = st.x * (1.0f - st.y), d = st.x * st.y;
            return ld3(t.value[0]) * a + ld3(t.value[1]) * b + ld3(t.value[2]) * c + ld3(t.value[3]) * d;
        }
        default: return ld3(t.value[0]);
    }
}
// prog[0] = n; prog[1..n]: node index | child slots (positions in this list, or PT_TEX_CHILD_CONST) << 16 / 20 / 24
__device__ __noinline__ V3 tex_eval(const pt_texture* textures, const uint32_t* prog, const TexHit& si, const PtImage* images) {
    V3 val[PT_TEX_PROG_MAX];
    const uint32_t n = prog[0];
    V3 last = mk3(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < n && i < PT_TEX_PROG_MAX; i++) {
        const uint32_t e = prog[1 + i];
        const pt_texture& t = textures[e & 0xffffu];
        V3 c[3];
        for (int k = 0; k < 3; k++) {
            const uint32_t slot = (e >> (16 + 4 * k)) & 15u;
            c[k] = slot == PT_TEX_CHILD_CONST ? ld3(t.value[k]) : val[slot];
        }
        last = tex_node(t, si, c[0], c[1], c[2], images);
        val[i] = last;
    }
    return last;
}
