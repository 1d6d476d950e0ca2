// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_texture.hpp: ray differentials at a hit and the procedural textures.
//   follows src/core/interaction/surface_interaction.rs:212-282 (compute_differentials),
//           src/core/transform/matrix4x4.rs:9-20 (solve_linear_system_2x2),
//           src/core/texture/mapping2d.rs, mapping3d.rs,
//           src/textures/{constant,scale,mix,checkerboard,uv,bilerp}.rs
#pragma once
#include "orc_accel.hpp"
#include "../include/pbrtgpu.h"
#include "../include/pbrtgpu_noise_perm.h"
#include "../include/pbrtgpu_ewa_lut.h"
#include <vector>

namespace orc {

// RayDifferential's offset rays (core/geometry/ray/ray_differential.rs).  Only a camera ray has them: path.rs replaces the
// ray by isect.spawn_ray(..).into() after the first vertex, which clears has_differentials.
struct RayDiff {
    bool has = false;
    V3 rx_o, ry_o, rx_d, ry_d;
};

struct TexHit {                 // the SurfaceInteraction fields textures read
    V3 p, dpdx, dpdy;
    V2 uv;
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
};

inline bool solve_2x2(const Float a[2][2], const Float b[2], Float* x0, Float* x1) {
    Float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (std::fabs(det) < 1e-10f) return false;
    Float r0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det;
    Float r1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
    if (std::isnan(r0) || std::isnan(r1)) return false;
    *x0 = r0; *x1 = r1;
    return true;
}
// SurfaceInteraction::compute_differentials (surface_interaction.rs:221-282)
inline TexHit compute_differentials(const SurfHit& si, const RayDiff& rd) {
    TexHit t;
    t.p = si.p; t.uv = si.uv;
    t.dpdx = V3(0, 0, 0); t.dpdy = V3(0, 0, 0);
    if (!rd.has) return t;
    V3 p = si.p, n = si.n;
    Float d = dot(n, p);
    Float tx = -(dot(n, rd.rx_o) - d) / dot(n, rd.rx_d);
    if (!std::isfinite(tx)) return t;
    V3 px = rd.rx_o + tx * rd.rx_d;
    Float ty = -(dot(n, rd.ry_o) - d) / dot(n, rd.ry_d);
    if (!std::isfinite(ty)) return t;
    V3 py = rd.ry_o + ty * rd.ry_d;
    t.dpdx = px - p;
    t.dpdy = py - p;
    int d0, d1;
    if (std::fabs(n.x) > std::fabs(n.y) && std::fabs(n.x) > std::fabs(n.z)) { d0 = 1; d1 = 2; }
    else if (std::fabs(n.y) > std::fabs(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    const Float a[2][2] = {{si.dpdu[d0], si.dpdv[d0]}, {si.dpdu[d1], si.dpdv[d1]}};
    const Float bx[2] = {px[d0] - p[d0], px[d1] - p[d1]};
    const Float by[2] = {py[d0] - p[d0], py[d1] - p[d1]};
    if (!solve_2x2(a, bx, &t.dudx, &t.dvdx)) { t.dudx = 0.0f; t.dvdx = 0.0f; }
    if (!solve_2x2(a, by, &t.dudy, &t.dvdy)) { t.dudy = 0.0f; t.dvdy = 0.0f; }
    return t;
}

// ---- TextureMapping2D::map (mapping2d.rs)
inline V2 sphere_st(const Mat4& w2t, V3 p) {
    V3 vec = normalize(w2t.transform_point(p) - V3(0.0f, 0.0f, 0.0f));
    Float theta = std::acos(clampf(vec.z, -1.0f, 1.0f));          // spherical_theta / spherical_phi (misc.rs:96-104)
    Float ph = std::atan2(vec.y, vec.x);
    Float phi = ph < 0.0f ? ph + 2.0f * kPi : ph;
    return V2(theta * kInvPi, phi * (kInvPi * 0.5f));
}
inline V2 cylinder_st(const Mat4& w2t, V3 p) {
    V3 vec = normalize(w2t.transform_point(p) - V3(0.0f, 0.0f, 0.0f));
    return V2((kPi + std::atan2(vec.y, vec.x)) * (kInvPi * 0.5f), vec.z);
}
inline void fix_wrap(V2* d) {
    if (d->y > 0.5f) d->y = 1.0f - d->y;
    else if (d->y < -0.5f) d->y = -(d->y + 1.0f);
}
inline void map2d(const pt_texture& t, const TexHit& si, V2* st, V2* dstdx, V2* dstdy) {
    Mat4 w2t;
    std::memcpy(w2t.m, t.world_to_texture, sizeof(w2t.m));
    switch (t.mapping) {
        case PT_MAPPING_SPHERICAL:
        case PT_MAPPING_CYLINDRICAL: {
            const bool sph = t.mapping == PT_MAPPING_SPHERICAL;
            const Float delta = 0.1f;
            V2 s0 = sph ? sphere_st(w2t, si.p) : cylinder_st(w2t, si.p);
            V2 sx = sph ? sphere_st(w2t, si.p + delta * si.dpdx) : cylinder_st(w2t, si.p + delta * si.dpdx);
            V2 dx = (sx - s0) * (1.0f / delta);
            V2 sy = sph ? sphere_st(w2t, si.p + delta * si.dpdy) : cylinder_st(w2t, si.p + delta * si.dpdy);
            V2 dy = (sy - s0) * (1.0f / delta);
            fix_wrap(&dx); fix_wrap(&dy);
            *st = s0; *dstdx = dx; *dstdy = dy;
            return;
        }
        case PT_MAPPING_PLANAR: {
            V3 vs(t.v1[0], t.v1[1], t.v1[2]), vt(t.v2[0], t.v2[1], t.v2[2]);
            *st = V2(t.du + dot(si.p, vs), t.dv + dot(si.p, vt));
            *dstdx = V2(dot(si.dpdx, vs), dot(si.dpdx, vt));
            *dstdy = V2(dot(si.dpdy, vs), dot(si.dpdy, vt));
            return;
        }
        default:
            *dstdx = V2(t.su * si.dudx, t.sv * si.dvdx);
            *dstdy = V2(t.su * si.dudy, t.sv * si.dvdy);
            *st = V2(t.su * si.uv.x + t.du, t.sv * si.uv.y + t.dv);
            return;
    }
}
// Rust's `f as i32`: saturating, NaN -> 0
inline int32_t f2i(Float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
inline Float bump_int(Float x) {                      // checkerboard.rs:36-39
    return std::floor(x / 2.0f) + 2.0f * fmax_(x / 2.0f - std::floor(x / 2.0f) - 0.5f, 0.0f);
}

// ---- Perlin noise, fBm, turbulence (core/texture/noise.rs)
static const uint8_t kNoisePerm[512] = {PT_NOISE_PERM_256, PT_NOISE_PERM_256};
inline Float noise_grad(uint32_t x, uint32_t y, uint32_t z, Float dx, Float dy, Float dz) {     // noise.rs:10-19
    uint32_t h = kNoisePerm[kNoisePerm[kNoisePerm[x] + y] + z] & 15u;
    Float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    Float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
inline Float noise_weight(Float t) { Float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
inline Float noise3(Float x, Float y, Float z) {                                              // noise.rs:64-96
    int32_t ixi = f2i(std::floor(x)), iyi = f2i(std::floor(y)), izi = f2i(std::floor(z));
    Float dx = x - (Float)ixi, dy = y - (Float)iyi, dz = z - (Float)izi;
    uint32_t ix = (uint32_t)ixi & 255u, iy = (uint32_t)iyi & 255u, iz = (uint32_t)izi & 255u;
    Float w000 = noise_grad(ix, iy, iz, dx, dy, dz);
    Float w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    Float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz);
    Float w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    Float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f);
    Float w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    Float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f);
    Float w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    Float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    Float x00 = lerpf(wx, w000, w100), x10 = lerpf(wx, w010, w110), x01 = lerpf(wx, w001, w101), x11 = lerpf(wx, w011, w111);
    Float y0 = lerpf(wy, x00, x10), y1 = lerpf(wy, x01, x11);
    return lerpf(wz, y0, y1);
}
inline Float smooth_step(Float mn, Float mx, Float value) {
    Float v = clampf((value - mn) / (mx - mn), 0.0f, 1.0f);
    return v * v * (-2.0f * v + 3.0f);
}
// fbm / turbulence (noise.rs:98-150); `turb` adds |noise| and the tail of 0.2-weighted octaves
inline Float fbm_turb(V3 p, V3 dpdx, V3 dpdy, Float omega, uint32_t max_octaves, bool turb) {
    Float len2 = fmax_(length_squared(dpdx), length_squared(dpdy));
    Float n = clampf(-1.0f - 0.5f * std::log2(len2), 0.0f, (Float)max_octaves);
    Float nf = std::floor(n);
    size_t n_int = nf != nf ? 0 : (nf <= 0.0f ? 0 : (size_t)nf);          // `as usize`: saturating
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (size_t i = 0; i < n_int; i++) {
        V3 lp = lambda * p;
        Float nz = noise3(lp.x, lp.y, lp.z);
        sum += o * (turb ? std::fabs(nz) : nz);
        lambda *= 1.99f;
        o *= omega;
    }
    Float n_partial = n - (Float)n_int;
    V3 lp = lambda * p;
    sum += o * smooth_step(0.3f, 0.7f, n_partial) * noise3(lp.x, lp.y, lp.z);
    if (turb)
        for (size_t i = 0; i < n_int; i++) { sum += o * 0.2f; o *= omega; }
    return sum;
}
// IdentityMapping3D::map (mapping3d.rs:24-31): both differentials come back as dpdx (as written)
inline void map3d(const pt_texture& t, const TexHit& si, V3* p, V3* dpdx, V3* dpdy) {
    Mat4 m;
    std::memcpy(m.m, t.world_to_texture, sizeof(m.m));
    *p = m.transform_point(si.p);
    *dpdx = m.transform_vector(si.dpdx);
    *dpdy = m.transform_vector(si.dpdx);
}

// ---- MIPMap (core/texture/mipmap.rs) over a caller-built pyramid
struct MipImage {
    uint32_t channels = 3;
    int swrap = 0, twrap = 0;
    std::vector<int32_t> w, h;
    std::vector<const float*> level;
    void init(const pt_image& im) {
        channels = im.channels;
        w.clear(); h.clear(); level.clear();
        int32_t cw = (int32_t)im.width, ch = (int32_t)im.height;
        const float* p = im.texels;
        for (uint32_t l = 0; l < im.n_levels; l++) {          // make_pyramid (:406-441): halve what is still > 1
            w.push_back(cw); h.push_back(ch); level.push_back(p);
            p += (size_t)cw * ch * channels;
            if (cw > 1) cw /= 2;
            if (ch > 1) ch /= 2;
        }
    }
    size_t levels() const { return w.size(); }
    RGB fetch(size_t l, size_t i) const {                     // MIPMapTexel::lookup (:83-96)
        const float* d = level[l];
        return channels == 1 ? RGB(d[i]) : RGB(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    }
    // texel_static (:503-544)
    RGB texel(size_t l, int32_t s, int32_t t) const {
        int32_t ww = w[l], hh = h[l];
        if (swrap == PT_WRAP_REPEAT) s &= ww - 1;
        else if (swrap == PT_WRAP_CLAMP) s = s < 0 ? 0 : (s > ww - 1 ? ww - 1 : s);
        else if (s < 0 || ww <= s) return RGB();
        if (twrap == PT_WRAP_REPEAT) t &= hh - 1;
        else if (twrap == PT_WRAP_CLAMP) t = t < 0 ? 0 : (t > hh - 1 ? hh - 1 : t);
        else if (t < 0 || hh <= t) return RGB();
        return fetch(l, (size_t)t * ww + s);
    }
    // triangle (:711-765).  With a "black" wrap mode on either axis the reference indexes the level unwrapped (and panics when
    // that leaves the image); texel() semantics are used there instead.
    RGB triangle(size_t l, V2 st) const {
        if (l > levels() - 1) l = levels() - 1;
        int32_t ww = w[l], hh = h[l];
        Float s = st.x * (Float)ww - 0.5f, t = st.y * (Float)hh - 0.5f;
        int32_t s0 = f2i(std::floor(s)), t0 = f2i(std::floor(t));
        Float ds = s - (Float)s0, dt = t - (Float)t0;
        int32_t s1 = s0 + 1, t1 = t0 + 1;
        return texel(l, s0, t0) * ((1.0f - ds) * (1.0f - dt)) + texel(l, s0, t1) * ((1.0f - ds) * dt) + texel(l, s1, t0) * (ds * (1.0f - dt)) +
               texel(l, s1, t1) * (ds * dt);
    }
    // lookup (:620-637): trilinear
    RGB lookup(V2 st, Float width) const {
        Float max_level = (Float)(levels() - 1);
        Float lvl = max_level + std::log2(fmax_(width, 1e-8f));
        if (lvl < 0.0f) return triangle(0, st);
        if (lvl >= max_level) return texel(levels() - 1, 0, 0);
        size_t il = (size_t)std::floor(lvl);
        Float delta = clampf(lvl - (Float)il, 0.0f, 1.0f);
        RGB a = triangle(il, st), b = triangle(il + 1, st);
        return a * (1.0f - delta) + b * delta;
    }
    // ewa_rgb / ewa_float (:785-817, :876-911) = make_ewa_params + ewa_core (:141-217)
    RGB ewa(size_t l, V2 st0, V2 d0, V2 d1) const {
        static const Float lut[PT_EWA_LUT_SIZE] = {PT_EWA_LUT_VALUES};
        if (l >= levels()) return texel(levels() - 1, 0, 0);
        Float ww = (Float)w[l], hh = (Float)h[l];
        V2 st(st0.x * ww - 0.5f, st0.y * hh - 0.5f);
        V2 dst0(d0.x * ww, d0.y * hh), dst1(d1.x * ww, d1.y * hh);
        Float a = dst0.y * dst0.y + dst1.y * dst1.y + 1.0f;
        Float b = -2.0f * (dst0.x * dst0.y + dst1.x * dst1.y);
        Float c = dst0.x * dst0.x + dst1.x * dst1.x + 1.0f;
        Float inv_f = 1.0f / (a * c - b * b * 0.25f);
        a = a * inv_f; b = b * inv_f; c = c * inv_f;
        Float det = -b * b + 4.0f * a * c;
        Float inv_det = 1.0f / det;
        Float u_sqrt = std::sqrt(det * c), v_sqrt = std::sqrt(det * a);
        int32_t s0 = f2i(std::ceil(st.x - 2.0f * inv_det * u_sqrt)), s1 = f2i(std::floor(st.x + 2.0f * inv_det * u_sqrt));
        int32_t t0 = f2i(std::ceil(st.y - 2.0f * inv_det * v_sqrt)), t1 = f2i(std::floor(st.y + 2.0f * inv_det * v_sqrt));
        RGB sum;
        Float sum_wts = 0.0f;
        for (int64_t it = t0; it <= t1; it++) {
            Float tt = (Float)(int32_t)it - st.y;
            for (int64_t is = s0; is <= s1; is++) {
                Float ss = (Float)(int32_t)is - st.x;
                Float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
                if (r2 < 1.0f) {
                    Float fi = r2 * (Float)PT_EWA_LUT_SIZE;
                    size_t index = fi != fi ? 0 : (fi <= 0.0f ? 0 : (size_t)fi);
                    if (index > PT_EWA_LUT_SIZE - 1) index = PT_EWA_LUT_SIZE - 1;
                    Float weight = lut[index];
                    RGB v = texel(l, (int32_t)is, (int32_t)it);
                    for (int k = 0; k < 3; k++) sum.c[k] += v.c[k] * weight;
                    sum_wts += weight;
                }
            }
        }
        Float inv_sum = 1.0f / sum_wts;
        return RGB(sum.c[0] * inv_sum, sum.c[1] * inv_sum, sum.c[2] * inv_sum);
    }
    // lookup_delta_rgb / lookup_delta_float (:819-852, :913-946)
    RGB lookup_delta(V2 st, V2 dst0, V2 dst1, bool trilinear, Float max_aniso) const {
        if (trilinear) {
            Float width = fmax_(fmax_(std::fabs(dst0.x), std::fabs(dst0.y)), fmax_(std::fabs(dst1.x), std::fabs(dst1.y)));
            return lookup(st, width);
        }
        if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) std::swap(dst0, dst1);
        Float major_length = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
        Float minor_length = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
        if (minor_length * max_aniso < major_length && minor_length > 0.0f) {
            Float scale = major_length / (minor_length * max_aniso);
            dst1 = dst1 * scale;
            minor_length *= scale;
        }
        if (minor_length <= 0.0f) return lookup(st, 0.0f);
        Float lod = fmax_(0.0f, (Float)levels() - 1.0f + std::log2(minor_length));
        size_t ilod = (size_t)std::floor(lod);
        Float t = lod - (Float)ilod;
        RGB e0 = ewa(ilod, st, dst0, dst1), e1 = ewa(ilod + 1, st, dst0, dst1);
        return RGB(lerpf(t, e0.c[0], e1.c[0]), lerpf(t, e0.c[1], e1.c[1]), lerpf(t, e0.c[2], e1.c[2]));
    }
};
struct TexCtx { const pt_texture* tex; const MipImage* images; };

// Texture<T>::evaluate over the flattened texture array; float textures are RGB with equal channels.
inline RGB texture_eval(const pt_texture* tex, int32_t index, const TexHit& si, const MipImage* images = nullptr);
inline RGB texture_child_i(const pt_texture* tex, const pt_texture& t, int k, const TexHit& si, const MipImage* images) {
    if (t.tex[k] >= 0) return texture_eval(tex, t.tex[k], si, images);
    return RGB(t.value[k][0], t.value[k][1], t.value[k][2]);
}
inline RGB texture_eval(const pt_texture* tex, int32_t index, const TexHit& si, const MipImage* images) {
    const pt_texture& t = tex[index];
    auto texture_child = [&](const pt_texture* tx, const pt_texture& tt, int k, const TexHit& s2) { return texture_child_i(tx, tt, k, s2, images); };
    switch (t.type) {
        case PT_TEX_SCALE: {
            RGB a = texture_child(tex, t, 0, si), b = texture_child(tex, t, 1, si);
            return a * b;
        }
        case PT_TEX_MIX: {
            RGB t1 = texture_child(tex, t, 0, si), t2 = texture_child(tex, t, 1, si);
            Float amt = texture_child(tex, t, 2, si).c[0];
            return t1 * (1.0f - amt) + t2 * amt;
        }
        case PT_TEX_CHECKERBOARD_2D: {
            V2 st, dstdx, dstdy;
            map2d(t, si, &st, &dstdx, &dstdy);
            auto point = [&]() {
                return ((int32_t)((uint32_t)f2i(std::floor(st.x)) + (uint32_t)f2i(std::floor(st.y))) % 2 == 0) ? texture_child(tex, t, 0, si) : texture_child(tex, t, 1, si);
            };
            if (t.aa_none) return point();
            Float ds = fmax_(std::fabs(dstdx.x), std::fabs(dstdy.x));
            Float dt = fmax_(std::fabs(dstdx.y), std::fabs(dstdy.y));
            Float s0 = st.x - ds, s1 = st.x + ds, t0 = st.y - dt, t1 = st.y + dt;
            if (std::floor(s0) == std::floor(s1) && std::floor(t0) == std::floor(t1)) return point();
            Float sint = (bump_int(s1) - bump_int(s0)) / (2.0f * ds);
            Float tint = (bump_int(t1) - bump_int(t0)) / (2.0f * dt);
            Float area2 = sint + tint - 2.0f * sint * tint;
            if (ds > 1.0f || dt > 1.0f) area2 = 0.5f;
            return texture_child(tex, t, 0, si) * (1.0f - area2) + texture_child(tex, t, 1, si) * area2;
        }
        case PT_TEX_CHECKERBOARD_3D: {
            // IdentityMapping3D (mapping3d.rs:16-32): the matrix handed over is tex2world itself (checkerboard.rs:159)
            Mat4 m;
            std::memcpy(m.m, t.world_to_texture, sizeof(m.m));
            V3 p = m.transform_point(si.p);
            if ((int32_t)((uint32_t)f2i(std::floor(p.x)) + (uint32_t)f2i(std::floor(p.y)) + (uint32_t)f2i(std::floor(p.z))) % 2 == 0) return texture_child(tex, t, 0, si);
            return texture_child(tex, t, 1, si);
        }
        case PT_TEX_UV: {
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            return RGB(st.x - std::floor(st.x), st.y - std::floor(st.y), 0.0f);
        }
        case PT_TEX_IMAGEMAP: {                                   // imagemap.rs:57-70
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            MipImage im = images[t.image];
            im.swrap = t.swrap; im.twrap = t.twrap;
            return im.lookup_delta(st, dx, dy, t.trilinear != 0, t.max_anisotropy);
        }
        case PT_TEX_DOTS: {                                       // dots.rs:27-43
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            Float s_cell = std::floor(st.x + 0.5f), t_cell = std::floor(st.y + 0.5f);
            if (noise3(s_cell + 0.5f, t_cell + 0.5f, 0.0f) > 0.0f) {
                Float radius = 0.35f;
                Float max_shift = 0.5f - radius;
                Float s_center = s_cell + max_shift * noise3(s_cell + 1.5f, t_cell + 2.8f, 0.0f);
                Float t_center = t_cell + max_shift * noise3(s_cell + 4.5f, t_cell + 9.8f, 0.0f);
                V2 dst = st - V2(s_center, t_center);
                if (dst.x * dst.x + dst.y * dst.y < radius * radius) return texture_child(tex, t, 1, si);
            }
            return texture_child(tex, t, 0, si);
        }
        case PT_TEX_FBM: case PT_TEX_WRINKLED: {
            V3 p, dpdx, dpdy;
            map3d(t, si, &p, &dpdx, &dpdy);
            return RGB(fbm_turb(p, dpdx, dpdy, t.omega, (uint32_t)t.octaves, t.type == PT_TEX_WRINKLED));
        }
        case PT_TEX_WINDY: {                                      // windy.rs:14-19
            V3 p, dpdx, dpdy;
            map3d(t, si, &p, &dpdx, &dpdy);
            Float wind_strength = fbm_turb(0.1f * p, 0.1f * dpdx, 0.1f * dpdy, 0.5f, 3, false);
            Float wave_height = fbm_turb(p, dpdx, dpdy, 0.5f, 6, false);
            return RGB(std::fabs(wind_strength) * wave_height);
        }
        case PT_TEX_MARBLE: {                                     // marble.rs:35-62
            static const Float C[9][3] = {{0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.5f, 0.5f, 0.5f}, {0.6f, 0.59f, 0.58f},
                                          {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.2f, 0.2f, 0.33f}, {0.58f, 0.58f, 0.6f}};
            V3 p, dpdx, dpdy;
            map3d(t, si, &p, &dpdx, &dpdy);
            p = t.scale * p;
            Float marble = p.y + t.variation * fbm_turb(p, t.scale * dpdx, t.scale * dpdy, t.omega, (uint32_t)t.octaves, false);
            Float tt = 0.5f + 0.5f * std::sin(marble);
            const int nseg = 9 - 3;
            Float fl = std::floor(tt * (Float)nseg);
            size_t first = std::min<size_t>(1, fl != fl ? 0 : (fl <= 0.0f ? 0 : (size_t)fl));
            tt = tt * (Float)nseg - (Float)first;
            auto col = [&](size_t k) { return RGB(C[k][0], C[k][1], C[k][2]); };
            auto lerps = [](RGB c0, RGB c1, Float u) { return c0 * (1.0f - u) + c1 * u; };
            RGB c0 = col(first), c1 = col(first + 1), c2 = col(first + 2), c3 = col(first + 3);
            RGB s0 = lerps(c0, c1, tt), s1 = lerps(c1, c2, tt), s2 = lerps(c2, c3, tt);
            s0 = lerps(s0, s1, tt); s1 = lerps(s1, s2, tt);
            return lerps(s0, s1, tt) * 1.5f;
        }
        case PT_TEX_BILERP: {
            V2 st, dx, dy;
            map2d(t, si, &st, &dx, &dy);
            Float a = (1.0f - st.x) * (1.0f - st.y), b = (1.0f - st.x) * st.y, c = st.x * (1.0f - st.y), d = st.x * st.y;
            auto v = [&](int k) { return RGB(t.value[k][0], t.value[k][1], t.value[k][2]); };
            return v(0) * a + v(1) * b + v(2) * c + v(3) * d;
        }
        default:
            return RGB(t.value[0][0], t.value[0][1], t.value[0][2]);
    }
}

}  // namespace orc
