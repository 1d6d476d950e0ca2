// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_texture.hpp: ray differentials at a hit and the procedural textures.
//   follows src/core/interaction/surface_interaction.rs:212-282 (compute_differentials),
//           src/core/transform/matrix4x4.rs:9-20 (solve_linear_system_2x2),
//           src/core/texture/mapping2d.rs, mapping3d.rs,
//           src/textures/{constant,scale,mix,checkerboard,uv,bilerp}.rs
#pragma once
#include "orc_accel.hpp"
#include "../include/pbrtgpu.h"

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

// Texture<T>::evaluate over the flattened texture array; float textures are RGB with equal channels.
inline RGB texture_eval(const pt_texture* tex, int32_t index, const TexHit& si);
inline RGB texture_child(const pt_texture* tex, const pt_texture& t, int k, const TexHit& si) {
    if (t.tex[k] >= 0) return texture_eval(tex, t.tex[k], si);
    return RGB(t.value[k][0], t.value[k][1], t.value[k][2]);
}
inline RGB texture_eval(const pt_texture* tex, int32_t index, const TexHit& si) {
    const pt_texture& t = tex[index];
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
