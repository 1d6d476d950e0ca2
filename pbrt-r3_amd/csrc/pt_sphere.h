// pt_sphere.h -- the reference's Sphere shape on the device (src/shapes/sphere.rs), tested with its
// EFloat running-error intervals (src/core/efloat/efloat.rs) in object space.  Spheres are rare
// primitives (pbrt-v3's killeroo-simple lights its scene with two of them), so the code favours the
// reference's exact arithmetic over speed: every product and sum below is the one the Rust source
// writes, in its order, and the kernels that can meet a sphere are separate instantiations so the
// triangle-only kernels do not pay registers for it.
#pragma once
#include "pt_device.h"
#include "pt_device_math.h"

// core/efloat/efloat.rs:6-219
struct PtEF { float v, lo, hi; };
PT_DEV PtEF ef_make(float v, float err) {        // from_float :14-24
    PtEF r; r.v = v;
    if (err == 0.0f) { r.lo = v; r.hi = v; }
    else { r.lo = next_float_down(v - err); r.hi = next_float_up(v + err); }
    return r;
}
PT_DEV PtEF ef_add(PtEF a, PtEF b) { PtEF r; r.v = a.v + b.v; r.lo = next_float_down(a.lo + b.lo); r.hi = next_float_up(a.hi + b.hi); return r; }
PT_DEV PtEF ef_sub(PtEF a, PtEF b) { PtEF r; r.v = a.v - b.v; r.lo = next_float_down(a.lo - b.hi); r.hi = next_float_up(a.hi - b.lo); return r; }
PT_DEV PtEF ef_mul(PtEF a, PtEF b) {
    PtEF r; r.v = a.v * b.v;
    float p0 = a.lo * b.lo, p1 = a.hi * b.lo, p2 = a.lo * b.hi, p3 = a.hi * b.hi;
    r.lo = next_float_down(fminf(fminf(p0, p1), fminf(p2, p3)));
    r.hi = next_float_up(fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)));
    return r;
}
PT_DEV PtEF ef_div(PtEF a, PtEF b) {
    PtEF r; r.v = a.v / b.v;
    if (b.lo < 0.0f && b.hi > 0.0f) { r.lo = -PT_INF; r.hi = PT_INF; return r; }
    float d0 = a.lo / b.lo, d1 = a.hi / b.lo, d2 = a.lo / b.hi, d3 = a.hi / b.hi;
    r.lo = next_float_down(fminf(fminf(d0, d1), fminf(d2, d3)));
    r.hi = next_float_up(fmaxf(fmaxf(d0, d1), fmaxf(d2, d3)));
    return r;
}
PT_DEV PtEF ef_mulf(PtEF a, float f) { return ef_mul(a, ef_make(f, 0.0f)); }
PT_DEV bool ef_eq(PtEF a, PtEF b) { return a.v == b.v && a.lo == b.lo && a.hi == b.hi; }
// EFloat::quadratic (efloat.rs:71-100): discriminant and its root in f64
PT_DEV bool ef_quadratic(PtEF a, PtEF b, PtEF c, PtEF* t0, PtEF* t1) {
    double av = (double)a.v, bv = (double)b.v, cv = (double)c.v;
    double discrim = bv * bv - 4.0 * av * cv;
    if (discrim < 0.0) return false;
    double root = sqrt(discrim);
    PtEF frd = ef_make((float)root, (float)(2.220446049250313e-16 * root));
    PtEF q = b.v < 0.0f ? ef_mulf(ef_sub(b, frd), -0.5f) : ef_mulf(ef_add(b, frd), -0.5f);
    PtEF r0 = ef_div(q, a), r1 = ef_div(c, q);
    if (r0.v <= r1.v) { *t0 = r0; *t1 = r1; } else { *t0 = r1; *t1 = r0; }
    return true;
}

// affine Transform rows (matrix4x4.rs:284-317); the last row is 0 0 0 1 (checked at upload), so
// transform_point's homogeneous divide never runs
PT_DEV V3 sph_point(const float* m, V3 p) {
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
PT_DEV V3 sph_vector(const float* m, V3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// Transform::transform_normal = m_inv transposed
PT_DEV V3 sph_normal(const float* mi, V3 n) {
    return mk3(mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z);
}
// transform_point_with_abs_error (transform.rs:205-243)
PT_DEV V3 sph_point_abs_error(const float* m, V3 p, V3 pe) {
    const float g3 = PT_GAMMA(3.0f);
    V3 e;
    e.x = (g3 + 1.0f) * (fabsf(m[0]) * pe.x + fabsf(m[1]) * pe.y + fabsf(m[2]) * pe.z) + g3 * (fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]));
    e.y = (g3 + 1.0f) * (fabsf(m[4]) * pe.x + fabsf(m[5]) * pe.y + fabsf(m[6]) * pe.z) + g3 * (fabsf(m[4] * p.x) + fabsf(m[5] * p.y) + fabsf(m[6] * p.z) + fabsf(m[7]));
    e.z = (g3 + 1.0f) * (fabsf(m[8]) * pe.x + fabsf(m[9]) * pe.y + fabsf(m[10]) * pe.z) + g3 * (fabsf(m[8] * p.x) + fabsf(m[9] * p.y) + fabsf(m[10] * p.z) + fabsf(m[11]));
    return e;
}

// Object-space ray and refined hit point, plus the two interval ends the t_max tests compared against: the hit stands for a
// given t_max iff !(a_hi > t_max) && !(b_hi > t_max)  (sphere.rs:87-95, :110-113: t0's upper bound always, t1's when t1 is
// the root taken).  A caller that passes t_max = +inf and applies that test later gets the decision of any finite t_max.
struct SphHit { V3 o, d, p_hit; float t, phi, a_hi, b_hi; };

PT_DEV bool sph_clipped(const PtSphere& s, V3 p_hit, float phi) {
    return (s.z_min > -s.radius && p_hit.z < s.z_min) || (s.z_max < s.radius && p_hit.z > s.z_max) || (phi > s.phi_max);
}
PT_DEV V3 sph_refine(const PtSphere& s, V3 o, V3 d, float t, float wrap, float* phi) {     // sphere.rs:95-103
    V3 p = o + d * t;
    p = p * (s.radius / length(p));
    if (p.x == 0.0f && p.y == 0.0f) p.x = 1e-5f * s.radius;
    float ph = pt_atan2f(p.y, p.x);
    if (ph < 0.0f) ph += wrap;
    *phi = ph;
    return p;
}
// Front of Sphere::intersect / intersect_p (sphere.rs:61-128, :200-263).  second_wrap: what the
// retry at t1 adds to a negative phi -- PI in intersect (sphere.rs:121, as written), 2*PI in intersect_p.
// (inline: for the pooled-leaf traversal kernel's sphere round, which then needs no scratch frame -- k_trace_sph_dist 44.4 -> 43.5 ms per
// launch on RT1M lit by a sphere; everybody else calls sph_hit_test below: inlined into the shading kernels it costs them 40 spilled registers)
__device__ __forceinline__ bool sph_hit_test_inl(const PtSphere& s, V3 ro, V3 rd, float t_max, float second_wrap, SphHit* h) {
    // Transform::transform_ray with world_to_object (transform.rs:184-203, :245-282)
    const float* m = s.w2o;
    const float g3 = PT_GAMMA(3.0f);
    V3 o = sph_point(m, ro);
    V3 oe = g3 * mk3(fabsf(m[0] * ro.x) + fabsf(m[1] * ro.y) + fabsf(m[2] * ro.z) + fabsf(m[3]),
                     fabsf(m[4] * ro.x) + fabsf(m[5] * ro.y) + fabsf(m[6] * ro.z) + fabsf(m[7]),
                     fabsf(m[8] * ro.x) + fabsf(m[9] * ro.y) + fabsf(m[10] * ro.z) + fabsf(m[11]));
    V3 de = g3 * mk3(fabsf(m[0] * rd.x) + fabsf(m[1] * rd.y) + fabsf(m[2] * rd.z), fabsf(m[4] * rd.x) + fabsf(m[5] * rd.y) + fabsf(m[6] * rd.z),
                     fabsf(m[8] * rd.x) + fabsf(m[9] * rd.y) + fabsf(m[10] * rd.z));
    V3 d = sph_vector(m, rd);
    float ls = length_squared(d);
    if (ls > 0.0f) {
        float dt = dot(vabs(d), oe) / ls;
        o = o + d * dt;
    }
    {   // The value lane of the interval computation below, alone -- the same operations in the same order, a third of the work and none of the
        // next_float_up / down calls: a miss IT proves is a miss of the whole test (every EFloat keeps lo <= v <= hi, so t0.v > t_max implies
        // t0.hi > t_max; the discriminant and the infinity checks read the value lanes only).  Nearly every sphere test of a scene lit by a
        // sphere ends here: shadow rays stop just short of the light's surface, BSDF-sampled directions miss a small light altogether.
        const float av = (d.x * d.x + d.y * d.y) + d.z * d.z;
        const float bv = ((d.x * o.x + d.y * o.y) + d.z * o.z) * 2.0f;
        const float cv = ((o.x * o.x + o.y * o.y) + o.z * o.z) - s.radius * s.radius;
        const double discrim = (double)bv * (double)bv - 4.0 * (double)av * (double)cv;
        if (discrim < 0.0) return false;
        const float fr = (float)sqrt(discrim);
        const float qv = bv < 0.0f ? (bv - fr) * -0.5f : (bv + fr) * -0.5f;
        const float r0 = qv / av, r1 = cv / qv;
        const float t0v = r0 <= r1 ? r0 : r1, t1v = r0 <= r1 ? r1 : r0;
        if (isinf(t0v) || isinf(t1v)) return false;
        if (t0v > t_max) return false;
    }
    PtEF ox = ef_make(o.x, oe.x), oy = ef_make(o.y, oe.y), oz = ef_make(o.z, oe.z);
    PtEF dx = ef_make(d.x, de.x), dy = ef_make(d.y, de.y), dz = ef_make(d.z, de.z);
    PtEF rad = ef_make(s.radius, 0.0f);
    PtEF a = ef_add(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(dz, dz));
    PtEF b = ef_mulf(ef_add(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(dz, oz)), 2.0f);
    PtEF c = ef_sub(ef_add(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(oz, oz)), ef_mul(rad, rad));
    PtEF t0, t1;
    if (!ef_quadratic(a, b, c, &t0, &t1)) return false;
    if (isinf(t0.v) || isinf(t1.v)) return false;
    if (t0.hi > t_max || t1.lo <= 0.0f) return false;
    PtEF th = t0;
    if (th.lo <= 0.0f) {
        th = t1;
        if (t_max < th.hi) return false;
    }
    float phi;
    V3 p_hit = sph_refine(s, o, d, th.v, 2.0f * PT_PI, &phi);
    if (sph_clipped(s, p_hit, phi)) {
        if (ef_eq(th, t1)) return false;
        if (t1.hi > t_max) return false;
        th = t1;
        p_hit = sph_refine(s, o, d, th.v, second_wrap, &phi);
        if (sph_clipped(s, p_hit, phi)) return false;
    }
    h->o = o; h->d = d; h->p_hit = p_hit; h->t = th.v; h->phi = phi;
    h->a_hi = t0.hi;
    h->b_hi = ef_eq(th, t0) ? -PT_INF : t1.hi;
    return true;
}
__device__ __noinline__ bool sph_hit_test(const PtSphere& s, V3 ro, V3 rd, float t_max, float second_wrap, SphHit* h) {
    return sph_hit_test_inl(s, ro, rd, t_max, second_wrap, h);
}
// World-space interaction of a hit (sphere.rs:130-198 + transform_surface_interaction, transform.rs:299-323).
// Only what the path consumes: p, p_error, n, wo, shading n and dpdu (u, v, dndu, dndv feed textures).
PT_DEV void sph_interaction(const PtSphere& s, const SphHit& h, V3* p, V3* p_error, V3* n, V3* wo, V3* sh_n, V3* dpdu_w, V3* dpdv_w, V2* uv,
                            V3* dndu_w, V3* dndv_w) {
    V3 ph = h.p_hit;
    float dtheta = s.theta_max - s.theta_min;
    float theta = pt_acosf(clampf(ph.z / s.radius, -1.0f, 1.0f));
    float z_radius = sqrtf(ph.x * ph.x + ph.y * ph.y);
    float inv_z_radius = 1.0f / z_radius;
    float cos_phi = ph.x * inv_z_radius, sin_phi = ph.y * inv_z_radius;
    V3 dpdu = mk3(-s.phi_max * ph.y, s.phi_max * ph.x, 0.0f);
    float sn, cs;
    pt_sincosf(theta, &sn, &cs);
    V3 dpdv = mk3(ph.z * cos_phi, ph.z * sin_phi, -s.radius * sn) * dtheta;
    V3 nn = normalize(cross(dpdu, dpdv));                        // BaseShape::calc_normal (base_shape.rs:27-33)
    if (s.flags & PT_SPH_FLIP) nn = nn * -1.0f;
    V3 pe = PT_GAMMA(5.0f) * vabs(ph);
    *p = sph_point(s.o2w, ph);
    *p_error = sph_point_abs_error(s.o2w, ph, pe);
    V3 nw = normalize(sph_normal(s.w2o, nn));
    *n = nw;
    *wo = normalize(sph_vector(s.o2w, -h.d));
    *sh_n = face_forward(nw, nw);                                // shading.n = n before the transform
    *dpdu_w = sph_vector(s.o2w, dpdu);
    *dpdv_w = sph_vector(s.o2w, dpdv);
    *uv = mk2(h.phi / s.phi_max, (theta - s.theta_min) / dtheta);      // sphere.rs:136-140 (read by textures only)
    {   // Weingarten equations (sphere.rs:156-176); read by bump mapping only
        V3 d2pduu = mk3(ph.x, ph.y, 0.0f) * (-s.phi_max * s.phi_max);
        V3 d2pduv = mk3(-sin_phi, cos_phi, 0.0f) * (ph.z * dtheta * s.phi_max);
        V3 d2pdvv = mk3(ph.x, ph.y, ph.z) * (-dtheta * dtheta);
        float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        float ee = dot(nn, d2pduu), ff = dot(nn, d2pduv), gg = dot(nn, d2pdvv);
        float inv_egf2 = 1.0f / (E * G - F * F);
        V3 dndu = dpdu * ((ff * F - ee * G) * inv_egf2) + dpdv * ((ee * F - ff * E) * inv_egf2);
        V3 dndv = dpdu * ((gg * F - ff * G) * inv_egf2) + dpdv * ((ff * F - gg * E) * inv_egf2);
        *dndu_w = sph_normal(s.w2o, dndu);
        *dndv_w = sph_normal(s.w2o, dndv);
    }
}
// Sphere::sample (sphere.rs:286-304)
PT_DEV void sph_sample(const PtSphere& s, V2 u, V3* p, V3* n, V3* p_error) {
    float z = 1.0f - 2.0f * u.x;                                 // uniform_sample_sphere (sampling.rs:97-102)
    float rr = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float phi = 2.0f * PT_PI * u.y;
    float sn, cs;
    pt_sincosf(phi, &sn, &cs);
    V3 po = mk3(0.0f, 0.0f, 0.0f) + s.radius * mk3(rr * cs, rr * sn, z);
    V3 nn = normalize(sph_normal(s.w2o, po));
    if (s.flags & PT_SPH_REVERSE) nn = nn * -1.0f;
    po = po * (s.radius / length(po));
    V3 pe = PT_GAMMA(5.0f) * vabs(po);
    *p = sph_point(s.o2w, po);
    *p_error = sph_point_abs_error(s.o2w, po, pe);
    *n = nn;
}
// Sphere::sample_from (sphere.rs:306-387): uniform over the sphere from inside, over the subtended cone from outside
__device__ __noinline__ bool sph_sample_from(const PtSphere& s, V3 ref_p, V3 ref_pe, V3 ref_n, V2 u, V3* p, V3* n, V3* p_error, float* pdf_out) {
    V3 pc = sph_point(s.o2w, mk3(0.0f, 0.0f, 0.0f));            // object_to_world(0,0,0)
    V3 p_origin = offset_ray_origin(ref_p, ref_pe, ref_n, pc - ref_p);
    if (distance_squared(p_origin, pc) <= s.radius * s.radius) {
        sph_sample(s, u, p, n, p_error);
        float pdf = 1.0f / s.area;
        V3 wi = *p - ref_p;
        if (length_squared(wi) == 0.0f) return false;
        wi = normalize(wi);
        pdf = pdf * distance_squared(*p, ref_p) / abs_dot(*n, -wi);
        if (pdf <= 0.0f || isinf(pdf)) return false;
        *pdf_out = pdf;
        return true;
    }
    float dc = length(ref_p - pc);
    float inv_dc = 1.0f / dc;
    V3 wc = (pc - ref_p) * inv_dc;
    V3 wcx, wcy;
    coordinate_system(wc, &wcx, &wcy);
    float sin_theta_max = s.radius * inv_dc;
    float sin_theta_max2 = sin_theta_max * sin_theta_max;
    float inv_sin_theta_max = 1.0f / sin_theta_max;
    float cos_theta_max = sqrtf(fmaxf(0.0f, 1.0f - sin_theta_max2));
    float pdf = 1.0f / (2.0f * PT_PI * (1.0f - cos_theta_max));
    if (pdf <= 0.0f || isinf(pdf)) return false;
    float cos_theta = (cos_theta_max - 1.0f) * u.x + 1.0f;
    float sin_theta2 = 1.0f - cos_theta * cos_theta;
    if (sin_theta_max2 < 0.00068523f) {
        sin_theta2 = fmaxf(0.0f, sin_theta_max2 * u.x);
        cos_theta = sqrtf(1.0f - sin_theta2);
    }
    float cos_alpha = sin_theta2 * inv_sin_theta_max + cos_theta * sqrtf(fmaxf(0.0f, 1.0f - sin_theta2 * inv_sin_theta_max * inv_sin_theta_max));
    float sin_alpha = sqrtf(fmaxf(0.0f, 1.0f - cos_alpha * cos_alpha));
    float phi = u.y * 2.0f * PT_PI;
    float sn, cs;
    pt_sincosf(phi, &sn, &cs);
    V3 nw = (sin_alpha * cs) * (-wcx) + (sin_alpha * sn) * (-wcy) + cos_alpha * (-wc);    // spherical_direction_axes (misc.rs:82-93)
    V3 pw = pc + s.radius * nw;
    *p = pw;
    *p_error = PT_GAMMA(5.0f) * vabs(pw);
    if (s.flags & PT_SPH_REVERSE) nw = nw * -1.0f;
    *n = nw;
    *pdf_out = pdf;
    return true;
}
