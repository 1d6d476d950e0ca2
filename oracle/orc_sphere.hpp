// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_sphere.hpp: the Sphere shape with its EFloat running-error intervals.
//   follows src/shapes/sphere.rs, src/core/efloat/efloat.rs, src/core/transform/transform.rs:122-323,
//           the Shape trait defaults in src/core/shape/shape.rs:19-74
// Pinned by the reference's sphere_solid_angle / full_sphere_reintersect / partial_sphere_reintersect
// (tests/shapes.rs:280-329, :416-457), restated in kat_main.cpp.  Included by orc_accel.hpp after
// Ray / SurfHit / Bounds3 / Mat4 are defined.
#pragma once

namespace orc {

// core/efloat/efloat.rs:6-219: a value with a conservative [low, high] interval.
struct EFloat {
    Float v = 0, low = 0, high = 0;
    EFloat() {}
    EFloat(Float v_, Float err) {             // from_float (efloat.rs:14-24)
        v = v_;
        if (err == 0.0f) { low = v_; high = v_; }
        else { low = next_float_down(v_ - err); high = next_float_up(v_ + err); }
    }
    static EFloat from_f64(double v_, double err) { return EFloat((Float)v_, (Float)err); }
    Float upper_bound() const { return high; }
    Float lower_bound() const { return low; }
    bool operator==(const EFloat& o) const { return v == o.v && low == o.low && high == o.high; }   // derive(PartialEq)
};
inline EFloat operator+(EFloat a, EFloat b) {       // efloat.rs:103-112
    EFloat r; r.v = a.v + b.v; r.low = next_float_down(a.low + b.low); r.high = next_float_up(a.high + b.high); return r;
}
inline EFloat operator-(EFloat a, EFloat b) {       // efloat.rs:114-123
    EFloat r; r.v = a.v - b.v; r.low = next_float_down(a.low - b.high); r.high = next_float_up(a.high - b.low); return r;
}
inline EFloat operator*(EFloat a, EFloat b) {       // efloat.rs:125-146
    EFloat r; r.v = a.v * b.v;
    Float p0 = a.low * b.low, p1 = a.high * b.low, p2 = a.low * b.high, p3 = a.high * b.high;
    r.low = next_float_down(fmin_(fmin_(p0, p1), fmin_(p2, p3)));
    r.high = next_float_up(fmax_(fmax_(p0, p1), fmax_(p2, p3)));
    return r;
}
inline EFloat operator/(EFloat a, EFloat b) {       // efloat.rs:148-175
    EFloat r; r.v = a.v / b.v;
    if (b.low < 0.0f && b.high > 0.0f) { r.low = -kInfinity; r.high = kInfinity; return r; }
    Float d0 = a.low / b.low, d1 = a.high / b.low, d2 = a.low / b.high, d3 = a.high / b.high;
    r.low = next_float_down(fmin_(fmin_(d0, d1), fmin_(d2, d3)));
    r.high = next_float_up(fmax_(fmax_(d0, d1), fmax_(d2, d3)));
    return r;
}
inline EFloat operator*(EFloat a, Float f) { return a * EFloat(f, 0.0f); }   // efloat.rs:177-182
// EFloat::quadratic (efloat.rs:71-100): discriminant in f64
inline bool efloat_quadratic(EFloat a, EFloat b, EFloat c, EFloat* t0, EFloat* t1) {
    const double eps = std::numeric_limits<double>::epsilon();
    double av = (double)a.v, bv = (double)b.v, cv = (double)c.v;
    double discrim = bv * bv - 4.0 * av * cv;
    if (discrim < 0.0) return false;
    double root_discrim = std::sqrt(discrim);
    EFloat frd = EFloat::from_f64(root_discrim, eps * root_discrim);
    EFloat q = b.v < 0.0f ? (b - frd) * -0.5f : (b + frd) * -0.5f;
    EFloat r0 = q / a, r1 = c / q;
    if (r0.v <= r1.v) { *t0 = r0; *t1 = r1; } else { *t0 = r1; *t1 = r0; }
    return true;
}

struct Sphere {
    Mat4 o2w, w2o;                 // Transform{m, m_inv} of object_to_world
    bool reverse_orientation = false, swaps_handedness = false;
    Float radius = 1, z_min = -1, z_max = 1, theta_min = 0, theta_max = 0, phi_max = 0;

    // Sphere::new (sphere.rs:18-41), BaseShape::new (swaps_handedness, transform.rs:113-120)
    void init(const Float* o2w_m, const Float* w2o_m, bool ro, Float r, Float zmin, Float zmax, Float phimax_deg) {
        std::memcpy(o2w.m, o2w_m, sizeof(o2w.m));
        std::memcpy(w2o.m, w2o_m, sizeof(w2o.m));
        reverse_orientation = ro;
        const Float* m = o2w.m;
        Float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
        swaps_handedness = det < 0.0f;
        radius = r;
        z_min = clampf(fmin_(zmin, zmax), -r, r);
        z_max = clampf(fmax_(z_min, zmax), -r, r);          // sic: uses the already clamped z_min (sphere.rs:28)
        theta_min = std::acos(clampf(z_min / r, -1.0f, 1.0f));
        theta_max = std::acos(clampf(z_max / r, -1.0f, 1.0f));
        phi_max = radians(clampf(phimax_deg, 0.0f, 360.0f));
    }
    // transform.rs:130-132 -> matrix4x4.rs:309-317 with m_inv
    V3 normal_to_world(V3 n) const {
        const Float* m = w2o.m;
        return V3(m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z, m[2] * n.x + m[6] * n.y + m[10] * n.z);
    }
    Bounds3 object_bound() const {       // sphere.rs:45-52
        Float r = radius * 1.001f;
        Float diff = r - radius;
        return Bounds3(V3(-r, -r, z_min - diff), V3(r, r, z_max + diff));
    }
    Bounds3 world_bound() const {        // transform.rs:134-182
        Bounds3 b = object_bound();
        V3 mn, mx;
        for (int i = 0; i < 8; i++) {
            V3 c((i & 4) ? b.max.x : b.min.x, (i & 2) ? b.max.y : b.min.y, (i & 1) ? b.max.z : b.min.z);
            V3 q = o2w.transform_point(c);
            if (i == 0) { mn = q; mx = q; }
            else {
                mn = V3(fmin_(mn.x, q.x), fmin_(mn.y, q.y), fmin_(mn.z, q.z));
                mx = V3(fmax_(mx.x, q.x), fmax_(mx.y, q.y), fmax_(mx.z, q.z));
            }
        }
        return Bounds3(mn, mx);
    }
    Float area() const { return phi_max * radius * (z_max - z_min); }    // sphere.rs:282-284

    // Transform::transform_ray (transform.rs:184-203, :245-282) with world_to_object
    void ray_to_object(const Ray& r, V3* o, V3* d, V3* o_err, V3* d_err) const {
        const Float* m = w2o.m;
        V3 p = r.o, v = r.d;
        V3 op = w2o.transform_point(p);
        *o_err = kGamma3 * V3(std::fabs(m[0] * p.x) + std::fabs(m[1] * p.y) + std::fabs(m[2] * p.z) + std::fabs(m[3]),
                              std::fabs(m[4] * p.x) + std::fabs(m[5] * p.y) + std::fabs(m[6] * p.z) + std::fabs(m[7]),
                              std::fabs(m[8] * p.x) + std::fabs(m[9] * p.y) + std::fabs(m[10] * p.z) + std::fabs(m[11]));
        *d_err = kGamma3 * V3(std::fabs(m[0] * v.x) + std::fabs(m[1] * v.y) + std::fabs(m[2] * v.z),
                              std::fabs(m[4] * v.x) + std::fabs(m[5] * v.y) + std::fabs(m[6] * v.z),
                              std::fabs(m[8] * v.x) + std::fabs(m[9] * v.y) + std::fabs(m[10] * v.z));
        V3 dd = w2o.transform_vector(v);
        Float ls = length_squared(dd);
        if (ls > 0.0f) {
            Float dt = dot(vabs(dd), *o_err) / ls;
            op += dd * dt;
        }
        *o = op; *d = dd;
    }
    bool clipped(V3 p_hit, Float phi) const {
        return (z_min > -radius && p_hit.z < z_min) || (z_max < radius && p_hit.z > z_max) || (phi > phi_max);
    }
    // p_hit refinement + phi (sphere.rs:95-103)
    V3 refine(V3 o, V3 d, Float t, Float* phi, Float wrap) const {
        V3 p_hit = o + d * t;
        p_hit = p_hit * (radius / length(p_hit));
        if (p_hit.x == 0.0f && p_hit.y == 0.0f) p_hit.x = 1e-5f * radius;
        Float ph = std::atan2(p_hit.y, p_hit.x);
        if (ph < 0.0f) ph += wrap;
        *phi = ph;
        return p_hit;
    }
    // Shared front of intersect / intersect_p (sphere.rs:61-128, :200-263).  `second_wrap` is what
    // the retry at t1 adds to a negative phi: PI in intersect (sphere.rs:121, sic), 2*PI in intersect_p.
    bool hit_test(const Ray& r, Float second_wrap, V3* o_obj, V3* d_obj, Float* t_out, V3* p_hit_out, Float* phi_out) const {
        V3 o, d, oe, de;
        ray_to_object(r, &o, &d, &oe, &de);
        EFloat ox(o.x, oe.x), oy(o.y, oe.y), oz(o.z, oe.z);
        EFloat dx(d.x, de.x), dy(d.y, de.y), dz(d.z, de.z);
        EFloat rad(radius, 0.0f);
        EFloat a = dx * dx + dy * dy + dz * dz;
        EFloat b = (dx * ox + dy * oy + dz * oz) * 2.0f;
        EFloat c = ox * ox + oy * oy + oz * oz - rad * rad;
        Float t_max = r.t_max;
        EFloat t0, t1;
        if (!efloat_quadratic(a, b, c, &t0, &t1)) return false;
        if (std::isinf(t0.v) || std::isinf(t1.v)) return false;
        if (t0.upper_bound() > t_max || t1.lower_bound() <= 0.0f) return false;
        EFloat t_shape_hit = t0;
        if (t_shape_hit.lower_bound() <= 0.0f) {
            t_shape_hit = t1;
            if (t_max < t_shape_hit.upper_bound()) return false;
        }
        Float phi;
        V3 p_hit = refine(o, d, t_shape_hit.v, &phi, 2.0f * kPi);
        if (clipped(p_hit, phi)) {
            if (t_shape_hit == t1) return false;
            if (t1.upper_bound() > t_max) return false;
            t_shape_hit = t1;
            p_hit = refine(o, d, t_shape_hit.v, &phi, second_wrap);
            if (clipped(p_hit, phi)) return false;
        }
        *o_obj = o; *d_obj = d; *t_out = t_shape_hit.v; *p_hit_out = p_hit; *phi_out = phi;
        return true;
    }
    bool intersect_p(const Ray& r) const {
        V3 o, d, p; Float t, phi;
        return hit_test(r, 2.0f * kPi, &o, &d, &t, &p, &phi);
    }
    // sphere.rs:61-198 + Transform::transform_surface_interaction (transform.rs:299-323)
    bool intersect(const Ray& r, Float* t_hit, SurfHit* si) const {
        V3 o, d, p_hit; Float t, phi;
        if (!hit_test(r, kPi, &o, &d, &t, &p_hit, &phi)) return false;
        Float dtheta = theta_max - theta_min;
        Float u = phi / phi_max;
        Float theta = std::acos(clampf(p_hit.z / radius, -1.0f, 1.0f));
        Float v = (theta - theta_min) / dtheta;
        Float z_radius = std::sqrt(p_hit.x * p_hit.x + p_hit.y * p_hit.y);
        Float inv_z_radius = 1.0f / z_radius;
        Float cos_phi = p_hit.x * inv_z_radius;
        Float sin_phi = p_hit.y * inv_z_radius;
        V3 dpdu(-phi_max * p_hit.y, phi_max * p_hit.x, 0.0f);
        V3 dpdv = V3(p_hit.z * cos_phi, p_hit.z * sin_phi, -radius * std::sin(theta)) * dtheta;
        // Weingarten equations (sphere.rs:156-176)
        V3 d2pduu = V3(p_hit.x, p_hit.y, 0.0f) * (-phi_max * phi_max);
        V3 d2pduv = V3(-sin_phi, cos_phi, 0.0f) * (p_hit.z * dtheta * phi_max);
        V3 d2pdvv = V3(p_hit.x, p_hit.y, p_hit.z) * (-dtheta * dtheta);
        Float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        V3 n = normalize(cross(dpdu, dpdv));                       // BaseShape::calc_normal (base_shape.rs:27-33)
        if (reverse_orientation ^ swaps_handedness) n = n * -1.0f;
        Float ee = dot(n, d2pduu), ff = dot(n, d2pduv), gg = dot(n, d2pdvv);
        Float inv_egf2 = 1.0f / (E * G - F * F);
        V3 dndu = dpdu * ((ff * F - ee * G) * inv_egf2) + dpdv * ((ee * F - ff * E) * inv_egf2);
        V3 dndv = dpdu * ((gg * F - ff * G) * inv_egf2) + dpdv * ((ff * F - gg * E) * inv_egf2);
        V3 p_error = kGamma5 * vabs(p_hit);
        // -> world
        const Float* m = o2w.m;
        si->p = o2w.transform_point(p_hit);
        V3 e;
        for (int i = 0; i < 3; i++) {
            const Float* row = m + 4 * i;
            Float v1 = (kGamma3 + 1.0f) * (std::fabs(row[0]) * p_error.x + std::fabs(row[1]) * p_error.y + std::fabs(row[2]) * p_error.z) +
                       kGamma3 * (std::fabs(row[0] * p_hit.x) + std::fabs(row[1] * p_hit.y) + std::fabs(row[2] * p_hit.z) + std::fabs(row[3]));
            if (i == 0) e.x = v1; else if (i == 1) e.y = v1; else e.z = v1;
        }
        si->p_error = e;
        si->n = normalize(normal_to_world(n));
        si->wo = normalize(o2w.transform_vector(-d));
        si->uv = V2(u, v);
        si->dpdu = o2w.transform_vector(dpdu);
        si->dpdv = o2w.transform_vector(dpdv);
        si->sh_n = face_forward(normalize(normal_to_world(n)), si->n);
        si->sh_dpdu = si->dpdu; si->sh_dpdv = si->dpdv;
        si->dndu = normal_to_world(dndu); si->dndv = normal_to_world(dndv);      // transform_normal, not renormalised (transform.rs:309-310)
        si->sh_dndu = si->dndu; si->sh_dndv = si->dndv;
        si->b0 = si->b1 = si->b2 = 0.0f;
        *t_hit = t;
        return true;
    }
    // sphere.rs:286-304
    void sample(V2 u, V3* p, V3* n, V3* p_error, Float* pdf) const {
        Float z = 1.0f - 2.0f * u.x;                               // uniform_sample_sphere (sampling.rs:97-102)
        Float rr = std::sqrt(fmax_(0.0f, 1.0f - z * z));
        Float phi = 2.0f * kPi * u.y;
        V3 p_obj = V3(0.0f, 0.0f, 0.0f) + radius * V3(rr * std::cos(phi), rr * std::sin(phi), z);
        V3 nn = normalize(normal_to_world(p_obj));
        if (reverse_orientation) nn = nn * -1.0f;
        p_obj = p_obj * (radius / length(p_obj));
        V3 pe = kGamma5 * vabs(p_obj);
        const Float* m = o2w.m;
        *p = o2w.transform_point(p_obj);
        V3 e;
        for (int i = 0; i < 3; i++) {
            const Float* row = m + 4 * i;
            Float v1 = (kGamma3 + 1.0f) * (std::fabs(row[0]) * pe.x + std::fabs(row[1]) * pe.y + std::fabs(row[2]) * pe.z) +
                       kGamma3 * (std::fabs(row[0] * p_obj.x) + std::fabs(row[1] * p_obj.y) + std::fabs(row[2] * p_obj.z) + std::fabs(row[3]));
            if (i == 0) e.x = v1; else if (i == 1) e.y = v1; else e.z = v1;
        }
        *p_error = e;
        *n = nn;
        *pdf = 1.0f / area();
    }
    // sphere.rs:306-387
    bool sample_from(V3 ref_p, V3 ref_p_error, V3 ref_n, V2 u, V3* p, V3* n, V3* p_error, Float* pdf_out) const {
        V3 p_center = o2w.transform_point(V3(0.0f, 0.0f, 0.0f));
        V3 p_origin = offset_ray_origin(ref_p, ref_p_error, ref_n, p_center - ref_p);
        if (distance_squared(p_origin, p_center) <= radius * radius) {
            Float pdf;
            sample(u, p, n, p_error, &pdf);
            V3 wi = *p - ref_p;
            if (length_squared(wi) == 0.0f) return false;
            wi = normalize(wi);
            pdf = pdf * distance_squared(*p, ref_p) / abs_dot(*n, -wi);
            if (pdf <= 0.0f || std::isinf(pdf)) return false;
            *pdf_out = pdf;
            return true;
        }
        Float dc = length(ref_p - p_center);
        Float inv_dc = 1.0f / dc;
        V3 wc = (p_center - ref_p) * inv_dc;
        V3 wc_x, wc_y;
        coordinate_system(wc, &wc_x, &wc_y);
        Float sin_theta_max = radius * inv_dc;
        Float sin_theta_max2 = sin_theta_max * sin_theta_max;
        Float inv_sin_theta_max = 1.0f / sin_theta_max;
        Float cos_theta_max = std::sqrt(fmax_(0.0f, 1.0f - sin_theta_max2));
        Float pdf = 1.0f / (2.0f * kPi * (1.0f - cos_theta_max));
        if (pdf <= 0.0f || std::isinf(pdf)) return false;
        Float cos_theta = (cos_theta_max - 1.0f) * u.x + 1.0f;
        Float sin_theta2 = 1.0f - cos_theta * cos_theta;
        if (sin_theta_max2 < 0.00068523f) {
            sin_theta2 = fmax_(0.0f, sin_theta_max2 * u.x);
            cos_theta = std::sqrt(1.0f - sin_theta2);
        }
        Float cos_alpha = sin_theta2 * inv_sin_theta_max +
                          cos_theta * std::sqrt(fmax_(0.0f, 1.0f - sin_theta2 * inv_sin_theta_max * inv_sin_theta_max));
        Float sin_alpha = std::sqrt(fmax_(0.0f, 1.0f - cos_alpha * cos_alpha));
        Float phi = u.y * 2.0f * kPi;
        // spherical_direction_axes (core/geometry/misc.rs:82-93)
        V3 n_world = (sin_alpha * std::cos(phi)) * (-wc_x) + (sin_alpha * std::sin(phi)) * (-wc_y) + cos_alpha * (-wc);
        V3 p_world = p_center + radius * n_world;
        *p = p_world;
        *p_error = kGamma5 * vabs(p_world);
        V3 nn = n_world;
        if (reverse_orientation) nn = nn * -1.0f;
        *n = nn;
        *pdf_out = pdf;
        return true;
    }
    // Shape::pdf_from default (shape.rs:40-54): Sphere does not override it (pbrt-v3's Sphere::Pdf
    // with the cone pdf is absent), so MIS weights use the area-measure conversion.
    Float pdf_from(V3 ref_p, V3 ref_p_error, V3 ref_n, V3 wi) const {
        Ray ray(offset_ray_origin(ref_p, ref_p_error, ref_n, wi), wi, kInfinity);
        Float t;
        SurfHit isect;
        if (!intersect(ray, &t, &isect)) return 0.0f;
        Float pdf = distance_squared(ref_p, isect.p) / (abs_dot(isect.n, -wi) * area());
        if (std::isinf(pdf)) return 0.0f;
        return pdf;
    }
};

}  // namespace orc
