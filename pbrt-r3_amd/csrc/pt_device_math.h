// pt_device_math.h -- f32 device helpers for the gfx950 kernels.
// Compiled with -ffp-contract=off and correctly rounded divide/sqrt so that the
// arithmetic is IEEE-identical to the reference's scalar Rust (which never fuses
// a*b+c): src/core/geometry/vector3.rs, geometry/misc.rs, misc/float.rs,
// spectrum/{rgb,convert}.rs, base/constants.rs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

struct V3 { float x, y, z; };
struct V2 { float x, y; };

PT_DEV V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V2 mk2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
PT_DEV V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
PT_DEV V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
PT_DEV V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_DEV V3 vabs(V3 a) { return mk3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
PT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV float abs_dot(V3 a, V3 b) { return fabsf(dot(a, b)); }
PT_DEV float length_squared(V3 a) { return dot(a, a); }
PT_DEV float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_DEV V3 normalize(V3 a) { float l = length(a); return mk3(a.x / l, a.y / l, a.z / l); }
PT_DEV float distance_squared(V3 a, V3 b) { V3 v = a - b; return dot(v, v); }
PT_DEV V3 cross(V3 a, V3 b) { return mk3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }
PT_DEV V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0.0f ? n * -1.0f : n; }
PT_DEV float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
PT_DEV float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
PT_DEV float clampf(float x, float lo, float hi) { if (x < lo) return lo; if (x > hi) return hi; return x; }
PT_DEV float lerpf(float t, float a, float b) { return (1.0f - t) * a + t * b; }
PT_DEV V3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
PT_DEV V3 f4_3(float4 v) { return mk3(v.x, v.y, v.z); }

#define PT_PI 3.14159265358979323846f
#define PT_INV_PI 0.31830988618379067154f
#define PT_PI_OVER_2 (PT_PI / 2.0f)
#define PT_PI_OVER_4 (PT_PI / 4.0f)
#define PT_ONE_MINUS_EPS 0.99999994f
#define PT_SHADOW_EPS 0.0001f
#define PT_INF __builtin_huge_valf()
#define PT_MACH_EPS (1.1920928955078125e-07f * 0.5f)
#define PT_GAMMA(n) (((n) * PT_MACH_EPS) / (1.0f - ((n) * PT_MACH_EPS)))
#define PT_GAMMA6_REF ((6.0f * PT_MACH_EPS) / (1.0f - (5.0f * PT_MACH_EPS)))   // triangle.rs:34 as written

// misc/float.rs:23-56
PT_DEV float next_float_down(float v) {
    if (isinf(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0.0f) ui = ui == 0u ? 0u : ui - 1u;
    else ui = ui == 0xffffffffu ? ui : ui + 1u;
    return __uint_as_float(ui);
}
PT_DEV float next_float_up(float v) {
    if (isinf(v) && v > 0.0f) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0.0f) ui = ui == 0xffffffffu ? ui : ui + 1u;
    else ui = ui == 0u ? 0u : ui - 1u;
    return __uint_as_float(ui);
}
// geometry/misc.rs:5-23
PT_DEV V3 offset_ray_origin(V3 p, V3 p_error, V3 n, V3 w) {
    float d = dot(vabs(n), p_error);
    V3 off = d * n;
    if (dot(w, n) < 0.0f) off = -off;
    V3 po = p + off;
    if (off.x > 0.0f) po.x = next_float_up(po.x); else if (off.x < 0.0f) po.x = next_float_down(po.x);
    if (off.y > 0.0f) po.y = next_float_up(po.y); else if (off.y < 0.0f) po.y = next_float_down(po.y);
    if (off.z > 0.0f) po.z = next_float_up(po.z); else if (off.z < 0.0f) po.z = next_float_down(po.z);
    return po;
}
// geometry/misc.rs:62-70
PT_DEV void coordinate_system(V3 v1, V3* v2, V3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y)) *v2 = mk3(-v1.z, 0.0f, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
    else *v2 = mk3(0.0f, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
    *v3 = normalize(cross(v1, *v2));
}

// sin/cos.  The reference calls libm's sinf/cosf (Rust f32::sin -> glibc), which is NOT correctly
// rounded: about 1.2 % of results differ from round(sin(x)) in the last bit, enough to send ~1.5 % of
// camera samples down a different path.  glibc's algorithm (ARM optimized-routines sincosf: reduce by
// pi/2 in f64, degree-7/8 f64 polynomials, one final rounding) is restated here with its published
// coefficients, so the device returns the same bits as the host libm for |x| < 120 (validated
// against libm.so.6 on >1e6 inputs, tools/check_sincosf_port.py).  Larger / non-finite arguments,
// which this path never produces, fall back to f64 sin/cos rounded once.
struct PtSinCosTab { double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
PT_DEV float pt_sincos_poly(double x, double x2, bool neg_cos, int n) {
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    if (neg_cos) { c0 = -c0; c1 = -c1; c2 = -c2; c3 = -c3; c4 = -c4; }     // __sincosf_table[1]
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double t1 = s2 + x2 * s3;
        double x5 = x3 * x2;
        double s = x + x3 * s1;
        return (float)(s + x5 * t1);
    }
    double x4 = x2 * x2;
    double t2 = c3 + x2 * c4;
    double t1 = c0 + x2 * c1;
    double x6 = x4 * x2;
    double c = t1 + x4 * c2;
    return (float)(c + x6 * t2);
}
PT_DEV uint32_t pt_abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
PT_DEV void pt_sincosf(float y, float* s_out, float* c_out) {
    const uint32_t top = pt_abstop12(y);
    double x = (double)y;
    if (top < 0x3f4u) {                      // |y| < pi/4   (abstop12(0x1.921FB6p-1f))
        double x2 = x * x;
        if (top < 0x398u) { *s_out = y; *c_out = 1.0f; return; }      // |y| < 2^-12
        *s_out = pt_sincos_poly(x, x2, false, 0);
        *c_out = pt_sincos_poly(x, x2, false, 1);
        return;
    }
    if (top < 0x42fu) {                      // |y| < 120
        const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
        double r = x * hpi_inv;
        int n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * hpi;
        double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;       // sign[n & 3] = {1,-1,-1,1}
        bool neg = (n & 2) != 0;
        *s_out = pt_sincos_poly(x * sg, x * x, neg, n);
        *c_out = pt_sincos_poly(x * sg, x * x, neg, n ^ 1);
        return;
    }
    *s_out = (float)sin(x);
    *c_out = (float)cos(x);
}

// spectrum
PT_DEV float lum_y(V3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; }
PT_DEV bool is_black(V3 c) { return c.x == 0.0f && c.y == 0.0f && c.z == 0.0f; }
PT_DEV V3 rgb_to_xyz(V3 c) {
    return mk3(0.412453f * c.x + 0.357580f * c.y + 0.180423f * c.z, 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z,
               0.019334f * c.x + 0.119193f * c.y + 0.950227f * c.z);
}
PT_DEV V3 xyz_to_rgb(V3 c) {
    return mk3(3.240479f * c.x - 1.537150f * c.y - 0.498535f * c.z, -0.969256f * c.x + 1.875991f * c.y + 0.041556f * c.z,
               0.055648f * c.x - 0.204043f * c.y + 1.057311f * c.z);
}

// core/sampling/sampling.rs:109-159
PT_DEV V2 uniform_sample_triangle(V2 u) { float su0 = sqrtf(u.x); return mk2(1.0f - su0, u.y * su0); }
PT_DEV V2 concentric_sample_disk(V2 u) {
    float ox = u.x * 2.0f - 1.0f, oy = u.y * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return mk2(0.0f, 0.0f);
    float r, theta;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = PT_PI_OVER_4 * (oy / ox); }
    else { r = oy; theta = PT_PI_OVER_2 - PT_PI_OVER_4 * (ox / oy); }
    float sn, cs;
    pt_sincosf(theta, &sn, &cs);
    return mk2(r * cs, r * sn);
}
PT_DEV V3 cosine_sample_hemisphere(V2 u) {
    V2 d = concentric_sample_disk(u);
    float z = sqrtf(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return mk3(d.x, d.y, z);
}
PT_DEV float power_heuristic(float f_pdf, float g_pdf) {   // nf = ng = 1
    float f = 1.0f * f_pdf, g = 1.0f * g_pdf;
    return (f * f) / (f * f + g * g);
}
