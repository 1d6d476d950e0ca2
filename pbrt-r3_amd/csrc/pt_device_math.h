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

// acosf / atanf / atan2f.  Same story as sinf: Rust's f32::acos / atan2 are glibc's, which (2.35, the
// image's) are the fdlibm single-precision routines in plain f32 arithmetic, not correctly rounded.
// Restated from the published fdlibm algorithm (e_acosf.c, s_atanf.c, e_atan2f.c) with its
// coefficients; tools/check_atan_acos_port.c compares them against libm.so.6 over every float
// (acosf on [-1,1], atanf on all finite inputs) and 6e8 atan2f pairs: 0 mismatches.
PT_DEV float pt_acosf(float x) {
    const float one = 1.0f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f, qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f,
                qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        float z = x * x;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {
        float z = (one + x) * 0.5f;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float s = sqrtf(z);
        float r = p / q;
        float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    float z = (one - x) * 0.5f;
    float s = sqrtf(z);
    float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    float c = (z - df * df) / (s + df);
    float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    float r = p / q;
    float w = r * s + c;
    return 2.0f * (df + w);
}
PT_DEV float pt_atanf(float x) {
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const float a0 = 3.3333334327e-01f, a1 = -2.0000000298e-01f, a2 = 1.4285714924e-01f, a3 = -1.1111110449e-01f, a4 = 9.0908870101e-02f,
                a5 = -7.6918758452e-02f, a6 = 6.6610731184e-02f, a7 = -5.8335702866e-02f, a8 = 4.9768779427e-02f, a9 = -3.6531571299e-02f,
                a10 = 1.6285819933e-02f;
    int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                 // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    float s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    float hi = id == 0 ? hi0 : (id == 1 ? hi1 : (id == 2 ? hi2 : hi3));
    float lo = id == 0 ? lo0 : (id == 1 ? lo1 : (id == 2 ? lo2 : lo3));
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}
PT_DEV float pt_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    int32_t hx = (int32_t)__float_as_uint(x), hy = (int32_t)__float_as_uint(y), ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return pt_atanf(y);
    int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = pt_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// log2f: glibc's (ARM optimized-routines log2f: 16-entry table of 1/c and log2(c), degree-4 polynomial in f64, one final
// rounding), restated with its published constants; compared with libm.so.6 over every positive float: 0 mismatches
// (tools/check_log2f_port.c).
PT_DEV float pt_log2f(float x) {
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, {0x1.49539f0f010bp+0, -0x1.7418b0a1fb77bp-2},
        {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8eap+0, -0x1.97c1d1b3b7afp-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5},
        {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4}, {0x1.ca4b31f026aap-1, 0x1.476a9543891bap-3},
        {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},
        {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
    const double A0 = -0x1.712b6f70a7e4dp-2, A1 = 0x1.ecabf496832ep-2, A2 = -0x1.715479ffae3dep-1, A3 = 0x1.715475f35c8b8p0;
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -PT_INF;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __uint_as_float(0x7fc00000u);
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    int i = (int)((tmp >> 19) % 16u);
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)tmp >> 23;
    double invc = T[i][0], logc = T[i][1];
    double z = (double)__uint_as_float(iz);
    double r = z * invc - 1.0;
    double y0 = logc + (double)k;
    double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    double p = A3 * r + y0;
    y = y * r2 + p;
    return (float)y;
}

// logf: glibc's (ARM optimized-routines logf: 16-entry table of 1/c and ln(c), degree-3 polynomial in f64, one final
// rounding), restated with its published constants; compared with libm.so.6 over every positive float: 0 mismatches, with and
// without the FMA contractions of glibc's x86-64 variant (tools/check_logf_port.c).
PT_DEV float pt_logf(float x) {
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
        {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
        {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
        {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, LN2 = 0x1.62e42fefa39efp-1;
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2u == 0u) return -PT_INF;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __uint_as_float(0x7fc00000u);
        ix = __float_as_uint(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * LN2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
// TrowbridgeReitzDistribution::roughness_to_alpha (core/distribution/trowbridge_reitz.rs:113-121)
PT_DEV float pt_roughness_to_alpha(float roughness) {
    roughness = roughness > 1e-3f ? roughness : 1e-3f;       // f32::max: a NaN becomes 1e-3
    const float x = pt_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
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
