/* Checks the fdlibm restatements used on the device (pt_device_math.h: pt_acosf, pt_atanf, pt_atan2f)
 * against the host libm the reference's f32::acos / f32::atan2 resolve to.
 *   gcc -O2 -ffp-contract=off -o /tmp/chk tools/check_atan_acos_port.c -lm && /tmp/chk
 * acosf: every float in [-1, 1]; atanf: every finite float; atan2f: 6e8 pseudo-random pairs
 * (half arbitrary bit patterns, half in [-1,1]^2 with small-magnitude variants).  Expect 0 mismatches
 * on glibc 2.35 (about 2 minutes). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t fu(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float uf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define PT_DEV static inline
#define __float_as_uint fu
#define __uint_as_float uf
/* ---- verbatim copy of the three functions in pbrt-r3_amd/csrc/pt_device_math.h ---- */
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


static uint64_t st = 88172645463325252ull;
static inline uint64_t xs(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
int main(void) {
    uint64_t bad = 0;
    for (uint64_t u = 0; u <= 0x3f800000u; u++)
        for (int sg = 0; sg < 2; sg++) { float x = uf((uint32_t)u | (sg ? 0x80000000u : 0)); if (fu(pt_acosf(x)) != fu(acosf(x))) bad++; }
    printf("acosf mismatches %llu\n", (unsigned long long)bad);
    uint64_t bad2 = 0;
    for (uint64_t u = 0; u < 0x7f800000u; u++)
        for (int sg = 0; sg < 2; sg++) { float x = uf((uint32_t)u | (sg ? 0x80000000u : 0)); if (fu(pt_atanf(x)) != fu(atanf(x))) bad2++; }
    printf("atanf mismatches %llu\n", (unsigned long long)bad2);
    uint64_t bad3 = 0;
    for (uint64_t i = 0; i < 600000000ull; i++) {
        uint64_t r = xs();
        float x, y;
        if (i & 1) { x = uf((uint32_t)r); y = uf((uint32_t)(r >> 32)); }
        else {
            x = (float)((double)(r & 0xffffff) / 8388608.0 - 1.0); y = (float)((double)((r >> 32) & 0xffffff) / 8388608.0 - 1.0);
            if (i & 2) x *= 1e-3f;
            if (i & 4) y *= 1e-4f;
        }
        float a = pt_atan2f(y, x), b = atan2f(y, x);
        if (fu(a) != fu(b) && !(a != a && b != b)) bad3++;
    }
    printf("atan2f mismatches %llu\n", (unsigned long long)bad3);
    return (bad || bad2 || bad3) ? 1 : 0;
}
