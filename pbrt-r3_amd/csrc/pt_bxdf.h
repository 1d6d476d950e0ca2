// pt_bxdf.h -- device BxDFs of the non-matte materials and the multi-lobe BSDF over them.
//
//   lobes        src/core/reflection/{specular,fresnel,microfacet,fresnel_blend,math}.rs
//   distribution src/core/distribution/trowbridge_reitz.rs (visible-normal sampling, as every material asks)
//   BSDF         src/core/reflection/bsdf.rs:92-270
//
// Parameter textures are constant, so Material::compute_scattering_functions yields the same lobe list
// at every hit of a material: the host builds it once (build_lobes, pt_context.cpp) and the kernels walk
// the PtLobe records.  Arithmetic is kept operation-for-operation as in the reference (no contraction,
// IEEE divide/sqrt); sin/cos and powf(x, 5) restate glibc's algorithms so results match the CPU bit for bit.
#pragma once
#include "pt_device.h"
#include "pt_device_math.h"

#define PT_BSDF_REFLECTION 1u
#define PT_BSDF_TRANSMISSION 2u
#define PT_BSDF_DIFFUSE 4u
#define PT_BSDF_GLOSSY 8u
#define PT_BSDF_SPECULAR 16u
#define PT_BSDF_ALL 31u

// ---- powf(x, 5.0f) as glibc computes it (sysdeps/ieee754/flt-32/e_powf.c, FMA build: the variant its
// ifunc picks on every x86-64 CPU with FMA): log2 via a 16-entry table + degree-5 polynomial, times 5,
// exp2 via a 32-entry table + cubic, all in double.  Tables are glibc's __powf_log2_data / __exp2f_data.
__device__ const double pt_powf_log2_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2},
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2},
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4},
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3},
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2},
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}};
__device__ const unsigned long long pt_exp2f_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
PT_DEV float pt_pow5(float x) {
    uint32_t ix = __float_as_uint(x), sign_bias = 0;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {      // zero, inf, nan, negative or subnormal
        if ((ix << 1) == 0) return x;
        if (ix == 0x7f800000u || ix == 0xff800000u) return x;
        if ((ix << 1) > 0xff000000u) return x + x;
        if (ix & 0x80000000u) { sign_bias = 1u << 16; ix &= 0x7fffffffu; }    // 5 is an odd integer
        if (ix < 0x00800000u) {
            ix = __float_as_uint(__uint_as_float(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    int k = (int32_t)top >> 23;
    double z = (double)__uint_as_float(ix - top);
    double r = __builtin_fma(z, pt_powf_log2_tab[i][0], -1.0);
    double y0 = pt_powf_log2_tab[i][1] + (double)k;
    double r2 = r * r;
    double y = __builtin_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    double p = __builtin_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    double r4 = r2 * r2;
    double q = __builtin_fma(0x1.71547652ab82bp+0, r, y0);
    q = __builtin_fma(p, r2, q);
    y = __builtin_fma(y, r4, q);
    double ylogx = 5.0 * y;
    if ((((unsigned long long)__double_as_longlong(ylogx) >> 47) & 0xffffull) >= (0x405f800000000000ull >> 47)) {   // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -PT_INF : PT_INF;
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
    }
    double kd = ylogx + 0x1.8p+47;
    unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd -= 0x1.8p+47;
    double rr = ylogx - kd;
    unsigned long long t = pt_exp2f_tab[ki & 31u];
    t += (ki + sign_bias) << 47;
    double s = __longlong_as_double((long long)t);
    double zz = __builtin_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    double rr2 = rr * rr;
    double yy = __builtin_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    yy = yy * s;
    return (float)yy;
}

// ---- reflection/math.rs
PT_DEV float bx_cos2_theta(V3 w) { return w.z * w.z; }
PT_DEV float bx_sin2_theta(V3 w) { return fmaxf(0.0f, 1.0f - w.z * w.z); }
PT_DEV float bx_sin_theta(V3 w) { return sqrtf(bx_sin2_theta(w)); }
PT_DEV float bx_tan_theta(V3 w) { return bx_sin_theta(w) / w.z; }
PT_DEV float bx_tan2_theta(V3 w) { return bx_sin2_theta(w) / bx_cos2_theta(w); }
PT_DEV float bx_cos_phi(V3 w) { float s = bx_sin_theta(w); return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
PT_DEV float bx_sin_phi(V3 w) { float s = bx_sin_theta(w); return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
PT_DEV bool bx_same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }
PT_DEV V3 bx_reflect(V3 wo, V3 n) { return (2.0f * dot(wo, n)) * n + (-wo); }
PT_DEV bool bx_refract(V3 wi, V3 n, float eta, V3* wt) {
    float cos_theta_i = dot(n, wi);
    float sin2_theta_i = fmaxf(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0f) return false;
    float cos_theta_t = sqrtf(1.0f - sin2_theta_t);
    *wt = eta * (-wi) + (eta * cos_theta_i - cos_theta_t) * n;
    return true;
}
PT_DEV V3 rgb_div(V3 a, V3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV V3 rgb_sqrt(V3 a) { return mk3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
PT_DEV V3 rgb1(float v) { return mk3(v, v, v); }

// ---- fresnel.rs:16-82
PT_DEV float fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    if (!(cos_theta_i > 0.0f)) {
        float t = eta_i; eta_i = eta_t; eta_t = t;
        cos_theta_i = fabsf(cos_theta_i);
    }
    float sin_theta_i = sqrtf(fmaxf(0.0f, 1.0f - cos_theta_i * cos_theta_i));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = sqrtf(fmaxf(0.0f, 1.0f - sin_theta_t * sin_theta_t));
    float rparl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    float rperp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (rparl * rparl + rperp * rperp) / 2.0f;
}
PT_DEV V3 fr_conductor(float cos_theta_i, V3 eta_t, V3 k) {     // eta_i = 1 (metal.rs:70)
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    V3 one = rgb1(1.0f);
    V3 eta = rgb_div(eta_t, one), etak = rgb_div(k, one);
    float cos2 = cos_theta_i * cos_theta_i;
    float sin2 = 1.0f - cos2;
    float sin2_2 = sin2 * sin2;
    V3 eta2 = eta * eta, etak2 = etak * etak;
    V3 c2 = rgb1(cos2), s2 = rgb1(sin2), s22 = rgb1(sin2_2);
    V3 t0 = eta2 - etak2 - s2;
    V3 a2plusb2 = rgb_sqrt(t0 * t0 + eta2 * etak2 * 4.0f);
    V3 t1 = a2plusb2 + c2;
    V3 a = rgb_sqrt((a2plusb2 + t0) * 0.5f);
    V3 t2 = rgb1(cos_theta_i) * a * 2.0f;
    V3 rs = rgb_div(t1 - t2, t1 + t2);
    V3 t3 = c2 * a2plusb2 + s22;
    V3 t4 = t2 * s2;
    V3 rp = rgb_div(rs * (t3 - t4), t3 + t4);
    return (rp + rs) * 0.5f;
}
PT_DEV V3 lobe_fresnel(const PtLobe& l, float cos_i) {
    if (l.fresnel == PT_FR_DIELECTRIC) return rgb1(fr_dielectric(cos_i, l.fr_eta_i, l.fr_eta_t));
    if (l.fresnel == PT_FR_CONDUCTOR) return fr_conductor(fabsf(cos_i), ld3(l.t), ld3(l.k));
    return rgb1(1.0f);
}

// ---- TrowbridgeReitzDistribution (trowbridge_reitz.rs:7-98, :145-221)
PT_DEV float tr_d(float ax, float ay, V3 wh) {
    float t2 = bx_tan2_theta(wh);
    if (isinf(t2)) return 0.0f;
    float c2 = bx_cos2_theta(wh);
    float cos4 = c2 * c2;
    float cp = bx_cos_phi(wh), sp = bx_sin_phi(wh);
    float e = ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay)) * t2;
    float e2 = (1.0f + e) * (1.0f + e);
    return 1.0f / (PT_PI * ax * ay * cos4 * e2);
}
PT_DEV float tr_lambda(float ax, float ay, V3 w) {
    float abs_tan = fabsf(bx_tan_theta(w));
    if (isinf(abs_tan)) return 0.0f;
    float cp = bx_cos_phi(w), sp = bx_sin_phi(w);
    float alpha = sqrtf((cp * cp) * ax * ax + (sp * sp) * ay * ay);
    float a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
PT_DEV float tr_g1(float ax, float ay, V3 w) { return 1.0f / (1.0f + tr_lambda(ax, ay, w)); }
PT_DEV float tr_g(float ax, float ay, V3 wo, V3 wi) { return 1.0f / (1.0f + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }
PT_DEV float tr_pdf(float ax, float ay, V3 wo, V3 wh) { return tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * abs_dot(wo, wh) / fabsf(wo.z); }
PT_DEV V3 tr_sample_wh(float ax, float ay, V3 wo, V2 u) {
    bool flip = wo.z < 0.0f;
    if (flip) wo = -wo;
    V3 ws = normalize(mk3(ax * wo.x, ay * wo.y, wo.z));
    float cos_t = ws.z, u1 = u.x, u2 = u.y, sx, sy;
    if (cos_t > 0.9999f) {
        float r = sqrtf(u1 / (1.0f - u1));
        float phi = 2.0f * PT_PI * u2;
        float sin_p, cos_p;
        pt_sincosf(phi, &sin_p, &cos_p);
        sx = r * cos_p;
        sy = r * sin_p;
    } else {
        float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
        float tan_t = sin_t / cos_t;
        float a = 1.0f / tan_t;
        float g1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
        a = 2.0f * u1 / g1 - 1.0f;
        float tmp = fminf(1e10f, 1.0f / (a * a - 1.0f));
        float b = tan_t;
        float dd = sqrtf(fmaxf(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
        float slope_x_1 = b * tmp - dd;
        float slope_x_2 = b * tmp + dd;
        sx = (a < 0.0f || slope_x_2 > 1.0f / tan_t) ? slope_x_1 : slope_x_2;
        float s;
        if (u2 > 0.5f) { s = 1.0f; u2 = 2.0f * (u2 - 0.5f); }
        else { s = -1.0f; u2 = 2.0f * (0.5f - u2); }
        float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        sy = s * z * sqrtf(1.0f + sx * sx);
    }
    float cp = bx_cos_phi(ws), sp = bx_sin_phi(ws);
    float tmp = cp * sx - sp * sy;
    sy = sp * sx + cp * sy;
    sx = tmp;
    sx *= ax;
    sy *= ay;
    V3 wh = normalize(mk3(-sx, -sy, 1.0f));
    if (flip) wh = -wh;
    return wh;
}

// ---- the lobes
PT_DEV bool lobe_matches(const PtLobe& l, uint32_t flags) { return (l.type & flags) == l.type; }

__device__ __noinline__ V3 globe_f(const PtLobe& l, V3 wo, V3 wi) {
    const V3 zero = mk3(0.0f, 0.0f, 0.0f);
    switch (l.kind) {
        case PT_LOBE_LAMBERT: return ld3(l.r) * PT_INV_PI;
        case PT_LOBE_OREN_NAYAR: {
            float sti = bx_sin_theta(wi), sto = bx_sin_theta(wo);
            float max_cos = 0.0f;
            if (sti > 1e-4f && sto > 1e-4f) {
                float d_cos = bx_cos_phi(wi) * bx_cos_phi(wo) + bx_sin_phi(wi) * bx_sin_phi(wo);
                max_cos = fmaxf(0.0f, d_cos);
            }
            float sin_alpha, tan_beta;
            if (fabsf(wi.z) > fabsf(wo.z)) { sin_alpha = sto; tan_beta = sti / fabsf(wi.z); }
            else { sin_alpha = sti; tan_beta = sto / fabsf(wo.z); }
            return (ld3(l.r) * PT_INV_PI) * (l.oa + l.ob * max_cos * sin_alpha * tan_beta);
        }
        case PT_LOBE_MF_REFL: {
            float cos_o = fabsf(wo.z), cos_i = fabsf(wi.z);
            V3 wh = wi + wo;
            if (cos_i == 0.0f || cos_o == 0.0f) return zero;
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return zero;
            wh = normalize(wh);
            V3 f = lobe_fresnel(l, dot(wi, face_forward(wh, mk3(0.0f, 0.0f, 1.0f))));
            return ld3(l.r) * f * (tr_d(l.ax, l.ay, wh) * tr_g(l.ax, l.ay, wo, wi) / (4.0f * cos_i * cos_o));
        }
        case PT_LOBE_MF_TRANS: {
            if (bx_same_hemisphere(wo, wi)) return zero;
            float cos_o = wo.z, cos_i = wi.z;
            if (cos_i == 0.0f || cos_o == 0.0f) return zero;
            float eta = cos_o > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
            V3 wh = normalize(wo + (wi * eta));
            if (wh.z < 0.0f) wh = -wh;
            float wo_wh = dot(wo, wh), wi_wh = dot(wi, wh);
            if (wo_wh * wi_wh > 0.0f) return zero;
            V3 f = rgb1(fr_dielectric(dot(wo, wh), l.eta_a, l.eta_b));
            float sqrt_denom = wo_wh + eta * wi_wh;
            float factor = 1.0f / eta;
            float d = fabsf(tr_d(l.ax, l.ay, wh) * tr_g(l.ax, l.ay, wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor /
                            (cos_i * cos_o * sqrt_denom * sqrt_denom));
            return ((rgb1(1.0f) - f) * ld3(l.r)) * d;
        }
        case PT_LOBE_FRESNEL_BLEND: {
            V3 rd = ld3(l.r), rs = ld3(l.t);
            V3 diffuse = rd * (rgb1(1.0f) - rs) * (1.0f - pt_pow5(1.0f - 0.5f * fabsf(wi.z))) * (1.0f - pt_pow5(1.0f - 0.5f * fabsf(wo.z))) *
                         (28.0f / (23.0f * PT_PI));
            V3 wh = wi + wo;
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return zero;
            wh = normalize(wh);
            V3 schlick = rs + (rgb1(1.0f) - rs) * pt_pow5(1.0f - dot(wi, wh));
            V3 specular = schlick * (tr_d(l.ax, l.ay, wh) / (4.0f * abs_dot(wi, wh) * fmaxf(fabsf(wi.z), fabsf(wo.z))));
            return diffuse + specular;
        }
        default: return zero;
    }
}

__device__ __noinline__ float globe_pdf(const PtLobe& l, V3 wo, V3 wi) {
    switch (l.kind) {
        case PT_LOBE_LAMBERT: case PT_LOBE_OREN_NAYAR: return bx_same_hemisphere(wo, wi) ? fabsf(wi.z) * PT_INV_PI : 0.0f;
        case PT_LOBE_MF_REFL: {
            if (!bx_same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            if (dot(wo, wh) < 0.0f) return 0.0f;
            return tr_pdf(l.ax, l.ay, wo, wh) / (4.0f * dot(wo, wh));
        }
        case PT_LOBE_MF_TRANS: {
            if (bx_same_hemisphere(wo, wi)) return 0.0f;
            float eta = wo.z > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
            V3 wh = normalize(wo + (wi * eta));
            float wo_wh = dot(wo, wh), wi_wh = dot(wi, wh);
            if (wo_wh * wi_wh > 0.0f) return 0.0f;
            float sqrt_denom = wo_wh + eta * wi_wh;
            float dwh_dwi = fabsf((eta * eta * wi_wh) / (sqrt_denom * sqrt_denom));
            return tr_pdf(l.ax, l.ay, wo, wh) * dwh_dwi;
        }
        case PT_LOBE_FRESNEL_BLEND: {
            if (!bx_same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            if (dot(wo, wh) < 0.0f) return 0.0f;
            float pdf_wh = tr_pdf(l.ax, l.ay, wo, wh);
            return 0.5f * (fabsf(wi.z) * PT_INV_PI + pdf_wh / (4.0f * dot(wo, wh)));
        }
        default: return 0.0f;
    }
}

// BxDF::sample_f; *sampled_type = 0 means "the lobe's own type"
__device__ __noinline__ bool globe_sample_f(const PtLobe& l, V3 wo, V2 u, V3* f, V3* wi_out, float* pdf, uint32_t* sampled_type) {
    *sampled_type = 0;
    switch (l.kind) {
        case PT_LOBE_LAMBERT: case PT_LOBE_OREN_NAYAR: {
            V3 wi = cosine_sample_hemisphere(u);
            if (wo.z < 0.0f) wi.z *= -1.0f;
            *pdf = globe_pdf(l, wo, wi);
            *f = globe_f(l, wo, wi);
            *wi_out = wi;
            return true;
        }
        case PT_LOBE_SPEC_REFL: {
            V3 wi = mk3(-wo.x, -wo.y, wo.z);
            *pdf = 1.0f;
            *f = (lobe_fresnel(l, wi.z) * ld3(l.r)) / fabsf(wi.z);
            *wi_out = wi;
            return true;
        }
        case PT_LOBE_SPEC_TRANS: {
            bool entering = wo.z > 0.0f;
            float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
            V3 wi;
            if (!bx_refract(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, &wi)) return false;
            *pdf = 1.0f;
            V3 ft = ld3(l.r) * (rgb1(1.0f) - rgb1(fr_dielectric(wi.z, l.eta_a, l.eta_b)));
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
            *f = ft / fabsf(wi.z);
            *wi_out = wi;
            return true;
        }
        case PT_LOBE_FRESNEL_SPEC: {
            float fr = fr_dielectric(wo.z, l.eta_a, l.eta_b);
            if (u.x < fr) {
                V3 wi = mk3(-wo.x, -wo.y, wo.z);
                *pdf = fr;
                *sampled_type = PT_BSDF_SPECULAR | PT_BSDF_REFLECTION;
                *f = ld3(l.r) * (fr / fabsf(wi.z));
                *wi_out = wi;
                return true;
            }
            bool entering = wo.z > 0.0f;
            float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
            V3 wi;
            if (!bx_refract(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, &wi)) return false;
            V3 ft = ld3(l.t) * (1.0f - fr);
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
            *sampled_type = PT_BSDF_SPECULAR | PT_BSDF_TRANSMISSION;
            *pdf = 1.0f - fr;
            *f = ft / fabsf(wi.z);
            *wi_out = wi;
            return true;
        }
        case PT_LOBE_MF_REFL: {
            if (wo.z == 0.0f) return false;
            V3 wh = tr_sample_wh(l.ax, l.ay, wo, u);
            if (dot(wo, wh) < 0.0f) return false;
            V3 wi = bx_reflect(wo, wh);
            if (!bx_same_hemisphere(wo, wi)) return false;
            float p = tr_pdf(l.ax, l.ay, wo, wh) / (4.0f * dot(wo, wh));
            if (p == 0.0f) return false;
            *f = globe_f(l, wo, wi);
            *pdf = p;
            *wi_out = wi;
            return true;
        }
        case PT_LOBE_MF_TRANS: {
            if (wo.z == 0.0f) return false;
            V3 wh = tr_sample_wh(l.ax, l.ay, wo, u);
            if (dot(wo, wh) < 0.0f) return false;
            float eta = wo.z > 0.0f ? l.eta_a / l.eta_b : l.eta_b / l.eta_a;
            V3 wi;
            if (!bx_refract(wo, wh, eta, &wi)) return false;
            float p = globe_pdf(l, wo, wi);
            if (!(p > 0.0f)) return false;
            *f = globe_f(l, wo, wi);
            *pdf = p;
            *wi_out = wi;
            return true;
        }
        case PT_LOBE_FRESNEL_BLEND: {
            float ux = u.x;
            V3 wi;
            if (ux < 0.5f) {
                ux = fminf(2.0f * ux, PT_ONE_MINUS_EPS);
                wi = cosine_sample_hemisphere(mk2(ux, u.y));
                if (wo.z < 0.0f) wi.z *= -1.0f;
            } else {
                ux = fminf(2.0f * (ux - 0.5f), PT_ONE_MINUS_EPS);
                V3 wh = tr_sample_wh(l.ax, l.ay, wo, mk2(ux, u.y));
                wi = bx_reflect(wo, wh);
                if (!bx_same_hemisphere(wo, wi)) return false;
            }
            float p = globe_pdf(l, wo, wi);
            if (!(p > 0.0f)) return false;
            *f = globe_f(l, wo, wi);
            *pdf = p;
            *wi_out = wi;
            return true;
        }
        default: return false;
    }
}

// ---- BSDF over a material's lobe list (bsdf.rs)
struct GBsdf {
    V3 ns, ng, ss, ts;
    const PtLobe* lobes;
    uint32_t n_lobes;
};
PT_DEV V3 gw2l(const GBsdf& b, V3 v) { return mk3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
PT_DEV V3 gl2w(const GBsdf& b, V3 v) {
    return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
PT_DEV bool gfinite3(V3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }
PT_DEV uint32_t gbsdf_num_components(const GBsdf& b, uint32_t flags) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < b.n_lobes; i++) n += lobe_matches(b.lobes[i], flags) ? 1u : 0u;
    return n;
}
PT_DEV V3 gbsdf_f_local(const GBsdf& b, V3 wo, V3 wi, bool reflect, uint32_t flags) {
    V3 r = mk3(0.0f, 0.0f, 0.0f);
    for (uint32_t i = 0; i < b.n_lobes; i++) {
        const PtLobe& l = b.lobes[i];
        if (lobe_matches(l, flags) && ((reflect && (l.type & PT_BSDF_REFLECTION)) || (!reflect && (l.type & PT_BSDF_TRANSMISSION))))
            r = r + globe_f(l, wo, wi);
    }
    return r;
}
PT_DEV V3 gbsdf_f(const GBsdf& b, V3 wo_w, V3 wi_w, uint32_t flags) {           // bsdf.rs:208-236
    V3 wi = gw2l(b, wi_w), wo = gw2l(b, wo_w);
    if (wo.z == 0.0f || !gfinite3(wo)) return mk3(0.0f, 0.0f, 0.0f);
    bool reflect = (dot(wi_w, b.ng) * dot(wo_w, b.ng)) > 0.0f;
    return gbsdf_f_local(b, wo, wi, reflect, flags);
}
PT_DEV float gbsdf_pdf(const GBsdf& b, V3 wo_w, V3 wi_w, uint32_t flags) {      // bsdf.rs:238-270
    V3 wi = gw2l(b, wi_w), wo = gw2l(b, wo_w);
    if (wo.z == 0.0f || !gfinite3(wo)) return 0.0f;
    uint32_t count = 0;
    float p = 0.0f;
    for (uint32_t i = 0; i < b.n_lobes; i++)
        if (lobe_matches(b.lobes[i], flags)) { p = p + globe_pdf(b.lobes[i], wo, wi); count++; }
    return count > 0 ? p / (float)count : 0.0f;
}
PT_DEV bool gbsdf_sample_f(const GBsdf& b, V3 wo_w, V2 u, uint32_t flags, V3* f_out, V3* wi_out, float* pdf_out, uint32_t* type_out) {   // bsdf.rs:92-206
    int matching = (int)gbsdf_num_components(b, flags);
    if (matching == 0) return false;
    int comp = (int)floorf(u.x * (float)matching);
    if (comp > matching - 1) comp = matching - 1;
    uint32_t index = 0;
    int count = comp;
    for (uint32_t i = 0; i < b.n_lobes; i++) {
        if (lobe_matches(b.lobes[i], flags)) {
            if (count == 0) { index = i; break; }
            count--;
        }
    }
    const PtLobe& lb = b.lobes[index];
    V2 remapped = mk2(fminf((u.x * (float)matching) - (float)comp, PT_ONE_MINUS_EPS), u.y);
    V3 wo = gw2l(b, wo_w);
    if (wo.z == 0.0f || !gfinite3(wo)) return false;
    V3 f, wi;
    float pdf;
    uint32_t t;
    if (!globe_sample_f(lb, wo, remapped, &f, &wi, &pdf, &t)) return false;
    if (pdf <= 0.0f) return false;
    uint32_t sampled_type = t != 0 ? t : lb.type;
    V3 wi_world = gl2w(b, wi);
    if ((lb.type & PT_BSDF_SPECULAR) == 0 && matching > 1)
        for (uint32_t i = 0; i < b.n_lobes; i++)
            if (i != index && lobe_matches(b.lobes[i], flags)) pdf += globe_pdf(b.lobes[i], wo, wi);
    if (matching > 1) pdf /= (float)matching;
    if ((lb.type & PT_BSDF_SPECULAR) == 0) {
        bool reflect = (dot(wi_world, b.ng) * dot(wo_w, b.ng)) > 0.0f;
        f = gbsdf_f_local(b, wo, wi, reflect, flags);
    }
    *f_out = f; *wi_out = wi_world; *pdf_out = pdf; *type_out = sampled_type;
    return true;
}
