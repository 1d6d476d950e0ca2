// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_bxdf.hpp: the BxDFs and microfacet distribution behind the materials of the accelerated path.
//   follows src/core/reflection/{bxdf,lambertian,oren_nayar,specular,fresnel,microfacet,fresnel_blend,math}.rs,
//           src/core/distribution/{microfacet,trowbridge_reitz}.rs
#pragma once
#include "orc_sampling.hpp"

namespace orc {

// ---- BxDF flags (core/reflection/bxdf.rs:8-14)
enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };

enum LobeKind {
    LOBE_LAMBERT = 0,        // LambertianReflection   lambertian.rs:16-47
    LOBE_OREN_NAYAR = 1,     // OrenNayar              oren_nayar.rs:12-70
    LOBE_SPEC_REFL = 2,      // SpecularReflection     specular.rs:21-57
    LOBE_SPEC_TRANS = 3,     // SpecularTransmission   specular.rs:59-124
    LOBE_FRESNEL_SPEC = 4,   // FresnelSpecular        fresnel.rs:118-216
    LOBE_MF_REFL = 5,        // MicrofacetReflection   microfacet.rs:10-100
    LOBE_MF_TRANS = 6,       // MicrofacetTransmission microfacet.rs:102-263
    LOBE_FRESNEL_BLEND = 7   // FresnelBlend           fresnel_blend.rs:10-113
};
enum FresnelKind { FR_NOOP = 0, FR_DIELECTRIC = 1, FR_CONDUCTOR = 2 };

inline RGB operator-(RGB a, RGB b) { return RGB(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); }
inline RGB operator/(RGB a, RGB b) { return RGB(a.c[0] / b.c[0], a.c[1] / b.c[1], a.c[2] / b.c[2]); }
inline RGB rgb_sqrt(RGB a) { return RGB(std::sqrt(a.c[0]), std::sqrt(a.c[1]), std::sqrt(a.c[2])); }
inline RGB rgb_clamp_zero(RGB a) {          // f32::clamp(0, inf): NaN stays NaN
    auto cl = [](Float v) { return v < 0.0f ? 0.0f : (v > kInfinity ? kInfinity : v); };
    return RGB(cl(a.c[0]), cl(a.c[1]), cl(a.c[2]));
}

// ---- reflection/math.rs
inline bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; }
inline Float cos_theta(V3 w) { return w.z; }
inline Float cos2_theta(V3 w) { return w.z * w.z; }
inline Float abs_cos_theta(V3 w) { return std::fabs(w.z); }
inline Float sin2_theta(V3 w) { return fmax_(0.0f, 1.0f - cos2_theta(w)); }
inline Float sin_theta(V3 w) { return std::sqrt(sin2_theta(w)); }
inline Float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
inline Float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
inline Float cos_phi(V3 w) { Float s = sin_theta(w); return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f); }
inline Float sin_phi(V3 w) { Float s = sin_theta(w); return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f); }
inline Float cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
inline Float sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
inline V3 reflect(V3 wo, V3 n) { return (2.0f * dot(wo, n)) * n + (-wo); }
inline bool refract(V3 wi, V3 n, Float eta, V3* wt) {
    Float cos_theta_i = dot(n, wi);
    Float sin2_theta_i = fmax_(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    Float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0f) return false;
    Float cos_theta_t = std::sqrt(1.0f - sin2_theta_t);
    *wt = eta * (-wi) + (eta * cos_theta_i - cos_theta_t) * n;
    return true;
}

// ---- fresnel.rs:16-82
inline Float fr_dielectric(Float cos_theta_i, Float eta_i, Float eta_t) {
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    bool entering = cos_theta_i > 0.0f;
    if (!entering) {
        std::swap(eta_i, eta_t);
        cos_theta_i = std::fabs(cos_theta_i);
    }
    Float sin_theta_i = std::sqrt(fmax_(0.0f, 1.0f - cos_theta_i * cos_theta_i));
    Float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    Float cos_theta_t = std::sqrt(fmax_(0.0f, 1.0f - sin_theta_t * sin_theta_t));
    Float rparl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    Float rperp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (rparl * rparl + rperp * rperp) / 2.0f;
}
inline RGB fr_conductor(Float cos_theta_i, RGB eta_i, RGB eta_t, RGB k) {
    cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
    RGB eta = eta_t / eta_i;
    RGB etak = k / eta_i;
    Float cos_theta_i2 = cos_theta_i * cos_theta_i;
    Float sin_theta_i2 = 1.0f - cos_theta_i2;
    Float sin_theta_i2_2 = sin_theta_i2 * sin_theta_i2;
    RGB eta2 = eta * eta;
    RGB etak2 = etak * etak;
    RGB c2(cos_theta_i2), s2(sin_theta_i2), s22(sin_theta_i2_2);
    RGB t0 = eta2 - etak2 - s2;
    RGB a2plusb2 = rgb_sqrt(t0 * t0 + eta2 * etak2 * 4.0f);
    RGB t1 = a2plusb2 + c2;
    RGB a = rgb_sqrt((a2plusb2 + t0) * 0.5f);
    RGB t2 = RGB(cos_theta_i) * a * 2.0f;
    RGB rs = (t1 - t2) / (t1 + t2);
    RGB t3 = c2 * a2plusb2 + s22;
    RGB t4 = t2 * s2;
    RGB rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * 0.5f;
}

// ---- TrowbridgeReitzDistribution (distribution/trowbridge_reitz.rs), samplevis = true as every material builds it
struct TRDist {
    Float alphax = 0.001f, alphay = 0.001f;
    void init(Float ax, Float ay) { alphax = fmax_(0.001f, ax); alphay = fmax_(0.001f, ay); }
    static Float roughness_to_alpha(Float roughness) {        // :104-113
        roughness = fmax_(roughness, 1e-3f);
        Float x = std::log(roughness);
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    }
    Float d(V3 wh) const {                                     // :145-159
        Float t2 = tan2_theta(wh);
        if (std::isinf(t2)) return 0.0f;
        Float c2 = cos2_theta(wh);
        Float cos4 = c2 * c2;
        Float e = (cos2_phi(wh) / (alphax * alphax) + sin2_phi(wh) / (alphay * alphay)) * t2;
        Float e2 = (1.0f + e) * (1.0f + e);
        return 1.0f / (kPi * alphax * alphay * cos4 * e2);
    }
    Float lambda(V3 w) const {                                 // :161-174
        Float abs_tan = std::fabs(tan_theta(w));
        if (std::isinf(abs_tan)) return 0.0f;
        Float alpha = std::sqrt(cos2_phi(w) * alphax * alphax + sin2_phi(w) * alphay * alphay);
        Float a2t2 = (alpha * abs_tan) * (alpha * abs_tan);
        return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
    }
    Float g1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
    Float g(V3 wo, V3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    static void sample_11(Float cos_t, Float u1, Float u2, Float* sx, Float* sy) {      // :7-64
        if (cos_t > 0.9999f) {
            Float r = std::sqrt(u1 / (1.0f - u1));
            Float phi = 2.0f * kPi * u2;
            Float sin_p = std::sin(phi), cos_p = std::cos(phi);
            *sx = r * cos_p;
            *sy = r * sin_p;
            return;
        }
        Float sin_t = std::sqrt(fmax_(0.0f, 1.0f - cos_t * cos_t));
        Float tan_t = sin_t / cos_t;
        Float a = 1.0f / tan_t;
        Float g1 = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (a * a)));
        a = 2.0f * u1 / g1 - 1.0f;
        Float tmp = fmin_(1e10f, 1.0f / (a * a - 1.0f));
        Float b = tan_t;
        Float dd = std::sqrt(fmax_(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
        Float slope_x_1 = b * tmp - dd;
        Float slope_x_2 = b * tmp + dd;
        Float slope_x = (a < 0.0f || slope_x_2 > 1.0f / tan_t) ? slope_x_1 : slope_x_2;
        Float s;
        if (u2 > 0.5f) { s = 1.0f; u2 = 2.0f * (u2 - 0.5f); }
        else { s = -1.0f; u2 = 2.0f * (0.5f - u2); }
        Float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *sx = slope_x;
        *sy = s * z * std::sqrt(1.0f + slope_x * slope_x);
    }
    V3 sample_wh(V3 wo, V2 u) const {                          // :66-98, :176-213 (samplevis branch)
        bool flip = wo.z < 0.0f;
        if (flip) wo = -wo;
        V3 ws = normalize(V3(alphax * wo.x, alphay * wo.y, wo.z));
        Float sx, sy;
        sample_11(cos_theta(ws), u.x, u.y, &sx, &sy);
        Float tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;
        sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
        sx = tmp;
        sx *= alphax;
        sy *= alphay;
        V3 wh = normalize(V3(-sx, -sy, 1.0f));
        if (flip) wh = -wh;
        return wh;
    }
    Float pdf(V3 wo, V3 wh) const { return d(wh) * g1(wo) * abs_dot(wo, wh) / abs_cos_theta(wo); }   // :215-221
};

struct Lobe {
    int kind = LOBE_LAMBERT;
    RGB r;                       // R (reflection lobes), T (transmission lobes), Rd (FresnelBlend)
    RGB t;                       // T of FresnelSpecular, Rs of FresnelBlend
    Float a = 0, b = 0;          // OrenNayar coefficients
    TRDist dist;                 // microfacet lobes
    int fresnel = FR_NOOP;       // Fresnel object of SpecularReflection / MicrofacetReflection
    Float fr_eta_i = 1, fr_eta_t = 1;
    RGB fr_cond_eta, fr_cond_k;
    Float eta_a = 1, eta_b = 1;  // transmission lobes / FresnelSpecular
    uint32_t type() const {
        switch (kind) {
            case LOBE_LAMBERT: case LOBE_OREN_NAYAR: return BSDF_REFLECTION | BSDF_DIFFUSE;
            case LOBE_SPEC_REFL: return BSDF_REFLECTION | BSDF_SPECULAR;
            case LOBE_SPEC_TRANS: return BSDF_TRANSMISSION | BSDF_SPECULAR;
            case LOBE_FRESNEL_SPEC: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
            case LOBE_MF_TRANS: return BSDF_TRANSMISSION | BSDF_GLOSSY;
            default: return BSDF_REFLECTION | BSDF_GLOSSY;
        }
    }
    bool matches(uint32_t t_) const { uint32_t tp = type(); return (tp & t_) == tp; }
    RGB fresnel_eval(Float cos_i) const {
        if (fresnel == FR_DIELECTRIC) return RGB(fr_dielectric(cos_i, fr_eta_i, fr_eta_t));
        if (fresnel == FR_CONDUCTOR) return fr_conductor(std::fabs(cos_i), RGB(1.0f), fr_cond_eta, fr_cond_k);
        return RGB(1.0f);
    }
};

inline Float pow5(Float v) { return std::pow(v, 5.0f); }      // fresnel_blend.rs:16-18 (powf)

inline RGB lobe_f(const Lobe& l, V3 wo, V3 wi) {
    switch (l.kind) {
        case LOBE_LAMBERT: return l.r * kInvPi;
        case LOBE_OREN_NAYAR: {
            Float sin_theta_i = sin_theta(wi), sin_theta_o = sin_theta(wo);
            Float max_cos = 0.0f;
            if (sin_theta_i > 1e-4f && sin_theta_o > 1e-4f) {
                Float sin_phi_i = sin_phi(wi), cos_phi_i = cos_phi(wi);
                Float sin_phi_o = sin_phi(wo), cos_phi_o = cos_phi(wo);
                Float d_cos = cos_phi_i * cos_phi_o + sin_phi_i * sin_phi_o;
                max_cos = fmax_(0.0f, d_cos);
            }
            Float sin_alpha, tan_beta;
            if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sin_theta_o; tan_beta = sin_theta_i / abs_cos_theta(wi); }
            else { sin_alpha = sin_theta_i; tan_beta = sin_theta_o / abs_cos_theta(wo); }
            return (l.r * kInvPi) * (l.a + l.b * max_cos * sin_alpha * tan_beta);
        }
        case LOBE_MF_REFL: {
            Float cos_theta_o = abs_cos_theta(wo), cos_theta_i = abs_cos_theta(wi);
            V3 wh = wi + wo;
            if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return RGB();
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB();
            wh = normalize(wh);
            RGB f = l.fresnel_eval(dot(wi, face_forward(wh, V3(0.0f, 0.0f, 1.0f))));
            return l.r * f * (l.dist.d(wh) * l.dist.g(wo, wi) / (4.0f * cos_theta_i * cos_theta_o));
        }
        case LOBE_MF_TRANS: {
            if (same_hemisphere(wo, wi)) return RGB();
            Float cos_theta_o = cos_theta(wo), cos_theta_i = cos_theta(wi);
            if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return RGB();
            Float eta = cos_theta_o > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
            V3 wh = normalize(wo + (wi * eta));
            if (wh.z < 0.0f) wh = -wh;
            Float wo_wh = dot(wo, wh), wi_wh = dot(wi, wh);
            if (wo_wh * wi_wh > 0.0f) return RGB();
            RGB f = RGB(fr_dielectric(dot(wo, wh), l.eta_a, l.eta_b));
            Float sqrt_denom = wo_wh + eta * wi_wh;
            Float factor = 1.0f / eta;       // TransportMode::Radiance
            Float d = std::fabs(l.dist.d(wh) * l.dist.g(wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor /
                                (cos_theta_i * cos_theta_o * sqrt_denom * sqrt_denom));
            return ((RGB(1.0f) - f) * l.r) * d;
        }
        case LOBE_FRESNEL_BLEND: {
            RGB diffuse = l.r * (RGB(1.0f) - l.t) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo))) *
                          (28.0f / (23.0f * kPi));
            V3 wh = wi + wo;
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return RGB();
            wh = normalize(wh);
            RGB schlick = l.t + (RGB(1.0f) - l.t) * pow5(1.0f - dot(wi, wh));
            RGB specular = schlick * (l.dist.d(wh) / (4.0f * abs_dot(wi, wh) * fmax_(abs_cos_theta(wi), abs_cos_theta(wo))));
            return diffuse + specular;
        }
        default: return RGB();    // specular lobes
    }
}

inline Float lobe_pdf(const Lobe& l, V3 wo, V3 wi) {
    switch (l.kind) {
        case LOBE_LAMBERT: case LOBE_OREN_NAYAR:              // bxdf.rs:88-94
            return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f;
        case LOBE_MF_REFL: {
            if (!same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            if (dot(wo, wh) < 0.0f) return 0.0f;
            return l.dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
        }
        case LOBE_MF_TRANS: {
            if (same_hemisphere(wo, wi)) return 0.0f;
            Float eta = cos_theta(wo) > 0.0f ? l.eta_b / l.eta_a : l.eta_a / l.eta_b;
            V3 wh = normalize(wo + (wi * eta));
            Float wo_wh = dot(wo, wh), wi_wh = dot(wi, wh);
            if (wo_wh * wi_wh > 0.0f) return 0.0f;
            Float sqrt_denom = wo_wh + eta * wi_wh;
            Float dwh_dwi = std::fabs((eta * eta * wi_wh) / (sqrt_denom * sqrt_denom));
            return l.dist.pdf(wo, wh) * dwh_dwi;
        }
        case LOBE_FRESNEL_BLEND: {
            if (!same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            if (dot(wo, wh) < 0.0f) return 0.0f;
            Float pdf_wh = l.dist.pdf(wo, wh);
            return 0.5f * (abs_cos_theta(wi) * kInvPi + pdf_wh / (4.0f * dot(wo, wh)));
        }
        default: return 0.0f;
    }
}

// BxDF::sample_f; sampled_type is the BxDF-reported type (0 = "use get_type()")
inline bool lobe_sample_f(const Lobe& l, V3 wo, V2 u, RGB* f, V3* wi_out, Float* pdf, uint32_t* sampled_type) {
    *sampled_type = 0;
    switch (l.kind) {
        case LOBE_LAMBERT: case LOBE_OREN_NAYAR: {            // bxdf.rs:74-86
            V3 wi = cosine_sample_hemisphere(u);
            if (wo.z < 0.0f) wi.z *= -1.0f;
            *pdf = lobe_pdf(l, wo, wi);
            *f = lobe_f(l, wo, wi);
            *wi_out = wi;
            return true;
        }
        case LOBE_SPEC_REFL: {
            V3 wi(-wo.x, -wo.y, wo.z);
            *pdf = 1.0f;
            *f = (l.fresnel_eval(cos_theta(wi)) * l.r) / abs_cos_theta(wi);
            *wi_out = wi;
            return true;
        }
        case LOBE_SPEC_TRANS: {
            bool entering = cos_theta(wo) > 0.0f;
            Float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
            V3 wi;
            if (!refract(wo, face_forward(V3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, &wi)) return false;
            *pdf = 1.0f;
            RGB ft = l.r * (RGB(1.0f) - RGB(fr_dielectric(cos_theta(wi), l.eta_a, l.eta_b)));
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
            *f = ft / abs_cos_theta(wi);
            *wi_out = wi;
            return true;
        }
        case LOBE_FRESNEL_SPEC: {
            Float fr = fr_dielectric(cos_theta(wo), l.eta_a, l.eta_b);
            if (u.x < fr) {
                V3 wi(-wo.x, -wo.y, wo.z);
                *pdf = fr;
                *sampled_type = BSDF_SPECULAR | BSDF_REFLECTION;
                *f = l.r * (fr / abs_cos_theta(wi));
                *wi_out = wi;
                return true;
            }
            bool entering = cos_theta(wo) > 0.0f;
            Float eta_i = entering ? l.eta_a : l.eta_b, eta_t = entering ? l.eta_b : l.eta_a;
            V3 wi;
            if (!refract(wo, face_forward(V3(0.0f, 0.0f, 1.0f), wo), eta_i / eta_t, &wi)) return false;
            RGB ft = l.t * (1.0f - fr);
            ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));
            *sampled_type = BSDF_SPECULAR | BSDF_TRANSMISSION;
            *pdf = 1.0f - fr;
            *f = ft / abs_cos_theta(wi);
            *wi_out = wi;
            return true;
        }
        case LOBE_MF_REFL: {
            if (wo.z == 0.0f) return false;
            V3 wh = l.dist.sample_wh(wo, u);
            if (dot(wo, wh) < 0.0f) return false;
            V3 wi = reflect(wo, wh);
            if (!same_hemisphere(wo, wi)) return false;
            Float p = l.dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
            if (p == 0.0f) return false;
            *f = lobe_f(l, wo, wi);
            *pdf = p;
            *wi_out = wi;
            return true;
        }
        case LOBE_MF_TRANS: {
            if (wo.z == 0.0f) return false;
            V3 wh = l.dist.sample_wh(wo, u);
            if (dot(wo, wh) < 0.0f) return false;
            Float eta = cos_theta(wo) > 0.0f ? l.eta_a / l.eta_b : l.eta_b / l.eta_a;
            V3 wi;
            if (!refract(wo, wh, eta, &wi)) return false;
            Float p = lobe_pdf(l, wo, wi);
            if (!(p > 0.0f)) return false;
            *f = lobe_f(l, wo, wi);
            *pdf = p;
            *wi_out = wi;
            return true;
        }
        case LOBE_FRESNEL_BLEND: {
            Float ux = u.x, uy = u.y;
            V3 wi;
            if (ux < 0.5f) {
                ux = fmin_(2.0f * ux, kOneMinusEpsilon);
                wi = cosine_sample_hemisphere(V2(ux, uy));
                if (wo.z < 0.0f) wi.z *= -1.0f;
            } else {
                ux = fmin_(2.0f * (ux - 0.5f), kOneMinusEpsilon);
                V3 wh = l.dist.sample_wh(wo, V2(ux, uy));
                wi = reflect(wo, wh);
                if (!same_hemisphere(wo, wi)) return false;
            }
            Float p = lobe_pdf(l, wo, wi);
            if (!(p > 0.0f)) return false;
            *f = lobe_f(l, wo, wi);
            *pdf = p;
            *wi_out = wi;
            return true;
        }
    }
    return false;
}

}  // namespace orc
