// pt_lobes.h -- Material::compute_scattering_functions (src/materials/*.rs) as "parameters in, BxDF list out", shared by the host
// (constant materials: once per material at upload) and the device (textured materials: at every hit, after the parameter
// textures have been evaluated).  One source, so both produce the same lobes.
#pragma once
#include "pt_device.h"
#include "../../include/pbrtgpu.h"
#ifdef __HIP__
#define PT_HD __host__ __device__
#else
#define PT_HD
#endif

// ---- Material::compute_scattering_functions for constant parameter textures (src/materials/*.rs), evaluated
// once per material: the list of BxDFs the BSDF receives, in the order the material adds them.
static const uint32_t kRefl = 1, kTrans = 2, kDiffuse = 4, kGlossy = 8, kSpecular = 16;    // bxdf.rs:8-14
PT_HD inline float clamp_zero(float v) { return v < 0.0f ? 0.0f : v; }                            // Spectrum::clamp_zero per channel
PT_HD inline bool black(const float* c) { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
PT_HD inline PtLobe* push_lobe(PtMaterial& m, uint32_t kind, uint32_t type, const float* r) {
    PtLobe* l = &m.lobes[m.n_lobes++];
    l->fresnel = 0;
    l->t[0] = l->t[1] = l->t[2] = 0.0f; l->k[0] = l->k[1] = l->k[2] = 0.0f;
    l->oa = l->ob = 0.0f;
    l->kind = kind; l->type = type;
    l->r[0] = r[0]; l->r[1] = r[1]; l->r[2] = r[2];
    l->eta_a = l->eta_b = l->fr_eta_i = l->fr_eta_t = 1.0f;
    l->ax = l->ay = 0.001f;
    return l;
}
PT_HD inline void set_distribution(PtLobe* l, float ax, float ay) {                               // TrowbridgeReitzDistribution::new
    l->ax = 0.001f > ax ? 0.001f : ax;     // f32::max: a NaN roughness becomes 0.001
    l->ay = 0.001f > ay ? 0.001f : ay;
    if (ax != ax) l->ax = 0.001f;
    if (ay != ay) l->ay = 0.001f;
}
// a_r / a_u / a_v: "roughness" / "uroughness" / "vroughness" after the optional roughness_to_alpha remap (done by the caller on
// the host: it needs logf).  Every field of `m` read afterwards is written here.
PT_HD inline void build_lobes(const pt_material& in, float a_r, float a_u, float a_v, PtMaterial& m) {
    m.type = in.type;
    m.n_lobes = 0; m.nonspecular = 0; m.oren_a = 0.0f; m.oren_b = 0.0f;      // every field read afterwards is written here
    m.spec_mask = 0;
    m.kd[0] = in.kd[0]; m.kd[1] = in.kd[1]; m.kd[2] = in.kd[2];
    m.sigma = in.sigma;
    m.bsdf_eta = 1.0f;
    m.has_bsdf = in.type != PT_MATERIAL_NONE;
    const float pick_u = in.uroughness == PT_ROUGHNESS_UNSET ? a_r : a_u;      // Metal / Uber: "uroughness" falls back to "roughness"
    const float pick_v = in.vroughness == PT_ROUGHNESS_UNSET ? a_r : a_v;
    auto cz = [](const float* c, float* out) { for (int i = 0; i < 3; i++) out[i] = clamp_zero(c[i]); };
    auto mul = [](const float* a, const float* b, float* out) { for (int i = 0; i < 3; i++) out[i] = a[i] * b[i]; };
    switch (in.type) {
        case PT_MATERIAL_MATTE: {                       // matte.rs:25-53 (Kd is not clamped here)
            float sig = in.sigma < 0.0f ? 0.0f : (in.sigma > 90.0f ? 90.0f : in.sigma);
            const float pi = 3.14159265358979323846f;
            float sigma = sig * (pi / 180.0f);
            float sigma2 = sigma * sigma;
            m.oren_a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
            m.oren_b = 0.45f * sigma2 / (sigma2 + 0.09f);
            if (!black(in.kd)) {
                PtLobe* l = push_lobe(m, sig == 0.0f ? PT_LOBE_LAMBERT : PT_LOBE_OREN_NAYAR, kRefl | kDiffuse, in.kd);
                l->oa = m.oren_a; l->ob = m.oren_b;
            }
            break;
        }
        case PT_MATERIAL_PLASTIC: {                     // plastic.rs:31-71
            float kd[3], ks[3];
            cz(in.kd, kd); cz(in.ks, ks);
            if (!black(kd)) push_lobe(m, PT_LOBE_LAMBERT, kRefl | kDiffuse, kd);
            if (!black(ks)) {
                float rough = a_r;
                PtLobe* l = push_lobe(m, PT_LOBE_MF_REFL, kRefl | kGlossy, ks);
                set_distribution(l, rough, rough);
                l->fresnel = PT_FR_DIELECTRIC; l->fr_eta_i = 1.5f; l->fr_eta_t = 1.0f;
            }
            break;
        }
        case PT_MATERIAL_MIRROR: {                      // mirror.rs:19-41
            float kr[3];
            cz(in.kr, kr);
            if (!black(kr)) push_lobe(m, PT_LOBE_SPEC_REFL, kRefl | kSpecular, kr)->fresnel = PT_FR_NOOP;
            break;
        }
        case PT_MATERIAL_GLASS: {                       // glass.rs:46-110 (Kr, Kt unclamped; allow_multiple_lobes = true)
            if (black(in.kr) && black(in.kt)) { m.has_bsdf = 0; break; }
            m.bsdf_eta = in.eta;
            if (in.uroughness == 0.0f && in.vroughness == 0.0f) {
                PtLobe* l = push_lobe(m, PT_LOBE_FRESNEL_SPEC, kRefl | kTrans | kSpecular, in.kr);
                for (int i_ = 0; i_ < 3; i_++) (l->t)[i_] = (in.kt)[i_];
                l->eta_a = 1.0f; l->eta_b = in.eta;
            } else {
                float ur = a_u, vr = a_v;
                if (!black(in.kr)) {
                    PtLobe* l = push_lobe(m, PT_LOBE_MF_REFL, kRefl | kGlossy, in.kr);
                    set_distribution(l, ur, vr);
                    l->fresnel = PT_FR_DIELECTRIC; l->fr_eta_i = 1.0f; l->fr_eta_t = in.eta;
                }
                if (!black(in.kt)) {
                    PtLobe* l = push_lobe(m, PT_LOBE_MF_TRANS, kTrans | kGlossy, in.kt);
                    set_distribution(l, ur, vr);
                    l->eta_a = 1.0f; l->eta_b = in.eta;
                }
            }
            break;
        }
        case PT_MATERIAL_METAL: {                       // metal.rs:51-85
            const float one[3] = {1.0f, 1.0f, 1.0f};
            PtLobe* l = push_lobe(m, PT_LOBE_MF_REFL, kRefl | kGlossy, one);
            set_distribution(l, pick_u, pick_v);
            l->fresnel = PT_FR_CONDUCTOR;
            for (int i_ = 0; i_ < 3; i_++) (l->t)[i_] = (in.metal_eta)[i_];
            for (int i_ = 0; i_ < 3; i_++) (l->k)[i_] = (in.metal_k)[i_];
            break;
        }
        case PT_MATERIAL_UBER: {                        // uber.rs:63-127
            float t[3], c[3], tmp[3];
            for (int i = 0; i < 3; i++) t[i] = clamp_zero(1.0f - in.opacity[i]);
            m.bsdf_eta = !black(t) ? 1.0f : in.eta;
            if (!black(t)) push_lobe(m, PT_LOBE_SPEC_TRANS, kTrans | kSpecular, t);       // eta_a = eta_b = 1
            cz(in.kd, tmp); mul(in.opacity, tmp, c);
            if (!black(c)) push_lobe(m, PT_LOBE_LAMBERT, kRefl | kDiffuse, c);
            cz(in.ks, tmp); mul(in.opacity, tmp, c);
            if (!black(c)) {
                PtLobe* l = push_lobe(m, PT_LOBE_MF_REFL, kRefl | kGlossy, c);
                set_distribution(l, pick_u, pick_v);
                l->fresnel = PT_FR_DIELECTRIC; l->fr_eta_i = 1.0f; l->fr_eta_t = in.eta;
            }
            cz(in.kr, tmp); mul(in.opacity, tmp, c);
            if (!black(c)) {
                PtLobe* l = push_lobe(m, PT_LOBE_SPEC_REFL, kRefl | kSpecular, c);
                l->fresnel = PT_FR_DIELECTRIC; l->fr_eta_i = 1.0f; l->fr_eta_t = in.eta;
            }
            cz(in.kt, tmp); mul(in.opacity, tmp, c);
            if (!black(c)) {
                PtLobe* l = push_lobe(m, PT_LOBE_SPEC_TRANS, kTrans | kSpecular, c);
                l->eta_a = 1.0f; l->eta_b = in.eta;
            }
            break;
        }
        case PT_MATERIAL_SUBSTRATE: {                   // substrate.rs:34-68
            float d[3], s[3];
            cz(in.kd, d); cz(in.ks, s);
            if (!black(d) && !black(s)) {
                PtLobe* l = push_lobe(m, PT_LOBE_FRESNEL_BLEND, kRefl | kGlossy, d);
                for (int i_ = 0; i_ < 3; i_++) (l->t)[i_] = (s)[i_];
                set_distribution(l, a_u, a_v);
            }
            break;
        }
        default: break;
    }
    for (uint32_t i = 0; i < m.n_lobes; i++) {
        if (!(m.lobes[i].type & kSpecular)) m.nonspecular++;
        if ((m.lobes[i].type & (kRefl | kSpecular)) == m.lobes[i].type) m.spec_mask |= 1u;          // matches_flags (bxdf.rs:17-20) against what specular_reflect asks for
        if ((m.lobes[i].type & (kTrans | kSpecular)) == m.lobes[i].type) m.spec_mask |= 2u;         // ... and specular_transmit
    }
}

