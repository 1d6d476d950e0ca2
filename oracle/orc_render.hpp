// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_render.hpp: scene, camera, lights, light distributions, materials -> BSDF,
// PathIntegrator::li, film and the tile render driver.
//   follows src/cameras/perspective.rs, src/core/camera/projective.rs,
//           src/core/interaction/{surface_interaction,interaction}.rs,
//           src/materials/{matte,plastic,mirror,glass,metal,uber,substrate}.rs, src/core/reflection/bsdf.rs,
//           src/lights/diffuse.rs, src/core/light/visibility_tester.rs,
//           src/core/lightdistrib/{spatial,power,uniform,create_light_sample_distribution}.rs,
//           src/core/integrator/{sample_lights,sampler}.rs, src/integrators/path.rs,
//           src/core/film/{film,film_tile}.rs
#pragma once
#include "../include/pbrtgpu.h"
#include "orc_accel.hpp"
#include "orc_bxdf.hpp"
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <unordered_map>

namespace orc {

// core/reflection/bsdf.rs
struct BSDF {
    Float eta = 1.0f;
    V3 ns, ng, ss, ts;
    Lobe lobes[8];               // MAX_BXDFS
    int n_lobes = 0;
    Lobe& add() { return lobes[n_lobes++]; }
    void init(const SurfHit& si, Float eta_) {   // bsdf.rs:40-53
        eta = eta_;
        ns = si.sh_n;
        ng = si.n;
        ss = normalize(si.sh_dpdu);
        ts = normalize(cross(ns, ss));
        n_lobes = 0;
    }
    int num_components(uint32_t t) const { int n = 0; for (int i = 0; i < n_lobes; i++) if (lobes[i].matches(t)) n++; return n; }
    V3 world_to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    V3 local_to_world(V3 v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    static bool finite3(V3 v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); }
    // bsdf.rs:92-206
    bool sample_f(V3 wo_w, V2 u, uint32_t flags, RGB* f_out, V3* wi_out, Float* pdf_out, uint32_t* type_out) const {
        int matching = num_components(flags);
        if (matching == 0) return false;
        int comp = (int)std::floor(u.x * (Float)matching);
        if (comp > matching - 1) comp = matching - 1;
        int index = -1, count = comp;
        for (int i = 0; i < n_lobes; i++) {
            if (lobes[i].matches(flags)) {
                if (count == 0) { index = i; break; }
                count--;
            }
        }
        const Lobe& lb = lobes[index];
        V2 remapped(fmin_((u.x * (Float)matching) - (Float)comp, kOneMinusEpsilon), u.y);
        V3 wo = world_to_local(wo_w);
        if (wo.z == 0.0f || !finite3(wo)) return false;
        RGB f;
        V3 wi;
        Float pdf;
        uint32_t t = 0;
        if (!lobe_sample_f(lb, wo, remapped, &f, &wi, &pdf, &t)) return false;
        if (pdf <= 0.0f) return false;
        uint32_t sampled_type = t != 0 ? t : lb.type();
        V3 wi_world = local_to_world(wi);
        if ((lb.type() & BSDF_SPECULAR) == 0 && matching > 1)
            for (int i = 0; i < n_lobes; i++)
                if (i != index && lobes[i].matches(flags)) pdf += lobe_pdf(lobes[i], wo, wi);
        if (matching > 1) pdf /= (Float)matching;
        if ((lb.type() & BSDF_SPECULAR) == 0) {
            bool reflect = (dot(wi_world, ng) * dot(wo_w, ng)) > 0.0f;
            f = RGB();
            for (int i = 0; i < n_lobes; i++) {
                uint32_t tp = lobes[i].type();
                if (lobes[i].matches(flags) && ((reflect && (tp & BSDF_REFLECTION)) || (!reflect && (tp & BSDF_TRANSMISSION))))
                    f = f + lobe_f(lobes[i], wo, wi);
            }
        }
        *f_out = f; *wi_out = wi_world; *pdf_out = pdf; *type_out = sampled_type;
        return true;
    }
    // bsdf.rs:208-236
    RGB f(V3 wo_w, V3 wi_w, uint32_t flags) const {
        V3 wi = world_to_local(wi_w), wo = world_to_local(wo_w);
        if (wo.z == 0.0f || !finite3(wo)) return RGB();
        bool reflect = (dot(wi_w, ng) * dot(wo_w, ng)) > 0.0f;
        RGB r;
        for (int i = 0; i < n_lobes; i++) {
            uint32_t tp = lobes[i].type();
            if (lobes[i].matches(flags) && ((reflect && (tp & BSDF_REFLECTION)) || (!reflect && (tp & BSDF_TRANSMISSION))))
                r = r + lobe_f(lobes[i], wo, wi);
        }
        return r;
    }
    // bsdf.rs:238-270
    Float pdf(V3 wo_w, V3 wi_w, uint32_t flags) const {
        V3 wi = world_to_local(wi_w), wo = world_to_local(wo_w);
        if (wo.z == 0.0f || !finite3(wo)) return 0.0f;
        int count = 0;
        Float p = 0.0f;
        for (int i = 0; i < n_lobes; i++)
            if (lobes[i].matches(flags)) { p = p + lobe_pdf(lobes[i], wo, wi); count++; }
        if (count > 0) return p / (Float)count;
        return 0.0f;
    }
};

}  // namespace orc
#include "orc_texture.hpp"
namespace orc {

struct AreaLight {          // lights/diffuse.rs:5-11
    uint32_t shape;         // Geometry::ref encoding: triangle index, or PRIM_SPHERE | sphere index
    RGB lemit;
    bool two_sided;
    Float area;
    uint32_t n_samples = 1;     // "nsamples" (diffuse.rs:177-189): Light::get_sample_count
};

struct Scene {
    Geometry geom;
    QBVH bvh;
    std::vector<pt_material> materials;
    std::vector<pt_texture> textures;
    std::vector<MipImage> images;
    std::vector<int32_t> mesh_material, mesh_light_params;
    std::vector<int32_t> sphere_material;
    std::vector<int32_t> prim_light;    // light index per primitive or -1
    std::vector<std::unique_ptr<QBVH>> objects;     // accelerators of the ObjectBegin .. ObjectEnd groups
    int32_t shape_material(uint32_t ref) const {
        return (ref & PRIM_SPHERE) ? sphere_material[ref & ~PRIM_SPHERE] : mesh_material[geom.tri_mesh[ref]];
    }
    std::vector<AreaLight> lights;
    Bounds3 world_bound;

    // camera
    Transform raster_to_camera;
    Mat4 camera_to_world;
    Float lens_radius = 0, focal_distance = 0, shutter_open = 0, shutter_close = 1;
    // film
    int32_t xres = 0, yres = 0;
    int32_t crop[4] = {0, 0, 0, 0};     // x0 y0 x1 y1
    int32_t sample_bounds[4] = {0, 0, 0, 0};
    Float filter_radius[2] = {0.5f, 0.5f};
    Float filter_table[256];
    Float film_scale = 1.0f, max_sample_luminance = kInfinity;
    // sampler / integrator
    SobolTables sobol;
    int32_t spp = 1, max_depth = 5, light_strategy = PT_LIGHTS_SPATIAL;
    int32_t integrator = PT_INTEGRATOR_PATH, ao_samples = 64, direct_strategy = PT_DIRECT_ALL;
    bool ao_cos_sample = true;
    int32_t sampler_kind = PT_SAMPLER_SOBOL;
    bool halton_at_center = false;
    void init_sampler(SobolSampler& sm) const {
        if (sampler_kind == PT_SAMPLER_HALTON) sm.init_halton((uint32_t)spp, sample_bounds, halton_at_center);
        else sm.init(&sobol, (uint32_t)spp, sample_bounds);
        sm.array2d_n = integrator == PT_INTEGRATOR_AO ? (uint32_t)ao_samples : 0u;
        sm.n_arrays1 = (integrator == PT_INTEGRATOR_DIRECTLIGHTING && direct_strategy == PT_DIRECT_ALL) ? 2u * (uint32_t)lights.size() * (uint32_t)std::max(max_depth, 0) : 0u;
    }
    Float rr_threshold = 1.0f;

    bool build(const pt_scene_desc& d, const std::string& data_dir, std::string* err);
};

// DiffuseAreaLight::l (diffuse.rs:155-163)
inline RGB light_L(const AreaLight& l, V3 n, V3 w) { return (l.two_sided || dot(n, w) > 0.0f) ? l.lemit : RGB(); }

// DiffuseAreaLight::sample_li (diffuse.rs:70-87).  Outputs wi, pdf, the sampled
// point (p, p_error, n) for the visibility tester.
inline bool light_sample_li(const Scene& sc, const AreaLight& l, V3 ref_p, V3 ref_p_error, V3 ref_n, V2 u, RGB* li, V3* wi, Float* pdf, V3* lp, V3* lperr, V3* ln) {
    if (l.shape & PRIM_SPHERE) {
        if (!sc.geom.spheres[l.shape & ~PRIM_SPHERE].sample_from(ref_p, ref_p_error, ref_n, u, lp, ln, lperr, pdf)) return false;
    } else {
        if (!TriRef(&sc.geom, l.shape).sample_from(ref_p, u, lp, ln, lperr, pdf)) return false;
    }
    if (*pdf <= 0.0f || length_squared(*lp - ref_p) <= 0.0f) return false;
    *wi = normalize(*lp - ref_p);
    *li = light_L(l, *ln, -*wi);
    return true;
}

// ---- light distributions (core/lightdistrib)
struct LightDistribution {
    const Scene* sc = nullptr;
    int strategy = PT_LIGHTS_SPATIAL;
    Distribution1D fixed;          // uniform / power
    uint32_t voxels[3] = {1, 1, 1};
    std::unordered_map<uint64_t, std::unique_ptr<Distribution1D>> table;   // keyed by packed voxel id
    std::shared_mutex mu;      // the reference guards each slot with an RwLock (spatial.rs:199-260)

    void init(const Scene* s) {    // create_light_sample_distribution.rs:11-50
        sc = s;
        strategy = s->light_strategy;
        size_t nl = s->lights.size();
        if (strategy == PT_LIGHTS_UNIFORM && nl != 1) strategy = PT_LIGHTS_SPATIAL;   // quirk Q12
        if (strategy == PT_LIGHTS_UNIFORM) {
            fixed = Distribution1D(std::vector<Float>(nl, 1.0f));
        } else if (strategy == PT_LIGHTS_POWER) {   // power.rs:9-17, diffuse.rs:65-68
            std::vector<Float> pw;
            for (const AreaLight& l : s->lights) {
                Float n = l.two_sided ? 2.0f : 1.0f;
                RGB p = l.lemit * (n * l.area * kPi);
                pw.push_back(p.y());
            }
            fixed = Distribution1D(pw);
        } else {                                     // spatial.rs:36-63, max_voxels = 64
            const uint32_t max_voxels = 64;
            V3 diag = s->world_bound.diagonal();
            Float bmax = diag[s->world_bound.maximum_extent()];
            for (int i = 0; i < 3; i++) {
                Float c = std::ceil(diag[i] / bmax * (Float)max_voxels);
                uint32_t v = c > 0.0f ? (c >= 4294967296.0f ? 0xffffffffu : (uint32_t)c) : 0u;   // `as u32` saturates, NaN -> 0
                if (v < 1) v = 1;
                if (v > max_voxels) v = max_voxels;
                voxels[i] = v;
            }
        }
    }
    void voxel_of(V3 p, uint32_t pi[3]) const {      // spatial.rs:84-111
        V3 off = sc->world_bound.offset(p);
        for (int i = 0; i < 3; i++) {
            Float o = clampf(off[i], 0.0f, 1.0f);
            Float f = o * (Float)voxels[i];
            uint32_t v = f > 0.0f ? (uint32_t)f : 0u;     // NaN -> 0
            if (v > voxels[i] - 1) v = voxels[i] - 1;
            pi[i] = v;
        }
    }
    std::unique_ptr<Distribution1D> compute_distribution(const uint32_t pi[3]) const {   // spatial.rs:113-196
        const Bounds3& wb = sc->world_bound;
        V3 p0((Float)pi[0] / (Float)voxels[0], (Float)pi[1] / (Float)voxels[1], (Float)pi[2] / (Float)voxels[2]);
        V3 p1((Float)(pi[0] + 1) / (Float)voxels[0], (Float)(pi[1] + 1) / (Float)voxels[1], (Float)(pi[2] + 1) / (Float)voxels[2]);
        Bounds3 vb(wb.lerp(p0), wb.lerp(p1));
        const size_t n_samples = 128;
        size_t lsz = sc->lights.size();
        std::vector<Float> contrib(lsz, 0.0f);
        for (size_t i = 0; i < n_samples; i++) {
            V3 t(radical_inverse(0, i), radical_inverse(1, i), radical_inverse(2, i));
            V3 po = vb.lerp(t);
            V2 u(radical_inverse(3, i), radical_inverse(4, i));
            for (size_t j = 0; j < lsz; j++) {
                RGB li; V3 wi, lp, le, ln; Float pdf;
                if (light_sample_li(*sc, sc->lights[j], po, V3(0.0f, 0.0f, 0.0f), V3(0.0f, 0.0f, 0.0f), u, &li, &wi, &pdf, &lp, &le, &ln))
                    if (pdf > 0.0f) contrib[j] += li.y() / pdf;
            }
        }
        Float sum = 0.0f;
        for (Float c : contrib) sum += c;
        Float avg = sum / (Float)(n_samples * lsz);
        Float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
        for (size_t i = 0; i < lsz; i++) contrib[i] = fmax_(min_contrib, contrib[i]);
        return std::unique_ptr<Distribution1D>(new Distribution1D(contrib));
    }
    const Distribution1D* lookup(V3 p) {             // spatial.rs:199-260 (hash collisions only change probing, not results)
        if (strategy != PT_LIGHTS_SPATIAL) return &fixed;
        uint32_t pi[3];
        voxel_of(p, pi);
        uint64_t key = ((uint64_t)pi[0] << 40) | ((uint64_t)pi[1] << 20) | (uint64_t)pi[2];
        {
            std::shared_lock<std::shared_mutex> g(mu);
            auto it = table.find(key);
            if (it != table.end()) return it->second.get();
        }
        std::unique_ptr<Distribution1D> d = compute_distribution(pi);
        std::unique_lock<std::shared_mutex> g(mu);
        auto it = table.find(key);
        if (it != table.end()) return it->second.get();
        const Distribution1D* r = d.get();
        table[key] = std::move(d);
        return r;
    }
};

// ---- camera (perspective.rs:121-183 reduced to the main ray; the differential
// rays only feed texture filtering, which constant textures ignore)
inline void transform_ray(const Mat4& m, V3* o, V3* d) {   // transform.rs:216-282
    V3 p = *o;
    V3 op = m.transform_point(p);
    Float x_abs = std::fabs(m.m[0] * p.x) + std::fabs(m.m[1] * p.y) + std::fabs(m.m[2] * p.z) + std::fabs(m.m[3]);
    Float y_abs = std::fabs(m.m[4] * p.x) + std::fabs(m.m[5] * p.y) + std::fabs(m.m[6] * p.z) + std::fabs(m.m[7]);
    Float z_abs = std::fabs(m.m[8] * p.x) + std::fabs(m.m[9] * p.y) + std::fabs(m.m[10] * p.z) + std::fabs(m.m[11]);
    V3 o_error = kGamma3 * V3(x_abs, y_abs, z_abs);
    V3 dd = m.transform_vector(*d);
    Float ls = length_squared(dd);
    if (ls > 0.0f) {
        Float dt = dot(vabs(dd), o_error) / ls;
        op += dd * dt;
    }
    *o = op;
    *d = dd;
}
struct CameraSample { V2 p_film, p_lens; Float time; };
inline Ray generate_ray(const Scene& sc, const CameraSample& s, RayDiff* rdiff = nullptr);
inline Ray generate_ray_main(const Scene& sc, const CameraSample& s) {
    V3 p_camera = sc.raster_to_camera.transform_point(V3(s.p_film.x, s.p_film.y, 0.0f));
    V3 o(0.0f, 0.0f, 0.0f);
    V3 d = normalize(p_camera);
    if (sc.lens_radius > 0.0f) {
        V2 p_lens = concentric_sample_disk(s.p_lens) * sc.lens_radius;
        Float ft = sc.focal_distance / d.z;
        V3 p_focus = o + d * ft;
        o = V3(p_lens.x, p_lens.y, 0.0f);
        d = normalize(p_focus - o);
    }
    transform_ray(sc.camera_to_world, &o, &d);
    return Ray(o, d, kInfinity);
}
// PerspectiveCamera::generate_ray_differential (perspective.rs:121-183) + the 1/sqrt(spp) scaling of render_tile
// (sampler.rs:218,233; ray_differential.rs:26-35)
inline Ray generate_ray(const Scene& sc, const CameraSample& s, RayDiff* rdiff) {
    Ray ray = generate_ray_main(sc, s);
    if (!rdiff) return ray;
    V3 p_camera = sc.raster_to_camera.transform_point(V3(s.p_film.x, s.p_film.y, 0.0f));
    V3 dx_camera = sc.raster_to_camera.transform_point(V3(1.0f, 0.0f, 0.0f)) - sc.raster_to_camera.transform_point(V3(0.0f, 0.0f, 0.0f));
    V3 dy_camera = sc.raster_to_camera.transform_point(V3(0.0f, 1.0f, 0.0f)) - sc.raster_to_camera.transform_point(V3(0.0f, 0.0f, 0.0f));
    V3 rxo, ryo, rxd, ryd;
    if (sc.lens_radius > 0.0f) {
        V2 p_lens = concentric_sample_disk(s.p_lens) * sc.lens_radius;
        {
            V3 dx = normalize(p_camera + dx_camera);
            Float ft = sc.focal_distance / dx.z;
            V3 p_focus = V3(0.0f, 0.0f, 0.0f) + (ft * dx);
            rxo = V3(p_lens.x, p_lens.y, 0.0f);
            rxd = normalize(p_focus - rxo);
        }
        {
            V3 dy = normalize(p_camera + dy_camera);
            Float ft = sc.focal_distance / dy.z;
            V3 p_focus = V3(0.0f, 0.0f, 0.0f) + (ft * dy);
            ryo = V3(p_lens.x, p_lens.y, 0.0f);
            ryd = normalize(p_focus - ryo);
        }
    } else {
        rxo = V3(0.0f, 0.0f, 0.0f); ryo = rxo;          // = ray.o before the lens / camera transform
        rxd = normalize(p_camera + dx_camera);
        ryd = normalize(p_camera + dy_camera);
    }
    // transform_ray_differential (transform.rs:284-297): plain point / vector transforms for the offset rays
    rxo = sc.camera_to_world.transform_point(rxo); ryo = sc.camera_to_world.transform_point(ryo);
    rxd = sc.camera_to_world.transform_vector(rxd); ryd = sc.camera_to_world.transform_vector(ryd);
    Float scale = std::sqrt(1.0f / (Float)sc.spp);
    rdiff->has = true;
    rdiff->rx_o = ray.o + (rxo - ray.o) * scale;
    rdiff->ry_o = ray.o + (ryo - ray.o) * scale;
    rdiff->rx_d = ray.d + (rxd - ray.d) * scale;
    rdiff->ry_d = ray.d + (ryd - ray.d) * scale;
    return ray;
}

struct RayCounters { uint64_t camera = 0, regular = 0, shadow = 0, nodes = 0, tris = 0, vertices = 0; };

// ---- Material::compute_scattering_functions with constant textures, mode = Radiance,
// allow_multiple_lobes = true (path.rs:106).  Returns false when the interaction gets no BSDF.
inline void add_microfacet_refl(BSDF* b, RGB r, Float ax, Float ay, int fresnel, Float eta_i, Float eta_t) {
    Lobe& l = b->add();
    l.kind = LOBE_MF_REFL; l.r = r; l.dist.init(ax, ay);
    l.fresnel = fresnel; l.fr_eta_i = eta_i; l.fr_eta_t = eta_t;
}
inline bool make_bsdf_from_material(const pt_material& m, const SurfHit& si, BSDF* b) {
    if (m.type == PT_MATERIAL_NONE) return false;
    auto rgb = [](const float* c) { return RGB(c[0], c[1], c[2]); };
    auto pick = [](Float specific, Float general) { return specific == PT_ROUGHNESS_UNSET ? general : specific; };
    switch (m.type) {
        case PT_MATERIAL_MATTE: {                      // materials/matte.rs:25-53
            b->init(si, 1.0f);
            RGB r = rgb(m.kd);                         // no clamp_zero here, unlike the other materials
            Float sig = clampf(m.sigma, 0.0f, 90.0f);
            if (!r.is_black()) {
                Lobe& l = b->add();
                l.r = r;
                if (sig == 0.0f) {
                    l.kind = LOBE_LAMBERT;
                } else {                               // oren_nayar.rs:12-24
                    l.kind = LOBE_OREN_NAYAR;
                    Float sigma = radians(sig);
                    Float sigma2 = sigma * sigma;
                    l.a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
                    l.b = 0.45f * sigma2 / (sigma2 + 0.09f);
                }
            }
            return true;
        }
        case PT_MATERIAL_PLASTIC: {                    // materials/plastic.rs:31-71
            b->init(si, 1.0f);
            RGB kd = rgb_clamp_zero(rgb(m.kd));
            if (!kd.is_black()) { Lobe& l = b->add(); l.kind = LOBE_LAMBERT; l.r = kd; }
            RGB ks = rgb_clamp_zero(rgb(m.ks));
            if (!ks.is_black()) {
                Float rough = m.roughness;
                if (m.remap_roughness) rough = TRDist::roughness_to_alpha(rough);
                add_microfacet_refl(b, ks, rough, rough, FR_DIELECTRIC, 1.5f, 1.0f);
            }
            return true;
        }
        case PT_MATERIAL_MIRROR: {                     // materials/mirror.rs:19-41
            b->init(si, 1.0f);
            RGB r = rgb_clamp_zero(rgb(m.kr));
            if (!r.is_black()) { Lobe& l = b->add(); l.kind = LOBE_SPEC_REFL; l.r = r; l.fresnel = FR_NOOP; }
            return true;
        }
        case PT_MATERIAL_GLASS: {                      // materials/glass.rs:46-110
            Float eta = m.eta, u_rough = m.uroughness, v_rough = m.vroughness;
            RGB r = rgb(m.kr), t = rgb(m.kt);
            if (r.is_black() && t.is_black()) return false;
            b->init(si, eta);
            bool is_specular = u_rough == 0.0f && v_rough == 0.0f;
            if (is_specular) {                         // allow_multiple_lobes
                Lobe& l = b->add();
                l.kind = LOBE_FRESNEL_SPEC; l.r = r; l.t = t; l.eta_a = 1.0f; l.eta_b = eta;
            } else {
                if (m.remap_roughness) { u_rough = TRDist::roughness_to_alpha(u_rough); v_rough = TRDist::roughness_to_alpha(v_rough); }
                if (!r.is_black()) add_microfacet_refl(b, r, u_rough, v_rough, FR_DIELECTRIC, 1.0f, eta);
                if (!t.is_black()) {
                    Lobe& l = b->add();
                    l.kind = LOBE_MF_TRANS; l.r = t; l.dist.init(u_rough, v_rough); l.eta_a = 1.0f; l.eta_b = eta;
                }
            }
            return true;
        }
        case PT_MATERIAL_METAL: {                      // materials/metal.rs:51-85
            b->init(si, 1.0f);
            Float u_rough = pick(m.uroughness, m.roughness), v_rough = pick(m.vroughness, m.roughness);
            if (m.remap_roughness) { u_rough = TRDist::roughness_to_alpha(u_rough); v_rough = TRDist::roughness_to_alpha(v_rough); }
            Lobe& l = b->add();
            l.kind = LOBE_MF_REFL; l.r = RGB(1.0f); l.dist.init(u_rough, v_rough);
            l.fresnel = FR_CONDUCTOR; l.fr_cond_eta = rgb(m.metal_eta); l.fr_cond_k = rgb(m.metal_k);
            return true;
        }
        case PT_MATERIAL_UBER: {                       // materials/uber.rs:63-127
            Float e = m.eta;
            RGB op = rgb(m.opacity);
            RGB t = rgb_clamp_zero(RGB(1.0f) - op);
            b->init(si, !t.is_black() ? 1.0f : e);
            if (!t.is_black()) { Lobe& l = b->add(); l.kind = LOBE_SPEC_TRANS; l.r = t; l.eta_a = 1.0f; l.eta_b = 1.0f; }
            RGB kd = op * rgb_clamp_zero(rgb(m.kd));
            if (!kd.is_black()) { Lobe& l = b->add(); l.kind = LOBE_LAMBERT; l.r = kd; }
            RGB ks = op * rgb_clamp_zero(rgb(m.ks));
            if (!ks.is_black()) {
                Float u_rough = pick(m.uroughness, m.roughness), v_rough = pick(m.vroughness, m.roughness);
                if (m.remap_roughness) { u_rough = TRDist::roughness_to_alpha(u_rough); v_rough = TRDist::roughness_to_alpha(v_rough); }
                add_microfacet_refl(b, ks, u_rough, v_rough, FR_DIELECTRIC, 1.0f, e);
            }
            RGB kr = op * rgb_clamp_zero(rgb(m.kr));
            if (!kr.is_black()) { Lobe& l = b->add(); l.kind = LOBE_SPEC_REFL; l.r = kr; l.fresnel = FR_DIELECTRIC; l.fr_eta_i = 1.0f; l.fr_eta_t = e; }
            RGB kt = op * rgb_clamp_zero(rgb(m.kt));
            if (!kt.is_black()) { Lobe& l = b->add(); l.kind = LOBE_SPEC_TRANS; l.r = kt; l.eta_a = 1.0f; l.eta_b = e; }
            return true;
        }
        case PT_MATERIAL_SUBSTRATE: {                  // materials/substrate.rs:34-68
            b->init(si, 1.0f);
            RGB d = rgb_clamp_zero(rgb(m.kd)), s = rgb_clamp_zero(rgb(m.ks));
            if (!d.is_black() && !s.is_black()) {
                Float u_rough = m.uroughness, v_rough = m.vroughness;
                if (m.remap_roughness) { u_rough = TRDist::roughness_to_alpha(u_rough); v_rough = TRDist::roughness_to_alpha(v_rough); }
                Lobe& l = b->add();
                l.kind = LOBE_FRESNEL_BLEND; l.r = d; l.t = s; l.dist.init(u_rough, v_rough);
            }
            return true;
        }
    }
    return false;
}
// Parameter textures are evaluated at the hit (TextureParams::get_spectrum_texture / get_float_texture bound at material
// creation; Texture::evaluate(si) inside each compute_scattering_functions), then the constant-parameter code runs.
// material_bump (core/material.rs:31-72): the displacement texture at the hit and at two shifted points bends the shading
// frame.  The shifted evaluation points differ from the hit in p and uv only (the differentials are copied, :15-29).
inline void material_bump(const pt_texture* tex, int32_t d, const TexHit& th, SurfHit& si, const MipImage* images) {
    Float du = 0.5f * (std::fabs(th.dudx) + std::fabs(th.dudy));
    if (du == 0.0f) du = 0.0005f;
    TexHit ev = th;
    ev.p = si.p + du * si.sh_dpdu;
    ev.uv = si.uv + V2(du, 0.0f);
    Float u_displace = texture_eval(tex, d, ev, images).c[0];
    Float dv = 0.5f * (std::fabs(th.dvdx) + std::fabs(th.dvdy));
    if (dv == 0.0f) dv = 0.0005f;
    ev.p = si.p + dv * si.sh_dpdv;
    ev.uv = si.uv + V2(0.0f, dv);
    Float v_displace = texture_eval(tex, d, ev, images).c[0];
    Float displace = texture_eval(tex, d, th, images).c[0];
    V3 dpdu = si.sh_dpdu + (u_displace - displace) / du * si.sh_n + displace * si.sh_dndu;
    V3 dpdv = si.sh_dpdv + (v_displace - displace) / dv * si.sh_n + displace * si.sh_dndv;
    // set_shading_geometry(.., orientation_is_authoritative = false) (surface_interaction.rs:140-161)
    si.sh_n = face_forward(normalize(cross(dpdu, dpdv)), si.n);
    si.sh_dpdu = dpdu; si.sh_dpdv = dpdv;
}
inline bool make_bsdf(const Scene& sc, SurfHit& si, BSDF* b, const RayDiff& rd = RayDiff()) {
    int32_t mid = sc.shape_material(si.shape_ref);
    if (mid < 0) return false;
    const pt_material& m0 = sc.materials[mid];
    if (m0.type == PT_MATERIAL_NONE ||
        !(m0.tex_kd | m0.tex_ks | m0.tex_kr | m0.tex_kt | m0.tex_opacity | m0.tex_sigma | m0.tex_metal_eta | m0.tex_metal_k | m0.tex_bump |
          m0.tex_roughness | m0.tex_uroughness | m0.tex_vroughness | m0.tex_eta))
        return make_bsdf_from_material(m0, si, b);
    pt_material m = m0;
    TexHit th = compute_differentials(si, rd);                  // SurfaceInteraction::compute_scattering_functions :284-295
    if (m.tex_bump) material_bump(sc.textures.data(), (int32_t)m.tex_bump - 1, th, si, sc.images.data());      // first thing every material does
    auto spec = [&](uint32_t t, float* out) {
        if (!t) return;
        RGB v = texture_eval(sc.textures.data(), (int32_t)t - 1, th, sc.images.data());
        out[0] = v.c[0]; out[1] = v.c[1]; out[2] = v.c[2];
    };
    spec(m.tex_kd, m.kd); spec(m.tex_ks, m.ks); spec(m.tex_kr, m.kr); spec(m.tex_kt, m.kt); spec(m.tex_opacity, m.opacity);
    spec(m.tex_metal_eta, m.metal_eta); spec(m.tex_metal_k, m.metal_k);
    // float textures: Matte's sigma, and the roughness / eta parameters (plastic.rs:57-62, glass.rs:57-81, metal.rs:58-69,
    // uber.rs:66,96-104, substrate.rs:45-54): evaluated at the hit, then roughness_to_alpha (`ln`) inside the constant code
    auto flt = [&](uint32_t t, float* out) { if (t) *out = texture_eval(sc.textures.data(), (int32_t)t - 1, th, sc.images.data()).c[0]; };
    flt(m.tex_sigma, &m.sigma); flt(m.tex_roughness, &m.roughness); flt(m.tex_uroughness, &m.uroughness); flt(m.tex_vroughness, &m.vroughness);
    flt(m.tex_eta, &m.eta);
    return make_bsdf_from_material(m, si, b);
}

// SurfaceInteraction::le (surface_interaction.rs:297-306)
inline RGB surf_le(const Scene& sc, const SurfHit& si, V3 w) {
    int32_t li = sc.prim_light[si.prim];
    if (li < 0) return RGB();
    return light_L(sc.lights[li], si.n, w);
}

// estimate_direct_surface + uniform_sample_one_light_surface
// (core/integrator/sample_lights.rs:129-176, :330-453), handle_media = false, specular = false
// estimate_direct / estimate_direct_surface (sample_lights.rs:178-328, :330-453: the same arithmetic), handle_media = false, specular = false
inline RGB estimate_direct(const Scene& sc, const SurfHit& it, const BSDF& bsdf, size_t light_num, V2 u_light, V2 u_scattering, RayCounters& rc) {
    const AreaLight& light = sc.lights[light_num];
    const uint32_t bsdf_flags = BSDF_ALL & ~BSDF_SPECULAR;
    RGB ld;
    // -- sample the light
    {
        RGB li; V3 wi, lp, lperr, ln; Float lpdf;
        if (light_sample_li(sc, light, it.p, it.p_error, it.n, u_light, &li, &wi, &lpdf, &lp, &lperr, &ln)) {
            if (lpdf > 0.0f && !li.is_black()) {
                RGB f = bsdf.f(it.wo, wi, bsdf_flags) * abs_dot(wi, it.sh_n);
                Float scattering_pdf = bsdf.pdf(it.wo, wi, bsdf_flags);
                if (!f.is_black()) {
                    // VisibilityTester::unoccluded -> Interaction::spawn_ray_to (interaction.rs:118-127)
                    V3 origin = offset_ray_origin(it.p, it.p_error, it.n, lp - it.p);
                    V3 target = offset_ray_origin(lp, lperr, ln, origin - lp);
                    Ray sr(origin, target - origin, 1.0f - kShadowEpsilon);
                    rc.shadow++;
                    QBVH::Stats st;
                    bool occluded = sc.bvh.intersect_p(sr, &st);
                    rc.nodes += st.nodes; rc.tris += st.tris;
                    if (occluded) li = RGB();
                    if (!li.is_black()) {
                        Float weight = power_heuristic(1, lpdf, 1, scattering_pdf);
                        ld += f * li * (weight / lpdf);
                    }
                }
            }
        }
    }
    // -- sample the BSDF
    {
        RGB f; V3 wi; Float scattering_pdf = 0.0f; uint32_t sampled_type = 0;
        bool sampled_specular = false;
        RGB f2;
        if (bsdf.sample_f(it.wo, u_scattering, bsdf_flags, &f2, &wi, &scattering_pdf, &sampled_type)) {
            f = f2 * abs_dot(wi, it.sh_n);
            sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
        } else {
            scattering_pdf = 0.0f;
        }
        if (!f.is_black() && scattering_pdf > 0.0f) {
            Float weight = 1.0f;
            bool skip = false;
            if (!sampled_specular) {
                Float lpdf = (light.shape & PRIM_SPHERE) ? sc.geom.spheres[light.shape & ~PRIM_SPHERE].pdf_from(it.p, it.p_error, it.n, wi)
                                                         : TriRef(&sc.geom, light.shape).pdf_from(it.p, it.p_error, it.n, wi);   // pdf_li, diffuse.rs:89-94
                if (lpdf == 0.0f) skip = true;     // `return ld`
                else weight = power_heuristic(1, scattering_pdf, 1, lpdf);
            }
            if (!skip) {
                Ray ray(offset_ray_origin(it.p, it.p_error, it.n, wi), wi, kInfinity);
                rc.regular++;
                SurfHit lh;
                QBVH::Stats st;
                bool found = sc.bvh.intersect(ray, &lh, &st);
                rc.nodes += st.nodes; rc.tris += st.tris;
                RGB li;
                if (found) {
                    if (sc.prim_light[lh.prim] == (int32_t)light_num) li = surf_le(sc, lh, -wi);
                }   // else light.le(ray): zero for area lights (light.rs:33-35)
                if (!li.is_black()) ld += f * li * (weight / scattering_pdf);
            }
        }
    }
    return ld;
}
// uniform_sample_one_light_surface (sample_lights.rs:129-176) with the integrator's light distribution
inline RGB uniform_sample_one_light(const Scene& sc, const SurfHit& it, const BSDF& bsdf, SobolSampler& sampler,
                                    const Distribution1D* distrib, RayCounters& rc) {
    size_t n_lights = sc.lights.size();
    if (n_lights == 0) return RGB();
    Float light_pdf;
    size_t light_num = distrib->sample_discrete(sampler.get_1d(), &light_pdf, nullptr);
    if (light_pdf <= 0.0f) return RGB();
    V2 u_light = sampler.get_2d();
    V2 u_scattering = sampler.get_2d();
    return estimate_direct(sc, it, bsdf, light_num, u_light, u_scattering, rc) / light_pdf;
}

// PathIntegrator::li (integrators/path.rs:61-241)
inline RGB path_li(const Scene& sc, LightDistribution& ldist, Ray ray, SobolSampler& sampler, RayCounters& rc, RayDiff rdiff = RayDiff()) {
    // no lights => create_light_sample_distribution fails => li returns zero (path.rs:71-74)
    if (sc.lights.empty()) return RGB();
    RGB l, beta(1.0f);
    Float eta_scale = 1.0f;
    bool specular_bounce = false;
    int bounces = 0;
    for (;;) {
        SurfHit isect;
        rc.regular++;
        QBVH::Stats st;
        bool found = sc.bvh.intersect(ray, &isect, &st);
        rc.nodes += st.nodes; rc.tris += st.tris;
        if (bounces == 0 || specular_bounce) {
            if (found) l += beta * surf_le(sc, isect, -ray.d);
        }
        if (!found || bounces >= sc.max_depth) break;
        BSDF bsdf;
        bool has_bsdf = make_bsdf(sc, isect, &bsdf, rdiff);
        rdiff.has = false;                 // every later ray is a plain Ray (`.into()`, path.rs:109,:209)
        if (!has_bsdf) {
            ray = Ray(offset_ray_origin(isect.p, isect.p_error, isect.n, ray.d), ray.d, kInfinity);
            continue;
        }
        rc.vertices++;
        const Distribution1D* distrib = ldist.lookup(isect.p);
        if (bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
            RGB ld = beta * uniform_sample_one_light(sc, isect, bsdf, sampler, distrib, rc);
            l += ld;
        }
        V3 wo = -ray.d;
        RGB f; V3 wi; Float pdf; uint32_t flags;
        if (!bsdf.sample_f(wo, sampler.get_2d(), BSDF_ALL, &f, &wi, &pdf, &flags)) break;
        if (f.is_black() || pdf == 0.0f) break;
        beta = beta * (f * (abs_dot(wi, isect.sh_n) / pdf));
        specular_bounce = (flags & BSDF_SPECULAR) != 0;
        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
            Float eta = bsdf.eta;
            eta_scale *= dot(wo, isect.n) > 0.0f ? eta * eta : 1.0f / (eta * eta);
        }
        ray = Ray(offset_ray_origin(isect.p, isect.p_error, isect.n, wi), wi, kInfinity);
        RGB rr_beta = beta * eta_scale;
        if (rr_beta.max_component_value() < sc.rr_threshold && bounces > 3) {
            Float q = fmax_(0.05f, 1.0f - rr_beta.max_component_value());
            if (sampler.get_1d() < q) break;
            beta = beta / (1.0f - q);
        }
        bounces++;
    }
    return l;
}

// AOIntegrator::li (integrators/ao.rs:49-110).  A hit without a material makes the reference ask for the sample array a second
// time and unwrap None; such scenes are refused before they get here, the loop body below is the first iteration.
inline RGB ao_li(const Scene& sc, Ray ray, SobolSampler& sampler, RayCounters& rc) {
    RGB l;
    SurfHit isect;
    rc.regular++;
    QBVH::Stats st;
    bool found = sc.bvh.intersect(ray, &isect, &st);
    rc.nodes += st.nodes; rc.tris += st.tris;
    if (!found) return l;
    rc.vertices++;
    const V3 n = face_forward(isect.n, ray.d * -1.0f);
    const V3 s = normalize(isect.dpdu);
    const V3 t = cross(isect.n, s);
    const uint32_t n_samples = sampler.array2d_n;
    for (uint32_t i = 0; i < n_samples; i++) {
        const V2 u = sampler.array2d(i);
        V3 wi;
        Float pdf;
        if (sc.ao_cos_sample) { wi = cosine_sample_hemisphere(u); pdf = std::fabs(wi.z) * kInvPi; }
        else { wi = uniform_sample_hemisphere(u); pdf = kInv2Pi; }
        wi = V3(wi.x * s.x + wi.y * t.x + wi.z * n.x, wi.x * s.y + wi.y * t.y + wi.z * n.y, wi.x * s.z + wi.y * t.z + wi.z * n.z);
        Ray sh(offset_ray_origin(isect.p, isect.p_error, isect.n, wi), wi, kInfinity);
        rc.shadow++;
        QBVH::Stats st2;
        bool occluded = sc.bvh.intersect_p(sh, &st2);
        rc.nodes += st2.nodes; rc.tris += st2.tris;
        if (!occluded) l = l + RGB(dot(wi, n) / (pdf * (Float)n_samples));
    }
    return l;
}

// ---- DirectLightingIntegrator::li (integrators/directlighting.rs:66-135) and WhittedIntegrator::li (integrators/whitted.rs:38-110)
// with SamplerIntegrator::specular_reflect / specular_transmit (core/integrator/sampler.rs:37-150).  Recursive like the reference,
// so the sampler is consumed depth first.
inline RGB rec_li(const Scene& sc, Ray ray, RayDiff rd, SobolSampler& sampler, RayCounters& rc, int depth) {
    const bool whitted = sc.integrator == PT_INTEGRATOR_WHITTED;
    SurfHit isect;
    rc.regular++;
    QBVH::Stats st;
    bool found = sc.bvh.intersect(ray, &isect, &st);
    rc.nodes += st.nodes; rc.tris += st.tris;
    if (!found) return RGB();                       // sum of light.le(ray): zero for area lights (light.rs:33-35)
    const V3 n_before = isect.sh_n;                 // whitted.rs:52: the shading normal before compute_scattering_functions (bump mapping)
    const V3 wo = isect.wo;
    const TexHit th = compute_differentials(isect, rd);      // what compute_scattering_functions leaves in isect.dpdx / dudx .. (surface_interaction.rs:284-295)
    BSDF bsdf;
    if (!make_bsdf(sc, isect, &bsdf, rd)) {
        if (whitted) return RGB();
        // directlighting.rs:113-116: through the surface with a plain ray, same depth
        return rec_li(sc, Ray(offset_ray_origin(isect.p, isect.p_error, isect.n, ray.d), ray.d, kInfinity), RayDiff(), sampler, rc, depth);
    }
    rc.vertices++;
    RGB l;
    const size_t n_lights = sc.lights.size();
    if (whitted) {
        for (size_t j = 0; j < n_lights; j++) {
            const AreaLight& light = sc.lights[j];
            const V2 u = sampler.get_2d();
            RGB li; V3 wi, lp, lperr, ln; Float pdf;
            RGB term;
            if (light_sample_li(sc, light, isect.p, isect.p_error, isect.n, u, &li, &wi, &pdf, &lp, &lperr, &ln) && !(pdf <= 0.0f || li.is_black())) {
                RGB f = bsdf.f(wo, wi, BSDF_ALL);
                if (!f.is_black()) {
                    V3 origin = offset_ray_origin(isect.p, isect.p_error, isect.n, lp - isect.p);
                    V3 target = offset_ray_origin(lp, lperr, ln, origin - lp);
                    Ray sr(origin, target - origin, 1.0f - kShadowEpsilon);
                    rc.shadow++;
                    QBVH::Stats st2;
                    bool occluded = sc.bvh.intersect_p(sr, &st2);
                    rc.nodes += st2.nodes; rc.tris += st2.tris;
                    if (!occluded) term = f * li * (abs_dot(wi, n_before) / pdf);
                }
            }
            l += term;
        }
    } else {
        l = surf_le(sc, isect, wo);
        if (n_lights > 0) {
            RGB ld_all;
            if (sc.direct_strategy == PT_DIRECT_ALL) {              // uniform_sample_all_lights (sample_lights.rs:24-78)
                for (size_t j = 0; j < n_lights; j++) {
                    const uint32_t n_samples = sc.lights[j].n_samples;       // n_light_samples[j] = round_count(get_sample_count()) = nsamples
                    std::vector<V2> u_light_array, u_scattering_array;
                    const bool a = sampler.get_2d_array(n_samples, &u_light_array), b = sampler.get_2d_array(n_samples, &u_scattering_array);
                    if (a && b) {
                        RGB ld;
                        for (uint32_t k = 0; k < n_samples; k++) ld += estimate_direct(sc, isect, bsdf, j, u_light_array[k], u_scattering_array[k], rc);
                        ld_all += ld / (Float)n_samples;
                    } else {
                        const V2 u_light = sampler.get_2d();
                        const V2 u_scattering = sampler.get_2d();
                        ld_all += estimate_direct(sc, isect, bsdf, j, u_light, u_scattering, rc);
                    }
                }
            } else {                                                 // uniform_sample_one_light (sample_lights.rs:84-127), no distribution
                Float fl = sampler.get_1d() * (Float)n_lights;
                size_t light_num = std::min((size_t)fl, n_lights - 1);
                Float light_pdf = 1.0f / (Float)n_lights;
                V2 u_light = sampler.get_2d();
                V2 u_scattering = sampler.get_2d();
                ld_all = estimate_direct(sc, isect, bsdf, light_num, u_light, u_scattering, rc) / light_pdf;
            }
            l += ld_all;
        }
    }
    if (depth + 1 < sc.max_depth) {
        const V3 ns0 = isect.sh_n;
        {   // specular_reflect (sampler.rs:37-82)
            const V2 u = sampler.get_2d();
            RGB f; V3 wi; Float pdf; uint32_t ty;
            RGB term;
            if (bsdf.sample_f(wo, u, BSDF_REFLECTION | BSDF_SPECULAR, &f, &wi, &pdf, &ty)) {
                const V3 ns = ns0;
                const Float wi_ns = abs_dot(wi, ns);
                if (pdf > 0.0f && !f.is_black() && wi_ns != 0.0f) {
                    Ray r2(offset_ray_origin(isect.p, isect.p_error, isect.n, wi), wi, kInfinity);
                    RayDiff d2;
                    if (rd.has) {
                        d2.has = true;
                        d2.rx_o = isect.p + th.dpdx;
                        d2.ry_o = isect.p + th.dpdy;
                        const V3 dndx = isect.sh_dndu * th.dudx + isect.sh_dndv * th.dvdx;
                        const V3 dndy = isect.sh_dndu * th.dudy + isect.sh_dndv * th.dvdy;
                        const V3 dwodx = -rd.rx_d - wo, dwody = -rd.ry_d - wo;
                        const Float d_dndx = dot(dwodx, ns) + dot(wo, dndx);
                        const Float d_dndy = dot(dwody, ns) + dot(wo, dndy);
                        const Float wo_ns = dot(wo, ns);
                        d2.rx_d = wi - dwodx + 2.0f * (wo_ns * dndx + d_dndx * ns);
                        d2.ry_d = wi - dwody + 2.0f * (wo_ns * dndy + d_dndy * ns);
                    }
                    term = f * rec_li(sc, r2, d2, sampler, rc, depth + 1) * (wi_ns / pdf);
                }
            }
            l += term;
        }
        {   // specular_transmit (sampler.rs:84-143)
            const V2 u = sampler.get_2d();
            RGB f; V3 wi; Float pdf; uint32_t ty;
            RGB term;
            if (bsdf.sample_f(wo, u, BSDF_TRANSMISSION | BSDF_SPECULAR, &f, &wi, &pdf, &ty)) {
                V3 ns = ns0;
                Float wi_ns = abs_dot(wi, ns);
                Float wo_ns = dot(wo, ns);
                if (pdf > 0.0f && !f.is_black() && wi_ns != 0.0f) {
                    Ray r2(offset_ray_origin(isect.p, isect.p_error, isect.n, wi), wi, kInfinity);
                    RayDiff d2;
                    if (rd.has) {
                        d2.has = true;
                        d2.rx_o = isect.p + th.dpdx;
                        d2.ry_o = isect.p + th.dpdy;
                        V3 dndx = isect.sh_dndu * th.dudx + isect.sh_dndv * th.dvdx;
                        V3 dndy = isect.sh_dndu * th.dudy + isect.sh_dndv * th.dvdy;
                        Float eta = 1.0f / bsdf.eta;
                        if (dot(wo, ns) < 0.0f) {
                            eta = 1.0f / eta;
                            ns = -ns; dndx = -dndx; dndy = -dndy;
                            wi_ns = abs_dot(wi, ns);
                            wo_ns = dot(wo, ns);
                        }
                        const V3 dwodx = -rd.rx_d - wo, dwody = -rd.ry_d - wo;
                        const Float d_dndx = dot(dwodx, ns) + dot(wo, dndx);
                        const Float d_dndy = dot(dwody, ns) + dot(wo, dndy);
                        const Float mu = eta * wo_ns - wi_ns;
                        const Float dmudx = (eta - (eta * eta * wo_ns) / wi_ns) * d_dndx;
                        const Float dmudy = (eta - (eta * eta * wo_ns) / wi_ns) * d_dndy;
                        d2.rx_d = wi - eta * dwodx + (mu * dndx + dmudx * ns);
                        d2.ry_d = wi - eta * dwody + (mu * dndy + dmudy * ns);
                    }
                    term = f * rec_li(sc, r2, d2, sampler, rc, depth + 1) * (wi_ns / pdf);
                }
            }
            l += term;
        }
    }
    return l;
}

// validate_radiance_result (core/integrator/sampler.rs:151-176)
inline RGB validate_radiance(RGB l) {
    if (!l.is_valid()) return RGB();
    if (l.y() < -1e-5f) return RGB();
    if (std::isinf(l.y())) return RGB();
    return l;
}

// ---- film (core/film/film_tile.rs:84-183, film.rs:203-241, :440-484)
struct FilmTile {
    int32_t pb[4];                       // pixel bounds x0 y0 x1 y1
    std::vector<Float> contrib;          // 3 per pixel
    std::vector<Float> weight;
    const Scene* sc;
    FilmTile(const Scene* s, const int32_t sample_tile[4]) : sc(s) {
        // Film::get_film_tile
        int32_t p0x = (int32_t)std::floor((Float)sample_tile[0] - s->filter_radius[0]);
        int32_t p0y = (int32_t)std::floor((Float)sample_tile[1] - s->filter_radius[1]);
        int32_t p1x = (int32_t)std::ceil((Float)sample_tile[2] + s->filter_radius[0]);
        int32_t p1y = (int32_t)std::ceil((Float)sample_tile[3] + s->filter_radius[1]);
        pb[0] = std::max(p0x, s->crop[0]); pb[1] = std::max(p0y, s->crop[1]);
        pb[2] = std::min(p1x, s->crop[2]); pb[3] = std::min(p1y, s->crop[3]);
        int64_t w = std::max(0, pb[2] - pb[0]), h = std::max(0, pb[3] - pb[1]);
        contrib.assign((size_t)(w * h * 3), 0.0f);
        weight.assign((size_t)(w * h), 0.0f);
    }
    void add_sample(V2 p_film, RGB l, Float sample_weight) {
        if (l.y() > sc->max_sample_luminance) l = l * (sc->max_sample_luminance / l.y());
        Float rx = sc->filter_radius[0], ry = sc->filter_radius[1];
        Float irx = 1.0f / rx, iry = 1.0f / ry;
        int32_t p0x = (int32_t)std::floor(p_film.x - rx), p0y = (int32_t)std::floor(p_film.y - ry);
        int32_t p1x = (int32_t)std::ceil(p_film.x + rx), p1y = (int32_t)std::ceil(p_film.y + ry);
        p0x = std::max(p0x, pb[0]); p0y = std::max(p0y, pb[1]);
        p1x = std::min(p1x, pb[2]); p1y = std::min(p1y, pb[3]);
        int32_t dx = p1x - p0x, dy = p1y - p0y;
        if (dx <= 0 || dy <= 0) return;
        const int FT = 16;
        std::vector<int32_t> ifx(dx), ify(dy);
        Float lx = irx * (Float)(FT - 1), ly = iry * (Float)(FT - 1);
        for (int32_t x = p0x; x < p1x; x++) {
            Float d = std::fabs((Float)x + 0.5f - p_film.x);
            ifx[x - p0x] = d <= rx ? std::min((int32_t)std::floor(d * lx), FT - 1) : -1;
        }
        for (int32_t y = p0y; y < p1y; y++) {
            Float d = std::fabs((Float)y + 0.5f - p_film.y);
            ify[y - p0y] = d <= ry ? std::min((int32_t)std::floor(d * ly), FT - 1) : -1;
        }
        std::vector<Float> w((size_t)dx * dy, 0.0f);
        for (int32_t y = 0; y < dy; y++) {
            if (ify[y] < 0) continue;
            for (int32_t x = 0; x < dx; x++) {
                if (ifx[x] < 0) continue;
                w[(size_t)y * dx + x] = sc->filter_table[ify[y] * FT + ifx[x]];
            }
        }
        Float sum = 0.0f;
        for (Float v : w) sum += v;
        if (sum <= 0.0f) return;
        Float isum = 1.0f / sum;
        for (Float& v : w) v *= isum;
        int32_t width = pb[2] - pb[0];
        for (int32_t y = p0y; y < p1y; y++)
            for (int32_t x = p0x; x < p1x; x++) {
                Float fw = w[(size_t)(y - p0y) * dx + (x - p0x)];
                size_t pi = (size_t)(y - pb[1]) * width + (x - pb[0]);
                RGB c = l * sample_weight * fw;
                contrib[3 * pi] += c.c[0]; contrib[3 * pi + 1] += c.c[1]; contrib[3 * pi + 2] += c.c[2];
                weight[pi] += fw;
            }
    }
};

struct Film {
    const Scene* sc;
    std::vector<Float> xyzw;     // 4 per cropped pixel
    std::mutex mu;
    explicit Film(const Scene* s) : sc(s) {
        xyzw.assign((size_t)(s->crop[2] - s->crop[0]) * (s->crop[3] - s->crop[1]) * 4, 0.0f);
    }
    void merge(const FilmTile& t) {   // film.rs:219-241
        std::lock_guard<std::mutex> g(mu);
        int32_t tw = t.pb[2] - t.pb[0], fw = sc->crop[2] - sc->crop[0];
        for (int32_t y = t.pb[1]; y < t.pb[3]; y++)
            for (int32_t x = t.pb[0]; x < t.pb[2]; x++) {
                size_t src = (size_t)(y - t.pb[1]) * tw + (x - t.pb[0]);
                size_t dst = (size_t)(y - sc->crop[1]) * fw + (x - sc->crop[0]);
                Float xyz[3];
                rgb_to_xyz(&t.contrib[3 * src], xyz);
                for (int i = 0; i < 3; i++) xyzw[4 * dst + i] += xyz[i];
                xyzw[4 * dst + 3] += t.weight[src];
            }
    }
    void resolve_rgb(Float* rgb) const {   // film.rs:440-484
        resolve_xyzw(xyzw.data(), xyzw.size() / 4, sc->film_scale, rgb);
    }
    static void resolve_xyzw(const Float* xyzw, size_t n, Float scale, Float* rgb) {
        for (size_t i = 0; i < n; i++) {
            Float c[3];
            xyz_to_rgb(&xyzw[4 * i], c);
            Float w = xyzw[4 * i + 3];
            if (w > 0.0f) {
                Float inv = 1.0f / w;
                c[0] = fmax_(0.0f, c[0] * inv); c[1] = fmax_(0.0f, c[1] * inv); c[2] = fmax_(0.0f, c[2] * inv);
            }
            // + splat (zero for PathIntegrator), then * scale
            rgb[3 * i] = (c[0] + 0.0f) * scale; rgb[3 * i + 1] = (c[1] + 0.0f) * scale; rgb[3 * i + 2] = (c[2] + 0.0f) * scale;
        }
    }
};

// SampleIntegratorCore::render_tile (sampler.rs:201-257).  If radiance_out is
// non-null it receives validate(L) per (pixel, sample), pixel-major.
inline void render_tile(const Scene& sc, LightDistribution& ldist, const int32_t tb[4], Film* film, Float* radiance_out, RayCounters& rc) {
    SobolSampler sampler;
    sc.init_sampler(sampler);
    FilmTile tile(&sc, tb);
    size_t k = 0;
    for (int32_t yy = tb[1]; yy < tb[3]; yy++)
        for (int32_t xx = tb[0]; xx < tb[2]; xx++) {
            sampler.start_pixel(xx, yy);
            do {
                CameraSample cs;
                cs.p_film = V2((Float)xx, (Float)yy) + sampler.get_2d();
                cs.p_lens = sampler.get_2d();
                cs.time = sampler.get_1d();
                RayDiff rdiff;
                Ray ray = generate_ray(sc, cs, &rdiff);
                rc.camera++;
                RGB l;
                if (sc.integrator == PT_INTEGRATOR_AO) l = ao_li(sc, ray, sampler, rc);
                else if (sc.integrator == PT_INTEGRATOR_DIRECTLIGHTING || sc.integrator == PT_INTEGRATOR_WHITTED) l = rec_li(sc, ray, rdiff, sampler, rc, 0);
                else l = path_li(sc, ldist, ray, sampler, rc, rdiff);
                l = validate_radiance(l);
                if (radiance_out) { radiance_out[3 * k] = l.c[0]; radiance_out[3 * k + 1] = l.c[1]; radiance_out[3 * k + 2] = l.c[2]; k++; }
                tile.add_sample(cs.p_film, l, 1.0f);
            } while (sampler.start_next_sample());
        }
    if (film) film->merge(tile);
}

// SampleIntegratorCore::render (sampler.rs:259-325): 16x16 tiles pulled from an
// atomic counter (stand-in for rayon's par_iter).
inline void default_tiles(const Scene& sc, std::vector<pt_tile>* out) {
    const int32_t TS = 16;
    int32_t ex = sc.sample_bounds[2] - sc.sample_bounds[0], ey = sc.sample_bounds[3] - sc.sample_bounds[1];
    int32_t ntx = (ex + TS - 1) / TS, nty = (ey + TS - 1) / TS;
    for (int32_t y = 0; y < nty; y++)
        for (int32_t x = 0; x < ntx; x++) {
            pt_tile t;
            t.x0 = sc.sample_bounds[0] + x * TS;
            t.x1 = std::min(t.x0 + TS, sc.sample_bounds[2]);
            t.y0 = sc.sample_bounds[1] + y * TS;
            t.y1 = std::min(t.y0 + TS, sc.sample_bounds[3]);
            out->push_back(t);
        }
}
inline void render(const Scene& sc, LightDistribution& ldist, const pt_tile* tiles, size_t n_tiles, int n_threads, Film* film, RayCounters* total) {
    std::atomic<size_t> next(0);
    std::mutex mu;
    auto worker = [&]() {
        RayCounters rc;
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= n_tiles) break;
            int32_t tb[4] = {tiles[i].x0, tiles[i].y0, tiles[i].x1, tiles[i].y1};
            render_tile(sc, ldist, tb, film, nullptr, rc);
        }
        std::lock_guard<std::mutex> g(mu);
        total->camera += rc.camera; total->regular += rc.regular; total->shadow += rc.shadow;
        total->nodes += rc.nodes; total->tris += rc.tris; total->vertices += rc.vertices;
    };
    if (n_threads <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < n_threads; i++) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

// ---- scene assembly from the flattened description
inline bool Scene::build(const pt_scene_desc& d, const std::string& data_dir, std::string* err) {
    if (!sobol.load(data_dir + "/sobol_tables.bin")) { *err = "cannot load sobol_tables.bin from " + data_dir; return false; }
    geom.P.resize(d.n_vertices);
    for (uint32_t i = 0; i < d.n_vertices; i++) geom.P[i] = V3(d.P[3 * i], d.P[3 * i + 1], d.P[3 * i + 2]);
    if (d.N) { geom.N.resize(d.n_vertices); for (uint32_t i = 0; i < d.n_vertices; i++) geom.N[i] = V3(d.N[3 * i], d.N[3 * i + 1], d.N[3 * i + 2]); }
    if (d.S) { geom.S.resize(d.n_vertices); for (uint32_t i = 0; i < d.n_vertices; i++) geom.S[i] = V3(d.S[3 * i], d.S[3 * i + 1], d.S[3 * i + 2]); }
    if (d.UV) { geom.UV.resize(d.n_vertices); for (uint32_t i = 0; i < d.n_vertices; i++) geom.UV[i] = V2(d.UV[2 * i], d.UV[2 * i + 1]); }
    geom.idx.assign(d.indices, d.indices + 3 * (size_t)d.n_triangles);
    geom.tri_mesh.assign(d.tri_mesh, d.tri_mesh + d.n_triangles);
    geom.mesh.resize(d.n_meshes);
    mesh_material.resize(d.n_meshes);
    mesh_light_params.resize(d.n_meshes);
    for (uint32_t i = 0; i < d.n_meshes; i++) {
        uint32_t f = d.meshes[i].flags;
        MeshFlags& mf = geom.mesh[i];
        mf.two_sided = f & PT_MESH_TWO_SIDED; mf.reverse_orientation = f & PT_MESH_REVERSE_ORIENTATION;
        mf.swaps_handedness = f & PT_MESH_SWAPS_HANDEDNESS;
        mf.has_n = (f & PT_MESH_HAS_N) && d.N; mf.has_s = (f & PT_MESH_HAS_S) && d.S; mf.has_uv = (f & PT_MESH_HAS_UV) && d.UV;
        mesh_material[i] = d.meshes[i].material;
        mesh_light_params[i] = d.meshes[i].area_light;
    }
    materials.assign(d.materials, d.materials + d.n_materials);
    if (d.n_textures) textures.assign(d.textures, d.textures + d.n_textures);
    images.resize(d.n_images);
    for (uint32_t i = 0; i < d.n_images; i++) images[i].init(d.images[i]);
    for (uint32_t i = 0; i < d.n_textures; i++)
        if (textures[i].type == PT_TEX_IMAGEMAP && (textures[i].image < 0 || (uint32_t)textures[i].image >= d.n_images)) { if (err) *err = "imagemap image index out of range"; return false; }
    for (uint32_t i = 0; i < d.n_textures; i++)
        for (int k = 0; k < 3; k++)
            if (textures[i].tex[k] >= (int32_t)i) { if (err) *err = "texture child index must be smaller than the texture's own"; return false; }
    for (const pt_material& m : materials) {
        const uint32_t refs[13] = {m.tex_kd, m.tex_ks, m.tex_kr, m.tex_kt, m.tex_opacity, m.tex_sigma, m.tex_metal_eta, m.tex_metal_k, m.tex_bump,
                                   m.tex_roughness, m.tex_uroughness, m.tex_vroughness, m.tex_eta};
        for (uint32_t r : refs) if (r > d.n_textures) { if (err) *err = "material texture index out of range"; return false; }
    }
    // spheres and object instances, spliced into the primitive lists at before_triangle (ties: creation order)
    geom.spheres.resize(d.n_spheres);
    sphere_material.resize(d.n_spheres);
    for (uint32_t s = 0; s < d.n_spheres; s++) {
        const pt_sphere& ps = d.spheres[s];
        if (ps.before_triangle > d.n_triangles) { if (err) *err = "sphere before_triangle exceeds n_triangles"; return false; }
        geom.spheres[s].init(ps.object_to_world, ps.world_to_object, (ps.flags & PT_SPHERE_REVERSE_ORIENTATION) != 0, ps.radius, ps.zmin, ps.zmax, ps.phimax);
        sphere_material[s] = ps.material;
    }
    uint32_t n_objects = 0;
    for (uint32_t i = 0; i < d.n_meshes; i++) n_objects = std::max(n_objects, d.meshes[i].object);
    for (uint32_t s = 0; s < d.n_spheres; s++) n_objects = std::max(n_objects, d.spheres[s].object);
    geom.instances.resize(d.n_instances);
    for (uint32_t i = 0; i < d.n_instances; i++) {
        const pt_instance& pi = d.instances[i];
        if (pi.object >= n_objects || pi.before_triangle > d.n_triangles) { if (err) *err = "instance object / position out of range"; return false; }
        std::memcpy(geom.instances[i].m.m, pi.instance_to_world, 64);
        std::memcpy(geom.instances[i].minv.m, pi.world_to_instance, 64);
        geom.instances[i].object = pi.object;
    }
    // list 0 = the world, list k = object k - 1
    auto make_list = [&](uint32_t tag, std::vector<uint32_t>* out) {
        struct Extra { uint32_t before, order, ref; };
        std::vector<Extra> extra;
        for (uint32_t s = 0; s < d.n_spheres; s++) if (d.spheres[s].object == tag) extra.push_back({d.spheres[s].before_triangle, d.spheres[s].order, PRIM_SPHERE | s});
        if (tag == 0) for (uint32_t i = 0; i < d.n_instances; i++) extra.push_back({d.instances[i].before_triangle, d.instances[i].order, PRIM_INSTANCE | i});
        std::stable_sort(extra.begin(), extra.end(), [](const Extra& x, const Extra& y) { return x.before != y.before ? x.before < y.before : x.order < y.order; });
        size_t e = 0;
        for (uint32_t t = 0; t <= d.n_triangles; t++) {
            while (e < extra.size() && extra[e].before <= t) out->push_back(extra[e++].ref);
            if (t < d.n_triangles && d.meshes[d.tri_mesh[t]].object == tag) out->push_back(t);
        }
    };
    const bool plain = d.n_spheres == 0 && d.n_instances == 0 && n_objects == 0;
    if (!plain) make_list(0, &geom.prim_ref);
    objects.resize(n_objects);
    geom.object_bvh.resize(n_objects);
    for (uint32_t k = 0; k < n_objects; k++) {
        objects[k].reset(new QBVH());
        QBVH& ob = *objects[k];
        ob.own_refs = true;
        ob.geom = &geom;
        make_list(k + 1, &ob.refs);
        geom.object_bvh[k] = &ob;
        if (ob.refs.size() > 1 && !ob.build(&geom, (size_t)(d.max_node_prims > 0 ? d.max_node_prims : 4), (SplitMethod)d.split_method)) {
            if (err) *err = "hlbvh: degenerate treelet centroids in an object (the reference panics here)";
            return false;
        }
    }
    for (uint32_t i = 0; i < d.n_instances; i++)
        if (objects[d.instances[i].object]->refs.empty()) { if (err) *err = "instance of an empty object"; return false; }
    // one DiffuseAreaLight per emissive primitive, in primitive order (scene_context.rs:1218-1231)
    prim_light.assign(geom.n_prims(), -1);
    for (size_t p = 0; p < geom.n_prims(); p++) {
        uint32_t ref = geom.ref(p);
        if (ref & PRIM_INSTANCE) continue;              // TransformedPrimitive::get_area_light is None
        int32_t lp = (ref & PRIM_SPHERE) ? d.spheres[ref & ~PRIM_SPHERE].area_light : mesh_light_params[geom.tri_mesh[ref]];
        if (lp >= 0) {
            AreaLight al;
            al.shape = ref;
            al.lemit = RGB(d.area_lights[lp].L[0], d.area_lights[lp].L[1], d.area_lights[lp].L[2]);
            al.two_sided = d.area_lights[lp].two_sided != 0;
            al.n_samples = (uint32_t)std::max(1, d.area_lights[lp].n_samples);
            al.area = (ref & PRIM_SPHERE) ? geom.spheres[ref & ~PRIM_SPHERE].area() : TriRef(&geom, ref).area();
            prim_light[p] = (int32_t)lights.size();
            lights.push_back(al);
        }
    }
    if (!bvh.build(&geom, (size_t)(d.max_node_prims > 0 ? d.max_node_prims : 4), (SplitMethod)d.split_method)) {
        if (err) *err = "hlbvh: degenerate treelet centroids (the reference panics here)";
        return false;
    }
    world_bound = bvh.bounds;
    // film (film.rs:62-100, :166-179)
    xres = d.xres; yres = d.yres;
    crop[0] = std::max(0, (int32_t)std::floor((Float)xres * d.crop_window[0]));
    crop[1] = std::max(0, (int32_t)std::floor((Float)yres * d.crop_window[2]));
    crop[2] = std::min((int32_t)std::ceil((Float)xres * d.crop_window[1]), xres);
    crop[3] = std::min((int32_t)std::ceil((Float)yres * d.crop_window[3]), yres);
    filter_radius[0] = d.filter_radius[0]; filter_radius[1] = d.filter_radius[1];
    std::memcpy(filter_table, d.filter_table, sizeof(filter_table));
    sample_bounds[0] = (int32_t)std::floor((Float)crop[0] - filter_radius[0]);
    sample_bounds[1] = (int32_t)std::floor((Float)crop[1] - filter_radius[1]);
    sample_bounds[2] = (int32_t)std::ceil((Float)crop[2] + filter_radius[0]);
    sample_bounds[3] = (int32_t)std::ceil((Float)crop[3] + filter_radius[1]);
    film_scale = d.film_scale;
    max_sample_luminance = d.max_sample_luminance;
    // camera (projective.rs:23-53, perspective.rs:28-44)
    Transform camera_to_screen = Transform::perspective(d.fov, 1e-2f, 1000.0f);
    Float sx0 = d.screen_window[0], sx1 = d.screen_window[1], sy0 = d.screen_window[2], sy1 = d.screen_window[3];
    Transform screen_to_raster = Transform::scale((Float)xres, (Float)yres, 1.0f) *
                                 Transform::scale(1.0f / (sx1 - sx0), 1.0f / (sy0 - sy1), 1.0f) * Transform::translate(-sx0, -sy1, 0.0f);
    Transform raster_to_screen = screen_to_raster.inverse();
    raster_to_camera = camera_to_screen.inverse() * raster_to_screen;
    std::memcpy(camera_to_world.m, d.camera_to_world, sizeof(camera_to_world.m));
    lens_radius = d.lens_radius; focal_distance = d.focal_distance;
    shutter_open = d.shutter_open; shutter_close = d.shutter_close;
    sampler_kind = d.sampler;
    halton_at_center = d.halton_sample_at_center != 0;
    // Sobol' rounds the sample count up to a power of two (sobol.rs:23); Halton takes it as given
    spp = sampler_kind == PT_SAMPLER_HALTON ? std::max(1, d.spp) : (int32_t)round_up_pow2((uint32_t)std::max(1, d.spp));
    max_depth = d.max_depth; rr_threshold = d.rr_threshold; light_strategy = d.light_strategy;
    integrator = d.integrator; ao_samples = d.ao_samples > 0 ? d.ao_samples : 64; ao_cos_sample = d.ao_cos_sample != 0;
    direct_strategy = d.direct_strategy;
    return true;
}

}  // namespace orc
