// ORACLE -- test infrastructure only.  C entry points over the CPU restatement
// (orc_*.hpp) so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg can drive it through ctypes.  Nothing under pbrt-r3_amd/ may link this.
#include "orc_render.hpp"
#include <chrono>
#include <thread>

using namespace orc;

struct orc_scene {
    Scene sc;
    LightDistribution ld;
    std::string err;
};

// Rays are independent: large batches are cut into contiguous chunks over host threads (per-chunk stats, summed).
template <class F>
static void for_ray_chunks(uint32_t n, QBVH::Stats* total, F body) {
    unsigned nt = n >= 65536 ? std::min(32u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    std::vector<QBVH::Stats> st(nt);
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++) {
        uint32_t lo = (uint32_t)((uint64_t)n * k / nt), hi = (uint32_t)((uint64_t)n * (k + 1) / nt);
        if (nt == 1) body(lo, hi, &st[k]);
        else th.emplace_back([=, &st, &body] { body(lo, hi, &st[k]); });
    }
    for (auto& t : th) t.join();
    for (auto& x : st) { total->nodes += x.nodes; total->tris += x.tris; }
}

extern "C" {

const char* orc_header(void) { return "oracle: CPU restatement of pbrt-r3 (test infrastructure, not the product)"; }

int orc_scene_create(const pt_scene_desc* d, const char* data_dir, orc_scene** out) {
    orc_scene* s = new orc_scene;
    if (!s->sc.build(*d, data_dir, &s->err)) {
        std::fprintf(stderr, "orc_scene_create: %s\n", s->err.c_str());
        delete s;
        return 1;
    }
    s->ld.init(&s->sc);
    *out = s;
    return 0;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

void orc_scene_info(const orc_scene* s, pt_scene_info* o) {
    std::memset(o, 0, sizeof(*o));
    for (int i = 0; i < 4; i++) { o->sample_bounds[i] = s->sc.sample_bounds[i]; o->cropped_bounds[i] = s->sc.crop[i]; }
    o->spp = s->sc.spp;
    o->n_lights = (uint32_t)s->sc.lights.size();
    o->n_nodes = (uint32_t)s->sc.bvh.n_interior4;
    o->n_leaves = (uint32_t)s->sc.bvh.n_leaves;
    const Bounds3& b = s->sc.world_bound;
    o->world_bound[0] = b.min.x; o->world_bound[1] = b.min.y; o->world_bound[2] = b.min.z;
    o->world_bound[3] = b.max.x; o->world_bound[4] = b.max.y; o->world_bound[5] = b.max.z;
}

// ordered primitive numbers (BVH leaf order) for structural comparison
uint32_t orc_bvh_ordered_prims(const orc_scene* s, uint32_t* out, uint32_t cap) {
    uint32_t n = (uint32_t)s->sc.bvh.prims.size();
    for (uint32_t i = 0; i < n && i < cap; i++) out[i] = (uint32_t)s->sc.bvh.prims[i];
    return n;
}

// diagnostic: node visits by tree depth (tools/depth_hist.py).  enable = 1 computes the world tree's depths and starts counting.
void orc_depth_hist(orc_scene* s, int enable, uint64_t* out32) {
    if (enable) { s->sc.bvh.compute_depths(); for (auto& h : g_depth_hist) h = 0; g_depth_hist_on = true; return; }
    g_depth_hist_on = false;
    for (int i = 0; i < 32; i++) out32[i] = g_depth_hist[i];
}

// bits of orc::reference_panics() since the last reset (see orc_sampling.hpp)
uint32_t orc_reference_panics(int reset) { return reset ? reference_panics().exchange(0u) : reference_panics().load(); }

static void fill_counters(const RayCounters& rc, pt_counters* c) {
    if (!c) return;
    c->camera_rays += rc.camera; c->regular_rays += rc.regular; c->shadow_rays += rc.shadow;
    c->nodes_visited += rc.nodes; c->tris_tested += rc.tris; c->path_vertices += rc.vertices;
}

// Render `tiles` (or all 16x16 tiles when n_tiles == 0) on n_threads; returns
// wall seconds.  xyzw_out may be NULL (timing only).
double orc_render(orc_scene* s, const pt_tile* tiles, uint32_t n_tiles, int n_threads, float* xyzw_out, pt_counters* counters) {
    std::vector<pt_tile> all;
    if (n_tiles == 0) { default_tiles(s->sc, &all); tiles = all.data(); n_tiles = (uint32_t)all.size(); }
    Film film(&s->sc);
    RayCounters rc;
    auto t0 = std::chrono::steady_clock::now();
    render(s->sc, s->ld, tiles, n_tiles, n_threads, &film, &rc);
    auto t1 = std::chrono::steady_clock::now();
    if (xyzw_out) std::memcpy(xyzw_out, film.xyzw.data(), film.xyzw.size() * sizeof(float));
    fill_counters(rc, counters);
    return std::chrono::duration<double>(t1 - t0).count();
}

void orc_resolve_rgb(const float* xyzw, uint64_t n_pixels, float scale, float* rgb) { Film::resolve_xyzw(xyzw, (size_t)n_pixels, scale, rgb); }

void orc_radiance_samples(orc_scene* s, const pt_tile* tile, float* out_rgb) {
    RayCounters rc;
    int32_t tb[4] = {tile->x0, tile->y0, tile->x1, tile->y1};
    render_tile(s->sc, s->ld, tb, nullptr, out_rgb, rc);
}

void orc_trace_closest(const orc_scene* s, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out, pt_counters* counters) {
    QBVH::Stats st;
    for_ray_chunks(n, &st, [&](uint32_t lo, uint32_t hi, QBVH::Stats* cs) {
        for (uint32_t i = lo; i < hi; i++) {
            Ray r(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
            SurfHit h;
            if (s->sc.bvh.intersect(r, &h, cs)) { out[i].t = r.t_max; out[i].prim = h.prim; out[i].b0 = h.b0; out[i].b1 = h.b1; }
            else { out[i].t = 0.0f; out[i].prim = -1; out[i].b0 = 0.0f; out[i].b1 = 0.0f; }
        }
    });
    if (counters) { counters->regular_rays += n; counters->nodes_visited += st.nodes; counters->tris_tested += st.tris; }
}
void orc_trace_any(const orc_scene* s, uint32_t n, const float* o, const float* d, const float* tmax, uint8_t* out, pt_counters* counters) {
    QBVH::Stats st;
    for_ray_chunks(n, &st, [&](uint32_t lo, uint32_t hi, QBVH::Stats* cs) {
        for (uint32_t i = lo; i < hi; i++) {
            Ray r(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
            out[i] = s->sc.bvh.intersect_p(r, cs) ? 1 : 0;
        }
    });
    if (counters) { counters->shadow_rays += n; counters->nodes_visited += st.nodes; counters->tris_tested += st.tris; }
}
// accelerators/exhaustive cross-check: every primitive, no BVH
void orc_trace_exhaustive(const orc_scene* s, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out) {
    for (uint32_t i = 0; i < n; i++) {
        Ray r(V3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), V3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]);
        SurfHit h;
        if (s->sc.bvh.intersect_exhaustive(r, &h)) { out[i].t = r.t_max; out[i].prim = h.prim; out[i].b0 = h.b0; out[i].b1 = h.b1; }
        else { out[i].t = 0.0f; out[i].prim = -1; out[i].b0 = 0.0f; out[i].b1 = 0.0f; }
    }
}

void orc_generate_camera_rays(const orc_scene* s, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, float* out_o, float* out_d, float* out_pfilm) {
    SobolSampler sm;
    s->sc.init_sampler(sm);
    for (uint32_t i = 0; i < n; i++) {
        sm.start_pixel(pixel_xy[2 * i], pixel_xy[2 * i + 1]);
        sm.set_sample_number(sample_index[i]);
        CameraSample cs;
        cs.p_film = V2((Float)pixel_xy[2 * i], (Float)pixel_xy[2 * i + 1]) + sm.get_2d();
        cs.p_lens = sm.get_2d();
        cs.time = sm.get_1d();
        Ray r = generate_ray(s->sc, cs);
        out_o[3 * i] = r.o.x; out_o[3 * i + 1] = r.o.y; out_o[3 * i + 2] = r.o.z;
        out_d[3 * i] = r.d.x; out_d[3 * i + 1] = r.d.y; out_d[3 * i + 2] = r.d.z;
        out_pfilm[2 * i] = cs.p_film.x; out_pfilm[2 * i + 1] = cs.p_film.y;
    }
}
void orc_sobol_samples(const orc_scene* s, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim, float* out) {
    SobolSampler sm;
    s->sc.init_sampler(sm);
    for (uint32_t i = 0; i < n; i++) {
        sm.start_pixel(pixel_xy[2 * i], pixel_xy[2 * i + 1]);
        sm.set_sample_number(sample_index[i]);
        out[i] = sm.sample_dimension(sm.interval_sample_index, dim[i]);
    }
}

// light distribution at point p: writes n_lights func values + (n_lights+1) cdf values
uint32_t orc_light_distribution(orc_scene* s, const float p[3], float* func, float* cdf) {
    const Distribution1D* d = s->ld.lookup(V3(p[0], p[1], p[2]));
    for (size_t i = 0; i < d->func.size(); i++) func[i] = d->func[i];
    for (size_t i = 0; i < d->cdf.size(); i++) cdf[i] = d->cdf[i];
    return (uint32_t)d->func.size();
}
void orc_light_voxels(const orc_scene* s, uint32_t v[3]) { v[0] = s->ld.voxels[0]; v[1] = s->ld.voxels[1]; v[2] = s->ld.voxels[2]; }

// ---- scalar helpers for known-answer tests
uint32_t orc_order_entry(uint32_t hit_mask, uint32_t node_idx) { return order_entry(hit_mask, node_idx); }
float orc_radical_inverse(uint32_t base_index, uint64_t a) { return radical_inverse(base_index, a); }
void orc_rng_floats(uint64_t seq, int use_seq, uint32_t n, float* out_f, uint32_t* out_u) {
    RNG r;
    if (use_seq) r.set_sequence(seq);
    for (uint32_t i = 0; i < n; i++) { if (out_u) out_u[i] = r.uniform_uint32(); else out_f[i] = r.uniform_float(); }
}
// core/base/functions.rs:64-152, bounds3.rs:184-195 (the reference's tests/bitops.rs, find_interval.rs, bounds.rs)
uint32_t orc_log2int(uint32_t v) { return log2int(v); }
uint32_t orc_log2int64(uint64_t v) { return 63u - (uint32_t)__builtin_clzll(v); }
uint32_t orc_round_up_pow2(uint32_t v) { return round_up_pow2(v); }
uint32_t orc_ctz(uint32_t v) { return v ? (uint32_t)__builtin_ctz(v) : 32u; }
uint32_t orc_find_interval_cdf(const float* cdf, uint32_t n, float u) { return (uint32_t)Distribution1D::find_interval_cdf(std::vector<Float>(cdf, cdf + n), u); }
float orc_bounds_distance_squared(const float b[6], const float p[3]) { return Bounds3(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5])).distance_squared(V3(p[0], p[1], p[2])); }
void orc_bounds_union(const float a[6], int a_default, const float b[6], int b_is_point, float out[6]) {
    Bounds3 ba = a_default ? Bounds3() : Bounds3(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]));
    Bounds3 r = b_is_point == 1 ? bunion_p(ba, V3(b[0], b[1], b[2]))
                                : bunion(ba, b_is_point == 2 ? Bounds3() : Bounds3(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5])));      // 2: Bounds3::default()
    out[0] = r.min.x; out[1] = r.min.y; out[2] = r.min.z; out[3] = r.max.x; out[4] = r.max.y; out[5] = r.max.z;
}
float orc_next_float_up(float v) { return next_float_up(v); }
float orc_next_float_down(float v) { return next_float_down(v); }
// Distribution1D: returns offset; writes pdf / remapped
uint32_t orc_dist1d_sample_discrete(const float* f, uint32_t n, float u, float* pdf, float* remapped) {
    Distribution1D d(std::vector<Float>(f, f + n));
    return (uint32_t)d.sample_discrete(u, pdf, remapped);
}
float orc_dist1d_sample_continuous(const float* f, uint32_t n, float u, float* pdf, uint32_t* off) {
    Distribution1D d(std::vector<Float>(f, f + n));
    size_t o;
    float v = d.sample_continuous(u, pdf, &o);
    *off = (uint32_t)o;
    return v;
}
float orc_dist1d_discrete_pdf(const float* f, uint32_t n, uint32_t i) { return Distribution1D(std::vector<Float>(f, f + n)).discrete_pdf(i); }
void orc_cosine_sample_hemisphere(float u0, float u1, float out[3]) { V3 w = cosine_sample_hemisphere(V2(u0, u1)); out[0] = w.x; out[1] = w.y; out[2] = w.z; }

// Sobol' primitives straight off the table (tests/sampling.rs:110-135)
static SobolTables g_tab;
int orc_sobol_load(const char* data_dir) { return g_tab.load(std::string(data_dir) + "/sobol_tables.bin") ? 0 : 1; }
float orc_sobol_sample_float(int64_t a, uint32_t dim) { return sobol_sample_float(g_tab, a, dim, 0); }
uint64_t orc_sobol_interval_to_index(uint32_t m, uint64_t frame, int32_t px, int32_t py) { return sobol_interval_to_index(g_tab, m, frame, px, py); }

}  // extern "C"

// Standalone SobolSampler walk for tests/sampling.rs:137-201 (check_sampler): the first
// get_2d() of every sample of one pixel.
extern "C" uint32_t orc_sobol_pixel_first2d(uint32_t spp, const int32_t bounds[4], int32_t px, int32_t py, float* out_xy) {
    SobolSampler sm;
    sm.init(&g_tab, spp, bounds);
    sm.start_pixel(px, py);
    uint32_t n = 0;
    do {
        V2 u = sm.get_2d();
        out_xy[2 * n] = u.x; out_xy[2 * n + 1] = u.y;
        n++;
    } while (sm.start_next_sample());
    return n;
}

// Standalone HaltonSampler walk: every sample's first get_2d() of one pixel (tests/sampling.rs style checks)
extern "C" uint32_t orc_halton_pixel_first2d(uint32_t spp, const int32_t bounds[4], int32_t px, int32_t py, float* out_xy) {
    SobolSampler sm;
    sm.init_halton(spp, bounds, false);
    sm.start_pixel(px, py);
    uint32_t n = 0;
    do {
        V2 u = sm.get_2d();
        out_xy[2 * n] = u.x; out_xy[2 * n + 1] = u.y;
        n++;
    } while (sm.start_next_sample());
    return n;
}
// scrambled radical inverse with a caller-supplied permutation (tests/sampling.rs:24-64)
extern "C" float orc_scrambled_radical_inverse(uint32_t base, const uint16_t* perm, uint64_t a) { return scrambled_radical_inverse(base, perm, a); }
extern "C" float orc_halton_dimension(uint32_t dim, uint64_t index) {
    const HaltonTables& H = HaltonTables::get();
    return scrambled_radical_inverse(H.primes[dim], &H.perms[H.prime_sums[dim]], index);
}
extern "C" uint64_t orc_prime(uint32_t i) { return HaltonTables::get().primes[i]; }
extern "C" int64_t orc_halton_index(const int32_t bounds[4], int32_t px, int32_t py, int64_t sample_num) {
    SobolSampler sm;
    sm.init_halton(1, bounds, false);
    sm.pixel[0] = px; sm.pixel[1] = py;
    return sm.get_index_for_sample(sample_num);
}

// BSDF of one material on the canonical frame ns = ng = (0,0,1), ss = (1,0,0)  (bsdf.rs:92-270)
static bool canonical_bsdf(const orc_scene* s, uint32_t material, BSDF* b) {
    SurfHit si;
    si.n = V3(0.0f, 0.0f, 1.0f); si.sh_n = si.n;
    si.sh_dpdu = V3(1.0f, 0.0f, 0.0f);
    return make_bsdf_from_material(s->sc.materials[material], si, b);
}
extern "C" void orc_bsdf_eval(const orc_scene* s, uint32_t material, uint32_t n, const float* wo, const float* wi, uint32_t flags, float* f_out, float* pdf_out) {
    BSDF b;
    bool have = canonical_bsdf(s, material, &b);
    for (uint32_t i = 0; i < n; i++) {
        V3 o(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), w(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
        RGB f = have ? b.f(o, w, flags) : RGB();
        f_out[3 * i] = f.c[0]; f_out[3 * i + 1] = f.c[1]; f_out[3 * i + 2] = f.c[2];
        pdf_out[i] = have ? b.pdf(o, w, flags) : 0.0f;
    }
}
extern "C" void orc_bsdf_sample(const orc_scene* s, uint32_t material, uint32_t n, const float* wo, const float* u, uint32_t flags, float* f_out, float* wi_out,
                                float* pdf_out, uint32_t* type_out) {
    BSDF b;
    bool have = canonical_bsdf(s, material, &b);
    for (uint32_t i = 0; i < n; i++) {
        RGB f; V3 wi(0.0f, 0.0f, 0.0f); Float pdf = 0.0f; uint32_t t = 0;
        if (!have || !b.sample_f(V3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), V2(u[2 * i], u[2 * i + 1]), flags, &f, &wi, &pdf, &t)) {
            f = RGB(); wi = V3(0.0f, 0.0f, 0.0f); pdf = 0.0f; t = 0;
        }
        f_out[3 * i] = f.c[0]; f_out[3 * i + 1] = f.c[1]; f_out[3 * i + 2] = f.c[2];
        wi_out[3 * i] = wi.x; wi_out[3 * i + 1] = wi.y; wi_out[3 * i + 2] = wi.z;
        pdf_out[i] = pdf;
        type_out[i] = t;
    }
}
extern "C" float orc_roughness_to_alpha(float r) { return TRDist::roughness_to_alpha(r); }

// Diagnostic for tools/sim_wave_sched.py: render one tile single-threaded and return the traversal step stream of
// every ray the path integrator traced (0 = node visit, k = leaf with k triangle tests, 255 = end of ray).
extern "C" uint64_t orc_step_log(orc_scene* s, const pt_tile* tile, uint8_t* out, uint64_t cap) {
    std::vector<uint8_t> log;
    s->sc.bvh.step_log = &log;
    int32_t tb[4] = {tile->x0, tile->y0, tile->x1, tile->y1};
    RayCounters rc;
    std::vector<float> rad((size_t)(tb[2] - tb[0]) * (tb[3] - tb[1]) * s->sc.spp * 3);
    render_tile(s->sc, s->ld, tb, nullptr, rad.data(), rc);
    s->sc.bvh.step_log = nullptr;
    uint64_t n = std::min<uint64_t>(cap, log.size());
    std::memcpy(out, log.data(), n);
    return log.size();
}
