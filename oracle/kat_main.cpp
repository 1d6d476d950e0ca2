// ORACLE -- test infrastructure only.  The reference's own known-answer /
// property tests for the hot path, restated against the CPU restatement so the
// oracle is pinned before anything is compared with it:
//   tests/shapes.rs:17-122   triangle_watertight
//   tests/shapes.rs:156-196  triangle_reintersect (+ helper :369-414)
//   tests/shapes.rs:200-278  triangle_solid_angle
//   tests/shapes.rs:480-504  triangle_badcases
//   tests/shapes.rs:280-329  sphere_solid_angle (+ Shape::solid_angle, shape.rs:56-73)
//   tests/shapes.rs:416-457  full_sphere_reintersect, partial_sphere_reintersect
// Usage: orc_kat [name ...]   (no argument = all); exit code = number of failures.
#include "orc_render.hpp"
#include <cstdlib>
#include <string>

using namespace orc;

static int g_fail = 0;
#define CHECK(cond, ...)                                                    \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("  CHECK FAILED %s:%d: %s  ", __FILE__, __LINE__, #cond); \
            std::printf(__VA_ARGS__);                                       \
            std::printf("\n");                                              \
            return false;                                                   \
        }                                                                   \
    } while (0)

static Float p_exp(RNG& rng, Float e) { return std::pow(10.0f, lerpf(rng.uniform_float(), -e, e)); }
static Float p_unif(RNG& rng, Float range) { return lerpf(rng.uniform_float(), -range, range); }
// C++ leaves argument evaluation order unspecified; Rust is left-to-right.
static V2 rand2(RNG& rng) { Float a = rng.uniform_float(); Float b = rng.uniform_float(); return V2(a, b); }
static V3 spherical_direction(Float sin_theta, Float cos_theta, Float phi) {
    return V3(sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta);
}

static Geometry make_mesh(const std::vector<V3>& verts, const std::vector<uint32_t>& indices) {
    // create_triangle_mesh (triangle.rs:696-731): twosided defaults to true, degenerate triangles dropped
    Geometry g;
    g.P = verts;
    g.mesh.resize(1);
    for (size_t i = 0; i + 2 < indices.size(); i += 3) {
        Geometry tmp;
        tmp.P = verts;
        tmp.mesh.resize(1);
        tmp.idx = {indices[i], indices[i + 1], indices[i + 2]};
        tmp.tri_mesh = {0};
        if (TriRef(&tmp, 0).area() > 1e-16f) {
            g.idx.push_back(indices[i]); g.idx.push_back(indices[i + 1]); g.idx.push_back(indices[i + 2]);
            g.tri_mesh.push_back(0);
        }
    }
    return g;
}

static bool triangle_watertight() {
    RNG rng(12111);
    const uint32_t n_theta = 16, n_phi = 8;
    std::vector<V3> vertices;
    for (uint32_t t = 0; t < n_theta; t++) {
        Float theta = kPi * (Float)t / (Float)(n_theta - 1);
        Float cos_theta = std::cos(theta), sin_theta = std::sin(theta);
        for (uint32_t p = 0; p < n_phi; p++) {
            Float phi = 2.0f * kPi * (Float)p / (Float)(n_phi - 1);
            Float radius = 1.0f;
            if (t == 0) vertices.push_back(V3(0.0f, 0.0f, radius));
            else if (t == n_theta - 1) vertices.push_back(V3(0.0f, 0.0f, -radius));
            else if (p == n_phi - 1) vertices.push_back(vertices[vertices.size() - (n_phi - 1)]);
            else {
                radius += 5.0f * rng.uniform_float();
                vertices.push_back(V3() + radius * spherical_direction(sin_theta, cos_theta, phi));
            }
        }
    }
    std::vector<uint32_t> indices;
    auto offset = [&](uint32_t t, uint32_t p) { return t * n_phi + p; };
    for (uint32_t p = 0; p < n_phi - 1; p++) { indices.push_back(offset(0, 0)); indices.push_back(offset(1, p)); indices.push_back(offset(1, p + 1)); }
    for (uint32_t t = 1; t < n_theta - 2; t++)
        for (uint32_t p = 0; p < n_phi - 1; p++) {
            indices.push_back(offset(t, p)); indices.push_back(offset(t + 1, p)); indices.push_back(offset(t + 1, p + 1));
            indices.push_back(offset(t, p)); indices.push_back(offset(t + 1, p + 1)); indices.push_back(offset(t, p + 1));
        }
    for (uint32_t p = 0; p < n_phi - 1; p++) { indices.push_back(offset(n_theta - 1, 0)); indices.push_back(offset(n_theta - 2, p)); indices.push_back(offset(n_theta - 2, p + 1)); }
    Geometry g = make_mesh(vertices, indices);
    size_t nt = g.n_tris();
    // the same mesh through the BVH must agree with the exhaustive loop on "hit at all"
    QBVH bvh;
    bvh.build(&g, 4, SPLIT_SAH);
    for (int it = 0; it < 100000; it++) {
        V2 u = rand2(rng);
        V3 p = V3() + uniform_sample_sphere(u) * 0.5f;
        u = rand2(rng);
        Ray ray(p, uniform_sample_sphere(u), kInfinity);
        int n_hits = 0;
        for (size_t i = 0; i < nt; i++) {
            Ray r2 = ray;
            Float t; SurfHit si;
            if (TriRef(&g, (uint32_t)i).intersect(r2, &t, &si)) n_hits++;
        }
        CHECK(n_hits >= 1, "iteration %d", it);
        { Ray r2 = ray; SurfHit si; CHECK(bvh.intersect(r2, &si), "bvh missed, iteration %d", it); }
        V3 p_vertex = vertices[rng.uniform_uint32_threshold((uint32_t)vertices.size())];
        Ray ray2(p, p_vertex - p, kInfinity);
        n_hits = 0;
        for (size_t i = 0; i < nt; i++) {
            Ray r2 = ray2;
            Float t; SurfHit si;
            if (TriRef(&g, (uint32_t)i).intersect(r2, &t, &si)) n_hits++;
        }
        CHECK(n_hits >= 1, "vertex ray, iteration %d", it);
    }
    return true;
}

template <class F>
static Geometry get_random_triangle(F value) {
    for (;;) {
        std::vector<V3> v(3);
        for (int i = 0; i < 3; i++) { Float a = value(); Float b = value(); Float c = value(); v[i] = V3(a, b, c); }
        if (length_squared(cross(v[1] - v[0], v[2] - v[0])) < 1e-20f) continue;
        Geometry g = make_mesh(v, {0, 1, 2});
        if (g.n_tris() > 0) return g;
    }
}
static Ray spawn_ray(const SurfHit& si, V3 d) { return Ray(offset_ray_origin(si.p, si.p_error, si.n, d), d, kInfinity); }
static Ray spawn_ray_to_point(const SurfHit& si, V3 p2) {   // surface_interaction.rs:204-210
    V3 d = p2 - si.p;
    return Ray(offset_ray_origin(si.p, si.p_error, si.n, d), d, 1.0f - kShadowEpsilon);
}
static bool hits(const Geometry& g, const Ray& r) { Ray r2 = r; Float t; SurfHit si; return TriRef(&g, 0).intersect(r2, &t, &si); }
static bool hits_p(const Geometry& g, const Ray& r) { return TriRef(&g, 0).intersect_p(r); }

static bool test_reintersect_convex(const Geometry& g, RNG& rng) {   // tests/shapes.rs:369-414
    V3 o(0, 0, 0);
    o.x = p_exp(rng, 8.0f); o.y = p_exp(rng, 8.0f); o.z = p_exp(rng, 8.0f);
    Bounds3 bbox = TriRef(&g, 0).world_bound();
    V3 t(0, 0, 0);
    t.x = rng.uniform_float(); t.y = rng.uniform_float(); t.z = rng.uniform_float();
    V3 p2 = bbox.lerp(t);
    Ray r(o, p2 - o, kInfinity);
    if (rng.uniform_float() < 0.5f) r.d = normalize(r.d);
    Float th; SurfHit isect;
    if (TriRef(&g, 0).intersect(r, &th, &isect)) {
        for (int j = 0; j < 10000; j++) {
            V2 u = rand2(rng);
            V3 w = face_forward(uniform_sample_sphere(u), isect.n);
            Ray r_out = spawn_ray(isect, w);
            CHECK(!hits_p(g, r_out), "convex spawn_ray intersect_p j=%d", j);
            CHECK(!hits(g, r_out), "convex spawn_ray intersect j=%d", j);
            V3 q(0, 0, 0);
            q.x = p_exp(rng, 8.0f); q.y = p_exp(rng, 8.0f); q.z = p_exp(rng, 8.0f);
            V3 ww = face_forward(q - isect.p, isect.n);
            Ray r_out2 = spawn_ray_to_point(isect, isect.p + ww);
            CHECK(!hits_p(g, r_out2), "convex spawn_ray_to intersect_p j=%d", j);
            CHECK(!hits(g, r_out2), "convex spawn_ray_to intersect j=%d", j);
        }
    }
    return true;
}

static bool triangle_reintersect() {
    for (uint64_t i = 0; i < 1000; i++) {
        RNG rng(i);
        Geometry g = get_random_triangle([&]() { return p_unif(rng, 10.0f); });
        if (!test_reintersect_convex(g, rng)) return false;
        V2 u = rand2(rng);
        V3 p_tri, n_tri, perr; Float pdf;
        TriRef(&g, 0).sample(u, &p_tri, &n_tri, &perr, &pdf);
        V3 o(0, 0, 0);
        o.x = p_exp(rng, 8.0f); o.y = p_exp(rng, 8.0f); o.z = p_exp(rng, 8.0f);
        Ray r(o, p_tri - o, kInfinity);
        Float th; SurfHit isect;
        if (TriRef(&g, 0).intersect(r, &th, &isect)) {
            for (int j = 0; j < 10000; j++) {
                V2 uu = rand2(rng);
                V3 w = uniform_sample_sphere(uu);
                Ray r_out = spawn_ray(isect, w);
                CHECK(!hits_p(g, r_out), "tri %llu j=%d spawn_ray", (unsigned long long)i, j);
                V3 p2(0, 0, 0);
                p2.x = p_exp(rng, 8.0f); p2.y = p_exp(rng, 8.0f); p2.z = p_exp(rng, 8.0f);
                Ray r_out2 = spawn_ray_to_point(isect, p2);
                CHECK(!hits_p(g, r_out2), "tri %llu j=%d spawn_ray_to_point intersect_p", (unsigned long long)i, j);
                CHECK(!hits(g, r_out2), "tri %llu j=%d spawn_ray_to_point intersect", (unsigned long long)i, j);
            }
        }
    }
    return true;
}

static bool triangle_solid_angle() {
    for (uint64_t i = 0; i < 50; i++) {
        const Float range = 10.0f;
        RNG rng(100 + i);
        Geometry g = get_random_triangle([&]() { return p_unif(rng, range); });
        V3 pc(0, 0, 0);
        pc.x = p_unif(rng, range); pc.y = p_unif(rng, range); pc.z = p_unif(rng, range);
        uint32_t axis = rng.uniform_uint32() % 3;
        pc[axis] = rng.uniform_float() > 0.5f ? (-range - 3.0f) : (range + 3.0f);
        const int count = 512 * 1024;
        int n_hit = 0;
        for (int j = 0; j < count; j++) {
            V2 u(radical_inverse(0, (uint64_t)j), radical_inverse(1, (uint64_t)j));
            Ray ray(pc, uniform_sample_sphere(u), kInfinity);
            if (hits_p(g, ray)) n_hit++;
        }
        const Float inv4pi = kInvPi * 0.25f;
        Float unif_estimate = (Float)n_hit / ((Float)count * inv4pi);
        Float tri_sample_estimate = 0.0f;
        for (int j = 0; j < count; j++) {
            V2 u(radical_inverse(0, (uint64_t)j), radical_inverse(1, (uint64_t)j));
            V3 p, n, pe; Float pdf;
            bool ok = TriRef(&g, 0).sample_from(pc, u, &p, &n, &pe, &pdf);
            CHECK(ok, "sample_from returned None, tri %llu j=%d", (unsigned long long)i, j);
            CHECK(pdf > 0.0f, "pdf <= 0");
            tri_sample_estimate += 1.0f / ((Float)count * pdf);
        }
        auto error = [](Float a, Float b) { return (std::fabs(a) < 1e-4f || std::fabs(b) < 1e-4f) ? std::fabs(a - b) : std::fabs((a - b) / b); };
        if (tri_sample_estimate > 1e-3f)
            CHECK(error(tri_sample_estimate, unif_estimate) < 0.1f, "unif %g tri %g index %llu", unif_estimate, tri_sample_estimate, (unsigned long long)i);
    }
    return true;
}

static bool triangle_badcases() {
    std::vector<V3> p = {V3(-1113.45459f, -79.049614f, -56.2431908f), V3(-1113.45459f, -87.0922699f, -56.2431908f), V3(-1113.45459f, -79.2090149f, -56.2431908f)};
    Geometry g = make_mesh(p, {0, 1, 2});
    if (g.n_tris() > 0) {
        Ray ray(V3(-1081.47925f, 99.9999542f, 87.7701111f), V3(-32.1072998f, -183.355865f, -144.607635f), 0.9999f);
        CHECK(!hits(g, ray), "regression triangle must miss");
    }
    return true;
}

// ---- spheres
static bool sph_hits(const Sphere& s, const Ray& r) { Ray r2 = r; Float t; SurfHit si; return s.intersect(r2, &t, &si); }
static bool sphere_reintersect_convex(const Sphere& sp, RNG& rng, int* n_found) {   // tests/shapes.rs:369-414
    V3 o(0, 0, 0);
    o.x = p_exp(rng, 8.0f); o.y = p_exp(rng, 8.0f); o.z = p_exp(rng, 8.0f);
    Bounds3 bbox = sp.world_bound();
    V3 t(0, 0, 0);
    t.x = rng.uniform_float(); t.y = rng.uniform_float(); t.z = rng.uniform_float();
    V3 p2 = bbox.lerp(t);
    Ray r(o, p2 - o, kInfinity);
    if (rng.uniform_float() < 0.5f) r.d = normalize(r.d);
    Float th; SurfHit isect;
    if (sp.intersect(r, &th, &isect)) {
        (*n_found)++;
        for (int j = 0; j < 10000; j++) {
            V2 u = rand2(rng);
            V3 w = face_forward(uniform_sample_sphere(u), isect.n);
            Ray r_out = spawn_ray(isect, w);
            CHECK(!sp.intersect_p(r_out), "sphere spawn_ray intersect_p j=%d", j);
            CHECK(!sph_hits(sp, r_out), "sphere spawn_ray intersect j=%d", j);
            V3 q(0, 0, 0);
            q.x = p_exp(rng, 8.0f); q.y = p_exp(rng, 8.0f); q.z = p_exp(rng, 8.0f);
            V3 ww = face_forward(q - isect.p, isect.n);
            Ray r_out2 = spawn_ray_to_point(isect, isect.p + ww);
            CHECK(!sp.intersect_p(r_out2), "sphere spawn_ray_to intersect_p j=%d", j);
            CHECK(!sph_hits(sp, r_out2), "sphere spawn_ray_to intersect j=%d", j);
        }
    }
    return true;
}
static bool full_sphere_reintersect() {
    int n_found = 0;
    for (uint64_t i = 0; i < 100; i++) {
        RNG rng(i);
        Mat4 id = Mat4::identity();
        Float radius = p_exp(rng, 4.0f);
        Sphere sp;
        sp.init(id.m, id.m, false, radius, -radius, radius, 360.0f);
        if (!sphere_reintersect_convex(sp, rng, &n_found)) { std::printf("  (sphere %llu)\n", (unsigned long long)i); return false; }
    }
    CHECK(n_found > 20, "only %d of 100 rays found their sphere", n_found);
    return true;
}
static bool partial_sphere_reintersect() {
    int n_found = 0;
    for (uint64_t i = 0; i < 100; i++) {
        RNG rng(i);
        Mat4 id = Mat4::identity();
        Float radius = p_exp(rng, 4.0f);
        Float zmin, zmax, phi_max;
        if (rng.uniform_float() < 0.5f) zmin = -radius; else zmin = lerpf(rng.uniform_float(), -radius, radius);
        if (rng.uniform_float() < 0.5f) zmax = radius; else zmax = lerpf(rng.uniform_float(), -radius, radius);
        if (zmin > zmax) std::swap(zmin, zmax);
        if (rng.uniform_float() < 0.5f) phi_max = 360.0f; else phi_max = rng.uniform_float() * 360.0f;
        Sphere sp;
        sp.init(id.m, id.m, false, radius, zmin, zmax, phi_max);
        if (!sphere_reintersect_convex(sp, rng, &n_found)) { std::printf("  (sphere %llu)\n", (unsigned long long)i); return false; }
    }
    CHECK(n_found > 10, "only %d of 100 rays found their sphere", n_found);
    return true;
}
static Float mc_solid_angle(V3 p, const Sphere& sp, int n_samples) {   // tests/shapes.rs:282-293
    int n_hits = 0;
    for (int i = 0; i < n_samples; i++) {
        V2 u(radical_inverse(0, (uint64_t)i), radical_inverse(1, (uint64_t)i));
        Ray ray(p, uniform_sample_sphere(u), kInfinity);
        if (sp.intersect_p(ray)) n_hits++;
    }
    const Float inv4pi = kInvPi * 0.25f;
    return (Float)n_hits / (inv4pi * (Float)n_samples);
}
static Float shape_solid_angle(V3 p, const Sphere& sp, int n_samples) {   // Shape::solid_angle (shape.rs:56-73)
    Float solid_angle = 0.0f;
    for (int i = 0; i < n_samples; i++) {
        V2 u(radical_inverse(0, (uint64_t)i), radical_inverse(1, (uint64_t)i));
        V3 ps, ns, pe; Float pdf;
        if (sp.sample_from(p, V3(0, 0, 0), V3(0, 0, 0), u, &ps, &ns, &pe, &pdf)) {
            Ray r(p, ps - p, 0.999f);
            if (!sp.intersect_p(r)) solid_angle += 1.0f / pdf;
        }
    }
    return solid_angle / (Float)n_samples;
}
static bool sphere_solid_angle() {
    // Transform::translate(1, .5, -.8) * Transform::rotate_x(30) and its inverse (transform.rs:33-50, :276-281)
    Float sn = std::sin(radians(30.0f)), cs = std::cos(radians(30.0f));
    Mat4 rot = Mat4::identity();
    rot.m[5] = cs; rot.m[6] = -sn; rot.m[9] = sn; rot.m[10] = cs;
    Mat4 rot_inv = Mat4::identity();
    rot_inv.m[5] = cs; rot_inv.m[6] = sn; rot_inv.m[9] = -sn; rot_inv.m[10] = cs;
    Mat4 m = Mat4::translate(1.0f, 0.5f, -0.8f) * rot;
    Mat4 minv = rot_inv * Mat4::translate(-1.0f, -0.5f, 0.8f);
    Sphere sp;
    sp.init(m.m, minv.m, false, 1.0f, -1.0f, 1.0f, 360.0f);
    const int n_samples = 128 * 1024;
    V3 p_inside(1.0f, 0.9f, -0.8f);
    Float sa_mc = mc_solid_angle(p_inside, sp, n_samples);
    CHECK(std::fabs(sa_mc - 4.0f * kPi) < 0.01f, "solid_angle_mc %g", sa_mc);
    Float sa = shape_solid_angle(p_inside, sp, n_samples);
    CHECK(std::fabs(sa - 4.0f * kPi) < 0.01f, "solid_angle %g", sa);
    V3 p_outside(-0.25f, -1.0f, 0.8f);
    Float mc_sa = mc_solid_angle(p_outside, sp, n_samples);
    Float sphere_sa = shape_solid_angle(p_outside, sp, n_samples);
    CHECK(std::fabs(mc_sa - sphere_sa) < 0.001f, "mc_sa %g sphere_sa %g", mc_sa, sphere_sa);
    return true;
}

int main(int argc, char** argv) {
    struct T { const char* name; bool (*fn)(); } tests[] = {
        {"triangle_badcases", triangle_badcases},
        {"triangle_watertight", triangle_watertight},
        {"triangle_reintersect", triangle_reintersect},
        {"triangle_solid_angle", triangle_solid_angle},
        {"sphere_solid_angle", sphere_solid_angle},
        {"full_sphere_reintersect", full_sphere_reintersect},
        {"partial_sphere_reintersect", partial_sphere_reintersect},
    };
    for (auto& t : tests) {
        bool run = argc == 1;
        for (int i = 1; i < argc; i++) if (std::string(argv[i]) == t.name) run = true;
        if (!run) continue;
        bool ok = t.fn();
        std::printf("%s %s\n", ok ? "PASS" : "FAIL", t.name);
        std::fflush(stdout);
        if (!ok) g_fail++;
    }
    return g_fail;
}
