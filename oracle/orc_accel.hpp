// ORACLE -- test infrastructure only (see orc_math.hpp).
// orc_accel.hpp: triangle mesh, watertight ray/triangle test, BVH build, 4-wide
// flatten and SSE-semantics traversal.
//   follows src/shapes/triangle.rs, src/core/shape/shape.rs,
//           src/accelerators/bvh/build/{node,sah,middle,equal_counts,types}.rs,
//           src/accelerators/bvh/accel/qbvh/qbvh_x86.rs, src/core/geometry/intersect.rs
#pragma once
#include "orc_math.hpp"
#include "orc_sampling.hpp"
#include <algorithm>
#include <memory>
#include <atomic>
#include <vector>

namespace orc {

struct Ray {
    V3 o, d;
    mutable Float t_max;
    Ray() : t_max(kInfinity) {}
    Ray(V3 o_, V3 d_, Float t) : o(o_), d(d_), t_max(t) {}
};

// triangle.rs:30-35 (GAMMA6's denominator typo is the reference's, quirk Q5)
static const Float kGamma2 = (2.0f * kMachineEpsilon) / (1.0f - (2.0f * kMachineEpsilon));
static const Float kGamma3 = (3.0f * kMachineEpsilon) / (1.0f - (3.0f * kMachineEpsilon));
static const Float kGamma5 = (5.0f * kMachineEpsilon) / (1.0f - (5.0f * kMachineEpsilon));
static const Float kGamma6 = (6.0f * kMachineEpsilon) / (1.0f - (5.0f * kMachineEpsilon));
static const Float kGamma7 = (7.0f * kMachineEpsilon) / (1.0f - (7.0f * kMachineEpsilon));

struct MeshFlags {
    bool two_sided = true, reverse_orientation = false, swaps_handedness = false;
    bool has_n = false, has_s = false, has_uv = false;
};

// What SurfaceInteraction carries for the path (surface_interaction.rs:25-57).
struct SurfHit {
    V3 p, p_error, n, wo;
    V2 uv;
    V3 dpdu, dpdv;
    V3 sh_n, sh_dpdu, sh_dpdv;
    V3 dndu, dndv, sh_dndu, sh_dndv;         // read by bump mapping only (core/material.rs:31-72)
    Float b0 = 0, b1 = 0, b2 = 0;
    int32_t prim = -1;          // index into the scene's primitive list (Geometry::prim_ref)
    uint32_t shape_ref = 0;     // the shape that was hit: triangle index or PRIM_SPHERE | sphere index (inside an instance too)
};
struct BvhStats { uint64_t nodes = 0, tris = 0; };

}  // namespace orc
#include "orc_sphere.hpp"
namespace orc {

// Flattened scene geometry: one record per triangle (the reference's
// Triangle{mesh, v[3]} + TriangleMesh SoA, triangle.rs:10-22, :92-96) plus the spheres.
// prim_ref lists the scene's primitives in creation order (render_options.primitives,
// scene_context.rs:1301-1316): a triangle index, or PRIM_SPHERE | sphere index.
static const uint32_t PRIM_SPHERE = 0x80000000u;
static const uint32_t PRIM_INSTANCE = 0x40000000u;
struct QBVH;
// ObjectInstance -> TransformedPrimitive (core/primitive/transformed_primitive.rs) with a static transform
struct Instance { Mat4 m, minv; uint32_t object = 0; };
struct Geometry {
    std::vector<V3> P, N, S;
    std::vector<V2> UV;
    std::vector<uint32_t> idx;        // 3 per triangle
    std::vector<uint32_t> tri_mesh;   // mesh id per triangle
    std::vector<MeshFlags> mesh;
    std::vector<Sphere> spheres;
    std::vector<Instance> instances;
    std::vector<const QBVH*> object_bvh;     // per object: the accelerator over its primitives (owned by the Scene)
    std::vector<uint32_t> prim_ref;   // empty = triangles only, prim == triangle index
    size_t n_tris() const { return idx.size() / 3; }
    size_t n_prims() const { return prim_ref.empty() ? n_tris() : prim_ref.size(); }
    uint32_t ref(size_t prim) const { return prim_ref.empty() ? (uint32_t)prim : prim_ref[prim]; }
};

struct TriRef {
    const Geometry* g;
    uint32_t tri;
    V3 p0, p1, p2;
    uint32_t i0, i1, i2;
    const MeshFlags* mf;
    TriRef(const Geometry* g_, uint32_t t) : g(g_), tri(t) {
        i0 = g->idx[3 * t]; i1 = g->idx[3 * t + 1]; i2 = g->idx[3 * t + 2];
        p0 = g->P[i0]; p1 = g->P[i1]; p2 = g->P[i2];
        mf = &g->mesh[g->tri_mesh[t]];
    }
    // triangle.rs:115-130
    void get_uvs(V2 uv[3]) const {
        if (mf->has_uv) { uv[0] = g->UV[i0]; uv[1] = g->UV[i1]; uv[2] = g->UV[i2]; }
        else { uv[0] = V2(0.0f, 0.0f); uv[1] = V2(1.0f, 0.0f); uv[2] = V2(1.0f, 1.0f); }
    }
    // triangle.rs:132-186
    bool get_dpdu_dpdv(V2 uv[3], V3* dpdu, V3* dpdv) const {
        get_uvs(uv);
        V2 duv02 = uv[0] - uv[2], duv12 = uv[1] - uv[2];
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = std::fabs(determinant) < 1e-8f;
        if (!degenerate_uv) {
            Float invdet = 1.0f / determinant;
            V3 du = (duv12.y * dp02 - duv02.y * dp12) * invdet;
            V3 dv = (-duv12.x * dp02 + duv02.x * dp12) * invdet;
            if (!(length_squared(cross(du, dv)) <= 0.0f)) {
                *dpdu = du; *dpdv = dv;
                return true;
            }
        }
        V3 ng = cross(p2 - p0, p1 - p0);
        if (length_squared(ng) <= 0.0f) return false;
        coordinate_system(normalize(ng), dpdu, dpdv);
        return true;
    }
    // triangle.rs:579-588
    Float area() const { return 0.5f * length(cross(p1 - p0, p2 - p0)); }
    // triangle.rs:215-224 + union3 :189-200
    Bounds3 world_bound() const {
        Float mn[3] = {p0.x, p0.y, p0.z}, mx[3] = {p0.x, p0.y, p0.z};
        const V3 a[2] = {p1, p2};
        for (int j = 0; j < 2; j++)
            for (int i = 0; i < 3; i++) { mn[i] = fmin_(mn[i], a[j][i]); mx[i] = fmax_(mx[i], a[j][i]); }
        return Bounds3(V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]));
    }

    // Shared front half of intersect / intersect_p (triangle.rs:240-347, :466-571).
    // Returns false on miss; on hit fills t and the barycentrics.
    bool hit_test(const Ray& r, bool want_n, V3* n_out, Float* t_out, Float* b0o, Float* b1o, Float* b2o) const {
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        V3 n = cross(dp02, dp12);
        if (mf->reverse_orientation ^ mf->swaps_handedness) n = n * -1.0f;
        if (!mf->two_sided) {
            if (dot(n, r.d) >= 0.0f) return false;
        }
        if (want_n) *n_out = normalize(n);

        V3 p0t = p0 - r.o, p1t = p1 - r.o, p2t = p2 - r.o;
        static const int TRI[4] = {0, 1, 2, 0};
        int kz = max_dimension(vabs(r.d));
        int kx = TRI[kz + 1];
        int ky = TRI[kx + 1];
        V3 d = permute(r.d, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz);
        p1t = permute(p1t, kx, ky, kz);
        p2t = permute(p2t, kx, ky, kz);
        Float sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0f / d.z;
        p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
        p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
        p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
        Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
            double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (Float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (Float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (Float)(p1typ0tx - p1txp0ty);
        }
        if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
        Float det = e0 + e1 + e2;
        if (det == 0.0f) return false;
        p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
        Float t_max = r.t_max;
        Float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) return false;
        else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det)) return false;
        Float inv_det = 1.0f / det;
        Float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
        Float t = t_scaled * inv_det;
        Float max_zt = max_component(vabs(V3(p0t.z, p1t.z, p2t.z)));
        Float delta_z = kGamma3 * max_zt;
        Float max_xt = max_component(vabs(V3(p0t.x, p1t.x, p2t.x)));
        Float max_yt = max_component(vabs(V3(p0t.y, p1t.y, p2t.y)));
        Float delta_x = kGamma5 * (max_xt + max_zt);
        Float delta_y = kGamma5 * (max_yt + max_zt);
        Float delta_e = 2.0f * (kGamma2 * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
        Float max_e = max_component(vabs(V3(e0, e1, e2)));
        Float delta_t = 3.0f * (kGamma3 * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * std::fabs(inv_det);
        if (t <= delta_t) return false;
        *t_out = t; *b0o = b0; *b1o = b1; *b2o = b2;
        return true;
    }
    bool intersect_p(const Ray& r) const {
        Float t, b0, b1, b2;
        return hit_test(r, false, nullptr, &t, &b0, &b1, &b2);
    }
    // triangle.rs:226-450
    bool intersect(const Ray& r, Float* t_hit, SurfHit* si) const {
        Float t, b0, b1, b2;
        V3 n;
        if (!hit_test(r, true, &n, &t, &b0, &b1, &b2)) return false;
        V2 uv[3];
        V3 dpdu, dpdv;
        if (!get_dpdu_dpdv(uv, &dpdu, &dpdv)) return false;
        Float x_abs_sum = std::fabs(b0 * p0.x) + std::fabs(b1 * p1.x) + std::fabs(b2 * p2.x);
        Float y_abs_sum = std::fabs(b0 * p0.y) + std::fabs(b1 * p1.y) + std::fabs(b2 * p2.y);
        Float z_abs_sum = std::fabs(b0 * p0.z) + std::fabs(b1 * p1.z) + std::fabs(b2 * p2.z);
        si->p_error = kGamma7 * V3(x_abs_sum, y_abs_sum, z_abs_sum);
        si->p = b0 * p0 + b1 * p1 + b2 * p2;
        si->uv = b0 * uv[0] + b1 * uv[1] + b2 * uv[2];
        si->wo = -r.d;
        si->n = n;
        si->dpdu = dpdu; si->dpdv = dpdv;
        si->sh_n = n; si->sh_dpdu = dpdu; si->sh_dpdv = dpdv;
        si->dndu = V3(0, 0, 0); si->dndv = V3(0, 0, 0); si->sh_dndu = V3(0, 0, 0); si->sh_dndv = V3(0, 0, 0);
        si->b0 = b0; si->b1 = b1; si->b2 = b2;
        si->prim = (int32_t)tri;
        if (mf->has_n || mf->has_s) {
            V3 ns = si->n;
            if (mf->has_n) {
                V3 nns = b0 * g->N[i0] + b1 * g->N[i1] + b2 * g->N[i2];
                if (length_squared(nns) > 0.0f) ns = normalize(nns);
            }
            V3 ss = normalize(si->dpdu);
            if (mf->has_s) {
                V3 nns = b0 * g->S[i0] + b1 * g->S[i1] + b2 * g->S[i2];
                if (length_squared(nns) > 0.0f) ss = normalize(nns);
            }
            V3 ts = cross(ns, ss);  // pbrt-r3 order (triangle.rs:397)
            if (length_squared(ts) > 0.0f) {
                ts = normalize(ts);
                ss = normalize(cross(ts, ns));
            } else {
                coordinate_system(ns, &ss, &ts);
            }
            if (mf->has_n) {             // shading dndu / dndv from the vertex normals (triangle.rs:405-437)
                V2 duv02 = uv[0] - uv[2], duv12 = uv[1] - uv[2];
                V3 dn1 = g->N[i0] - g->N[i2], dn2 = g->N[i1] - g->N[i2];
                Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
                if (std::fabs(determinant) < 1e-8f) {
                    V3 dn = cross(g->N[i2] - g->N[i0], g->N[i1] - g->N[i0]);
                    if (length_squared(dn) != 0.0f) coordinate_system(dn, &si->sh_dndu, &si->sh_dndv);
                } else {
                    Float inv_det = 1.0f / determinant;
                    si->sh_dndu = (duv12.y * dn1 - duv02.y * dn2) * inv_det;
                    si->sh_dndv = (-duv12.x * dn1 + duv02.x * dn2) * inv_det;
                }
            }
            if (mf->reverse_orientation) ts = ts * -1.0f;
            // set_shading_geometry(ss, ts, .., orientation_is_authoritative=true)
            // surface_interaction.rs:140-161
            si->sh_n = normalize(cross(ss, ts));
            si->n = face_forward(si->n, si->sh_n);
            si->sh_dpdu = ss;
            si->sh_dpdv = ts;
        }
        *t_hit = t;
        return true;
    }
    // triangle.rs:590-619.  Returns (p, n, p_error), pdf = 1/area.
    void sample(V2 u, V3* p, V3* n, V3* p_error, Float* pdf) const {
        V2 b = uniform_sample_triangle(u);
        *p = b.x * p0 + b.y * p1 + (1.0f - b.x - b.y) * p2;
        V3 nn = normalize(cross(p1 - p0, p2 - p0));
        if (mf->has_n) {
            V3 ns = b.x * g->N[i0] + b.y * g->N[i1] + (1.0f - b.x - b.y) * g->N[i2];
            nn = face_forward(nn, ns);
        } else if (mf->reverse_orientation ^ mf->swaps_handedness) {
            nn = nn * -1.0f;
        }
        V3 p_abs_sum = vabs(b.x * p0) + vabs(b.y * p1) + vabs((1.0f - b.x - b.y) * p2);
        *p_error = kGamma6 * V3(p_abs_sum.x, p_abs_sum.y, p_abs_sum.z);
        *n = nn;
        *pdf = 1.0f / area();
    }
    // triangle.rs:621-651 (with the pbrt-r3 one-sided rejection, quirk Q6)
    bool sample_from(V3 ref_p, V2 u, V3* p, V3* n, V3* p_error, Float* pdf_out) const {
        Float pdf;
        sample(u, p, n, p_error, &pdf);
        V3 wi = *p - ref_p;
        if (length_squared(wi) <= 0.0f) return false;
        wi = normalize(wi);
        if (!mf->two_sided) {
            if (dot(*n, -wi) <= 0.0f) return false;
        }
        pdf = pdf * distance_squared(ref_p, *p) / abs_dot(*n, -wi);
        if (pdf <= 0.0f || std::isinf(pdf)) return false;
        *pdf_out = pdf;
        return true;
    }
    // core/shape/shape.rs:40-54; ray = Interaction::spawn_ray(wi) of the reference point
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

// Primitive dispatch (the reference's Arc<dyn Primitive> -> GeometricPrimitive -> dyn Shape chain).
bool instance_intersect(const Geometry* g, uint32_t inst, const Ray& r, Float* t, SurfHit* si, BvhStats* st);
bool instance_intersect_p(const Geometry* g, uint32_t inst, const Ray& r, BvhStats* st);
Bounds3 instance_world_bound(const Geometry* g, uint32_t inst);
inline bool ref_intersect(const Geometry* g, uint32_t ref, const Ray& r, Float* t, SurfHit* si, BvhStats* st) {
    if (ref & PRIM_INSTANCE) return instance_intersect(g, ref & ~PRIM_INSTANCE, r, t, si, st);
    bool hit = (ref & PRIM_SPHERE) ? g->spheres[ref & ~PRIM_SPHERE].intersect(r, t, si) : TriRef(g, ref).intersect(r, t, si);
    if (hit) si->shape_ref = ref;
    return hit;
}
inline bool ref_intersect_p(const Geometry* g, uint32_t ref, const Ray& r, BvhStats* st) {
    if (ref & PRIM_INSTANCE) return instance_intersect_p(g, ref & ~PRIM_INSTANCE, r, st);
    return (ref & PRIM_SPHERE) ? g->spheres[ref & ~PRIM_SPHERE].intersect_p(r) : TriRef(g, ref).intersect_p(r);
}
inline Bounds3 ref_world_bound(const Geometry* g, uint32_t ref) {
    if (ref & PRIM_INSTANCE) return instance_world_bound(g, ref & ~PRIM_INSTANCE);
    return (ref & PRIM_SPHERE) ? g->spheres[ref & ~PRIM_SPHERE].world_bound() : TriRef(g, ref).world_bound();
}

// ---------------------------------------------------------------- BVH build
enum SplitMethod { SPLIT_SAH = 0, SPLIT_HLBVH = 1, SPLIT_MIDDLE = 2, SPLIT_EQUAL = 3 };

struct PrimInfo { size_t prim; Bounds3 bounds; V3 centroid; };
struct BuildNode {
    Bounds3 bounds;
    std::unique_ptr<BuildNode> c[2];
    uint8_t split_axis = 0;
    size_t first = 0, n_prims = 0;
};

struct BVHBuilder {
    size_t max_prims;
    SplitMethod method;
    std::vector<size_t> ordered;
    size_t n_interior = 0, n_leaf = 0;

    std::unique_ptr<BuildNode> leaf(const PrimInfo* pi, size_t n, const Bounds3& b) {
        std::unique_ptr<BuildNode> nd(new BuildNode);
        nd->first = ordered.size();
        nd->n_prims = n;
        nd->bounds = b;
        for (size_t i = 0; i < n; i++) ordered.push_back(pi[i].prim);
        n_leaf++;
        return nd;
    }
    std::unique_ptr<BuildNode> interior(int axis, std::unique_ptr<BuildNode> c0, std::unique_ptr<BuildNode> c1) {
        std::unique_ptr<BuildNode> nd(new BuildNode);
        nd->bounds = bunion(c0->bounds, c1->bounds);
        nd->split_axis = (uint8_t)axis;
        nd->c[0] = std::move(c0);
        nd->c[1] = std::move(c1);
        n_interior++;
        return nd;
    }
    static void sort_by_centroid(PrimInfo* pi, size_t n, int dim) {
        std::stable_sort(pi, pi + n, [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
    }
    // build/node.rs:27-105
    std::unique_ptr<BuildNode> recursive_build(PrimInfo* pi, size_t n) {
        Bounds3 bounds = pi[0].bounds;
        for (size_t i = 1; i < n; i++) bounds = bunion(bounds, pi[i].bounds);
        if (n <= max_prims) return leaf(pi, n, bounds);
        Bounds3 cb(pi[0].centroid, pi[0].centroid);
        for (size_t i = 1; i < n; i++) cb = bunion_p(cb, pi[i].centroid);
        int dim = cb.maximum_extent();
        if (cb.min[dim] == cb.max[dim]) return leaf(pi, n, bounds);
        switch (method) {
            case SPLIT_MIDDLE: return split_middle(dim, (cb.min[dim] + cb.max[dim]) / 2.0f, pi, n);
            case SPLIT_SAH: return split_sah(dim, pi, n);
            default: return split_equal_counts(dim, pi, n);
        }
    }
    // build/equal_counts.rs:6-30
    std::unique_ptr<BuildNode> split_equal_counts(int dim, PrimInfo* pi, size_t n) {
        sort_by_centroid(pi, n, dim);
        size_t mid = n / 2;
        auto c0 = recursive_build(pi, mid);
        auto c1 = recursive_build(pi + mid, n - mid);
        return interior(dim, std::move(c0), std::move(c1));
    }
    // build/middle.rs:8-40
    std::unique_ptr<BuildNode> split_middle(int dim, Float p_mid, PrimInfo* pi, size_t n) {
        sort_by_centroid(pi, n, dim);
        size_t mid = 0;
        while (mid < n && !(p_mid <= pi[mid].centroid[dim])) mid++;
        if (mid == 0 || mid == n) return split_equal_counts(dim, pi, n);
        auto c0 = recursive_build(pi, mid);
        auto c1 = recursive_build(pi + mid, n - mid);
        return interior(dim, std::move(c0), std::move(c1));
    }
    static int bucket_of(const Bounds3& cb, V3 c, int dim) {
        const int NB = 12;
        int b = (int)std::floor((Float)NB * cb.offset(c)[dim]);
        if (b > NB - 1) b = NB - 1;
        if (b < 0) b = 0;
        return b;
    }
    // build/sah.rs:31-170
    std::unique_ptr<BuildNode> split_sah(int dim, PrimInfo* pi, size_t n) {
        if (n == 1) return leaf(pi, n, pi[0].bounds);
        if (n == 2) {
            sort_by_centroid(pi, n, dim);
            auto c0 = recursive_build(pi, 1);
            auto c1 = recursive_build(pi + 1, 1);
            return interior(dim, std::move(c0), std::move(c1));
        }
        Bounds3 bounds = pi[0].bounds;
        for (size_t i = 0; i < n; i++) bounds = bunion(bounds, pi[i].bounds);
        Bounds3 cb(pi[0].centroid, pi[0].centroid);
        for (size_t i = 0; i < n; i++) cb = bunion_p(cb, pi[i].centroid);
        const int NB = 12;
        int count[NB];
        Bounds3 bb[NB];
        for (int i = 0; i < NB; i++) count[i] = 0;
        for (size_t i = 0; i < n; i++) {
            int b = bucket_of(cb, pi[i].centroid, dim);
            count[b] += 1;
            bb[b] = bunion(bb[b], pi[i].bounds);
        }
        Float cost[NB - 1];
        for (int i = 0; i < NB - 1; i++) {
            int count0 = 0, count1 = 0;
            Bounds3 b0 = bb[i], b1 = bb[i + 1];
            for (int j = 0; j <= i; j++) { b0 = bunion(b0, bb[j]); count0 += count[j]; }
            for (int j = i + 1; j < NB; j++) { b1 = bunion(b1, bb[j]); count1 += count[j]; }
            cost[i] = 1.0f + ((Float)count0 * b0.surface_area() + (Float)count1 * b1.surface_area()) / bounds.surface_area();
        }
        Float min_cost = cost[0];
        int min_bucket = 0;
        for (int i = 1; i < NB - 1; i++)
            if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = i; }
        Float leaf_cost = (Float)n;
        if (n > max_prims || min_cost < leaf_cost) {
            std::vector<PrimInfo> left, right;
            for (size_t i = 0; i < n; i++) {
                if (bucket_of(cb, pi[i].centroid, dim) <= min_bucket) left.push_back(pi[i]);
                else right.push_back(pi[i]);
            }
            if (left.empty() || right.empty()) return split_equal_counts(dim, pi, n);
            auto c0 = recursive_build(left.data(), left.size());
            auto c1 = recursive_build(right.data(), right.size());
            return interior(dim, std::move(c0), std::move(c1));
        }
        return leaf(pi, n, bounds);
    }

    // ---- HLBVH (build/hlbvh.rs) -------------------------------------------------------------
    struct MortonPrim { uint32_t prim, code; };
    bool failed = false;          // the reference panics (index out of bounds) on these inputs
    static uint32_t left_shift3(uint32_t x) {          // hlbvh.rs:23-40 (input 1024 is clamped to 1023)
        if (x >= 1024u) x = 1023u;
        x = (x | (x << 16)) & 0x30000ffu;
        x = (x | (x << 8)) & 0x300f00fu;
        x = (x | (x << 4)) & 0x30c30c3u;
        x = (x | (x << 2)) & 0x9249249u;
        return x;
    }
    static uint32_t f2u_sat(Float f) {                 // Rust `as u32`: saturating, NaN -> 0
        if (!(f > 0.0f)) return 0;
        if (f >= 4294967296.0f) return 0xffffffffu;
        return (uint32_t)f;
    }
    static uint32_t encode_morton3(const Float v[3]) { // hlbvh.rs:42-47: note the ceil
        uint32_t x = f2u_sat(std::ceil(v[0])), y = f2u_sat(std::ceil(v[1])), z = f2u_sat(std::ceil(v[2]));
        return (left_shift3(z) << 2) | (left_shift3(y) << 1) | left_shift3(x);
    }
    static Float clamp01(Float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }   // f32::clamp keeps NaN
    std::unique_ptr<BuildNode> morton_leaf(const MortonPrim* mp, size_t n, const PrimInfo* info) {
        std::unique_ptr<BuildNode> nd(new BuildNode);
        nd->first = ordered.size();
        nd->n_prims = n;
        Bounds3 b = info[mp[0].prim].bounds;
        ordered.push_back(mp[0].prim);
        for (size_t i = 1; i < n; i++) {
            ordered.push_back(mp[i].prim);
            b = bunion(b, info[mp[i].prim].bounds);
        }
        nd->bounds = b;
        n_leaf++;
        return nd;
    }
    // hlbvh.rs:102-157: median split on centroids once the Morton bits are exhausted
    std::unique_ptr<BuildNode> split_node(MortonPrim* mp, size_t n, const PrimInfo* info, int dim) {
        if (n <= max_prims) return morton_leaf(mp, n, info);
        std::stable_sort(mp, mp + n, [&](const MortonPrim& a, const MortonPrim& b) { return info[a.prim].centroid[dim] < info[b.prim].centroid[dim]; });
        int next_dim = (dim + 3 - 1) % 3;
        size_t split = n / 2;
        auto c0 = split_node(mp, split, info, next_dim);
        auto c1 = split_node(mp + split, n - split, info, next_dim);
        return interior(dim, std::move(c0), std::move(c1));
    }
    // hlbvh.rs:159-246
    std::unique_ptr<BuildNode> emit_lbvh(const MortonPrim* mp, size_t n, const PrimInfo* info, int bit_index) {
        if (n <= max_prims) return morton_leaf(mp, n, info);
        if (bit_index == -1) {
            std::vector<MortonPrim> copy(mp, mp + n);
            return split_node(copy.data(), n, info, 2);
        }
        uint32_t mask = 1u << bit_index;
        if ((mp[0].code & mask) == (mp[n - 1].code & mask)) return emit_lbvh(mp, n, info, bit_index - 1);
        size_t search_start = 0, search_end = n - 1;
        while (search_start + 1 != search_end) {
            size_t mid = (search_start + search_end) / 2;
            if ((mp[search_start].code & mask) == (mp[mid].code & mask)) search_start = mid;
            else search_end = mid;
        }
        size_t split = search_end;
        auto c0 = emit_lbvh(mp, split, info, bit_index - 1);
        auto c1 = emit_lbvh(mp + split, n - split, info, bit_index - 1);
        return interior(bit_index % 3, std::move(c0), std::move(c1));
    }
    static size_t upper_bucket(const Bounds3& b, const Bounds3& cb, int dim) {
        Float c = (b.min[dim] + b.max[dim]) * 0.5f;
        Float f = 12.0f * ((c - cb.min[dim]) / (cb.max[dim] - cb.min[dim]));
        size_t bk = !(f > 0.0f) ? 0 : (f >= 1.8446744e19f ? ~(size_t)0 : (size_t)f);     // `as usize`
        return bk < 11 ? bk : 11;
    }
    // hlbvh.rs:254-352.  The cost loop is the reference's own: b0 starts from bucket i but count0 leaves
    // bucket i out, count1 starts at 1, and the constant is 0.125.
    std::unique_ptr<BuildNode> build_upper_sah(std::vector<std::unique_ptr<BuildNode>>& roots) {
        if (roots.empty()) { failed = true; return nullptr; }
        if (roots.size() == 1) return std::move(roots[0]);
        Bounds3 bounds = roots[0]->bounds;
        for (size_t i = 1; i < roots.size(); i++) bounds = bunion(bounds, roots[i]->bounds);
        V3 center = (roots[0]->bounds.min + roots[0]->bounds.max) * 0.5f;
        Bounds3 cb(center, center);
        for (size_t i = 1; i < roots.size(); i++) cb = bunion_p(cb, (roots[i]->bounds.min + roots[i]->bounds.max) * 0.5f);
        int dim = cb.maximum_extent();
        const int NB = 12;
        uint32_t count[NB];
        Bounds3 bb[NB];
        for (int i = 0; i < NB; i++) count[i] = 0;
        for (size_t i = 0; i < roots.size(); i++) {
            size_t b = upper_bucket(roots[i]->bounds, cb, dim);
            count[b] += 1;
            bb[b] = count[b] == 1 ? roots[i]->bounds : bunion(bb[b], roots[i]->bounds);
        }
        Float cost[NB - 1];
        for (int i = 0; i < NB - 1; i++) {
            Bounds3 b0 = bb[i], b1 = bb[i];
            uint32_t count0 = 0, count1 = 1;
            for (int j = 0; j < i; j++) { b0 = bunion(b0, bb[j]); count0 += count[j]; }
            for (int j = i + 1; j < NB; j++) { b1 = bunion(b1, bb[j]); count1 += count[j]; }
            cost[i] = 0.125f + (((Float)count0 * b0.surface_area() + (Float)count1 * b1.surface_area()) / bounds.surface_area());
        }
        Float min_cost = cost[0];
        size_t min_bucket = 0;
        for (int i = 1; i < NB - 1; i++)
            if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = (size_t)i; }
        std::vector<std::unique_ptr<BuildNode>> r0, r1;
        for (auto& r : roots) {
            if (upper_bucket(r->bounds, cb, dim) <= min_bucket) r0.push_back(std::move(r));
            else r1.push_back(std::move(r));
        }
        auto c0 = build_upper_sah(r0);
        auto c1 = build_upper_sah(r1);
        if (!c0 || !c1) { failed = true; return nullptr; }
        return interior(dim, std::move(c0), std::move(c1));
    }
    // hlbvh.rs:354-428
    std::unique_ptr<BuildNode> hlbvh_build(const PrimInfo* info, size_t n) {
        Bounds3 bounds = info[0].bounds;
        for (size_t i = 1; i < n; i++) bounds = bunion(bounds, info[i].bounds);
        std::vector<MortonPrim> mp(n), tmp(n);
        for (size_t i = 0; i < n; i++) {
            mp[i].prim = (uint32_t)info[i].prim;
            V3 o = bounds.offset(info[i].centroid);
            Float sc[3] = {clamp01(o.x) * 1024.0f, clamp01(o.y) * 1024.0f, clamp01(o.z) * 1024.0f};
            mp[i].code = encode_morton3(sc);
        }
        // radix_sort (hlbvh.rs:49-100) is a stable LSD sort on the 30-bit code
        std::stable_sort(mp.begin(), mp.end(), [](const MortonPrim& a, const MortonPrim& b) { return a.code < b.code; });
        std::vector<std::unique_ptr<BuildNode>> treelets;
        const uint32_t MASK = 0x3ffc0000u;
        size_t start = 0;
        for (size_t end = 1; end <= n; end++) {
            if (end == n || (mp[start].code & MASK) != (mp[end].code & MASK)) {
                treelets.push_back(emit_lbvh(&mp[start], end - start, info, 29 - 12));
                start = end;
            }
        }
        return build_upper_sah(treelets);
    }
};

// ------------------------------------------------- 4-wide node + traversal
static const size_t kEmpty = ~(size_t)0;
struct QNode {
    Float bb[2][3][4];      // [min/max][xyz][child]  (qbvh_x86.rs:15-24)
    size_t children[4];
    uint8_t axis_top = 0, axis_left = 0, axis_right = 0, is_leaf = 0;
    QNode() { std::memset(bb, 0, sizeof(bb)); children[0] = children[1] = children[2] = children[3] = 0; }
};

// ORDER_TABLE in closed form (qbvh_x86.rs:186-204, decoded in SURVEY.md section 2):
// nibbles are pushed low-to-high, so the LAST pushed child is visited first.
inline uint32_t order_entry(uint32_t hit_mask, uint32_t node_idx) {
    bool s_top = (node_idx & 4) != 0, s_left = (node_idx & 2) != 0, s_right = (node_idx & 1) != 0;
    int visit[4];
    int l0 = s_left ? 1 : 0, l1 = s_left ? 0 : 1, r0 = s_right ? 3 : 2, r1 = s_right ? 2 : 3;
    if (!s_top) { visit[0] = l0; visit[1] = l1; visit[2] = r0; visit[3] = r1; }
    else { visit[0] = r0; visit[1] = r1; visit[2] = l0; visit[3] = l1; }
    uint32_t order = 0x44444;
    int shift = 0;
    // push order = reverse visit order
    uint32_t nib[4];
    int cnt = 0;
    for (int k = 3; k >= 0; k--)
        if (hit_mask & (1u << visit[k])) nib[cnt++] = (uint32_t)visit[k];
    for (int k = 0; k < cnt; k++) { order = (order & ~(0xfu << shift)) | (nib[k] << shift); shift += 4; }
    return order;
}

// diagnostic (tools/depth_hist.py): node visits by 4-wide tree depth, summed over every traversal while enabled
static std::atomic<uint64_t> g_depth_hist[32];
static std::atomic<bool> g_depth_hist_on{false};

struct QBVH {
    const Geometry* geom = nullptr;
    mutable std::vector<uint8_t> depth_of_node;      // filled by compute_depths() for the diagnostic above
    void compute_depths() const {
        depth_of_node.assign(nodes.size(), 0);
        std::vector<size_t> cur{0}, nxt;
        for (uint8_t d = 0; !cur.empty(); d = d < 31 ? d + 1 : d) {
            nxt.clear();
            for (size_t n : cur) {
                depth_of_node[n] = d;
                if (nodes[n].is_leaf) continue;
                for (int k = 0; k < 4; k++) if (nodes[n].children[k] != kEmpty && !nodes[nodes[n].children[k]].is_leaf) nxt.push_back(nodes[n].children[k]);
            }
            cur.swap(nxt);
        }
    }
    std::vector<size_t> prims;     // ordered primitive numbers
    std::vector<QNode> nodes;
    Bounds3 bounds;
    size_t n_interior4 = 0, n_leaves = 0;
    // tools/sim_wave_sched.py: byte stream of traversal steps (0 = node visit, k = leaf with k triangles, 255 = ray end);
    // single-threaded diagnostic use only
    mutable std::vector<uint8_t>* step_log = nullptr;
    // traversal statistics (thread-unsafe; callers keep one copy per thread or ignore)
    typedef BvhStats Stats;
    std::vector<uint32_t> refs;    // an object's own primitive list (ObjectBegin .. ObjectEnd); empty: the scene's list (geom->ref)
    bool own_refs = false;
    uint32_t ref_of(size_t prim) const { return own_refs ? refs[prim] : geom->ref(prim); }
    size_t n_prims() const { return own_refs ? refs.size() : geom->n_prims(); }

    size_t flatten(const BuildNode* node) {  // qbvh_x86.rs:93-176
        size_t offset = nodes.size();
        nodes.push_back(QNode());
        if (node->n_prims > 0) {
            nodes[offset].is_leaf = 1;
            nodes[offset].children[0] = node->first;
            nodes[offset].children[1] = node->n_prims;
            n_leaves++;
        } else {
            n_interior4++;
            size_t indices[4] = {0, 0, 0, 0};
            V3 boxes[4][2];
            const BuildNode* c0 = node->c[0].get();
            const BuildNode* c1 = node->c[1].get();
            if (c0->n_prims > 0) {
                indices[0] = flatten(c0);
                indices[1] = kEmpty;
                boxes[0][0] = c0->bounds.min; boxes[0][1] = c0->bounds.max;
            } else {
                indices[0] = flatten(c0->c[0].get());
                indices[1] = flatten(c0->c[1].get());
                boxes[0][0] = c0->c[0]->bounds.min; boxes[0][1] = c0->c[0]->bounds.max;
                boxes[1][0] = c0->c[1]->bounds.min; boxes[1][1] = c0->c[1]->bounds.max;
            }
            if (c1->n_prims > 0) {
                indices[2] = flatten(c1);
                indices[3] = kEmpty;
                boxes[2][0] = c1->bounds.min; boxes[2][1] = c1->bounds.max;
            } else {
                indices[2] = flatten(c1->c[0].get());
                indices[3] = flatten(c1->c[1].get());
                boxes[2][0] = c1->c[0]->bounds.min; boxes[2][1] = c1->c[0]->bounds.max;
                boxes[3][0] = c1->c[1]->bounds.min; boxes[3][1] = c1->c[1]->bounds.max;
            }
            QNode& nd = nodes[offset];
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 4; k++) { nd.bb[0][j][k] = boxes[k][0][j]; nd.bb[1][j][k] = boxes[k][1][j]; }
            for (int k = 0; k < 4; k++) nd.children[k] = indices[k];
            nd.axis_top = node->split_axis;
            nd.axis_left = c0->split_axis;
            nd.axis_right = c1->split_axis;
        }
        return offset;
    }

    // QBVHAccel::new (qbvh_x86.rs:352-370) + create_bvh_node (build/node.rs:107-151)
    bool build(const Geometry* g, size_t max_prims_in_node, SplitMethod method) {
        geom = g;
        size_t n = n_prims();
        const Float eps = std::numeric_limits<Float>::epsilon() * 2.0f;  // BOUND_EPS node.rs:13
        std::vector<PrimInfo> info(n);
        for (size_t i = 0; i < n; i++) {
            Bounds3 b = ref_world_bound(g, ref_of(i));
            V3 mn = b.min, mx = b.max;
            mn.x -= eps; mn.y -= eps; mn.z -= eps;
            mx.x += eps; mx.y += eps; mx.z += eps;
            b = Bounds3(mn, mx);
            info[i].prim = i;
            info[i].bounds = b;
            info[i].centroid = (b.min + b.max) * 0.5f;
        }
        BVHBuilder bld;
        bld.max_prims = max_prims_in_node < 255 ? max_prims_in_node : 255;
        bld.method = method;
        std::unique_ptr<BuildNode> root = method == SPLIT_HLBVH ? bld.hlbvh_build(info.data(), n) : bld.recursive_build(info.data(), n);
        if (!root || bld.failed) return false;
        prims.swap(bld.ordered);
        nodes.clear();
        flatten(root.get());
        bounds = root->bounds;
        // iterative destruction is unnecessary: depth is O(log n) for these builders
        return true;
    }

    static int get_sign(Float x) { return std::signbit(x) ? 1 : 0; }

    // Bounds3f::intersect_p (bounds3.rs:154-163 -> intersect.rs:16-65): scalar, NaN-ignoring max/min
    bool root_test(const Ray& r, Float* tmin, Float* tmax) const {
        Float idir[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
        int sign[3] = {get_sign(r.d.x), get_sign(r.d.y), get_sign(r.d.z)};
        const V3* bnd[2] = {&bounds.min, &bounds.max};
        Float t0 = 0.0f, t1 = r.t_max;
        for (int i = 0; i < 3; i++) {
            t0 = fmax_(t0, ((*bnd[sign[i]])[i] - r.o[i]) * idir[i]);
            t1 = fmin_(t1, ((*bnd[1 - sign[i]])[i] - r.o[i]) * idir[i]);
        }
        if (t0 <= t1) { *tmin = t0; *tmax = t1; return true; }
        return false;
    }
    // _mm_max_ps / _mm_min_ps return the SECOND operand when either is NaN (quirk Q15)
    static Float sse_max(Float a, Float b) { return a > b ? a : b; }
    static Float sse_min(Float a, Float b) { return a < b ? a : b; }
    // test_aabb (qbvh_x86.rs:26-67)
    static uint32_t test_aabb(const QNode& nd, const Float org[3], const Float idir[3], const int sign[3], Float tmin0, Float tmax0) {
        uint32_t mask = 0;
        for (int k = 0; k < 4; k++) {
            Float tmin = tmin0, tmax = tmax0;
            for (int a = 0; a < 3; a++) {
                tmin = sse_max(tmin, (nd.bb[sign[a]][a][k] - org[a]) * idir[a]);
                tmax = sse_min(tmax, (nd.bb[1 - sign[a]][a][k] - org[a]) * idir[a]);
            }
            if (tmax >= tmin) mask |= 1u << k;
        }
        return mask;
    }

    // intersect_simd (qbvh_x86.rs:230-287) behind QBVHAccel::intersect (:378-387)
    bool intersect(const Ray& r, SurfHit* out, Stats* st = nullptr) const {
        Float tmin, tmax;
        if (!root_test(r, &tmin, &tmax)) return false;
        Float org[3] = {r.o.x, r.o.y, r.o.z};
        Float idir[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
        int sign[3] = {get_sign(r.d.x), get_sign(r.d.y), get_sign(r.d.z)};
        bool hit = false;
        std::vector<size_t> stack;
        stack.reserve(64);
        stack.push_back(0);
        while (!stack.empty()) {
            size_t cur = stack.back();
            stack.pop_back();
            const QNode& nd = nodes[cur];
            if (nd.is_leaf == 0) {
                if (st) st->nodes++;
                if (g_depth_hist_on.load(std::memory_order_relaxed) && !depth_of_node.empty()) g_depth_hist[depth_of_node[cur]].fetch_add(1, std::memory_order_relaxed);
                if (step_log) step_log->push_back(0);
                uint32_t hit_mask = test_aabb(nd, org, idir, sign, tmin, tmax);
                if (hit_mask != 0) {
                    uint32_t node_idx = (uint32_t)((sign[nd.axis_top] << 2) | (sign[nd.axis_left] << 1) | sign[nd.axis_right]);
                    uint32_t order = order_entry(hit_mask, node_idx);
                    while ((order & 0x4) == 0) {
                        size_t cidx = nd.children[order & 0x3];
                        if (cidx != kEmpty) stack.push_back(cidx);
                        order >>= 4;
                    }
                }
            } else {
                size_t start = nd.children[0], end = start + nd.children[1];
                if (step_log) step_log->push_back((uint8_t)std::min<size_t>(end - start, 250));
                bool leaf_hit = false;
                for (size_t i = start; i < end; i++) {   // intersect_primitives :206-218
                    if (st) st->tris++;
                    Float t;
                    SurfHit si;
                    if (ref_intersect(geom, ref_of(prims[i]), r, &t, &si, st)) {
                        si.prim = (int32_t)prims[i];
                        r.t_max = t;                      // GeometricPrimitive::intersect :41-58
                        *out = si;
                        leaf_hit = true;
                    }
                }
                if (leaf_hit) { tmax = r.t_max; hit = true; }
            }
        }
        if (step_log) step_log->push_back(255);
        return hit;
    }
    // intersect_simd_p (qbvh_x86.rs:289-343)
    bool intersect_p(const Ray& r, Stats* st = nullptr) const {
        Float tmin, tmax;
        if (!root_test(r, &tmin, &tmax)) return false;
        Float org[3] = {r.o.x, r.o.y, r.o.z};
        Float idir[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
        int sign[3] = {get_sign(r.d.x), get_sign(r.d.y), get_sign(r.d.z)};
        std::vector<size_t> stack;
        stack.reserve(64);
        stack.push_back(0);
        while (!stack.empty()) {
            size_t cur = stack.back();
            stack.pop_back();
            const QNode& nd = nodes[cur];
            if (nd.is_leaf == 0) {
                if (st) st->nodes++;
                if (g_depth_hist_on.load(std::memory_order_relaxed) && !depth_of_node.empty()) g_depth_hist[depth_of_node[cur]].fetch_add(1, std::memory_order_relaxed);
                if (step_log) step_log->push_back(0);
                uint32_t hit_mask = test_aabb(nd, org, idir, sign, tmin, tmax);
                if (hit_mask != 0) {
                    uint32_t node_idx = (uint32_t)((sign[nd.axis_top] << 2) | (sign[nd.axis_left] << 1) | sign[nd.axis_right]);
                    uint32_t order = order_entry(hit_mask, node_idx);
                    while ((order & 0x4) == 0) {
                        size_t cidx = nd.children[order & 0x3];
                        if (cidx != kEmpty) stack.push_back(cidx);
                        order >>= 4;
                    }
                }
            } else {
                size_t start = nd.children[0], end = start + nd.children[1];
                for (size_t i = start; i < end; i++) {
                    if (st) st->tris++;
                    if (ref_intersect_p(geom, ref_of(prims[i]), r, st)) {
                        if (step_log) { step_log->push_back((uint8_t)std::min<size_t>(i - start + 1, 250)); step_log->push_back(255); }
                        return true;
                    }
                }
                if (step_log) step_log->push_back((uint8_t)std::min<size_t>(end - start, 250));
            }
        }
        if (step_log) step_log->push_back(255);
        return false;
    }
    // accelerators/exhaustive: brute force over every primitive, used as a cross-check
    bool intersect_exhaustive(const Ray& r, SurfHit* out) const {
        bool hit = false;
        for (size_t i = 0; i < n_prims(); i++) {
            Float t;
            SurfHit si;
            if (ref_intersect(geom, ref_of(i), r, &t, &si, nullptr)) { si.prim = (int32_t)i; r.t_max = t; *out = si; hit = true; }
        }
        return hit;
    }
};

// ---- TransformedPrimitive (core/primitive/transformed_primitive.rs:26-66) for a static primitive_to_world
// Transform::transform_ray with the instance's inverse (transform.rs:184-203, :245-282)
inline Ray instance_ray(const Instance& in, const Ray& r) {
    const Float* m = in.minv.m;
    V3 p = r.o, v = r.d;
    V3 op = in.minv.transform_point(p);
    V3 o_err = kGamma3 * V3(std::fabs(m[0] * p.x) + std::fabs(m[1] * p.y) + std::fabs(m[2] * p.z) + std::fabs(m[3]),
                            std::fabs(m[4] * p.x) + std::fabs(m[5] * p.y) + std::fabs(m[6] * p.z) + std::fabs(m[7]),
                            std::fabs(m[8] * p.x) + std::fabs(m[9] * p.y) + std::fabs(m[10] * p.z) + std::fabs(m[11]));
    V3 dd = in.minv.transform_vector(v);
    Float ls = length_squared(dd);
    if (ls > 0.0f) {
        Float dt = dot(vabs(dd), o_err) / ls;
        op += dd * dt;
    }
    return Ray(op, dd, r.t_max);
}
// Transform::transform_surface_interaction (transform.rs:299-323)
inline void transform_surf(const Instance& in, SurfHit* si) {
    const Float* m = in.m.m;
    const Float* mi = in.minv.m;
    auto nrm = [&](V3 n) { return V3(mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z); };
    V3 p = si->p, pe = si->p_error;
    si->p = in.m.transform_point(p);
    V3 e;
    for (int i = 0; i < 3; i++) {
        const Float* row = m + 4 * i;
        Float v1 = (kGamma3 + 1.0f) * (std::fabs(row[0]) * pe.x + std::fabs(row[1]) * pe.y + std::fabs(row[2]) * pe.z) +
                   kGamma3 * (std::fabs(row[0] * p.x) + std::fabs(row[1] * p.y) + std::fabs(row[2] * p.z) + std::fabs(row[3]));
        if (i == 0) e.x = v1; else if (i == 1) e.y = v1; else e.z = v1;
    }
    si->p_error = e;
    si->n = normalize(nrm(si->n));
    si->wo = normalize(in.m.transform_vector(si->wo));
    si->dpdu = in.m.transform_vector(si->dpdu);
    si->dpdv = in.m.transform_vector(si->dpdv);
    si->dndu = nrm(si->dndu);
    si->dndv = nrm(si->dndv);
    si->sh_n = normalize(nrm(si->sh_n));
    si->sh_dpdu = in.m.transform_vector(si->sh_dpdu);
    si->sh_dpdv = in.m.transform_vector(si->sh_dpdv);
    si->sh_dndu = nrm(si->sh_dndu);
    si->sh_dndv = nrm(si->sh_dndv);
    si->sh_n = face_forward(si->sh_n, si->n);
}
inline bool instance_intersect(const Geometry* g, uint32_t inst, const Ray& r, Float* t, SurfHit* si, BvhStats* st) {
    const Instance& in = g->instances[inst];
    Ray ray = instance_ray(in, r);
    const QBVH* b = g->object_bvh[in.object];
    bool hit;
    if (b->n_prims() == 1) {            // a single primitive is wrapped without an accelerator (scene_context.rs:1370-1377)
        Float tt;
        hit = ref_intersect(g, b->ref_of(0), ray, &tt, si, st);
        if (hit) { si->prim = 0; ray.t_max = tt; }
    } else {
        hit = b->intersect(ray, si, st);
    }
    if (!hit) return false;
    *t = ray.t_max;                       // r.t_max.set(ray.t_max.get())
    transform_surf(in, si);
    return true;
}
inline bool instance_intersect_p(const Geometry* g, uint32_t inst, const Ray& r, BvhStats* st) {
    const Instance& in = g->instances[inst];
    Ray ray = instance_ray(in, r);
    const QBVH* b = g->object_bvh[in.object];
    if (b->n_prims() == 1) return ref_intersect_p(g, b->ref_of(0), ray, st);
    return b->intersect_p(ray, st);
}
inline Bounds3 instance_world_bound(const Geometry* g, uint32_t inst) {          // motion_bounds, not animated: transform_bounds
    const Instance& in = g->instances[inst];
    const QBVH* ob = g->object_bvh[in.object];
    Bounds3 b = ob->n_prims() == 1 ? ref_world_bound(g, ob->ref_of(0)) : ob->bounds;
    V3 mn, mx;
    for (int i = 0; i < 8; i++) {
        V3 c((i & 4) ? b.max.x : b.min.x, (i & 2) ? b.max.y : b.min.y, (i & 1) ? b.max.z : b.min.z);
        V3 q = in.m.transform_point(c);
        if (i == 0) { mn = q; mx = q; }
        else {
            mn = V3(fmin_(mn.x, q.x), fmin_(mn.y, q.y), fmin_(mn.z, q.z));
            mx = V3(fmax_(mx.x, q.x), fmax_(mx.y, q.y), fmax_(mx.z, q.z));
        }
    }
    return Bounds3(mn, mx);
}

}  // namespace orc
