// pt_bvh.h -- host BVH builder producing the device layout of pt_device.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <memory>
#include <thread>
#include <utility>
#include <vector>
#include "pt_device.h"
#include "../../include/pbrtgpu.h"

namespace ptbvh {

// fn(begin, end) over [0, n) on up to 16 host threads; ranges below 64K items run inline.  Only for loops whose iterations are independent.
template <class F> inline void parallel_for(size_t n, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? (hw < 16 ? hw : 16) : 4;
    if (n < 65536 || nt < 2) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t step = (n + nt - 1) / nt;
    for (size_t b = step; b < n; b += step) th.emplace_back(fn, b, b + step < n ? b + step : n);
    fn((size_t)0, step < n ? step : n);
    for (std::thread& t : th) t.join();
}

// std::vector that leaves new elements uninitialised (resize() of a 50 MB array would otherwise zero it on one thread -- and fault every page
// in on that thread -- before the parallel fill or the device copy overwrites it)
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using NoInitVec = std::vector<T, NoInitAlloc<T>>;

// fn(i) for i in [0, n) on up to 16 host threads, indices handed out one at a time (uneven tasks: subtrees)
template <class F> inline void parallel_tasks(size_t n, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw ? (hw < 16 ? hw : 16) : 4;
    if (nt > n) nt = n;
    if (nt < 2) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

struct Result {
    NoInitVec<PtNode> nodes;            // 4-wide interior nodes, root first
    NoInitVec<PtTri> tris;              // triangle records in leaf order + one zero pad record
    NoInitVec<uint32_t> rec_of_prim;    // primitive index -> record index
    uint32_t root_ref = PT_EMPTY_REF;
    uint32_t max_leaf = 0;             // most triangles in one leaf
    uint32_t n_leaves = 0;
    uint32_t max_stack = 1;             // upper bound on traversal stack entries
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
};

// tri_flags[t]: PT_TRI_ONE_SIDED / PT_TRI_FLIP / PT_TRI_HAS_ATTR bits and the material field of triangle t.  Returns false for an
// unsupported split method.
// spheres[]: analytic primitives spliced into the primitive list before triangle `before_triangle` (non-decreasing); lo / hi is
// Sphere::world_bound, flags the record flags (PT_TRI_SPHERE | material field).  Primitive numbering (PtTri::prim, rec_of_prim) is the
// merged list; with no spheres it is the triangle index.
struct SpherePrim { float lo[3], hi[3]; uint32_t flags; uint32_t before_triangle; };
// One primitive of a list: its world bound (before the builder's epsilon inflation) and its leaf record (prim / light1 / the LAST
// bit are filled in by the builder and the caller).
struct Prim { float lo[3], hi[3]; PtTri rec; };
void triangle_prim(const float* P, const uint32_t* indices, uint32_t t, uint32_t flags, Prim* out);
void sphere_record(uint32_t sphere_index, uint32_t flags, PtTri* rec);
void instance_record(uint32_t instance_index, PtTri* rec);
// Binary build node (BVHBuildNode, build/node.rs:20-60): a leaf covers items [first, first + count), an interior node has count 0.
struct LbvhNode {                       // plain data (no initialisers): arrays of it are filled by a device copy or by the threads that own a range
    float lo[3], hi[3];
    int32_t left, right;               // children (indices into the same array) or -1
    uint32_t first, count;
    uint8_t axis;
    static LbvhNode blank() { LbvhNode n; for (int i = 0; i < 3; i++) n.lo[i] = n.hi[i] = 0.0f; n.left = n.right = -1; n.first = n.count = 0; n.axis = 0; return n; }
};

// pt_hlbvh.hip: Morton codes, radix sort, treelets and emit_lbvh on the GPU.  raw_bounds = n x {lo[3], hi[3]} (host memory, not yet
// inflated).  On success (0) order[k] = primitive stored k-th, nodes[0 .. n_treelets) are the treelet roots in Morton order.  Returns 1
// when the host has to build instead (a range needs the centroid-median fallback, non-finite bounds, no memory), -1 on a HIP error.
int device_lbvh(hipStream_t st, const float* raw_bounds, uint32_t n, uint32_t max_prims, NoInitVec<uint32_t>* order, NoInitVec<LbvhNode>* nodes,
                uint32_t* n_treelets, hipError_t* err);
// The same with bounds that are on the device already and results that stay there (pt_sah.hip device_hlbvh_scene): order_dev[n] the
// primitive order, nodes_dev[0 .. *n_nodes) the binary nodes (room for node_cap >= 2 n), the first *n_treelets of them the treelet roots,
// copied to roots_host for the upper SAH.
int device_lbvh_keep(hipStream_t st, const float* raw_dev, uint32_t n, uint32_t max_prims, uint32_t* order_dev, LbvhNode* nodes_dev, uint32_t node_cap, uint32_t* n_nodes,
                     uint32_t* n_treelets, std::vector<LbvhNode>* roots_host, hipError_t* err);
// build_upper_sah (hlbvh.rs:254-352) over the treelet roots tree[0 .. n_roots): appends the joining nodes to `tree` and returns the index of
// the root, -1 where the reference panics (all centroids of a subset coincide along the split axis).
int32_t hlbvh_upper_tree(NoInitVec<LbvhNode>& tree, uint32_t n_roots);
// pt_sah.hip: the SAH binary build on the GPU, level by level.  Same conventions: 0 = order / nodes filled (nodes[0] is the root), 1 = the
// host has to build (a split needs the equal-counts fallback, non-finite bounds, maxnodeprims < 2), -1 = HIP error.
int device_sah(hipStream_t st, const float* raw_bounds, uint32_t n, uint32_t max_prims, NoInitVec<uint32_t>* order, NoInitVec<LbvhNode>* nodes, hipError_t* err);
// pt_sah.hip: a triangle-only world list built, collapsed and finished on the device (SAH).  In: the scene's arrays on the host (they are staged
// and uploaded here) and three per-mesh tables -- the record flag word of the mesh's triangles (PT_TRI_* | material field), its material, its
// PT_MESH_* flags with the attributes the scene does not carry masked out.  Out: device blocks the caller owns -- the finished PtNode array
// (breadth-first top, order tables, axis bits, inverted empty slots: what pt_scene_upload uploads on the host path, byte for byte), the
// PtTri records in leaf order + the pad record, the PtTriInfo records, the primitive -> record map (for the light list; free it after).
struct SceneIn {
    const float* P; uint32_t n_vertices;
    const uint32_t* indices; const uint32_t* tri_mesh; uint32_t n_tris;
    const uint32_t* mesh_triflags; const int32_t* mesh_material; const uint32_t* mesh_flags; uint32_t n_meshes;
    // Analytic spheres of the world list (round 4), in list order: sphere j is primitive sph_prim[j] of the merged list (ascending; triangle t is
    // primitive t + the number of spheres listed before it), sph_bounds[6 j ..] its Sphere::world_bound, sph_rec[4 j ..] = {index into the scene's
    // sphere array, record flags (PT_TRI_SPHERE | material field), material, 0}.  n_spheres == 0: a triangle-only list, as before.
    const uint32_t* sph_prim = nullptr; const float* sph_bounds = nullptr; const uint32_t* sph_rec = nullptr; uint32_t n_spheres = 0;
    uint32_t n_prims() const { return n_tris + n_spheres; }
};
struct SceneOut {
    void *d_nodes = nullptr, *d_tris = nullptr, *d_tinfo = nullptr, *d_rec_of_prim = nullptr;
    uint32_t n_nodes4 = 0, n_leaves = 0, max_leaf = 0, max_depth4 = 0, n_top = 0, any_one_sided = 0;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    void free_all();
};
// 0 = done, 1 = this list needs the host path (a fallback split, non-finite bounds), -1 = HIP error
int device_sah_scene(hipStream_t st, const SceneIn& in, uint32_t max_prims, SceneOut* out, hipError_t* err);
int device_hlbvh_scene(hipStream_t st, const SceneIn& in, uint32_t max_prims, SceneOut* out, hipError_t* err);      // the same under "splitmethod" "hlbvh": -2 = the reference panics on this input
int device_scene_lights(hipStream_t st, SceneOut* out, const uint32_t* prims, uint32_t n, uint32_t* recs_host, hipError_t* err);
// Where the lower half of an HLBVH build runs: PT_BVH_BUILD_AUTO picks the device from kDeviceMinPrims primitives up.
struct DeviceBuild { hipStream_t stream; int mode; bool used; hipError_t err; };
const uint32_t kDeviceMinPrims = 1u << 16;
bool build_prims(const Prim* prims, uint32_t n_prims, int split_method, int max_node_prims, Result* out, DeviceBuild* dev = nullptr);
bool build(const float* P, const uint32_t* indices, const uint32_t* tri_flags, uint32_t n_tris, const SpherePrim* spheres, uint32_t n_spheres,
           int split_method, int max_node_prims, Result* out);

}  // namespace ptbvh
