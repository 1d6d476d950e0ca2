// pt_bvh.h -- host BVH builder producing the device layout of pt_device.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "pt_device.h"
#include "../../include/pbrtgpu.h"

namespace ptbvh {

struct Result {
    std::vector<PtNode> nodes;          // 4-wide interior nodes, root first
    std::vector<PtTri> tris;            // triangle records in leaf order + one zero pad record
    std::vector<uint32_t> rec_of_prim;  // primitive index -> record index
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
bool build_prims(const Prim* prims, uint32_t n_prims, int split_method, int max_node_prims, Result* out);
bool build(const float* P, const uint32_t* indices, const uint32_t* tri_flags, uint32_t n_tris, const SpherePrim* spheres, uint32_t n_spheres,
           int split_method, int max_node_prims, Result* out);

}  // namespace ptbvh
