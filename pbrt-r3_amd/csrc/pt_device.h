// pt_device.h -- plain-old-data layouts shared by the host side (pt_context.cpp,
// pt_bvh.cpp) and the gfx950 kernels (pt_kernels.hip).  Everything here lives in
// HBM; layouts are chosen for 16-byte lane loads and 128-byte node fetches.
#pragma once
#include <stdint.h>
#include "../../include/pbrtgpu.h"

#define PT_WAVE 64
#define PT_BLOCK 256
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 32          // per-lane traversal stack entries kept in LDS
#endif
// PT_TRACE_WIDE 1: the pooled-leaf traversal kernels (k_trace, k_trace_sph_dist) run as ONE 1024-thread block per CU instead of four 256-thread
// blocks: same sixteen waves, same registers, but the LDS copy of the top of the tree is shared by all of them -- one copy of 341 nodes
// (levels 0..4) instead of four copies of 85 (levels 0..3).
#ifndef PT_TRACE_WIDE
#define PT_TRACE_WIDE 1
#endif
#define PT_TBLOCK (PT_TRACE_WIDE ? 1024 : PT_BLOCK)      // threads per block of the pooled-leaf traversal kernels
#define PT_SLOT (PT_TBLOCK * 4u)   // bytes between two entries of a lane's LDS traversal stack ([entry][lane] layout; pooled-leaf kernels)
// PT_NODE_STAGED 1: the pooled-leaf traversal kernels fetch a round's nodes cooperatively through LDS (global_load_lds_dwordx4;
// a quarter of the L1 requests, but 49 KB of LDS per block = three blocks per CU and a longer round); 0: every lane loads its
// own node (four blocks per CU).  Measured on RT1M: 0 is the faster one.
#ifndef PT_NODE_STAGED
#define PT_NODE_STAGED 0
#endif
// PT_TOP_NODES > 0: the pooled-leaf traversal kernels keep the first PT_TOP_NODES nodes of the world tree (the upload numbers the top
// of the tree breadth-first, so these are its top levels: 85 = levels 0..3) in LDS, 112 bytes each, and visits to them read LDS instead
// of going through the vector L1, whose request rate bounds the kernel.  Paid for with 8 of the 32 stack slots.
#ifndef PT_TOP_NODES
#define PT_TOP_NODES (PT_TRACE_WIDE ? 341 : 85)
#endif
#ifndef PT_SORT_CELL_BITS
#define PT_SORT_CELL_BITS 4      // shadow-ray sort key (pt_raysort.hip): bits per axis of the origin's cell; the key is 3 x this + 3 octant bits wide.
                                 // RT1M, k_trace ms per launch: no sort 164.9, 0 bits (octant only) 162.2, 1: 161.5, 2: 158.9, 3: 157.3, 4: 156.8, 5: 156.6, 6 .. 9: 156.8
#endif
#define PT_TOP_BFS_NODES 1365u   // what the upload renumbers breadth-first (levels 0..5), whatever PT_TOP_NODES the kernels were built with
// LDS stack slots per lane of the pooled-leaf traversal kernels (slot 0 holds a sentinel; deeper entries spill to HBM)
#if PT_NODE_STAGED
#define PT_FS_SLOTS (PT_LDS_STACK < 16 ? PT_LDS_STACK : 16)
#elif PT_TOP_NODES > 0
#define PT_FS_SLOTS (PT_LDS_STACK < 24 ? PT_LDS_STACK : 24)
#else
#define PT_FS_SLOTS PT_LDS_STACK
#endif
#define PT_DIAG_WORDS 2048u       // words at the head of the traversal spill buffer reserved for the diagnostic builds (phase clocks, histograms)
#define PT_EMPTY_REF 0xffffffffu
#define PT_LEAF_BIT 0x80000000u
#define PT_LEAF_COUNT_SHIFT 28          // bits 28..30 of a leaf reference: min(triangles in the leaf, 8) - 1
#define PT_LEAF_FIRST_MASK 0x03ffffffu   // bits 0..25: first triangle record
#define PT_REF_INDEX_MASK 0x03ffffffu    // interior reference: bits 0..25 node index
#define PT_REF_AXIS_SHIFT 26             // bits 26..27 of a CHILD reference stored in a node: one of the node's three split axes, so that the
                                         // lean visit needs no separate load for them -- child 0 carries axis_top, child 1 axis_left, child 3
                                         // axis_right (slots 0 and 2 are always occupied; slot 1 / 3 exactly when the left / right pair is one)

// One 4-wide BVH node = one 128-byte line.  Replaces the reference's 144-byte
// SIMDBVHNode + separate leaf nodes (qbvh_x86.rs:15-24, :93-176): a child that is
// a leaf is referenced as PT_LEAF_BIT | first_triangle_record, so leaves cost no
// node fetch; an empty slot is PT_EMPTY_REF.
struct PtNode {
    float bmin[3][4];            // [axis][child]
    float bmax[3][4];
    uint32_t child[4];
    uint32_t axes;               // axis_top | axis_left << 2 | axis_right << 4 | occupied-slot mask << 8
    uint32_t order_lut;          // written at upload (pt_context.cpp finish_nodes): bit o of byte 0 / 1 / 2 = the ray of direction octant o
                                 // (bit 0 x < 0, bit 1 y < 0, bit 2 z < 0) is negative along axis_top / axis_left / axis_right
    uint32_t pad[2];
};

// Triangle record in BVH leaf order (48 bytes = 3 x dwordx4).  The last record of
// a leaf carries PT_TRI_LAST, so a leaf reference needs no count.
#define PT_TRI_LAST 1u
#define PT_TRI_ONE_SIDED 2u      // !two_sided: cull when dot(n, d) >= 0 (triangle.rs:247-251)
#define PT_TRI_FLIP 4u           // reverse_orientation ^ swaps_handedness
#define PT_TRI_HAS_ATTR 8u       // mesh carries N / S / UV: shading must also read PtTriInfo
#define PT_TRI_SPHERE 16u        // the record stands for a sphere: p0[0] holds its index into PtScene::spheres (as bits)
#define PT_TRI_INSTANCE 32u      // the record stands for an object instance: p0[0] holds its index into PtScene::instances (as bits)
#define PT_TRI_MATERIAL_SHIFT 16 // bits 16..31: material index + 1 (0 = no material)
struct PtTri {
    float p0[3];
    uint32_t prim;               // caller's triangle index
    float p1[3];
    uint32_t flags;              // PT_TRI_* | (material + 1) << 16: shading needs no second record
    float p2[3];
    uint32_t light1;             // area light index + 1 (0 = not emissive)
};

// Per original triangle shading record.
struct PtTriInfo {
    uint32_t v[3];               // vertex indices (N / S / UV lookups)
    uint32_t mesh;
    int32_t light;               // index into lights[] or -1
    int32_t material;            // index into materials[] or -1
    uint32_t mesh_flags;         // PT_MESH_* of include/pbrtgpu.h
    uint32_t pad;
};

// One BxDF of a material's BSDF.  With constant textures Material::compute_scattering_functions adds
// the same BxDFs at every hit, so the host evaluates it once per material (pt_context.cpp build_lobes).
enum { PT_LOBE_LAMBERT = 0, PT_LOBE_OREN_NAYAR = 1, PT_LOBE_SPEC_REFL = 2, PT_LOBE_SPEC_TRANS = 3, PT_LOBE_FRESNEL_SPEC = 4,
       PT_LOBE_MF_REFL = 5, PT_LOBE_MF_TRANS = 6, PT_LOBE_FRESNEL_BLEND = 7 };
enum { PT_FR_NOOP = 0, PT_FR_DIELECTRIC = 1, PT_FR_CONDUCTOR = 2 };
struct PtLobe {                  // 80 bytes
    uint32_t kind;               // PT_LOBE_*
    uint32_t type;               // BxDFType bits (bxdf.rs:8-14)
    uint32_t fresnel;            // PT_FR_* of SpecularReflection / MicrofacetReflection
    float ax;                    // TrowbridgeReitz alpha_x (after max(0.001, .))
    float r[3];                  // R / T / Kd / Rd
    float ay;
    float t[3];                  // FresnelSpecular T, FresnelBlend Rs, conductor eta
    float eta_a;
    float k[3];                  // conductor k
    float eta_b;
    float fr_eta_i, fr_eta_t;    // FresnelDielectric
    float oa, ob;                // OrenNayar A, B
};
#define PT_MAX_LOBES 5           // UberMaterial adds at most five BxDFs
struct PtMaterial {
    int32_t type;
    float kd[3];
    float sigma;
    float oren_a, oren_b;        // OrenNayar::new (oren_nayar.rs:16-23), precomputed on the host
    uint32_t n_lobes;
    uint32_t has_bsdf;           // 0: compute_scattering_functions leaves bsdf = None (glass with Kr = Kt = 0)
    uint32_t nonspecular;        // num_components(BSDF_ALL & !BSDF_SPECULAR)
    float bsdf_eta;              // BSDF::eta
    uint32_t sort_bin;           // shade-queue bin: [0,128) Matte materials, [128,256) the others
    uint32_t textured;           // 1: parameters come from textures at each hit (PtMatParams), the lobes below are unused
    uint32_t spec_mask;          // bit 0: some lobe matches BSDF_REFLECTION | BSDF_SPECULAR, bit 1: ... BSDF_TRANSMISSION | BSDF_SPECULAR (what specular_reflect /
                                 // specular_transmit ask BSDF::sample_f for: with the bit clear the call returns None before it looks at anything)
    uint32_t pad[2];
    PtLobe lobes[PT_MAX_LOBES];
};
// One MIP pyramid (core/texture/mipmap.rs) in HBM: all levels back to back, level l at texels + level_off[l] floats.
#define PT_MAX_MIP_LEVELS 16
struct PtImage {
    const float* texels;
    uint32_t width, height;      // level 0 (powers of two); level l is max(1, width >> l) x max(1, height >> l)
    uint32_t channels;           // 1 or 3
    uint32_t n_levels;
    uint32_t level_off[PT_MAX_MIP_LEVELS];
    uint32_t pad[2];
};
#define PT_TEX_PROG_MAX 12          // nodes one parameter's texture graph may need
#define PT_TEX_CHILD_CONST 15u      // texture program entry, child slot: the node's own constant (pt_texture.h)
// A material with texture-driven parameters: the caller's parameter block, the roughness values after the optional
// roughness_to_alpha remap (host-side: logf), and per parameter the offset of its texture program in PtScene::tex_prog
// (0 = constant).  Parameter order: Kd Ks Kr Kt opacity sigma metal-eta metal-k bump roughness uroughness vroughness eta.
struct PtMatParams {
    pt_material m;
    float a_r, a_u, a_v;
    uint32_t prog[13];           // [8] = the bump map's displacement texture, [9..12] = roughness, uroughness, vroughness, eta
};

// One DiffuseAreaLight (one emissive triangle).
struct PtLight {
    float p0[3]; float area;
    float p1[3]; uint32_t mesh_flags;
    float p2[3]; int32_t two_sided;
    float L[3];  uint32_t tri_rec;   // index of the triangle's PtTri record
    float n0[3]; uint32_t prim;
    float n1[3]; uint32_t n_samples;   // the light's sample count (DirectLighting "all": size of its sample arrays), >= 1
    float n2[3]; uint32_t pad1;
};

// Sphere (shapes/sphere.rs:7-41), affine transforms only.
#define PT_SPH_REVERSE 1u        // reverse_orientation: flips sampled normals (sphere.rs:293-295, :380-382)
#define PT_SPH_FLIP 2u           // reverse_orientation ^ transform_swaps_handedness: flips the intersection normal
struct PtSphere {                // 144 bytes
    float o2w[12];               // rows 0..2 of object_to_world.m
    float w2o[12];               // rows 0..2 of object_to_world.m_inv
    float radius, z_min, z_max, phi_max;
    float theta_min, theta_max, area;
    uint32_t flags;
    uint32_t pad[4];
};
// ObjectInstance: TransformedPrimitive over an object's accelerator (affine, static).
struct PtInstance {              // 144 bytes
    float m[12];                 // rows 0..2 of instance_to_world
    float minv[12];              // rows 0..2 of its stored inverse
    uint32_t root_ref;           // the object's root (node index or leaf reference); direct: its single record's index
    uint32_t direct;             // 1: one primitive, wrapped without an accelerator (no root box test)
    float root_lo[3], root_hi[3];// the accelerator's bound (QBVHAccel::intersect starts with it)
    uint32_t world_prim;         // the instance's index in the world primitive list (pt_hit.prim)
    uint32_t pad[3];
};
#define PT_LIGHT_SPHERE 0x80000000u   // PtLight::mesh_flags: the light's shape is sphere number bits(p0[0])

struct PtCamera {
    float raster_to_camera[16];
    float camera_to_world[16];
    float lens_radius, focal_distance;
    float pad[2];
};

struct PtFilm {
    int32_t crop[4];             // x0 y0 x1 y1
    int32_t sample_bounds[4];
    float filter_radius[2];
    float inv_filter_radius[2];
    float max_sample_luminance;
    float scale;
    int32_t box_unit;            // 1 when every in-pixel sample has a one-pixel footprint of weight 1
    int32_t pad;
    float filter_table[256];
};

struct PtSobol {
    const uint32_t* m32;         // [n_dims * 52]
    const uint64_t* vdc;         // row (log2_resolution - 1) of VDC_SOBOL_MATRICES
    const uint64_t* vdc_inv;     // row (log2_resolution - 1) of VDC_SOBOL_MATRICES_INV
    uint32_t m32_len;
    const uint32_t* bytetab;     // [n_tab_dims][7][256]: XOR of the columns selected by one index byte
    uint32_t n_tab_dims;
    uint32_t pad;
    uint32_t log2_resolution;
    uint32_t resolution;
    uint32_t spp;                // Sobol': rounded up to a power of two; Halton: as given
    // ---- HaltonSampler (samplers/halton.rs:46-162); kind selects the sampler
    uint32_t kind;               // PT_SAMPLER_SOBOL / PT_SAMPLER_HALTON
    uint32_t h_center;           // "samplepixelcenter"
    uint32_t h_exp[2];           // base_exponents
    uint32_t h_scale1;           // base_scales[1]
    uint32_t h_stride;           // sample_stride
    uint32_t h_mul[2];           // (sample_stride / base_scales[i]) * mult_inverse[i]
    const uint4* h_dims;         // per dimension: {prime, offset into perms, magic lo, magic hi}; magic = floor(2^64/prime)+1
    const uint16_t* h_perms;     // radical-inverse digit permutations (halton.rs:12-20)
    uint32_t h_n_dims;
    uint32_t pad2;
};

struct PtLightGrid {
    // dense replacement for the reference's lazily filled hash table (spatial.rs:199-260):
    // voxel v owns floats [v*stride, (v+1)*stride): func[nl], cdf[nl+1], func_int
    const float* data;
    uint32_t voxels[3];
    uint32_t n_lights;
    uint32_t stride;
    uint32_t single;             // 1: uniform / power strategy, one table for every point
    float wb_min[3], wb_max[3];
    // Lazy fill (scenes whose dense grid would not fit: thousands of lights): row_of[voxel] = the row of `data` that holds the voxel's tables, or
    // a negative number while nobody has asked for it (spatial.rs:199-260 fills on first touch as well).  nullptr: dense, row = voxel.
    const int32_t* row_of;
};

struct PtCounters {
    unsigned long long regular_rays, shadow_rays, nodes, tris, vertices, camera_rays;
    unsigned long long nodes_lds;       // of `nodes`: visits k_trace served from its LDS copy of the top of the tree (no L1 request)
};

struct PtScene {
    uint32_t dist_leaves;        // 1: no leaf holds more than 8 triangles, k_trace spreads leaf tests over the wave
    uint32_t any_one_sided;      // 1: some triangle is one-sided ("twosided" false): leaf rounds also need the ray direction
    uint32_t general_materials;  // 1 when any material is not Matte: k_shade_general runs instead of k_shade
    const PtNode* nodes;
    const PtTri* tris;
    const PtTriInfo* tri_info;
    const float* N; const float* S; const float* UV;     // per-vertex attributes or null
    const PtMaterial* materials;
    const PtLight* lights;
    const PtSphere* spheres;
    const PtInstance* instances;
    uint32_t n_instances;        // > 0: k_trace_inst / k_shade_general_inst run
    uint32_t pad_inst;
    const pt_texture* textures;  // pt_scene_desc.textures as given
    const uint32_t* tex_prog;    // texture programs (pt_texture.h)
    const PtImage* images;       // MIP pyramids of the imagemap textures
    const PtMatParams* mat_params;   // per material; read for textured materials only
    uint32_t textured;           // 1: some material is textured (k_shade_general_full runs)
    uint32_t n_spheres;          // > 0: the sphere-capable kernel instantiations run
    uint32_t n_lights;
    uint32_t root_ref;           // node 0, or a leaf reference when the whole scene is one leaf
    uint32_t n_top;              // nodes 0 .. n_top-1 are the top of the world tree in breadth-first order (at most PT_TOP_BFS_NODES)
    float cell_scale[3];         // 2^PT_SORT_CELL_BITS / the world bound's extent per axis (0 for a flat axis): ray origin -> cell coordinate
    float wb_min[3], wb_max[3];  // BVH root bounds (Scene::world_bound)
    int32_t max_depth;
    float rr_threshold;
    int32_t integrator;          // pt_integrator_type
    int32_t direct_strategy;     // pt_direct_strategy (directlighting)
    int32_t ao_samples, ao_cos_sample;
    PtCamera cam;
    PtFilm film;
    PtSobol sobol;
    PtLightGrid grid;
};

// ---- wavefront path pool (SoA, one slot per in-flight camera sample)
struct PtPaths {
    float4* ray_o;       // xyz origin, w = t_max
    float4* ray_d;       // xyz direction
    float4* sh_o;        // shadow ray (NEE light sample), w = t_max
    float4* sh_d;
    float4* pr_o;        // MIS probe ray, w = t_max (inf)
    float4* pr_d;
    float4* beta;        // xyz, w = eta_scale
    float4* L;           // xyz radiance, w unused
    float4* pendA;       // light-sample term f*Li*(w/pdf), w = selection pdf
    float4* pendB;       // BSDF-sample term,              w = beta snapshot index unused
    float4* pbeta;       // beta at NEE time
    float2* p_film;
    uint64_t* sobol_index;
    uint32_t* pixel;     // x | y << 16 relative to sample_bounds.min
    uint32_t* state;     // dim (16) | bounces (8) | flags (8)
    float* hit_t;
    int32_t* hit_rec;    // triangle record index or -1
    uint32_t* nee;       // bit0 shadow ray live, bit1 probe live, bits 8.. light index
    uint8_t* occluded;
    int32_t* probe_rec;
    uint32_t* hit_inst;  // instance index + 1 of the closest hit (0 = a world primitive); allocated for scenes with instances only
    float4* tex_res;     // [n_paths][PT_TEX_RES_F4]: what k_tex_resolve evaluated at a textured hit (parameter values, alphas, the bump-mapped
                         // shading frame), read by k_shade_general_res; allocated for scenes with textured materials only
};
#define PT_TEX_RES_F4 9u

// ---- DirectLightingIntegrator / WhittedIntegrator on the wavefront (pt_kernels.hip, "recursive integrators"): the per-camera-sample
// state of the depth-first walk over the specular_reflect / specular_transmit tree
#define PT_REC_FRAME_F4 8u       // float4 per frame
struct PtRec {
    float4* diff;            // [4][n_paths]: the current ray's offset rays rx_o, ry_o, rx_d, ry_d (valid when PT_ST_DIFF is set)
    float4* frames;          // [max_depth][PT_REC_FRAME_F4][n_paths]: ray o | hit record, ray d | instance, rx_o | phase, ry_o, rx_d | pending scale,
                             //                                        ry_d, l, pending f
    // next-event entries, entry e = path * epp + j (j: light by light, one entry per light sample; 0 for "one"): shadow ray, MIS probe ray, the two
    // pending terms
    float4 *sh_o, *sh_d, *pr_o, *pr_d, *A, *B;       // A.w = what the light's summed estimate is divided by ("one": the selection pdf; "all": the sample count), B.w unused
    uint8_t* occ;            // shadow ray result
    int32_t* prec;           // probe ray result (closest record)
    uint32_t* flags;         // per entry: PT_NEE_SHADOW | PT_NEE_PROBE | light << 8
    uint32_t n_paths, max_depth, epp, n_arrays;      // epp: entries per path ("all": the lights' sample counts added up); n_arrays: 2-D sample arrays
                                                     // requested per pixel ("all": two per light and depth)
    uint32_t s0, n_pix;                              // of the pass: path p is sample number s0 + p / n_pix of its pixel (the arrays are indexed by it)
    uint32_t* panic;                                 // the context's error word: bit 2 (4) = a node asked the Halton sampler for a dimension past its
                                                     // table, where the reference panics (halton.rs:103-108)
};
#define PT_REC_OUT_FRAME 0u      // k_rec_enter outcomes (bits 2..3 of the state's flag byte): a frame was pushed, next-event rays may be pending
#define PT_REC_OUT_RETURN0 1u    // the ray left the scene (or Whitted met a surface without BSDF): the node returns zero
#define PT_REC_OUT_RETRACE 2u    // DirectLighting passed through a surface without BSDF: the new ray is traced at the same depth
#define PT_ST_DIFF 1u            // recursive integrators: the current ray carries differentials

#define PT_ST_SPECULAR 1u
#define PT_ST_CAMERA 2u         // the ray is still the camera ray: it has differentials (textures filter with them)
#define PT_NEE_SHADOW 1u
#define PT_NEE_PROBE 2u
#define PT_NEE_DIMS5 4u          // split shading (part 1 -> part 2): the vertex drew all five next-event dimensions (light choice, u_light, u_scattering), not just the light choice

struct PtQueues {
    uint32_t* cur;       // path ids to shade / whose continuation ray is traced
    uint32_t* next;
    uint32_t* nee;       // path ids with a pending NEE resolve (k_nee_resolve)
    uint32_t* shadow;    // path ids whose NEE shadow ray (any hit) is to be traced   (counts[7])
    uint32_t* probe;     // path ids whose MIS probe ray (closest hit) is to be traced (counts[8])
    uint32_t* counts;    // counters, PT_Q_* below.  The ones every wave bumps with an atomic sit 128 bytes apart:
                         // atomics on one line serialise in a single L2 channel (k_shade was bound by exactly that)
    uint32_t* sorted;    // cur re-ordered by material bin, misses dropped (scenes with non-Matte materials)
    uint16_t* bin;       // per entry of cur: its material bin (0xffff: a miss), written by k_sort_count for k_sort_scatter; nullptr: recomputed there
    uint32_t* shadow_key;// per entry of shadow: the sort key of its ray (origin cell | direction octant, pt_raysort.hip); nullptr: not wanted
};
#define PT_Q_CUR 0u          // items in cur
#define PT_Q_MATTE_END 1u    // material sort: end of the Matte segment of sorted
#define PT_Q_GENERAL_END 2u  // material sort: end of the sorted queue (= end of the textured segment)
#define PT_Q_TEX_BEGIN 3u    // material sort: end of the general segment = start of the textured-material segment
#define PT_Q_NEXT 32u        // items pushed to next
#define PT_Q_NEE 64u         // paths with a pending NEE resolve
#define PT_Q_TICKET 96u      // work ticket (k_trace, k_shade Matte segment)
#define PT_Q_TICKET2 128u    // work ticket of k_shade_general
#define PT_Q_TICKET3 224u    // work ticket of k_shade_general_tex (textured-material segment)
#define PT_Q_SHADOW 160u     // shadow rays queued
#define PT_Q_PROBE 192u      // probe rays queued
#define PT_SORT_BINS 256u
#define PT_SORT_GENERAL0 128u
#define PT_SORT_TEX0 224u       // bins [224, 256): materials with texture-driven parameters (lobes built per hit)
#define PT_SORT_COUNT0 256u                       // [+256) bin counts
#define PT_SORT_CURSOR0 (256u + 256u)             // [+256) bin cursors
#define PT_Q_SEG_TICKET0 768u                     // k_trace: one ticket per queue segment (8 segments, 128 B apart)
#define PT_Q_TICKET_N1 1024u                      // work tickets of the next-event halves of the split shading kernels (Matte / general / textured segment)
#define PT_Q_TICKET_N2 1056u
#define PT_Q_TICKET_N3 1088u
#define PT_COUNTS_WORDS (768u + 8u * 32u + 3u * 32u)
