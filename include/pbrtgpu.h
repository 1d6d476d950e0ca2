/*
 * pbrtgpu.h -- C ABI of the MI355X path-tracing core (libpbrtgpu.so).
 *
 * This is the drop-in boundary for pbrt-r3's per-pixel radiance loop.  pbrt-r3
 * has no FFI of its own; its extension seams are Rust traits.  Each entry point
 * below names the reference interface it stands in for (paths relative to the
 * reference tree):
 *
 *   pt_scene_upload      <- what SceneContext::pbrt_shape / make_scene /
 *                           make_integrator assemble
 *                           (src/core/api/scene_context/scene_context.rs:1201-1318,
 *                            :718-729, :675-716) + BVHAccel::new
 *                           (src/accelerators/bvh/accel/qbvh/qbvh_x86.rs:352-370)
 *   pt_render            <- Integrator::render (src/core/integrator/integrator.rs:6-9),
 *                           i.e. SampleIntegratorCore::render
 *                           (src/core/integrator/sampler.rs:259-325)
 *   pt_film_*            <- Film::merge_film_tile / write_image
 *                           (src/core/film/film.rs:219-241, :440-484)
 *   pt_trace_closest/any, pt_trace_wavefront <- Scene::intersect / intersect_p
 *                           (src/core/scene/scene.rs:44-54)
 *   pt_generate_camera_rays <- Sampler::get_camera_sample +
 *                           PerspectiveCamera::generate_ray_differential
 *                           (src/core/sampler/sampler.rs:26-36,
 *                            src/cameras/perspective.rs:121-183)
 *   pt_sobol_samples     <- SobolSampler::sample_dimension (src/samplers/sobol.rs:167-185)
 *   pt_bsdf_eval/sample  <- BSDF::f / pdf / sample_f (src/core/reflection/bsdf.rs:92-270)
 *   pt_get_counters      <- the stat counters "Intersections/Regular ray intersection
 *                           tests" / "Shadow ray intersection tests"
 *                           (src/core/scene/scene.rs:11-12)
 *
 * Conventions: plain C, caller-owned host buffers unless a parameter is named
 * dev_*, every function returns a pt_status, one context per device, contexts
 * are thread-compatible (not thread-safe).  No torch / HIP types appear here.
 */
#ifndef PBRTGPU_H
#define PBRTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 9

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1,
    PT_ERR_NO_DEVICE = 2,        /* no HIP device / runtime: there is NO CPU fallback */
    PT_ERR_DEVICE = 3,           /* a HIP call failed; see pt_last_error */
    PT_ERR_UNSUPPORTED = 4,      /* feature outside the accelerated path */
    PT_ERR_NO_SCENE = 5,
    PT_ERR_OUT_OF_MEMORY = 6
} pt_status;

typedef struct pt_context pt_context;

/* ---- flattened scene description ------------------------------------- */

/* Procedural textures (src/textures/, src/core/texture/mapping2d.rs, mapping3d.rs).  A texture evaluates to an RGB triple;
 * a float texture is one whose three channels are equal (consumers read channel 0), so scale / mix / checkerboard are the
 * same arithmetic per channel as the reference's generic Texture<T>.  textures[] is in definition order: a child index is
 * always smaller than the index of the texture that uses it. */
typedef enum {
    PT_TEX_CONSTANT = 0,         /* core/texture/constant.rs: value[0] */
    PT_TEX_SCALE = 1,            /* textures/scale.rs: tex1 * tex2 */
    PT_TEX_MIX = 2,              /* textures/mix.rs: tex1 * (1 - amount) + tex2 * amount; child 2 = "amount" */
    PT_TEX_CHECKERBOARD_2D = 3,  /* textures/checkerboard.rs:13-84: point-sampled or closed-form box-filtered */
    PT_TEX_CHECKERBOARD_3D = 4,  /* textures/checkerboard.rs:86-122 over the identity 3-D mapping */
    PT_TEX_UV = 5,               /* textures/uv.rs: (frac s, frac t, 0) */
    PT_TEX_BILERP = 6,           /* textures/bilerp.rs: value[0..3] = v00 v01 v10 v11 */
    PT_TEX_DOTS = 7,             /* textures/dots.rs: tex1 outside, tex2 inside noise-placed dots; 2-D mapping */
    PT_TEX_FBM = 8,              /* textures/fbm.rs: noise.rs fbm(p, dpdx, dpdy, omega, octaves) over the 3-D mapping */
    PT_TEX_WRINKLED = 9,         /* textures/wrinkled.rs: turbulence(...) */
    PT_TEX_WINDY = 10,           /* textures/windy.rs: |fbm(.1p, .., .5, 3)| * fbm(p, .., .5, 6) */
    PT_TEX_MARBLE = 11,          /* textures/marble.rs: spline over sin(scale*p.y + variation * fbm(...)) */
    PT_TEX_IMAGEMAP = 12         /* textures/imagemap.rs: MIPMap::lookup_delta over images[image]; 2-D mapping */
} pt_texture_type;
typedef enum { PT_WRAP_REPEAT = 0, PT_WRAP_BLACK = 1, PT_WRAP_CLAMP = 2 } pt_image_wrap;      /* core/texture/texture.rs:7-12 */
/* A MIP pyramid as MIPMap::new leaves it (core/texture/mipmap.rs:406-485): level 0 has power-of-two width x height (the
 * caller resamples, :316-404), each further level halves the dimensions that are still > 1 (box filter), down to 1 x 1.
 * texels: all levels back to back, level 0 first, row-major, `channels` floats per texel (1 = float image, 3 = RGB), already
 * inverse-gamma'd, scaled and flipped in y as ImageTexture does (textures/imagemap.rs). */
typedef struct {
    uint32_t width, height;     /* level 0; powers of two */
    uint32_t channels;          /* 1 or 3 */
    uint32_t n_levels;          /* 1 + log2(max(width, height)) */
    const float* texels;
} pt_image;
typedef enum {
    PT_MAPPING_UV = 0,           /* UVMapping2D: su sv du dv */
    PT_MAPPING_SPHERICAL = 1,    /* SphericalMapping2D: world_to_texture */
    PT_MAPPING_CYLINDRICAL = 2,  /* CylindricalMapping2D: world_to_texture */
    PT_MAPPING_PLANAR = 3        /* PlanarMapping2D: v1 v2, du dv as ds dt */
} pt_mapping_type;
typedef struct {
    int32_t type;               /* pt_texture_type */
    int32_t tex[3];             /* children "tex1", "tex2", "amount": index into textures[] (< own index), or -1 = value[i] */
    float value[4][3];          /* constants: see the type; a float is replicated over the three channels */
    int32_t mapping;            /* pt_mapping_type, 2-D textures ("mapping", default "uv") */
    int32_t aa_none;            /* checkerboard "aamode" "none" (default "closedform") */
    float su, sv, du, dv;       /* "uscale" "vscale" "udelta" "vdelta" (defaults 1 1 0 0) */
    float v1[3], v2[3];         /* planar "v1" (1 0 0), "v2" (0 1 0) */
    float world_to_texture[16]; /* inverse of the CTM at the Texture directive for spherical / cylindrical mappings; the 3-D
                                 * textures (checkerboard 3-D, fbm, wrinkled, windy, marble) take the CTM itself, which is what the
                                 * reference hands IdentityMapping3D (checkerboard.rs:159, fbm.rs:40) */
    int32_t octaves;            /* fbm / wrinkled / marble "octaves" (8) */
    float omega;                /* "roughness" (0.5) */
    float scale, variation;     /* marble "scale" (1), "variation" (0.2) */
    int32_t image;              /* imagemap: index into images[] */
    int32_t trilinear;          /* imagemap "trilinear" (false: EWA) */
    float max_anisotropy;       /* imagemap "maxanisotropy" (8) */
    int32_t swrap, twrap;       /* a pt_image_wrap value: "wrap" / "swrap" / "twrap", default repeat */
    int32_t reserved[3];
} pt_texture;

/* Material::compute_scattering_functions variants (src/materials/).  Colour parameters and Matte's sigma may be
 * textures (tex_* below), a procedural bump map is applied (tex_bump), and (ABI 6) the float parameters "roughness" /
 * "uroughness" / "vroughness" / "eta" may be textures too: they are evaluated at every hit and the roughness goes through
 * TrowbridgeReitzDistribution::roughness_to_alpha (`ln`, core/distribution/trowbridge_reitz.rs:113-121) on the device. */
typedef enum {
    PT_MATERIAL_NONE = 0,      /* GeometricPrimitive.material == None: ray passes through (path.rs:108-111) */
    PT_MATERIAL_MATTE = 1,     /* materials/matte.rs:25-53      Kd, sigma */
    PT_MATERIAL_PLASTIC = 2,   /* materials/plastic.rs:31-71    Kd, Ks, roughness, remaproughness */
    PT_MATERIAL_MIRROR = 3,    /* materials/mirror.rs:19-41     Kr */
    PT_MATERIAL_GLASS = 4,     /* materials/glass.rs:46-110     Kr, Kt, eta, uroughness, vroughness, remaproughness */
    PT_MATERIAL_METAL = 5,     /* materials/metal.rs:51-85      eta (spectrum -> RGB), k, roughness | uroughness/vroughness */
    PT_MATERIAL_UBER = 6,      /* materials/uber.rs:63-127      Kd, Ks, Kr, Kt, opacity, eta, roughness | u/v roughness */
    PT_MATERIAL_SUBSTRATE = 7  /* materials/substrate.rs:34-68  Kd, Ks, uroughness, vroughness */
} pt_material_type;

#define PT_ROUGHNESS_UNSET (-1.0f)   /* "uroughness"/"vroughness" not given: Metal and Uber fall back to "roughness" */

/* Field defaults are the reference's create_*_material defaults; a field a material type does not
 * read is ignored.  164 bytes. */
typedef struct {
    int32_t type;           /* pt_material_type */
    float kd[3];            /* "Kd": matte 0.5, plastic/uber 0.25, substrate 0.5 */
    float sigma;            /* matte "sigma" degrees, clamped to [0,90]; 0 => Lambertian */
    float ks[3];            /* "Ks": plastic/uber 0.25, substrate 0.5 */
    float kr[3];            /* "Kr": mirror 0.9, glass 1, uber 0 */
    float kt[3];            /* "Kt": glass 1, uber 0 */
    float opacity[3];       /* uber "opacity", default 1 */
    float eta;              /* glass/uber "eta" (alias "index"), default 1.5 */
    float roughness;        /* plastic 0.1, metal 0.01, uber 0.1 */
    float uroughness;       /* glass 0, substrate 0.1; metal/uber: PT_ROUGHNESS_UNSET unless given */
    float vroughness;
    int32_t remap_roughness;/* "remaproughness", default true */
    float metal_eta[3];     /* metal "eta" as RGB (the reference's default is the copper SPD) */
    float metal_k[3];       /* metal "k" as RGB */
    /* ABI 4: index + 1 of the texture that drives the parameter at each hit (Texture::evaluate(si)), 0 = the constant
     * above.  Zero-initialised materials are therefore constant. */
    uint32_t tex_kd, tex_ks, tex_kr, tex_kt, tex_opacity, tex_sigma, tex_metal_eta, tex_metal_k;
    uint32_t tex_bump;      /* "bumpmap" (core/material.rs:31-72): index + 1 of the displacement (float) texture, 0 = none */
    /* ABI 6: float textures (index + 1, 0 = the constant above) behind "roughness", "uroughness", "vroughness" and "eta" / "index"
     * (plastic.rs:57-62, glass.rs:57-81, metal.rs:58-69, uber.rs:66,96-104, substrate.rs:45-54).  A Metal / Uber material whose
     * "uroughness" ("vroughness") is neither a number nor a texture falls back to "roughness" -- constant or textured. */
    uint32_t tex_roughness, tex_uroughness, tex_vroughness, tex_eta;
} pt_material;

/* DiffuseAreaLight parameters shared by every triangle of one emissive shape
 * (src/lights/diffuse.rs:165-194).  One light is instantiated per triangle. */
typedef struct {
    float L[3];             /* "L" * "scale" */
    int32_t two_sided;      /* "twosided", default 0 */
    int32_t n_samples;      /* ABI 8: "nsamples" (diffuse.rs:177-189), default 1; 0 is read as 1.  Only DirectLighting's "all" strategy looks at
                             * it: every light of the shape is estimated from sample arrays of this size (directlighting.rs:51-64) */
} pt_area_light;

/* TriangleMesh flags (src/shapes/triangle.rs:10-22). */
#define PT_MESH_TWO_SIDED            1u  /* "twosided" shape param, default true (triangle.rs:707) */
#define PT_MESH_REVERSE_ORIENTATION  2u
#define PT_MESH_SWAPS_HANDEDNESS     4u
#define PT_MESH_HAS_N                8u
#define PT_MESH_HAS_S               16u
#define PT_MESH_HAS_UV              32u

typedef struct {
    uint32_t flags;
    int32_t material;       /* index into materials[], or -1 for none */
    int32_t area_light;     /* index into area_lights[], or -1 (ignored inside an object, scene_context.rs:1302-1304) */
    uint32_t object;        /* 0: the mesh is a world primitive; k: it belongs to object k - 1 (ObjectBegin ... ObjectEnd) */
} pt_mesh;

/* Sphere (src/shapes/sphere.rs:7-41, create_sphere_shape :401-420).  The only analytic shape on the
 * path: pbrt-v3's killeroo-simple lights its scene with two of them.  It lives in the same BVH as
 * the triangles (one primitive), is tested with the reference's EFloat interval arithmetic
 * (core/efloat/efloat.rs) in object space, and carries its own material / area light. */
#define PT_SPHERE_REVERSE_ORIENTATION 1u
typedef struct {
    float object_to_world[16];  /* row-major Transform.m of the CTM at the Shape directive */
    float world_to_object[16];  /* Transform.m_inv (the reference keeps both; it never re-inverts) */
    float radius;               /* "radius", default 1 */
    float zmin, zmax;           /* "zmin" / "zmax", defaults -radius / radius; clamped as Sphere::new does */
    float phimax;               /* "phimax" in degrees, default 360 */
    uint32_t flags;             /* PT_SPHERE_REVERSE_ORIENTATION */
    int32_t material;           /* index into materials[], or -1 */
    int32_t area_light;         /* index into area_lights[], or -1: one DiffuseAreaLight for the sphere */
    uint32_t before_triangle;   /* position in the primitive list: the sphere precedes triangle number
                                 * `before_triangle` (n_triangles = after the last one).  Non-decreasing
                                 * over spheres[]; equal values keep array order. */
    uint32_t object;            /* 0: world primitive; k: belongs to object k - 1 */
    uint32_t order;             /* creation order among spheres and instances that share a before_triangle value */
    uint32_t reserved[2];
} pt_sphere;

/* ObjectInstance (scene_context.rs:1349-1391): a TransformedPrimitive (core/primitive/transformed_primitive.rs) over the
 * accelerator built from the primitives of one object (the same Accelerator settings as the scene's; a single-primitive object
 * is wrapped directly).  Objects are the meshes / spheres tagged with `object`; their primitive order is their order in the
 * triangle array with their spheres spliced in by before_triangle, exactly as for the world list.  Rays are taken to instance
 * space with the stored inverse (Transform::inverse swaps m and m_inv) and hits brought back with
 * transform_surface_interaction.  Animated instance transforms are outside the accelerated path. */
typedef struct {
    float instance_to_world[16];    /* CTM at the ObjectInstance directive */
    float world_to_instance[16];
    uint32_t object;                /* index of the instanced object (0-based) */
    uint32_t before_triangle;       /* position in the WORLD primitive list, as for spheres */
    uint32_t order;                 /* creation order among spheres and instances sharing a before_triangle value */
    uint32_t reserved;
} pt_instance;

typedef enum { PT_SPLIT_SAH = 0, PT_SPLIT_HLBVH = 1, PT_SPLIT_MIDDLE = 2, PT_SPLIT_EQUAL_COUNTS = 3 } pt_split_method;
typedef enum {
    PT_SAMPLER_SOBOL = 0,   /* samplers/sobol.rs */
    PT_SAMPLER_HALTON = 1   /* samplers/halton.rs (the reference's default sampler) */
} pt_sampler_type;
/* PT_INTEGRATOR_AO: AOIntegrator::li (integrators/ao.rs:49-110) -- the camera ray's hit, then ao_samples occlusion rays over the
 * hemisphere of the geometric normal, directions from the sampler's 2-D sample array (dimensions 5, 6 of sample number
 * camera_sample * ao_samples + k; core/sampler/global_sampler.rs, samplers/sobol.rs:43-75).  Lights and materials play no part, except
 * that the reference panics on a surface without a material and lets a bump map flip the frame: both are refused at upload. */
/* PT_INTEGRATOR_DIRECTLIGHTING: DirectLightingIntegrator::li (integrators/directlighting.rs:66-135): emitted + direct light at the
 * hit ("strategy" all: every light, its u_light / u_scattering from the sampler's 2-D arrays while they last -- two arrays per light
 * and depth, area lights request one sample each; "one": uniform_sample_one_light with a uniform pick), then the trees of
 * specular_reflect / specular_transmit (core/integrator/sampler.rs:37-150) down to "maxdepth" (pt_scene_desc.max_depth, default 5),
 * ray differentials carried through the specular bounces.  PT_INTEGRATOR_WHITTED: WhittedIntegrator::li (integrators/whitted.rs:38-110):
 * one light sample per light and hit without MIS, weighted by the shading normal as it was before bump mapping, the same trees.
 * The sampler is consumed depth first (reflect subtree, then transmit), as the reference's recursion does. */
typedef enum { PT_INTEGRATOR_PATH = 0, PT_INTEGRATOR_AO = 1, PT_INTEGRATOR_DIRECTLIGHTING = 2, PT_INTEGRATOR_WHITTED = 3 } pt_integrator_type;
typedef enum { PT_DIRECT_ALL = 0, PT_DIRECT_ONE = 1 } pt_direct_strategy;      /* directlighting "strategy" (default "all") */
typedef enum { PT_LIGHTS_UNIFORM = 0, PT_LIGHTS_POWER = 1, PT_LIGHTS_SPATIAL = 2 } pt_light_strategy;

typedef struct {
    /* ---- geometry: world-space vertices (TriangleMesh::new pre-transforms, triangle.rs:49-66) */
    uint32_t n_vertices;
    const float* P;             /* 3*n_vertices */
    const float* N;             /* 3*n_vertices or NULL; read only where the mesh has PT_MESH_HAS_N */
    const float* S;             /* 3*n_vertices or NULL */
    const float* UV;            /* 2*n_vertices or NULL */
    uint32_t n_triangles;       /* degenerate triangles (area <= 1e-16) already dropped, triangle.rs:726 */
    const uint32_t* indices;    /* 3*n_triangles, into the vertex arrays */
    const uint32_t* tri_mesh;   /* n_triangles, index into meshes[] */
    uint32_t n_meshes;
    const pt_mesh* meshes;
    uint32_t n_materials;
    const pt_material* materials;
    uint32_t n_area_lights;
    const pt_area_light* area_lights;

    /* ---- accelerator (accelerators/bvh/create_bvh_accelerator.rs:13-27) */
    int32_t split_method;       /* pt_split_method, default SAH */
    int32_t max_node_prims;     /* default 4, clamped to 255 */

    /* ---- camera (cameras/perspective.rs:337-394) */
    float camera_to_world[16];  /* row-major Transform.m */
    float fov;                  /* degrees */
    float screen_window[4];     /* x0 x1 y0 y1 */
    float lens_radius;
    float focal_distance;
    float shutter_open, shutter_close;

    /* ---- film (core/film/film.rs:62-164, :520-580) */
    int32_t xres, yres;
    float crop_window[4];       /* x0 x1 y0 y1 in [0,1] */
    float filter_radius[2];
    float filter_table[256];    /* 16x16, Film::new's table (film.rs:102-120) */
    float film_scale;
    float max_sample_luminance; /* +inf = off */

    /* ---- sampler / integrator (samplers/sobol.rs:16-31, integrators/path.rs:252-271) */
    int32_t sampler;            /* pt_sampler_type */
    int32_t spp;                /* "pixelsamples"; Sobol' rounds up to a power of two, Halton takes it as given */
    int32_t max_depth;          /* default 5 */
    float rr_threshold;         /* default 1 */
    int32_t light_strategy;     /* pt_light_strategy, default spatial */
    int32_t halton_sample_at_center;  /* Halton "samplepixelcenter", default 0 */

    /* ---- analytic shapes (ABI 3).  The scene's primitive list is the triangles in order with the
     * spheres spliced in at `before_triangle`; BVH builders, light order (one light per emissive
     * primitive, scene_context.rs:1218-1231) and pt_hit.prim all use that merged numbering, which is
     * the plain triangle index when n_spheres == 0. */
    uint32_t n_spheres;
    const pt_sphere* spheres;

    /* ---- procedural textures (ABI 4), referenced by pt_material.tex_* */
    uint32_t n_textures;
    const pt_texture* textures;
    uint32_t n_images;
    const pt_image* images;     /* MIP pyramids of the imagemap textures */
    uint32_t n_instances;
    const pt_instance* instances;
    /* ---- ABI 5: the SamplerIntegrator that runs on the wavefront */
    int32_t integrator;         /* pt_integrator_type; 0 (zero-initialised descriptors) = path */
    int32_t ao_samples;         /* ao "nsamples", default 64: one 2-D sample array of this size per camera sample (ao.rs:14-36) */
    int32_t ao_cos_sample;      /* ao "cossample", default true */
    int32_t direct_strategy;    /* ABI 6: directlighting "strategy", a pt_direct_strategy */
} pt_scene_desc;

/* Axis-aligned block of film *sample* pixels, half-open: the unit the reference
 * hands to one rayon task (16x16, sampler.rs:271-289). */
typedef struct { int32_t x0, y0, x1, y1; } pt_tile;

typedef struct {
    float t;                /* hit distance (undefined when prim < 0) */
    int32_t prim;           /* index into the scene's primitive list (= the caller's triangle index
                             * when the scene has no spheres), -1 = miss */
    float b0, b1;           /* barycentrics of v0, v1 (triangle.rs:321-323); 0 for a sphere */
} pt_hit;

typedef struct {
    uint64_t camera_rays;       /* Integrator/Camera rays traced */
    uint64_t regular_rays;      /* Scene::intersect calls */
    uint64_t shadow_rays;       /* Scene::intersect_p calls */
    uint64_t nodes_visited;     /* interior 4-wide nodes popped */
    uint64_t tris_tested;       /* Triangle::intersect(_p) calls made by traversal */
    uint64_t path_vertices;     /* surface interactions shaded */
    uint64_t trace_launches;    /* launches of the traversal kernel */
    double   trace_ms;          /* device time in the traversal kernel (HIP events) */
    double   shade_ms;
    double   render_ms;         /* first ray-gen launch -> film ready on device */
    uint64_t nodes_from_lds;    /* of nodes_visited: visits the traversal kernel served from its LDS copy of the top of the tree */
} pt_counters;

typedef struct {
    int32_t sample_bounds[4];   /* x0 y0 x1 y1  Film::get_sample_bounds (film.rs:166-179) */
    int32_t cropped_bounds[4];  /* x0 y0 x1 y1 */
    int32_t spp;                /* rounded */
    uint32_t n_lights;
    uint32_t n_nodes;           /* 4-wide interior nodes */
    uint32_t n_leaves;
    float world_bound[6];
    double bvh_build_ms;
    double upload_ms;
    int32_t bvh_on_device;      /* ABI 5: 1 when the lower half of an HLBVH build ran on the GPU (pt_set_bvh_build) */
    int32_t reserved;
} pt_scene_info;

/* ---- context ---------------------------------------------------------- */
pt_status pt_context_create(int device, pt_context** out);
void      pt_context_destroy(pt_context* ctx);
const char* pt_last_error(const pt_context* ctx);   /* valid until the next call on ctx */
int       pt_abi_version(void);

/* Path of the Sobol' matrix table (data/sobol_tables.bin).  Must be set before
 * pt_scene_upload; pt_context_create looks next to the shared library first. */
pt_status pt_set_data_dir(pt_context* ctx, const char* dir);

/* ---- scene ------------------------------------------------------------ */
pt_status pt_scene_upload(pt_context* ctx, const pt_scene_desc* desc);
pt_status pt_scene_info_get(const pt_context* ctx, pt_scene_info* out);

/* ---- rendering -------------------------------------------------------- */
/* Zero the device film (RGB sums + weights over the cropped pixel bounds). */
pt_status pt_film_clear(pt_context* ctx);
/* Render `n_tiles` tiles (all spp) and accumulate into the device film.  Tiles
 * must lie inside the sample bounds and be pairwise disjoint (overlapping tiles
 * are rejected with PT_ERR_INVALID_ARGUMENT: a pixel's samples are folded by one
 * thread in sample order).  n_tiles == 0 with tiles == NULL renders
 * every 16x16 tile of the sample bounds (the reference's decomposition).
 * PT_ERR_UNSUPPORTED after the frame has been rendered: a node of the directlighting / whitted recursion asked the
 * Halton sampler for a dimension past its 1000 (every vertex draws two per light), where the reference panics
 * (samplers/halton.rs:103-108); the film then holds values taken from the last dimension and the context stays usable. */
pt_status pt_render(pt_context* ctx, const pt_tile* tiles, uint32_t n_tiles);
/* Film as {X,Y,Z,weight} float4 per cropped pixel: the quantity
 * Film::merge_film_tile accumulates and the unit of the multi-GPU sum-reduce. */
pt_status pt_film_download_xyzw(pt_context* ctx, float* xyzw_out /* 4*W*H */);
/* Same buffer, device pointer, for an RCCL reduce without a host round trip. */
pt_status pt_film_device_xyzw(pt_context* ctx, void** dev_ptr, size_t* n_floats);
/* Tell the context the XYZW buffer (possibly reduced in place) is authoritative. */
pt_status pt_film_commit_xyzw(pt_context* ctx);
/* Multi-GPU: sum the XYZW film over the ranks of an RCCL communicator the caller created (ncclComm_t passed as void*; one
 * rank = one context = one GPU) and mark the result authoritative -- the path's only exchange step (each rank has rendered
 * its share of the tiles into a zero-initialised film).  root < 0: ncclAllReduce; root >= 0: ncclReduce to that rank.
 * Runs on the library's stream; librccl.so.1 is resolved at run time (the copy already loaded in the process if any). */
pt_status pt_film_allreduce(pt_context* ctx, void* nccl_comm, int root);
/* ABI 9.  The same exchange step without a collective: add another rank's {X,Y,Z,weight} film (4*W*H floats in host memory, as
 * pt_film_download_xyzw returns it) to this context's film and mark the sum authoritative -- Film::merge_film_tile
 * (src/core/film/film.rs:219-241) across ranks that share no RCCL communicator (MPI / shared-memory hosts; several contexts on one
 * device, which ncclCommInitAll refuses).  Ranks added in rank order give the sums an all-reduce gives when tiles are disjoint. */
pt_status pt_film_add_xyzw(pt_context* ctx, const float* xyzw);
/* Film::write_image arithmetic: rgb = xyz_to_rgb(xyz)/weight, clamp >=0, *scale. */
pt_status pt_film_resolve_rgb(pt_context* ctx, float* rgb_out /* 3*W*H */);

/* ---- component hooks (parity tests; also usable by a host integrator) -- */
/* o,d: 3*n floats (AoS xyz); tmax: n floats. */
pt_status pt_trace_closest(pt_context* ctx, uint32_t n, const float* o, const float* d,
                           const float* tmax, pt_hit* out);
pt_status pt_trace_any(pt_context* ctx, uint32_t n, const float* o, const float* d,
                       const float* tmax, uint8_t* occluded_out);
/* The same queries through the WAVEFRONT's traversal kernel -- the one pt_render spends its time in (persistent lanes,
 * ray prefetch pipeline, per-XCD queue segments, pooled leaf rounds), not the plain one-ray-per-lane kernel behind
 * pt_trace_closest/any.  The rays are issued as the three kinds of work items one bounce mixes in a single launch:
 *   kind[i] = 1  continuation ray: Scene::intersect (scene.rs:44-48)   -> out[i] = {t, prim, b0, b1}
 *   kind[i] = 2  shadow ray:       Scene::intersect_p (scene.rs:50-54) -> occluded_out[i]
 *   kind[i] = 3  MIS probe ray:    Scene::intersect, only the primitive is kept (sample_lights.rs:419-447) -> out[i].prim
 * out and occluded_out hold n entries each (entries of the other kind are zeroed / prim = -1). */
pt_status pt_trace_wavefront(pt_context* ctx, uint32_t n, const float* o, const float* d, const float* tmax,
                             const uint8_t* kind, pt_hit* out, uint8_t* occluded_out);
/* Camera rays for pixel (px,py), sample index s: out_o/out_d 3 floats each per ray,
 * out_pfilm 2 floats per ray. */
pt_status pt_generate_camera_rays(pt_context* ctx, uint32_t n, const int32_t* pixel_xy,
                                  const uint32_t* sample_index, float* out_o, float* out_d,
                                  float* out_pfilm);
/* out[i] = sampler value for (pixel_xy[i], sample_index[i], dim[i]). */
pt_status pt_sobol_samples(pt_context* ctx, uint32_t n, const int32_t* pixel_xy,
                           const uint32_t* sample_index, const uint32_t* dim, float* out);
/* BSDF of material `material` of the uploaded scene on the canonical frame ns = ng = (0,0,1),
 * ss = (1,0,0): BSDF::f and BSDF::pdf (src/core/reflection/bsdf.rs:208-270) for n direction pairs
 * (world = local here), flags = BxDFType mask.  f_out 3 floats, pdf_out 1 float per pair. */
pt_status pt_bsdf_eval(pt_context* ctx, uint32_t material, uint32_t n, const float* wo, const float* wi,
                       uint32_t flags, float* f_out, float* pdf_out);
/* BSDF::sample_f (bsdf.rs:92-206): u is 2 floats per sample; type_out[i] = sampled BxDFType, or 0
 * when sample_f returns None (f/wi/pdf are then zero). */
pt_status pt_bsdf_sample(pt_context* ctx, uint32_t material, uint32_t n, const float* wo, const float* u,
                         uint32_t flags, float* f_out, float* wi_out, float* pdf_out, uint32_t* type_out);
/* Per-sample radiance (PathIntegrator::li after validate_radiance_result) for the
 * pixels of one tile: out is 3 floats per (pixel, sample), pixel-major. */
pt_status pt_radiance_samples(pt_context* ctx, const pt_tile* tile, float* out_rgb);

/* Host-only utility (no device, no context): builds the BVH exactly as pt_scene_upload
 * does and returns the leaf order (order_out[k] = primitive index stored k-th),
 * the 4-wide node / leaf counts and the traversal stack bound.  Lets a host check the
 * tree against the reference's ordered_prims (build/node.rs:138-151) without a GPU. */
pt_status pt_bvh_leaf_order(const pt_scene_desc* desc, uint32_t* order_out /* n_triangles + n_spheres */,
                            uint32_t* n_nodes, uint32_t* n_leaves, uint32_t* max_stack);

/* ABI 5.  Where the lower half of an HLBVH build runs (hlbvh.rs:354-428: Morton codes, radix sort, treelets, emit_lbvh): on the
 * GPU, or in the threaded host builder.  Both produce the same tree; AUTO takes the GPU from 65536 primitives per list up.  The
 * other split methods always build on the host.  Applies to the next pt_scene_upload. */
typedef enum { PT_BVH_BUILD_AUTO = 0, PT_BVH_BUILD_HOST = 1, PT_BVH_BUILD_DEVICE = 2 } pt_bvh_build_where;
pt_status pt_set_bvh_build(pt_context* ctx, int where);
/* FNV-1a digests of the uploaded 4-wide node array and of the leaf record array (read back from the device): two uploads of one scene
 * agree exactly when their trees and leaf orders do. */
pt_status pt_scene_bvh_digest(pt_context* ctx, uint64_t* nodes_digest, uint64_t* records_digest);

pt_status pt_get_counters(pt_context* ctx, pt_counters* out);
pt_status pt_reset_counters(pt_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PBRTGPU_H */
