// pt_context.cpp -- host side of libpbrtgpu.so: the C ABI of include/pbrtgpu.h.
//
// Plays the role of pbrt-r3's SceneContext::make_scene + make_integrator
// (src/core/api/scene_context/scene_context.rs:675-729) and of
// SampleIntegratorCore::render (src/core/integrator/sampler.rs:259-325), re-designed as
// a wavefront scheduler: camera samples are generated for a pass of S samples per pixel
// into an SoA path pool in HBM, then every bounce is one traversal launch (continuation
// rays + the previous bounce's shadow / MIS probe rays) and one shading launch; queues are
// compacted on the device and no host synchronisation happens inside a pass except one
// counter read at its end.  There is no CPU rendering path in this library.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <array>
#include <memory>
#include <vector>

#include "../../include/pbrtgpu.h"
#include "pt_bvh.h"
#include "pt_device.h"
#include "pt_host_math.h"
#include "pt_kernels.h"
#include "pt_lobes.h"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

uint32_t round_up_pow2(uint32_t v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
uint32_t log2int(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

}  // namespace

struct pt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err, data_dir;
    int n_cu = 256;

    // ---- scene
    bool have_scene = false;
    PtScene sc;
    pt_scene_info info;
    DevBuf d_nodes, d_tris, d_tri_info, d_spheres, d_instances, d_hit_inst, d_textures, d_tex_prog, d_mat_params, d_images, d_image_texels, d_N, d_S, d_UV, d_materials, d_lights, d_m32, d_vdc, d_vdc_inv, d_grid, d_bytetab, d_hdims, d_hperms;
    std::vector<uint32_t> sobol_m32;
    std::vector<uint64_t> sobol_vdc, sobol_inv;
    uint32_t sobol_n_vdc = 0, sobol_n_inv = 0, sobol_msize = 52;
    uint32_t n_materials = 0;
    uint32_t max_stack = 1;
    uint32_t film_w = 0, film_h = 0;
    int bvh_build_where = PT_BVH_BUILD_AUTO;
    size_t n_nodes_up = 0, n_tris_up = 0;

    // ---- traversal scratch
    DevBuf d_spill, d_err, d_counters, d_ticket, d_tex_res;
    uint32_t spill_depth = 0;
    int grid_trace = 1024, grid_trace_dist = 768, grid_shade = 512, grid_wide = 2048;

    // ---- path pool
    size_t pool_paths = 0;
    DevBuf d_pool;          // one slab carved into the SoA arrays of PtPaths
    PtPaths paths{};
    DevBuf d_qa, d_qb, d_qnee, d_qshadow, d_qprobe, d_qsorted, d_counts, d_pixels, d_tiles, d_tilebits;
    ptbvh::Result host_bvh;                       // upload scratch: the world tree's arrays
    std::unique_ptr<ptbvh::Prim[]> host_prims;    // upload scratch (see pt_scene_upload); released when it exceeds 4 M primitives
    size_t host_prims_cap = 0;
    DevBuf d_qbin;                                     // PtQueues::bin
    DevBuf d_sort_ids, d_sort_keys[2], d_sort_temp;   // pt_raysort.hip: the second id list, the key lists and rocprim's scratch
    size_t sort_cap = 0;
    DevBuf d_csort_ids, d_csort_keys[2], d_csort_temp; // the continuation rays' key list, the ordered copy (ids + keys) and rocprim's scratch
    size_t csort_cap = 0;
    // the lazily filled light grid (PtLightGrid::row_of): voxel -> row table, the list of voxels a bounce touched first, rows in use / allocated
    DevBuf d_grid_rows, d_grid_todo;
    size_t grid_rows_used = 0, grid_rows_cap = 0, grid_nvox = 0;
    bool grid_lazy = false;
    uint32_t* h_live = nullptr;                        // page-locked: the recursive integrators' live-sample count, read one level behind
    // k_trace_far (nodes fetched by lane pairs: scenes whose rays miss the caches): PBRTGPU_TRACE_FAR 0 never, 1 always, default -1 = decided per scene by
    // a timed trial on the scene's own rays -- the first incoherent bounce after an upload is traced twice, once by each kernel
    int trace_far = -1, trace_far_choice = -1;
    DevBuf d_cnt_save;
    // PBRTGPU_SHADE_LOCAL: the material sort inside runs of 4 096 queue entries + one lobe-list kernel (k_sort_local) instead of the global sort and a
    // kernel per class: 0 never, 1 wherever possible, default -1 = scenes that have lobe-list (non-Matte) constant materials, no textured ones and no
    // instances (mixed materials 889 -> 916 Mrays/s, with a sphere light as well 834 -> 843; a Matte scene lit by a sphere, where the sorted queue exists
    // only to select the sphere-capable kernels, loses 1.4 % and keeps the global sort)
    int shade_local = -1;
    bool scene_has_lobe_materials = false;
    int nee_split = 0;                                 // PBRTGPU_NEE_SPLIT: kernel families that shade a vertex in two kernels (ptk_shade)
    int sort_cont = -1;                                // continuation rays of a bounce ordered like the shadow rays: 0 never, 1 for the traversal kernel only
                                                       // (shading keeps path order), 2 for both, -1 (default): mode 1 for scenes larger than the Infinity Cache
    int sort_cont_min = 1 << 20;                       // ... and the continuation rays (when sort_cont is on) from this many up: its own threshold (PBRTGPU_SORT_CONT_MIN)
    int sort_shadow_min = 1 << 20;                     // shadow rays of a launch are ordered by origin cell from this many up (0: never)
    DevBuf d_rec, d_counts2; // recursive integrators (directlighting, whitted): frames, differentials, next-event entries and lists; the second counter block
    size_t rec_paths = 0;
    uint32_t rec_epp = 0, rec_depth = 0;
    uint32_t light_samples_total = 0;     // sum of the lights' sample counts (DirectLighting "all")
    DevBuf d_ao;             // AO integrator: occlusion-ray batch (o, d, tmax, weight, occluded) for ao_rays_cap rays
    size_t ao_rays_cap = 0;
    size_t pixels_cap = 0;

    // ---- film
    DevBuf d_own, d_spillfilm, d_xyzw, d_rgb;
    bool xyzw_committed = false;

    // ---- timing
    std::vector<hipEvent_t> ev;
    double trace_ms = 0, shade_ms = 0, render_ms = 0;
    uint64_t trace_launches = 0;

    pt_status fail(pt_status st, const std::string& m) { err = m; return st; }
    pt_status hip_fail(hipError_t e, const char* what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE;
    }
};

#define PT_HIP(call)                                             \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return ctx->hip_fail(e_, #call);   \
    } while (0)

namespace {

bool load_sobol(pt_context* ctx) {
    std::string path = ctx->data_dir + "/sobol_tables.bin";
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8];
    uint32_t hdr[4];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTSOBOL1", 8) == 0 && std::fread(hdr, 4, 4, f) == 4;
    if (ok) {
        ctx->sobol_msize = hdr[1];
        ctx->sobol_n_vdc = hdr[2];
        ctx->sobol_n_inv = hdr[3];
        ctx->sobol_m32.resize((size_t)hdr[0] * hdr[1]);
        ctx->sobol_vdc.resize((size_t)hdr[2] * hdr[1]);
        ctx->sobol_inv.resize((size_t)hdr[3] * hdr[1]);
        ok = std::fread(ctx->sobol_m32.data(), 4, ctx->sobol_m32.size(), f) == ctx->sobol_m32.size() &&
             std::fread(ctx->sobol_vdc.data(), 8, ctx->sobol_vdc.size(), f) == ctx->sobol_vdc.size() &&
             std::fread(ctx->sobol_inv.data(), 8, ctx->sobol_inv.size(), f) == ctx->sobol_inv.size();
    }
    std::fclose(f);
    return ok;
}

template <class T>
pt_status upload(pt_context* ctx, DevBuf& b, const T* src, size_t n) {
    PT_HIP(b.alloc(n * sizeof(T)));
    if (n) PT_HIP(hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}

pt_status ensure_pool(pt_context* ctx, size_t n_paths) {
    if (ctx->sc.n_instances && ctx->d_hit_inst.bytes < std::max(n_paths, ctx->pool_paths) * 4) {      // scenes with object instances only
        PT_HIP(ctx->d_hit_inst.alloc(std::max(n_paths, ctx->pool_paths) * 4));
        ctx->paths.hit_inst = ctx->d_hit_inst.as<uint32_t>();
    }
    const bool tex_split = ctx->sc.textured && !ctx->sc.n_instances;      // k_tex_resolve -> k_shade_general_res: 144 bytes per path, for such scenes only
    if (tex_split && ctx->d_tex_res.bytes < std::max(n_paths, ctx->pool_paths) * (size_t)(PT_TEX_RES_F4 * 16)) {
        PT_HIP(ctx->d_tex_res.alloc(std::max(n_paths, ctx->pool_paths) * (size_t)(PT_TEX_RES_F4 * 16)));
        ctx->paths.tex_res = ctx->d_tex_res.as<float4>();
    } else if (!tex_split && ctx->d_tex_res.p) {
        ctx->d_tex_res.release();
        ctx->paths.tex_res = nullptr;
    }
    if (ctx->pool_paths >= n_paths && ctx->d_pool.p) return PT_OK;
    // 11 float4 + float2 + u64 + 6 x 4-byte + 1 byte per path
    const size_t per_path = 11 * 16 + 8 + 8 + 6 * 4 + 4;
    PT_HIP(ctx->d_pool.alloc(n_paths * per_path + 16384));
    char* base = ctx->d_pool.as<char>();
    size_t off = 0;
    auto carve = [&](size_t elem) { void* p = base + off; off += (n_paths * elem + 255) & ~(size_t)255; return p; };
    PtPaths& P = ctx->paths;
    P.ray_o = (float4*)carve(16); P.ray_d = (float4*)carve(16);
    P.sh_o = (float4*)carve(16); P.sh_d = (float4*)carve(16);
    P.pr_o = (float4*)carve(16); P.pr_d = (float4*)carve(16);
    P.beta = (float4*)carve(16); P.L = (float4*)carve(16);
    P.pendA = (float4*)carve(16); P.pendB = (float4*)carve(16); P.pbeta = (float4*)carve(16);
    P.p_film = (float2*)carve(8);
    P.sobol_index = (uint64_t*)carve(8);
    P.pixel = (uint32_t*)carve(4); P.state = (uint32_t*)carve(4);
    P.hit_t = (float*)carve(4); P.hit_rec = (int32_t*)carve(4);
    P.nee = (uint32_t*)carve(4);
    P.probe_rec = (int32_t*)carve(4);
    P.occluded = (uint8_t*)carve(1);
    if (off > ctx->d_pool.bytes) return ctx->fail(PT_ERR_DEVICE, "internal: path pool carve overflow");
    PT_HIP(ctx->d_qa.alloc(n_paths * 4));
    PT_HIP(ctx->d_qb.alloc(n_paths * 4));
    PT_HIP(ctx->d_qnee.alloc(n_paths * 4));
    PT_HIP(ctx->d_qsorted.alloc(n_paths * 4));
    PT_HIP(ctx->d_qshadow.alloc(n_paths * 4));
    PT_HIP(ctx->d_qprobe.alloc(n_paths * 4));
    ctx->pool_paths = n_paths;
    return PT_OK;
}

pt_status ensure_traversal_scratch(pt_context* ctx) {
    // +1: the top entry is stored too; -1: the pooled-leaf kernels keep a sentinel in LDS slot 0
    uint32_t need = ctx->max_stack + 1 > PT_FS_SLOTS - 1 ? ctx->max_stack + 1 - (PT_FS_SLOTS - 1) : 0;
    size_t threads = (size_t)std::max(ctx->grid_trace, 2048) * PT_BLOCK;
    if (!ctx->d_spill.p || ctx->spill_depth < need) {
        PT_HIP(ctx->d_spill.alloc((size_t)std::max(need, 1u) * threads * 4 + PT_DIAG_WORDS * 4));
        PT_HIP(hipMemsetAsync(ctx->d_spill.p, 0, PT_DIAG_WORDS * 4, ctx->stream));
        ctx->spill_depth = need;
    }
    return PT_OK;
}

hipEvent_t get_event(pt_context* ctx, size_t i) {
    while (ctx->ev.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev.push_back(e);
    }
    return ctx->ev[i];
}

}  // namespace

extern "C" {

int pt_abi_version(void) { return PT_ABI_VERSION; }

pt_status pt_context_create(int device, pt_context** out) {
    if (!out) return PT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PT_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PT_ERR_NO_DEVICE;
    pt_context* ctx = new pt_context;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PT_ERR_DEVICE; }
    std::memset(&ctx->sc, 0, sizeof(ctx->sc));
    std::memset(&ctx->info, 0, sizeof(ctx->info));
    std::memset(&ctx->paths, 0, sizeof(ctx->paths));
    // occupancy-sized persistent grids: blocks per CU from register/LDS use x CU count
    ctx->grid_trace = ctx->n_cu * 4;
    ctx->grid_trace_dist = ctx->n_cu * ptk_trace_dist_blocks_per_cu();
    if (const char* e = std::getenv("PBRTGPU_TRACE_BLOCKS_PER_CU")) {
        ctx->grid_trace = ctx->n_cu * std::max(1, std::atoi(e));
        if (!ptk_trace_wide()) ctx->grid_trace_dist = ctx->grid_trace;          // (a PT_TRACE_WIDE build runs one 1024-thread block per CU: sixteen waves, all a CU holds)
    }
    ctx->grid_shade = ctx->n_cu * 2;
    if (const char* e = std::getenv("PBRTGPU_SHADE_BLOCKS_PER_CU")) ctx->grid_shade = ctx->n_cu * std::max(1, std::atoi(e));
    if (const char* e = std::getenv("PBRTGPU_SORT_SHADOW_MIN")) ctx->sort_shadow_min = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("PBRTGPU_SORT_CONT_MIN")) ctx->sort_cont_min = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("PBRTGPU_SORT_CONT")) ctx->sort_cont = std::min(2, std::max(-1, std::atoi(e)));
    if (const char* e = std::getenv("PBRTGPU_TRACE_FAR")) ctx->trace_far = std::atoi(e);
    if (const char* e = std::getenv("PBRTGPU_SHADE_LOCAL")) ctx->shade_local = std::atoi(e);
    ctx->nee_split = ptk_nee_split_default();
    if (const char* e = std::getenv("PBRTGPU_NEE_SPLIT")) ctx->nee_split = std::atoi(e);
    ctx->grid_wide = ctx->n_cu * 8;
    // Sobol' tables: $PBRTGPU_DATA_DIR, else <directory of this shared library>/../data
    const char* dd = std::getenv("PBRTGPU_DATA_DIR");
    if (dd) ctx->data_dir = dd;
    else {
        Dl_info di;
        if (dladdr((const void*)&pt_abi_version, &di) && di.dli_fname) {
            std::string so = di.dli_fname;
            size_t slash = so.find_last_of('/');
            ctx->data_dir = (slash == std::string::npos ? std::string(".") : so.substr(0, slash)) + "/../data";
        }
    }
    *out = ctx;
    return PT_OK;
}

void pt_context_destroy(pt_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->h_live) (void)hipHostFree(ctx->h_live);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* pt_last_error(const pt_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

pt_status pt_set_data_dir(pt_context* ctx, const char* dir) {
    if (!ctx || !dir) return PT_ERR_INVALID_ARGUMENT;
    ctx->data_dir = dir;
    return PT_OK;
}

namespace {
// trowbridge_reitz.rs:104-113 (host only: logf)
float roughness_to_alpha(float roughness) {
    roughness = roughness > 1e-3f ? roughness : 1e-3f;
    float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
// The three roughness parameters after the optional remap (plastic.rs:55-58, glass.rs:76-81, metal.rs:58-69, uber.rs:96-104, substrate.rs:49-54)
void material_alphas(const pt_material& in, float* a_r, float* a_u, float* a_v) {
    auto remap = [&](float r) { return in.remap_roughness ? roughness_to_alpha(r) : r; };
    *a_r = remap(in.roughness); *a_u = remap(in.uroughness); *a_v = remap(in.vroughness);
}
void build_lobes(const pt_material& in, PtMaterial& m) {
    std::memset(&m, 0, sizeof(m));
    float a_r, a_u, a_v;
    material_alphas(in, &a_r, &a_u, &a_v);
    ::build_lobes(in, a_r, a_u, a_v, m);
}

// PCG32 with the reference's default state (core/rng.rs:8-67) -- only the bounded draw that
// shuffle_array (core/sampling/sampling.rs:4-15) needs
struct Pcg32 {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t next() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((32u - rot) & 31u));
    }
    uint32_t below(uint32_t b) {
        uint32_t threshold = (0u - b) % b;
        for (;;) {
            uint32_t r = next();
            if (r >= threshold) return r % b;
        }
    }
};
int64_t floor_mod(int64_t a, int64_t b) { int64_t r = a % b; return r < 0 ? r + b : r; }
// x with a*x = 1 (mod n), by the extended Euclid recursion of halton.rs:31-44
int64_t mod_inverse(int64_t a, int64_t n) {
    int64_t r0 = a, r1 = n, x0 = 1, x1 = 0;
    while (r1 != 0) {
        int64_t q = r0 / r1, t = r0 - q * r1;
        r0 = r1; r1 = t;
        t = x0 - q * x1;
        x0 = x1; x1 = t;
    }
    return floor_mod(x0, n);
}
// HaltonSampler::new (samplers/halton.rs:57-112) + compute_radical_inverse_permutations (:12-20)
pt_status setup_halton(pt_context* ctx, PtSobol& sb, int32_t res_x, int32_t res_y, bool at_center) {
    constexpr uint32_t kDims = 1000;          // PRIMES / PRIME_SUMS hold 1000 entries (primes.rs)
    if (!ctx->d_hdims.p) {
        std::vector<uint32_t> primes;
        std::vector<uint8_t> composite(8000, 0);
        for (uint32_t i = 2; primes.size() < kDims; i++) {
            if (composite[i]) continue;
            primes.push_back(i);
            for (uint32_t j = i * i; j < composite.size(); j += i) composite[j] = 1;
        }
        std::vector<uint32_t> dims(4 * (size_t)kDims);
        std::vector<uint16_t> perms;
        Pcg32 rng;
        for (uint32_t i = 0; i < kDims; i++) {
            const uint32_t p = primes[i], off = (uint32_t)perms.size();
            for (uint32_t j = 0; j < p; j++) perms.push_back((uint16_t)j);
            for (uint32_t j = 0; j < p; j++) std::swap(perms[off + j], perms[off + j + rng.below(p - j)]);
            const uint64_t magic = ~0ull / p + 1;           // floor(2^64 / p) + 1 for p not a power of two; p = 2 is never divided this way
            dims[4 * i] = p; dims[4 * i + 1] = off; dims[4 * i + 2] = (uint32_t)magic; dims[4 * i + 3] = (uint32_t)(magic >> 32);
        }
        pt_status st;
        if ((st = upload(ctx, ctx->d_hdims, dims.data(), dims.size())) != PT_OK) return st;
        if ((st = upload(ctx, ctx->d_hperms, perms.data(), perms.size())) != PT_OK) return st;
    }
    sb.h_dims = ctx->d_hdims.as<uint4>();
    sb.h_perms = ctx->d_hperms.as<uint16_t>();
    sb.h_n_dims = kDims;

    sb.h_center = at_center ? 1u : 0u;
    const int32_t res[2] = {res_x, res_y}, bases[2] = {2, 3};
    int32_t scale[2], exp[2];
    for (int i = 0; i < 2; i++) {
        scale[i] = 1; exp[i] = 0;
        while (scale[i] < std::min(res[i], 128)) { scale[i] *= bases[i]; exp[i]++; }
    }
    const int32_t stride = scale[0] * scale[1];
    sb.h_exp[0] = (uint32_t)exp[0]; sb.h_exp[1] = (uint32_t)exp[1];
    sb.h_scale1 = (uint32_t)scale[1];
    sb.h_stride = (uint32_t)stride;
    sb.h_mul[0] = (uint32_t)((stride / scale[0]) * (int32_t)mod_inverse(scale[1], scale[0]));
    sb.h_mul[1] = (uint32_t)((stride / scale[1]) * (int32_t)mod_inverse(scale[0], scale[1]));
    return PT_OK;
}
// Sphere::new + world_bound (sphere.rs:18-59, transform.rs:134-182): device records and BVH build items
static void build_spheres(const pt_scene_desc* d, std::vector<PtSphere>& sph, std::vector<ptbvh::SpherePrim>& sprims) {
    sph.assign(d->n_spheres, PtSphere());
    sprims.assign(d->n_spheres, ptbvh::SpherePrim());
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const pt_sphere& in = d->spheres[i];
        PtSphere& s = sph[i];
        std::memset(&s, 0, sizeof(s));
        std::memcpy(s.o2w, in.object_to_world, 48);
        std::memcpy(s.w2o, in.world_to_object, 48);
        const float* m = in.object_to_world;
        const float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
        const bool swaps = det < 0.0f, reverse = (in.flags & PT_SPHERE_REVERSE_ORIENTATION) != 0;
        auto clampf = [](float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); };
        s.radius = in.radius;
        s.z_min = clampf(std::fmin(in.zmin, in.zmax), -in.radius, in.radius);
        s.z_max = clampf(std::fmax(s.z_min, in.zmax), -in.radius, in.radius);        // as written (sphere.rs:28)
        s.theta_min = std::acos(clampf(s.z_min / in.radius, -1.0f, 1.0f));
        s.theta_max = std::acos(clampf(s.z_max / in.radius, -1.0f, 1.0f));
        s.phi_max = clampf(in.phimax, 0.0f, 360.0f) * (3.14159265358979323846f / 180.0f);
        s.area = s.phi_max * s.radius * (s.z_max - s.z_min);
        s.flags = (reverse ? PT_SPH_REVERSE : 0u) | ((reverse ^ swaps) ? PT_SPH_FLIP : 0u);
        const float r = s.radius * 1.001f, diff = r - s.radius;
        const float lo[3] = {-r, -r, s.z_min - diff}, hi[3] = {r, r, s.z_max + diff};
        ptbvh::SpherePrim& sp = sprims[i];
        for (int c = 0; c < 8; c++) {
            const float x = (c & 4) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 1) ? hi[2] : lo[2];
            const float q[3] = {m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]};
            for (int k = 0; k < 3; k++) {
                sp.lo[k] = c == 0 ? q[k] : std::fmin(sp.lo[k], q[k]);
                sp.hi[k] = c == 0 ? q[k] : std::fmax(sp.hi[k], q[k]);
            }
        }
        sp.before_triangle = in.before_triangle;
        sp.flags = PT_TRI_SPHERE;
        if (in.material >= 0 && d->materials && d->materials[in.material].type != PT_MATERIAL_NONE) sp.flags |= (uint32_t)(in.material + 1) << PT_TRI_MATERIAL_SHIFT;
    }
}
}  // namespace

pt_status pt_scene_upload(pt_context* ctx, const pt_scene_desc* d) {
    if (!ctx || !d) return PT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    ctx->have_scene = false;
    // ---- validation (the kernels index these arrays unchecked)
    if (d->n_triangles == 0 && d->n_spheres == 0) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "scene has no primitives");
    if (d->n_instances > 0 && !d->instances) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "instances array missing");
    for (uint32_t i = 0; i < d->n_instances; i++) {
        const pt_instance& in = d->instances[i];
        if (in.before_triangle > d->n_triangles) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "instance before_triangle exceeds n_triangles");
        for (int k = 0; k < 2; k++) {
            const float* m = k ? in.world_to_instance : in.instance_to_world;
            if (m[12] != 0.0f || m[13] != 0.0f || m[14] != 0.0f || m[15] != 1.0f)
                return ctx->fail(PT_ERR_UNSUPPORTED, "object instance under a projective transform (last matrix row must be 0 0 0 1)");
        }
    }
    if (d->n_triangles > 0 && (!d->P || !d->indices || !d->tri_mesh || !d->meshes)) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "triangle arrays missing");
    if (d->n_spheres > 0 && !d->spheres) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "spheres array missing");
    if (d->xres <= 0 || d->yres <= 0 || d->spp <= 0 || d->max_depth < 0 || d->max_depth > 250) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "bad film / sampler / integrator parameters");
    if (d->integrator < PT_INTEGRATOR_PATH || d->integrator > PT_INTEGRATOR_WHITTED) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "unknown integrator");
    if (d->integrator == PT_INTEGRATOR_DIRECTLIGHTING && d->direct_strategy != PT_DIRECT_ALL && d->direct_strategy != PT_DIRECT_ONE)
        return ctx->fail(PT_ERR_INVALID_ARGUMENT, "unknown directlighting strategy");
    if ((d->integrator == PT_INTEGRATOR_DIRECTLIGHTING || d->integrator == PT_INTEGRATOR_WHITTED) && d->max_depth > 16)
        return ctx->fail(PT_ERR_UNSUPPORTED, "directlighting / whitted: maxdepth above 16 (one frame per level is kept per camera sample)");
    if (d->n_materials > 0 && !d->materials) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "materials array missing");
    if (d->n_area_lights > 0 && !d->area_lights) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "area lights array missing");
    for (uint32_t i = 0; i < d->n_area_lights; i++)
        if (d->area_lights[i].n_samples < 0 || d->area_lights[i].n_samples > 4096) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "area light nsamples outside [0, 4096]");
    if (d->sampler != PT_SAMPLER_SOBOL && d->sampler != PT_SAMPLER_HALTON)
        return ctx->fail(PT_ERR_UNSUPPORTED, "sampler not on the accelerated path: only the index-addressed samplers (sobol, halton) are reproducible on a wavefront");
    if (d->n_triangles >= 0x7fffffffu) return ctx->fail(PT_ERR_UNSUPPORTED, "too many triangles");
    for (uint32_t t = 0; t < d->n_triangles; t++) {
        for (int k = 0; k < 3; k++)
            if (d->indices[3 * (size_t)t + k] >= d->n_vertices) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "vertex index out of range");
        if (d->tri_mesh[t] >= d->n_meshes) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "mesh index out of range");
    }
    for (uint32_t m = 0; m < d->n_meshes; m++) {
        if (d->meshes[m].material >= (int32_t)d->n_materials) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "material index out of range");
        if (d->meshes[m].material >= 65535) return ctx->fail(PT_ERR_UNSUPPORTED, "more than 65534 materials");
        if (d->meshes[m].area_light >= (int32_t)d->n_area_lights) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "area light index out of range");

        int32_t mi = d->meshes[m].material;
        if (mi >= 0 && (d->materials[mi].type < PT_MATERIAL_NONE || d->materials[mi].type > PT_MATERIAL_SUBSTRATE))
            return ctx->fail(PT_ERR_UNSUPPORTED, "material type not on the accelerated path");
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const pt_material& m = d->materials[i];
        const uint32_t refs[13] = {m.tex_kd, m.tex_ks, m.tex_kr, m.tex_kt, m.tex_opacity, m.tex_sigma, m.tex_metal_eta, m.tex_metal_k, m.tex_bump,
                                   m.tex_roughness, m.tex_uroughness, m.tex_vroughness, m.tex_eta};
        for (uint32_t r : refs)
            if (r > d->n_textures || (r && !d->textures)) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "material texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const pt_texture& t = d->textures[i];
        if (t.type < PT_TEX_CONSTANT || t.type > PT_TEX_IMAGEMAP) return ctx->fail(PT_ERR_UNSUPPORTED, "texture class not on the accelerated path");
        if (t.mapping < PT_MAPPING_UV || t.mapping > PT_MAPPING_PLANAR) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "unknown texture mapping");
        for (int k = 0; k < 3; k++)
            if (t.tex[k] >= (int32_t)i) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "texture child index must be smaller than the texture's own (definition order)");
        if (t.type == PT_TEX_IMAGEMAP) {
            if (t.image < 0 || (uint32_t)t.image >= d->n_images || !d->images) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "imagemap image index out of range");
            if (t.swrap < PT_WRAP_REPEAT || t.swrap > PT_WRAP_CLAMP || t.twrap < PT_WRAP_REPEAT || t.twrap > PT_WRAP_CLAMP) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "unknown image wrap mode");
        }
    }
    for (uint32_t i = 0; i < d->n_images; i++) {
        const pt_image& im = d->images[i];
        auto pow2 = [](uint32_t v) { return v != 0 && (v & (v - 1)) == 0; };
        uint32_t lv = 1;
        for (uint32_t m = std::max(im.width, im.height); m > 1; m >>= 1) lv++;
        if (!pow2(im.width) || !pow2(im.height) || (im.channels != 1 && im.channels != 3) || !im.texels || im.n_levels != lv || lv > PT_MAX_MIP_LEVELS)
            return ctx->fail(PT_ERR_INVALID_ARGUMENT, "image pyramid: level 0 must be power-of-two sized, 1 or 3 channels, with 1 + log2(max(w, h)) levels");
    }
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        const pt_sphere& sp = d->spheres[i];
        if (sp.material >= (int32_t)d->n_materials || sp.material >= 65535) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "sphere material index out of range");
        if (sp.area_light >= (int32_t)d->n_area_lights) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "sphere area light index out of range");
        if (sp.material >= 0 && (d->materials[sp.material].type < PT_MATERIAL_NONE || d->materials[sp.material].type > PT_MATERIAL_SUBSTRATE))
            return ctx->fail(PT_ERR_UNSUPPORTED, "material type not on the accelerated path");
        if (sp.before_triangle > d->n_triangles) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "sphere before_triangle exceeds n_triangles");
        if (!(sp.radius > 0.0f)) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "sphere radius must be positive");
        for (int k = 0; k < 2; k++) {
            const float* m = k ? sp.world_to_object : sp.object_to_world;
            if (m[12] != 0.0f || m[13] != 0.0f || m[14] != 0.0f || m[15] != 1.0f)
                return ctx->fail(PT_ERR_UNSUPPORTED, "sphere under a projective transform (last matrix row must be 0 0 0 1)");
        }
    }
    // integrator-specific refusals come after the index range checks above (they dereference materials[])
    if (d->integrator == PT_INTEGRATOR_PATH && d->sampler == PT_SAMPLER_HALTON && d->max_depth > 124)
        return ctx->fail(PT_ERR_UNSUPPORTED, "path with the Halton sampler and maxdepth above 124: a path that long asks for more than the sampler's 1000 "
                                             "dimensions (5 for the camera, 8 per vertex), where the reference panics (halton.rs:103-108)");
    if (d->integrator == PT_INTEGRATOR_AO) {
        if (d->ao_samples < 1 || d->ao_samples > 4096) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "ao: nsamples outside [1, 4096]");
        for (uint32_t i = 0; i < d->n_meshes; i++)
            if (d->meshes[i].material < 0 || d->materials[d->meshes[i].material].type == PT_MATERIAL_NONE)
                return ctx->fail(PT_ERR_UNSUPPORTED, "ao: a surface without a material makes the reference request its sample array twice and panic (ao.rs:66-69, :78)");
        for (uint32_t i = 0; i < d->n_spheres; i++)
            if (d->spheres[i].material < 0 || d->materials[d->spheres[i].material].type == PT_MATERIAL_NONE)
                return ctx->fail(PT_ERR_UNSUPPORTED, "ao: a surface without a material makes the reference request its sample array twice and panic (ao.rs:66-69, :78)");
        for (uint32_t i = 0; i < d->n_materials; i++)
            if (d->materials[i].tex_bump) return ctx->fail(PT_ERR_UNSUPPORTED, "ao: bump-mapped materials (the bump can flip the frame's normal) are not on the accelerated path");
    }
    if (ctx->sobol_m32.empty() && !load_sobol(ctx)) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "cannot read sobol_tables.bin from data dir '" + ctx->data_dir + "'");

    double t0 = now_ms();
    const bool build_trace = std::getenv("PBRTGPU_BUILD_TRACE") != nullptr;
    double tm_last = t0;
    auto mark = [&](const char* what) { if (build_trace) { const double t = now_ms(); std::fprintf(stderr, "[upload] %s %.1f ms\n", what, t - tm_last); tm_last = t; } };
    // What either path below leaves for the rest of the upload.
    std::vector<PtSphere> sph;
    std::vector<PtInstance> dinst;
    std::vector<PtLight> lights;
    std::vector<PtTriInfo> tinfo;                   // host path: per leaf record (sphere / instance entries unused)
    std::vector<uint32_t> tri_flags;
    ptbvh::Result& bvh = ctx->host_bvh;             // host path: the build's output arrays live in the context between uploads (cleared, capacity retained) while they stay below 1 GB
    struct TrimBvh { pt_context* c; ~TrimBvh() { if (c->host_bvh.nodes.capacity() * sizeof(PtNode) > ((size_t)1 << 30)) c->host_bvh = ptbvh::Result(); } } trim_bvh{ctx};
    size_t n_world_nodes = 0;
    uint32_t n_top = 0, up_root_ref = PT_EMPTY_REF, up_max_leaf = 0, up_n_leaves = 0;
    size_t up_n_nodes = 0, up_n_tris = 0;
    float up_root_lo[3] = {0, 0, 0}, up_root_hi[3] = {0, 0, 0};
    bool any_one_sided = false;
    double t1 = t0;
    ptbvh::DeviceBuild dev_build{ctx->stream, ctx->bvh_build_where, false, hipSuccess};
    const int max_node_prims = d->max_node_prims > 0 ? d->max_node_prims : 4;

    // ---- BVH, records and nodes on the device --------------------------------
    // A world list of triangles only (no spheres, objects or instances) under the SAH or HLBVH split -- BASELINE's scenes -- never leaves the
    // device: vertices and indices go up, pt_sah.hip / pt_hlbvh.hip build the binary tree, pt_sah.hip writes the leaf and shading records in leaf order, collapses
    // the tree into the 4-wide node array in the reference's depth-first numbering and finishes it (breadth-first top, order tables, axis
    // bits, inverted empty slots).  Same arrays as the host path below, byte for byte (tests/test_gpu_hlbvh.py compares the digests).
    // A list that needs a fallback split, PT_BVH_BUILD_HOST or PBRTGPU_HOST_FINISH=1 take the host path.
    bool fast = false;
    {
        uint32_t any_object = 0;
        for (uint32_t i = 0; i < d->n_meshes; i++) any_object |= d->meshes[i].object;
        for (uint32_t i = 0; i < d->n_spheres; i++) any_object |= d->spheres[i].object;
        const int mnp = std::min(std::max(max_node_prims, 0), 255);
        const bool sah = d->split_method == PT_SPLIT_SAH && mnp >= 2, hl = d->split_method == PT_SPLIT_HLBVH && mnp >= 1;
        const uint64_t n_world = (uint64_t)d->n_triangles + d->n_spheres;
        // (round 4: analytic spheres of the world list ride along -- a handful of them among the triangles, killeroo-simple's two sphere lights --;
        // objects and instances still take the host path)
        if (d->n_instances == 0 && any_object == 0 && (sah || hl) && n_world >= 2 &&
            n_world < PT_LEAF_FIRST_MASK - 16u && ctx->bvh_build_where != PT_BVH_BUILD_HOST &&
            (ctx->bvh_build_where == PT_BVH_BUILD_DEVICE || n_world >= ptbvh::kDeviceMinPrims) && !std::getenv("PBRTGPU_HOST_FINISH")) {
            std::vector<uint32_t> m_triflags(d->n_meshes), m_flags(d->n_meshes);
            std::vector<int32_t> m_material(d->n_meshes);
            for (uint32_t i = 0; i < d->n_meshes; i++) {
                uint32_t mf = d->meshes[i].flags, f = 0;
                if (!(mf & PT_MESH_TWO_SIDED)) f |= PT_TRI_ONE_SIDED;
                if (((mf & PT_MESH_REVERSE_ORIENTATION) != 0) ^ ((mf & PT_MESH_SWAPS_HANDEDNESS) != 0)) f |= PT_TRI_FLIP;
                if (((mf & PT_MESH_HAS_N) && d->N) || ((mf & PT_MESH_HAS_S) && d->S) || ((mf & PT_MESH_HAS_UV) && d->UV)) f |= PT_TRI_HAS_ATTR;
                const int32_t mat = d->meshes[i].material;
                if (mat >= 0 && d->materials[mat].type != PT_MATERIAL_NONE) f |= (uint32_t)(mat + 1) << PT_TRI_MATERIAL_SHIFT;
                m_triflags[i] = f;
                if (!d->N) mf &= ~PT_MESH_HAS_N;
                if (!d->S) mf &= ~PT_MESH_HAS_S;
                if (!d->UV) mf &= ~PT_MESH_HAS_UV;
                m_flags[i] = mf;
                m_material[i] = mat;
            }
            // the world list's spheres in list order (spliced in before triangle `before_triangle`, ties in creation order: make_list below)
            std::vector<ptbvh::SpherePrim> sprims;
            std::vector<uint32_t> s_order(d->n_spheres), s_prim(d->n_spheres), s_rec(4 * (size_t)d->n_spheres);
            std::vector<float> s_bounds(6 * (size_t)d->n_spheres);
            if (d->n_spheres) {
                build_spheres(d, sph, sprims);
                for (uint32_t i = 0; i < d->n_spheres; i++) s_order[i] = i;
                std::stable_sort(s_order.begin(), s_order.end(), [&](uint32_t x, uint32_t y) {
                    return d->spheres[x].before_triangle != d->spheres[y].before_triangle ? d->spheres[x].before_triangle < d->spheres[y].before_triangle
                                                                                         : d->spheres[x].order < d->spheres[y].order;
                });
                for (uint32_t j = 0; j < d->n_spheres; j++) {
                    const uint32_t i = s_order[j];
                    s_prim[j] = d->spheres[i].before_triangle + j;          // the triangles before it and the j spheres listed before it
                    std::memcpy(&s_bounds[6 * (size_t)j], sprims[i].lo, 12); std::memcpy(&s_bounds[6 * (size_t)j + 3], sprims[i].hi, 12);
                    s_rec[4 * (size_t)j] = i; s_rec[4 * (size_t)j + 1] = sprims[i].flags; s_rec[4 * (size_t)j + 2] = (uint32_t)d->spheres[i].material; s_rec[4 * (size_t)j + 3] = 0;
                }
            }
            ptbvh::SceneIn in{d->P, d->n_vertices, d->indices, d->tri_mesh, d->n_triangles, m_triflags.data(), m_material.data(), m_flags.data(), d->n_meshes};
            in.sph_prim = s_prim.data(); in.sph_bounds = s_bounds.data(); in.sph_rec = s_rec.data(); in.n_spheres = d->n_spheres;
            ptbvh::SceneOut out;
            hipError_t herr = hipSuccess;
            const int rc = sah ? ptbvh::device_sah_scene(ctx->stream, in, (uint32_t)mnp, &out, &herr) : ptbvh::device_hlbvh_scene(ctx->stream, in, (uint32_t)mnp, &out, &herr);
            if (rc == -2) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "hlbvh: all treelet centroids coincide along the split axis (the reference panics on this input)");
            if (rc < 0) return ctx->hip_fail(herr, "scene build on the device");
            mark("device scene (bounds, SAH, records, collapse)");
            if (rc == 0) {
                // light i = the i-th emissive primitive in primitive order (scene_context.rs:1218-1231); `lit` holds primitive numbers of the merged list,
                // `lit_src` what each stands for (bit 31: a sphere, else the triangle)
                std::vector<uint32_t> lit, lit_src;
                bool any_light_mesh = false;
                for (uint32_t i = 0; i < d->n_meshes; i++) any_light_mesh |= d->meshes[i].area_light >= 0;
                if (any_light_mesh) {                     // the host's threads scan their share of the triangles; the shares are joined in order
                    const size_t n_chunks = 64, step = ((size_t)d->n_triangles + n_chunks - 1) / n_chunks;
                    std::vector<std::vector<uint32_t>> part(n_chunks);
                    ptbvh::parallel_tasks(n_chunks, [&](size_t c) {
                        const size_t a = c * step, b = std::min((size_t)d->n_triangles, a + step);
                        for (size_t t = a; t < b; t++) if (d->meshes[d->tri_mesh[t]].area_light >= 0) part[c].push_back((uint32_t)t);
                    });
                    for (const auto& v : part) lit_src.insert(lit_src.end(), v.begin(), v.end());
                }
                if (d->n_spheres) {                       // triangle numbers -> primitive numbers, the emissive spheres merged in at their places
                    std::vector<uint32_t> tri_lit;
                    tri_lit.swap(lit_src);
                    size_t ti = 0;
                    uint32_t j = 0;                       // spheres listed so far
                    auto flush_tris = [&](uint32_t before) {          // emissive triangles t < before: primitive t + j
                        while (ti < tri_lit.size() && tri_lit[ti] < before) { lit.push_back(tri_lit[ti] + j); lit_src.push_back(tri_lit[ti]); ti++; }
                    };
                    for (; j < d->n_spheres; j++) {
                        const uint32_t i = s_order[j];
                        flush_tris(d->spheres[i].before_triangle);
                        if (d->spheres[i].area_light >= 0) { lit.push_back(s_prim[j]); lit_src.push_back(0x80000000u | i); }
                    }
                    flush_tris(0xffffffffu);
                } else {
                    lit = lit_src;
                }
                if (lit.size() >= (1u << 24)) { out.free_all(); return ctx->fail(PT_ERR_UNSUPPORTED, "more than 2^24 emissive primitives"); }
                std::vector<uint32_t> lit_rec(lit.size());
                if (ptbvh::device_scene_lights(ctx->stream, &out, lit.data(), (uint32_t)lit.size(), lit_rec.data(), &herr) < 0) { out.free_all(); return ctx->hip_fail(herr, "light records on the device"); }
                for (size_t i = 0; i < lit.size(); i++) {
                    PtLight L;
                    std::memset(&L, 0, sizeof(L));
                    if (lit_src[i] & 0x80000000u) {       // a sphere: the host path's record for one (below)
                        const uint32_t si = lit_src[i] & 0x7fffffffu;
                        const pt_area_light& al = d->area_lights[d->spheres[si].area_light];
                        std::memcpy(&L.p0[0], &si, 4);
                        L.area = sph[si].area;
                        L.mesh_flags = PT_LIGHT_SPHERE;
                        L.two_sided = al.two_sided;
                        L.n_samples = (uint32_t)std::max(1, al.n_samples);
                        std::memcpy(L.L, al.L, 12);
                        L.tri_rec = lit_rec[i];
                        L.prim = lit[i];
                        lights.push_back(L);
                        continue;
                    }
                    const uint32_t t = lit_src[i];
                    const pt_mesh& m = d->meshes[d->tri_mesh[t]];
                    const pt_area_light& al = d->area_lights[m.area_light];
                    const uint32_t v0 = d->indices[3 * (size_t)t], v1 = d->indices[3 * (size_t)t + 1], v2 = d->indices[3 * (size_t)t + 2];
                    const float* p0 = d->P + 3 * (size_t)v0; const float* p1 = d->P + 3 * (size_t)v1; const float* p2 = d->P + 3 * (size_t)v2;
                    std::memcpy(L.p0, p0, 12); std::memcpy(L.p1, p1, 12); std::memcpy(L.p2, p2, 12);
                    // Triangle::area (triangle.rs:579-588)
                    float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
                    float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
                    float cx = (ay * bz) - (az * by), cy = (az * bx) - (ax * bz), cz = (ax * by) - (ay * bx);
                    L.area = 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz);
                    L.mesh_flags = m_flags[d->tri_mesh[t]];
                    L.two_sided = al.two_sided;
                    L.n_samples = (uint32_t)std::max(1, al.n_samples);
                    std::memcpy(L.L, al.L, 12);
                    L.tri_rec = lit_rec[i];
                    L.prim = lit[i];
                    if (L.mesh_flags & PT_MESH_HAS_N) { std::memcpy(L.n0, d->N + 3 * (size_t)v0, 12); std::memcpy(L.n1, d->N + 3 * (size_t)v1, 12); std::memcpy(L.n2, d->N + 3 * (size_t)v2, 12); }
                    lights.push_back(L);
                }
                (void)hipFree(out.d_rec_of_prim);
                out.d_rec_of_prim = nullptr;
                // the context's buffers take the blocks over
                ctx->d_nodes.release(); ctx->d_tris.release(); ctx->d_tri_info.release();
                ctx->d_nodes.p = out.d_nodes; ctx->d_nodes.bytes = (size_t)out.n_nodes4 * sizeof(PtNode);
                ctx->d_tris.p = out.d_tris; ctx->d_tris.bytes = ((size_t)n_world + 1) * sizeof(PtTri);
                ctx->d_tri_info.p = out.d_tinfo; ctx->d_tri_info.bytes = (size_t)n_world * sizeof(PtTriInfo);
                up_n_nodes = out.n_nodes4; up_n_tris = (size_t)n_world + 1;
                n_world_nodes = out.n_nodes4; n_top = out.n_top;
                up_root_ref = 0; up_max_leaf = out.max_leaf; up_n_leaves = out.n_leaves;
                std::memcpy(up_root_lo, out.root_lo, 12); std::memcpy(up_root_hi, out.root_hi, 12);
                any_one_sided = out.any_one_sided != 0;
                ctx->max_stack = 3u * out.max_depth4 + 2u;
                dev_build.used = true;
                fast = true;
                t1 = now_ms();
                mark("lights + hand-over");
            }
        }
    }

    if (!fast) {
    // ---- BVH (host) ---------------------------------------------------------
    tri_flags.resize(d->n_triangles);
    ptbvh::parallel_for(d->n_triangles, [&](size_t t0_, size_t t1_) {
    for (size_t t = t0_; t < t1_; t++) {
        uint32_t mf = d->meshes[d->tri_mesh[t]].flags, f = 0;
        if (!(mf & PT_MESH_TWO_SIDED)) f |= PT_TRI_ONE_SIDED;
        if (((mf & PT_MESH_REVERSE_ORIENTATION) != 0) ^ ((mf & PT_MESH_SWAPS_HANDEDNESS) != 0)) f |= PT_TRI_FLIP;
        if (((mf & PT_MESH_HAS_N) && d->N) || ((mf & PT_MESH_HAS_S) && d->S) || ((mf & PT_MESH_HAS_UV) && d->UV)) f |= PT_TRI_HAS_ATTR;
        int32_t mat = d->meshes[d->tri_mesh[t]].material;
        if (mat >= 0 && d->materials[mat].type != PT_MATERIAL_NONE) f |= (uint32_t)(mat + 1) << PT_TRI_MATERIAL_SHIFT;
        tri_flags[t] = f;
    }
    });
    mark("triangle flags");
    std::vector<ptbvh::SpherePrim> sprims;
    build_spheres(d, sph, sprims);
    // Primitive lists (render_options.primitives and each object's list, scene_context.rs:1301-1316): triangles in array order with
    // the spheres / instances of the list spliced in before triangle `before_triangle`, ties in creation order.
    struct Entry { uint32_t kind, idx; };          // 0 triangle, 1 sphere, 2 instance
    uint32_t n_objects = 0;
    for (uint32_t i = 0; i < d->n_meshes; i++) n_objects = std::max(n_objects, d->meshes[i].object);
    for (uint32_t i = 0; i < d->n_spheres; i++) n_objects = std::max(n_objects, d->spheres[i].object);
    auto make_list = [&](uint32_t tag) {
        struct Extra { uint32_t before, order, kind, idx; };
        std::vector<Extra> extra;
        for (uint32_t i = 0; i < d->n_spheres; i++) if (d->spheres[i].object == tag) extra.push_back({d->spheres[i].before_triangle, d->spheres[i].order, 1u, i});
        if (tag == 0) for (uint32_t i = 0; i < d->n_instances; i++) extra.push_back({d->instances[i].before_triangle, d->instances[i].order, 2u, i});
        std::stable_sort(extra.begin(), extra.end(), [](const Extra& x, const Extra& y) { return x.before != y.before ? x.before < y.before : x.order < y.order; });
        std::vector<Entry> out;
        if (extra.empty() && n_objects == 0) {            // triangles only, all in the world: the list is the triangle array
            out.resize(d->n_triangles);
            ptbvh::parallel_for(d->n_triangles, [&](size_t a, size_t b) { for (size_t t = a; t < b; t++) out[t] = {0u, (uint32_t)t}; });
            return out;
        }
        if (extra.empty()) {              // triangles only: the list's members counted and written by the host's threads, chunk by chunk, in triangle order
            const size_t n_chunks = 64, step = ((size_t)d->n_triangles + n_chunks - 1) / n_chunks;
            std::vector<size_t> cnt_c(n_chunks + 1, 0);
            ptbvh::parallel_tasks(n_chunks, [&](size_t c) {
                const size_t a = std::min((size_t)d->n_triangles, c * step), b = std::min((size_t)d->n_triangles, a + step);
                size_t k = 0;
                for (size_t t = a; t < b; t++) k += d->meshes[d->tri_mesh[t]].object == tag;
                cnt_c[c + 1] = k;
            });
            for (size_t c = 0; c < n_chunks; c++) cnt_c[c + 1] += cnt_c[c];
            out.resize(cnt_c[n_chunks]);
            ptbvh::parallel_tasks(n_chunks, [&](size_t c) {
                const size_t a = std::min((size_t)d->n_triangles, c * step), b = std::min((size_t)d->n_triangles, a + step);
                size_t k = cnt_c[c];
                for (size_t t = a; t < b; t++) if (d->meshes[d->tri_mesh[t]].object == tag) out[k++] = {0u, (uint32_t)t};
            });
            return out;
        }
        size_t e = 0;
        for (uint32_t t = 0; t <= d->n_triangles; t++) {
            while (e < extra.size() && extra[e].before <= t) { out.push_back({extra[e].kind, extra[e].idx}); e++; }
            if (t < d->n_triangles && d->meshes[d->tri_mesh[t]].object == tag) out.push_back({0u, t});
        }
        return out;
    };
    const char* hlbvh_msg = "hlbvh: all treelet centroids coincide along the split axis (the reference panics on this input)";
    // objects first: an instance's bound is its object's root bound under the instance transform
    struct ObjectBvh { std::vector<Entry> list; ptbvh::Result res; float lo[3], hi[3]; bool direct = false; };
    std::vector<ObjectBvh> objs(n_objects);
    auto fill_prim = [&](const Entry& en, ptbvh::Prim* pr) {
        if (en.kind == 0) ptbvh::triangle_prim(d->P, d->indices, en.idx, tri_flags[en.idx], pr);
        else {
            std::memcpy(pr->lo, sprims[en.idx].lo, 12); std::memcpy(pr->hi, sprims[en.idx].hi, 12);
            ptbvh::sphere_record(en.idx, sprims[en.idx].flags, &pr->rec);
        }
    };
    for (uint32_t k = 0; k < n_objects; k++) {
        ObjectBvh& ob = objs[k];
        ob.list = make_list(k + 1);
        // (the context's primitive buffer, shared with the world list below: uninitialised storage that survives from upload to upload)
        if (ctx->host_prims_cap < ob.list.size()) {
            ctx->host_prims.reset();
            ctx->host_prims.reset(new ptbvh::Prim[ob.list.size()]);
            ctx->host_prims_cap = ob.list.size();
        }
        struct { ptbvh::Prim* p; size_t n; ptbvh::Prim* data() const { return p; } size_t size() const { return n; } bool empty() const { return n == 0; }
                 ptbvh::Prim& operator[](size_t i) const { return p[i]; } } prims{ctx->host_prims.get(), ob.list.size()};
        ptbvh::parallel_for(ob.list.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) fill_prim(ob.list[i], &prims[i]); });
        if (prims.size() == 1) {             // a single primitive is wrapped without an accelerator (scene_context.rs:1370-1377)
            ob.direct = true;
            ob.res.tris.assign(1, prims[0].rec);
            ob.res.tris[0].prim = 0; ob.res.tris[0].flags |= PT_TRI_LAST; ob.res.tris[0].light1 = 0;
            ob.res.max_stack = 0; ob.res.max_leaf = 1;
            std::memcpy(ob.lo, prims[0].lo, 12); std::memcpy(ob.hi, prims[0].hi, 12);
        } else if (!prims.empty()) {
            if (!ptbvh::build_prims(prims.data(), (uint32_t)prims.size(), d->split_method, max_node_prims, &ob.res, &dev_build))
                return dev_build.err != hipSuccess ? ctx->hip_fail(dev_build.err, "HLBVH build on the device") : ctx->fail(PT_ERR_INVALID_ARGUMENT, hlbvh_msg);
            ob.res.tris.pop_back();          // the pad record: one for the whole array, below
            std::memcpy(ob.lo, ob.res.root_lo, 12); std::memcpy(ob.hi, ob.res.root_hi, 12);
        }
    }
    for (uint32_t i = 0; i < d->n_instances; i++)
        if (d->instances[i].object >= n_objects || objs[d->instances[i].object].list.empty()) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "instance of an unknown or empty object");
    dinst.resize(d->n_instances);
    mark("object trees");
    std::vector<Entry> world = make_list(0);
    mark("world primitive list");
    if (world.empty()) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "scene has no world primitives (objects are only rendered through ObjectInstance)");
    if ((uint64_t)world.size() >= PT_LEAF_FIRST_MASK - 16u) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "more than 2^26 primitives");
    {
        // the world's primitive list: uninitialised storage, first touched by the threads that fill it, kept by the context for the next
        // upload (allocating and returning 72 MB per million triangles costs more than the device-side build)
        if (ctx->host_prims_cap < world.size()) {
            ctx->host_prims.reset();
            ctx->host_prims.reset(new ptbvh::Prim[world.size()]);
            ctx->host_prims_cap = world.size();
        }
        struct PrimsView { ptbvh::Prim* p; size_t n; ptbvh::Prim* data() const { return p; } size_t size() const { return n; } ptbvh::Prim& operator[](size_t i) const { return p[i]; } };
        PrimsView prims{ctx->host_prims.get(), world.size()};
        ptbvh::parallel_for(world.size(), [&](size_t i0, size_t i1) {
            for (size_t i = i0; i < i1; i++) if (world[i].kind != 2) fill_prim(world[i], &prims[i]);
        });
        for (size_t i = 0; d->n_instances && i < world.size(); i++) {
            const Entry& en = world[i];
            if (en.kind != 2) continue;
            const pt_instance& in = d->instances[en.idx];
            const ObjectBvh& ob = objs[in.object];
            const float* m = in.instance_to_world;          // motion_bounds of a static transform = transform_bounds (transform.rs:134-182)
            for (int c = 0; c < 8; c++) {
                const float x = (c & 4) ? ob.hi[0] : ob.lo[0], y = (c & 2) ? ob.hi[1] : ob.lo[1], z = (c & 1) ? ob.hi[2] : ob.lo[2];
                float q[3] = {m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]};
                const float wq = m[12] * x + m[13] * y + m[14] * z + m[15];
                if (wq != 1.0f) { q[0] /= wq; q[1] /= wq; q[2] /= wq; }
                for (int a2 = 0; a2 < 3; a2++) {
                    prims[i].lo[a2] = c == 0 ? q[a2] : std::fmin(prims[i].lo[a2], q[a2]);
                    prims[i].hi[a2] = c == 0 ? q[a2] : std::fmax(prims[i].hi[a2], q[a2]);
                }
            }
            ptbvh::instance_record(en.idx, &prims[i].rec);
        }
        mark("world primitives (bounds + records)");
        if (!ptbvh::build_prims(prims.data(), (uint32_t)prims.size(), d->split_method, max_node_prims, &bvh, &dev_build))
            return dev_build.err != hipSuccess ? ctx->hip_fail(dev_build.err, "HLBVH build on the device") : ctx->fail(PT_ERR_INVALID_ARGUMENT, hlbvh_msg);
        mark("build_prims");
    }
    if (ctx->host_prims_cap > ((size_t)4 << 20)) { ctx->host_prims.reset(); ctx->host_prims_cap = 0; }
    mark("free primitive list");
    n_world_nodes = bvh.nodes.size();          // the objects' trees are appended behind these
    if (d->n_instances) for (size_t prim = 0; prim < world.size(); prim++) if (world[prim].kind == 2) dinst[world[prim].idx].world_prim = (uint32_t)prim;
    // one node array and one record array: the world first, then each object with its references shifted
    std::vector<Entry> rec_entry(bvh.tris.size() - 1);        // record -> what it stands for (shading records below)
    ptbvh::parallel_for(world.size(), [&](size_t a, size_t b) { for (size_t prim = a; prim < b; prim++) rec_entry[bvh.rec_of_prim[prim]] = world[prim]; });
    uint32_t max_inner_stack = 0;
    if (n_objects) {
        bvh.tris.pop_back();
        std::vector<uint32_t> node_off(n_objects), rec_off(n_objects);
        for (uint32_t k = 0; k < n_objects; k++) {
            ObjectBvh& ob = objs[k];
            node_off[k] = (uint32_t)bvh.nodes.size(); rec_off[k] = (uint32_t)bvh.tris.size();
            if ((uint64_t)rec_off[k] + ob.res.tris.size() >= PT_LEAF_FIRST_MASK - 16u) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "more than 2^26 primitives");
            auto shift = [&](uint32_t ref) {
                if (ref == PT_EMPTY_REF) return ref;
                return (ref & PT_LEAF_BIT) ? ((ref & ~PT_LEAF_FIRST_MASK) | ((ref & PT_LEAF_FIRST_MASK) + rec_off[k])) : ref + node_off[k];
            };
            bvh.nodes.resize((size_t)node_off[k] + ob.res.nodes.size());
            ptbvh::parallel_for(ob.res.nodes.size(), [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) { PtNode nd = ob.res.nodes[i]; for (int c = 0; c < 4; c++) nd.child[c] = shift(nd.child[c]); bvh.nodes[(size_t)node_off[k] + i] = nd; }
            });
            bvh.tris.resize((size_t)rec_off[k] + ob.res.tris.size());
            rec_entry.resize((size_t)rec_off[k] + ob.res.tris.size());
            ptbvh::parallel_for(ob.res.tris.size(), [&](size_t a, size_t b) {
                for (size_t r = a; r < b; r++) {
                    bvh.tris[(size_t)rec_off[k] + r] = ob.res.tris[r];
                    rec_entry[(size_t)rec_off[k] + r] = ob.direct ? ob.list[0] : ob.list[ob.res.tris[r].prim];
                }
            });
            ob.res.root_ref = ob.direct ? rec_off[k] : shift(ob.res.root_ref);
            max_inner_stack = std::max(max_inner_stack, ob.res.max_stack);
            bvh.max_leaf = std::max(bvh.max_leaf, ob.res.max_leaf);
        }
        PtTri pad;
        std::memset(&pad, 0, sizeof(pad));
        pad.flags = PT_TRI_LAST;
        bvh.tris.push_back(pad);
        for (uint32_t i = 0; i < d->n_instances; i++) {
            const pt_instance& in = d->instances[i];
            const ObjectBvh& ob = objs[in.object];
            PtInstance& o = dinst[i];
            const uint32_t wp = o.world_prim;
            std::memset(&o, 0, sizeof(o));
            o.world_prim = wp;
            std::memcpy(o.m, in.instance_to_world, 48); std::memcpy(o.minv, in.world_to_instance, 48);
            o.root_ref = ob.res.root_ref; o.direct = ob.direct ? 1u : 0u;
            std::memcpy(o.root_lo, ob.lo, 12); std::memcpy(o.root_hi, ob.hi, 12);
        }
    }
    mark("record map + objects");
    t1 = now_ms();
    ctx->max_stack = bvh.max_stack + max_inner_stack + (d->n_instances ? 4u : 0u);      // (+ what a ray leaves behind when it enters an object: the rest of its leaf, its slab interval, the marker)

    // ---- shading records ----------------------------------------------------
    tinfo.resize(bvh.tris.size() - 1);
    ptbvh::parallel_for(tinfo.size(), [&](size_t r0_, size_t r1_) {
    for (size_t r = r0_; r < r1_; r++) {
        PtTriInfo& ti = tinfo[r];
        std::memset(&ti, 0, sizeof(ti));
        ti.light = -1; ti.material = -1;
        const Entry& en = rec_entry[r];
        if (en.kind == 1) ti.material = d->spheres[en.idx].material;
        if (en.kind != 0) continue;
        const uint32_t t = en.idx;
        const pt_mesh& m = d->meshes[d->tri_mesh[t]];
        ti.v[0] = d->indices[3 * (size_t)t]; ti.v[1] = d->indices[3 * (size_t)t + 1]; ti.v[2] = d->indices[3 * (size_t)t + 2];
        ti.mesh = d->tri_mesh[t];
        ti.material = m.material;
        uint32_t mf = m.flags;
        if (!d->N) mf &= ~PT_MESH_HAS_N;
        if (!d->S) mf &= ~PT_MESH_HAS_S;
        if (!d->UV) mf &= ~PT_MESH_HAS_UV;
        ti.mesh_flags = mf;
    }
    });
    // one DiffuseAreaLight per emissive world primitive, in primitive order (scene_context.rs:1218-1231); lights inside objects are
    // dropped as in the reference (:1302-1304), instances carry none (TransformedPrimitive::get_area_light)
    for (uint32_t prim = 0; prim < world.size(); prim++) {
        const Entry& en = world[prim];
        const uint32_t rec = bvh.rec_of_prim[prim];
        if (en.kind == 1 && d->spheres[en.idx].area_light >= 0) {
            const pt_area_light& al = d->area_lights[d->spheres[en.idx].area_light];
            PtLight L;
            std::memset(&L, 0, sizeof(L));
            std::memcpy(&L.p0[0], &en.idx, 4);
            L.area = sph[en.idx].area;
            L.mesh_flags = PT_LIGHT_SPHERE;
            L.two_sided = al.two_sided;
            L.n_samples = (uint32_t)std::max(1, al.n_samples);
            std::memcpy(L.L, al.L, 12);
            L.tri_rec = rec;
            L.prim = prim;
            tinfo[rec].light = (int32_t)lights.size();
            bvh.tris[rec].light1 = (uint32_t)lights.size() + 1u;
            lights.push_back(L);
        } else if (en.kind == 0 && d->meshes[d->tri_mesh[en.idx]].area_light >= 0) {
            const pt_area_light& al = d->area_lights[d->meshes[d->tri_mesh[en.idx]].area_light];
            const PtTriInfo& ti = tinfo[rec];
            PtLight L;
            std::memset(&L, 0, sizeof(L));
            const float* p0 = d->P + 3 * (size_t)ti.v[0]; const float* p1 = d->P + 3 * (size_t)ti.v[1]; const float* p2 = d->P + 3 * (size_t)ti.v[2];
            std::memcpy(L.p0, p0, 12); std::memcpy(L.p1, p1, 12); std::memcpy(L.p2, p2, 12);
            // Triangle::area (triangle.rs:579-588)
            float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
            float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
            float cx = (ay * bz) - (az * by), cy = (az * bx) - (ax * bz), cz = (ax * by) - (ay * bx);
            L.area = 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz);
            L.mesh_flags = ti.mesh_flags;
            L.two_sided = al.two_sided;
            L.n_samples = (uint32_t)std::max(1, al.n_samples);
            std::memcpy(L.L, al.L, 12);
            L.tri_rec = rec;
            L.prim = prim;
            if (ti.mesh_flags & PT_MESH_HAS_N) {
                std::memcpy(L.n0, d->N + 3 * (size_t)ti.v[0], 12); std::memcpy(L.n1, d->N + 3 * (size_t)ti.v[1], 12); std::memcpy(L.n2, d->N + 3 * (size_t)ti.v[2], 12);
            }
            tinfo[rec].light = (int32_t)lights.size();
            bvh.tris[rec].light1 = (uint32_t)lights.size() + 1u;
            lights.push_back(L);
        }
    }
    if (lights.size() >= (1u << 24)) return ctx->fail(PT_ERR_UNSUPPORTED, "more than 2^24 emissive primitives");
    for (uint32_t f : tri_flags) if (f & PT_TRI_ONE_SIDED) { any_one_sided = true; break; }
    }          // host path
    std::vector<PtMaterial> mats(std::max<uint32_t>(d->n_materials, 1));
    std::memset(mats.data(), 0, mats.size() * sizeof(PtMaterial));
    bool general_materials = false;
    uint32_t n_matte_bins = 0, n_general_bins = 0, n_tex_bins = 0;
    // textured materials: parameter block + one evaluation program per texture-driven parameter (pt_texture.h)
    std::vector<PtMatParams> mparams(mats.size());
    std::memset(mparams.data(), 0, mparams.size() * sizeof(PtMatParams));
    std::vector<uint32_t> tex_prog(1, 0u);          // offset 0 = "constant"
    bool any_textured = false;
    auto add_program = [&](uint32_t root) -> uint32_t {        // post-order over the nodes the root needs; 0 on overflow
        std::vector<uint32_t> order;
        std::vector<int32_t> pos(d->n_textures, -1);
        std::vector<std::pair<uint32_t, int>> stack{{root, 0}};
        while (!stack.empty()) {
            auto& top = stack.back();
            const pt_texture& t = d->textures[top.first];
            const int n_children = t.type == PT_TEX_MIX ? 3 : ((t.type == PT_TEX_SCALE || t.type == PT_TEX_CHECKERBOARD_2D || t.type == PT_TEX_CHECKERBOARD_3D || t.type == PT_TEX_DOTS) ? 2 : 0);
            if (top.second < n_children) {
                int32_t c = t.tex[top.second++];
                if (c >= 0 && pos[c] < 0) stack.push_back({(uint32_t)c, 0});
                continue;
            }
            if (pos[top.first] < 0) { pos[top.first] = (int32_t)order.size(); order.push_back(top.first); }
            stack.pop_back();
        }
        if (order.size() > PT_TEX_PROG_MAX || d->n_textures > 0xffffu) return 0u;
        const uint32_t off = (uint32_t)tex_prog.size();
        tex_prog.push_back((uint32_t)order.size());
        for (uint32_t node : order) {
            const pt_texture& t = d->textures[node];
            uint32_t e = node;
            for (int k = 0; k < 3; k++) e |= (t.tex[k] >= 0 ? (uint32_t)pos[t.tex[k]] : PT_TEX_CHILD_CONST) << (16 + 4 * k);
            tex_prog.push_back(e);
        }
        return off;
    };
    for (uint32_t i = 0; i < d->n_materials; i++) {
        build_lobes(d->materials[i], mats[i]);
        {
            const pt_material& in = d->materials[i];
            const uint32_t refs[13] = {in.tex_kd, in.tex_ks, in.tex_kr, in.tex_kt, in.tex_opacity, in.tex_sigma, in.tex_metal_eta, in.tex_metal_k, in.tex_bump,
                                       in.tex_roughness, in.tex_uroughness, in.tex_vroughness, in.tex_eta};
            PtMatParams& mp = mparams[i];
            mp.m = in;
            material_alphas(in, &mp.a_r, &mp.a_u, &mp.a_v);
            for (int k = 0; k < 13; k++) {
                if (!refs[k] || in.type == PT_MATERIAL_NONE) continue;
                mp.prog[k] = add_program(refs[k] - 1);
                if (!mp.prog[k]) return ctx->fail(PT_ERR_UNSUPPORTED, "a material parameter's texture graph needs more than 12 nodes");
                mats[i].textured = 1;
                any_textured = true;
            }
        }
        // (with object instances every material rides in the general half: only k_shade_general_inst knows how to bring a hit back)
        const bool general = d->materials[i].type != PT_MATERIAL_NONE && (d->materials[i].type != PT_MATERIAL_MATTE || mats[i].textured || d->n_instances > 0);
        if (general) general_materials = true;
        // shade-queue bin: one per material while they last, the remainder of a class shares its last bin
        // (textured materials get the last 32 bins: their own queue segment and kernel; with object instances everything is
        // shaded by one kernel, which reads the whole general..end range)
        if (general && mats[i].textured && d->n_instances == 0) mats[i].sort_bin = PT_SORT_TEX0 + std::min(n_tex_bins++, PT_SORT_BINS - PT_SORT_TEX0 - 1u);
        else mats[i].sort_bin = general ? PT_SORT_GENERAL0 + std::min(n_general_bins++, PT_SORT_TEX0 - PT_SORT_GENERAL0 - 1u)
                                        : std::min(n_matte_bins++, PT_SORT_GENERAL0 - 1u);
    }

    PtScene& sc = ctx->sc;
    std::memset(&sc, 0, sizeof(sc));
    pt_status st;
    // What the lean node visit (pt_kernels.hip node_step_lean) reads beside the builder's output: the per-octant order tables, and
    // empty child slots as inverted boxes so that they fail the slab test by themselves (the builders leave them all-zero, as the
    // reference does; the general visit masks them out by the occupied-slot bits either way).
    if (!fast && bvh.nodes.size() >= (1u << 25)) return ctx->fail(PT_ERR_UNSUPPORTED, "more than 2^25 BVH nodes (32-bit node offsets)");
    // The builders emit nodes depth-first.  The top of the WORLD tree is renumbered breadth-first (the first PT_TOP_BFS_NODES nodes in
    // level order, everything else behind them in the old order), so that "node index < n" means "one of the top levels": k_trace keeps
    // those in LDS (PT_TOP_NODES).  Traversal order, boxes and references are untouched -- only where a node lives.
    if (!fast) {
    if (n_world_nodes > 0 && !(bvh.root_ref & PT_LEAF_BIT) && bvh.root_ref != PT_EMPTY_REF) {
        std::vector<uint32_t> bfs;
        bfs.reserve(PT_TOP_BFS_NODES);
        bfs.push_back(bvh.root_ref & PT_REF_INDEX_MASK);
        for (size_t head = 0; head < bfs.size() && bfs.size() < PT_TOP_BFS_NODES; head++)
            for (int ch = 0; ch < 4 && bfs.size() < PT_TOP_BFS_NODES; ch++) {
                const uint32_t r = bvh.nodes[bfs[head]].child[ch];
                if (r != PT_EMPTY_REF && !(r & PT_LEAF_BIT)) bfs.push_back(r & PT_REF_INDEX_MASK);
            }
        n_top = (uint32_t)bfs.size();
        std::vector<uint32_t> new_of(n_world_nodes, 0xffffffffu);
        for (uint32_t k = 0; k < n_top; k++) new_of[bfs[k]] = k;
        uint32_t next = n_top;
        for (size_t n = 0; n < n_world_nodes; n++) if (new_of[n] == 0xffffffffu) new_of[n] = next++;
        std::vector<PtNode> moved(n_world_nodes);
        ptbvh::parallel_for(n_world_nodes, [&](size_t n0, size_t n1) {
            for (size_t n = n0; n < n1; n++) {
                PtNode nd = bvh.nodes[n];
                for (int ch = 0; ch < 4; ch++)
                    if (nd.child[ch] != PT_EMPTY_REF && !(nd.child[ch] & PT_LEAF_BIT)) nd.child[ch] = (nd.child[ch] & ~PT_REF_INDEX_MASK) | new_of[nd.child[ch] & PT_REF_INDEX_MASK];
                moved[new_of[n]] = nd;
            }
        });
        std::copy(moved.begin(), moved.end(), bvh.nodes.begin());
        bvh.root_ref = (bvh.root_ref & ~PT_REF_INDEX_MASK) | new_of[bvh.root_ref & PT_REF_INDEX_MASK];
    }
    ptbvh::parallel_for(bvh.nodes.size(), [&](size_t n0, size_t n1) {
        for (size_t n = n0; n < n1; n++) {
            PtNode& nd = bvh.nodes[n];
            uint32_t lut = 0;
            for (uint32_t oct = 0; oct < 8; oct++)
                for (uint32_t k = 0; k < 3; k++)
                    if ((oct >> ((nd.axes >> (2 * k)) & 3u)) & 1u) lut |= 1u << (8 * k + oct);
            nd.order_lut = lut;
            // -0.0 and +0.0 bounds: the host builders keep whichever came first, the device builders order them; one spelling for both
            for (int a = 0; a < 3; a++)
                for (int ch = 0; ch < 4; ch++) { if (nd.bmin[a][ch] == 0.0f) nd.bmin[a][ch] = 0.0f; if (nd.bmax[a][ch] == 0.0f) nd.bmax[a][ch] = 0.0f; }
            const uint32_t ax_of[4] = {nd.axes & 3u, (nd.axes >> 2) & 3u, 0u, (nd.axes >> 4) & 3u};     // child 0: axis_top, 1: axis_left, 3: axis_right
            for (int ch = 0; ch < 4; ch++)
                if (nd.child[ch] != PT_EMPTY_REF) nd.child[ch] = (nd.child[ch] & ~(3u << PT_REF_AXIS_SHIFT)) | (ax_of[ch] << PT_REF_AXIS_SHIFT);
            for (int ch = 0; ch < 4; ch++)
                if (!((nd.axes >> (8 + ch)) & 1u))
                    for (int a = 0; a < 3; a++) { nd.bmin[a][ch] = INFINITY; nd.bmax[a][ch] = -INFINITY; }
        }
    });
    if ((st = upload(ctx, ctx->d_nodes, bvh.nodes.data(), bvh.nodes.size())) != PT_OK) return st;
    if ((st = upload(ctx, ctx->d_tris, bvh.tris.data(), bvh.tris.size())) != PT_OK) return st;
    if ((st = upload(ctx, ctx->d_tri_info, tinfo.data(), tinfo.size())) != PT_OK) return st;
    up_n_nodes = bvh.nodes.size(); up_n_tris = bvh.tris.size();
    up_root_ref = bvh.root_ref; up_max_leaf = bvh.max_leaf; up_n_leaves = bvh.n_leaves;
    std::memcpy(up_root_lo, bvh.root_lo, 12); std::memcpy(up_root_hi, bvh.root_hi, 12);
    }          // host path
    if (up_n_nodes >= (1u << 25)) return ctx->fail(PT_ERR_UNSUPPORTED, "more than 2^25 BVH nodes (32-bit node offsets)");
    ctx->n_nodes_up = up_n_nodes; ctx->n_tris_up = up_n_tris;
    if ((st = upload(ctx, ctx->d_materials, mats.data(), mats.size())) != PT_OK) return st;
    ctx->light_samples_total = 0;
    for (const PtLight& L : lights) ctx->light_samples_total += L.n_samples;
    if ((st = upload(ctx, ctx->d_lights, lights.data(), lights.size())) != PT_OK) return st;
    if (d->n_spheres) { if ((st = upload(ctx, ctx->d_spheres, sph.data(), sph.size())) != PT_OK) return st; } else ctx->d_spheres.release();
    if (d->n_instances) { if ((st = upload(ctx, ctx->d_instances, dinst.data(), dinst.size())) != PT_OK) return st; } else ctx->d_instances.release();
    std::vector<PtImage> dimages(d->n_images);
    if (any_textured && d->n_images) {           // all pyramids in one buffer
        size_t total = 0;
        for (uint32_t i = 0; i < d->n_images; i++) {
            const pt_image& im = d->images[i];
            PtImage& o = dimages[i];
            std::memset(&o, 0, sizeof(o));
            o.width = im.width; o.height = im.height; o.channels = im.channels; o.n_levels = im.n_levels;
            uint32_t w = im.width, h = im.height;
            size_t off = 0;
            for (uint32_t l = 0; l < im.n_levels; l++) {
                if (off > 0xffffffffull) return ctx->fail(PT_ERR_UNSUPPORTED, "image pyramid larger than 2^32 floats");
                o.level_off[l] = (uint32_t)off;
                off += (size_t)w * h * im.channels;
                if (w > 1) w /= 2;
                if (h > 1) h /= 2;
            }
            total += off;
        }
        std::vector<float> all(total);
        size_t at = 0;
        std::vector<size_t> base(d->n_images);
        for (uint32_t i = 0; i < d->n_images; i++) {
            size_t n = 0;
            uint32_t w = d->images[i].width, h = d->images[i].height;
            for (uint32_t l = 0; l < d->images[i].n_levels; l++) { n += (size_t)w * h * d->images[i].channels; if (w > 1) w /= 2; if (h > 1) h /= 2; }
            std::memcpy(all.data() + at, d->images[i].texels, n * sizeof(float));
            base[i] = at;
            at += n;
        }
        if ((st = upload(ctx, ctx->d_image_texels, all.data(), all.size())) != PT_OK) return st;
        for (uint32_t i = 0; i < d->n_images; i++) dimages[i].texels = ctx->d_image_texels.as<float>() + base[i];
        if ((st = upload(ctx, ctx->d_images, dimages.data(), dimages.size())) != PT_OK) return st;
    } else { ctx->d_image_texels.release(); ctx->d_images.release(); }
    if (any_textured) {
        if ((st = upload(ctx, ctx->d_textures, d->textures, d->n_textures)) != PT_OK) return st;
        if ((st = upload(ctx, ctx->d_tex_prog, tex_prog.data(), tex_prog.size())) != PT_OK) return st;
        if ((st = upload(ctx, ctx->d_mat_params, mparams.data(), mparams.size())) != PT_OK) return st;
    } else { ctx->d_textures.release(); ctx->d_tex_prog.release(); ctx->d_mat_params.release(); }
    if (d->N) { if ((st = upload(ctx, ctx->d_N, d->N, 3 * (size_t)d->n_vertices)) != PT_OK) return st; } else ctx->d_N.release();
    if (d->S) { if ((st = upload(ctx, ctx->d_S, d->S, 3 * (size_t)d->n_vertices)) != PT_OK) return st; } else ctx->d_S.release();
    if (d->UV) { if ((st = upload(ctx, ctx->d_UV, d->UV, 2 * (size_t)d->n_vertices)) != PT_OK) return st; } else ctx->d_UV.release();
    sc.nodes = ctx->d_nodes.as<PtNode>();
    sc.tris = ctx->d_tris.as<PtTri>();
    sc.tri_info = ctx->d_tri_info.as<PtTriInfo>();
    sc.N = d->N ? ctx->d_N.as<float>() : nullptr;
    sc.S = d->S ? ctx->d_S.as<float>() : nullptr;
    sc.UV = d->UV ? ctx->d_UV.as<float>() : nullptr;
    sc.materials = ctx->d_materials.as<PtMaterial>();
    sc.general_materials = general_materials ? 1u : 0u;
    ctx->scene_has_lobe_materials = general_materials;          // (before spheres / instances force the sorted queue on)
    sc.any_one_sided = any_one_sided ? 1u : 0u;
    sc.dist_leaves = (up_max_leaf <= 8 && !std::getenv("PBRTGPU_SEQ_LEAVES")) ? 1u : 0u;
    ctx->n_materials = d->n_materials;
    sc.lights = ctx->d_lights.as<PtLight>();
    sc.n_lights = (uint32_t)lights.size();
    sc.textured = any_textured ? 1u : 0u;
    sc.textures = any_textured ? ctx->d_textures.as<pt_texture>() : nullptr;
    sc.tex_prog = any_textured ? ctx->d_tex_prog.as<uint32_t>() : nullptr;
    sc.images = (any_textured && d->n_images) ? ctx->d_images.as<PtImage>() : nullptr;
    sc.mat_params = any_textured ? ctx->d_mat_params.as<PtMatParams>() : nullptr;
    sc.spheres = d->n_spheres ? ctx->d_spheres.as<PtSphere>() : nullptr;
    sc.instances = d->n_instances ? ctx->d_instances.as<PtInstance>() : nullptr;
    sc.n_instances = d->n_instances;
    sc.n_spheres = d->n_spheres;
    if (d->n_spheres) sc.general_materials = 1;     // sphere scenes run the sphere-capable kernel instantiations (sorted shade queue)
    if (d->n_instances) { sc.general_materials = 1; sc.dist_leaves = 0; }      // k_trace_inst walks leaves per lane; one shade kernel handles everything
    sc.root_ref = up_root_ref;
    sc.n_top = n_top;
    for (int a = 0; a < 3; a++) { const float ext = up_root_hi[a] - up_root_lo[a]; sc.cell_scale[a] = ext > 0.0f ? (float)(1u << PT_SORT_CELL_BITS) / ext : 0.0f; }
    std::memcpy(sc.wb_min, up_root_lo, 12);
    std::memcpy(sc.wb_max, up_root_hi, 12);
    sc.max_depth = d->max_depth;
    sc.integrator = d->integrator;
    sc.direct_strategy = d->direct_strategy;
    sc.ao_samples = d->ao_samples > 0 ? d->ao_samples : 64;
    sc.ao_cos_sample = d->ao_cos_sample != 0;
    sc.rr_threshold = d->rr_threshold;

    // ---- camera ---------------------------------------------------------------
    pth::M44 r2c;
    if (!pth::raster_to_camera(d->fov, d->screen_window, d->xres, d->yres, &r2c)) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "singular camera projection");
    std::memcpy(sc.cam.raster_to_camera, r2c.a, 64);
    std::memcpy(sc.cam.camera_to_world, d->camera_to_world, 64);
    sc.cam.lens_radius = d->lens_radius;
    sc.cam.focal_distance = d->focal_distance;

    // ---- film (Film::new, film.rs:62-100; get_sample_bounds :166-179) ---------
    PtFilm& fm = sc.film;
    fm.crop[0] = std::max(0, (int)std::floor((float)d->xres * d->crop_window[0]));
    fm.crop[1] = std::max(0, (int)std::floor((float)d->yres * d->crop_window[2]));
    fm.crop[2] = std::min((int)std::ceil((float)d->xres * d->crop_window[1]), d->xres);
    fm.crop[3] = std::min((int)std::ceil((float)d->yres * d->crop_window[3]), d->yres);
    if (fm.crop[2] <= fm.crop[0] || fm.crop[3] <= fm.crop[1]) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "empty crop window");
    fm.filter_radius[0] = d->filter_radius[0]; fm.filter_radius[1] = d->filter_radius[1];
    fm.inv_filter_radius[0] = 1.0f / d->filter_radius[0]; fm.inv_filter_radius[1] = 1.0f / d->filter_radius[1];
    fm.sample_bounds[0] = (int)std::floor((float)fm.crop[0] - fm.filter_radius[0]);
    fm.sample_bounds[1] = (int)std::floor((float)fm.crop[1] - fm.filter_radius[1]);
    fm.sample_bounds[2] = (int)std::ceil((float)fm.crop[2] + fm.filter_radius[0]);
    fm.sample_bounds[3] = (int)std::ceil((float)fm.crop[3] + fm.filter_radius[1]);
    fm.max_sample_luminance = d->max_sample_luminance;
    fm.scale = d->film_scale;
    std::memcpy(fm.filter_table, d->filter_table, sizeof(fm.filter_table));
    ctx->film_w = (uint32_t)(fm.crop[2] - fm.crop[0]);
    ctx->film_h = (uint32_t)(fm.crop[3] - fm.crop[1]);
    uint32_t sbw = (uint32_t)(fm.sample_bounds[2] - fm.sample_bounds[0]), sbh = (uint32_t)(fm.sample_bounds[3] - fm.sample_bounds[1]);
    if (sbw >= 65536 || sbh >= 65536) return ctx->fail(PT_ERR_UNSUPPORTED, "film larger than 65535 pixels on a side");

    // ---- sampler (SobolSampler::new, samplers/sobol.rs:16-31) -----------------
    PtSobol& sb = sc.sobol;
    sb.resolution = round_up_pow2(std::max(sbw, sbh));
    sb.log2_resolution = log2int(sb.resolution);
    sb.kind = (uint32_t)d->sampler;
    sb.spp = sb.kind == PT_SAMPLER_HALTON ? (uint32_t)d->spp : round_up_pow2((uint32_t)d->spp);
    if (sb.kind == PT_SAMPLER_HALTON) {
        if ((st = setup_halton(ctx, sb, (int32_t)sbw, (int32_t)sbh, d->halton_sample_at_center != 0)) != PT_OK) return st;
        sb.resolution = 1;
        sb.log2_resolution = 0;      // the Sobol' index tables are not consulted
    }
    if (sb.log2_resolution > 0 && (sb.log2_resolution - 1 >= ctx->sobol_n_vdc || sb.log2_resolution - 1 >= ctx->sobol_n_inv))
        return ctx->fail(PT_ERR_UNSUPPORTED, "film resolution beyond the VdC Sobol' tables");
    if ((st = upload(ctx, ctx->d_m32, ctx->sobol_m32.data(), ctx->sobol_m32.size())) != PT_OK) return st;
    size_t row = sb.log2_resolution > 0 ? (size_t)(sb.log2_resolution - 1) * ctx->sobol_msize : 0;
    if ((st = upload(ctx, ctx->d_vdc, ctx->sobol_vdc.data() + row, ctx->sobol_msize)) != PT_OK) return st;
    if ((st = upload(ctx, ctx->d_vdc_inv, ctx->sobol_inv.data() + row, ctx->sobol_msize)) != PT_OK) return st;
    sb.m32 = ctx->d_m32.as<uint32_t>();
    sb.m32_len = (uint32_t)ctx->sobol_m32.size();
    if (!ctx->d_bytetab.p) {
        // byte-sliced form of SOBOL_MATRICES_32: T[dim][k][x] = XOR over set bits j of x of column 8k+j,
        // with the reference's column indexing ((dim*52 + c) % len, sobol.rs:43-52) for c up to 55
        const uint32_t n_dims = (uint32_t)(ctx->sobol_m32.size() / ctx->sobol_msize);
        const size_t len = ctx->sobol_m32.size();
        std::vector<uint32_t> tab((size_t)n_dims * 7 * 256);
        for (uint32_t dim = 0; dim < n_dims; dim++) {
            size_t base = std::min((size_t)dim * 52, len - 1);
            for (uint32_t k = 0; k < 7; k++) {
                uint32_t* t = &tab[((size_t)dim * 7 + k) * 256];
                t[0] = 0;
                for (uint32_t x = 1; x < 256; x++) {
                    uint32_t low = x & (~x + 1u);              // lowest set bit
                    uint32_t j = (uint32_t)__builtin_ctz(x);
                    t[x] = t[x ^ low] ^ ctx->sobol_m32[(base + 8 * k + j) % len];
                }
            }
        }
        if ((st = upload(ctx, ctx->d_bytetab, tab.data(), tab.size())) != PT_OK) return st;
    }
    sb.bytetab = ctx->d_bytetab.as<uint32_t>();
    sb.n_tab_dims = (uint32_t)(ctx->sobol_m32.size() / ctx->sobol_msize);
    sb.vdc = ctx->d_vdc.as<uint64_t>();
    sb.vdc_inv = ctx->d_vdc_inv.as<uint64_t>();

    // ---- light sampling distribution (create_light_sample_distribution.rs:11-50) -
    PtLightGrid& g = sc.grid;
    g.n_lights = sc.n_lights;
    g.stride = (2 * sc.n_lights + 2 + 3) & ~3u;     // rows are 16-byte aligned
    std::memcpy(g.wb_min, sc.wb_min, 12);
    std::memcpy(g.wb_max, sc.wb_max, 12);
    int strategy = d->light_strategy;
    if (strategy == PT_LIGHTS_UNIFORM && sc.n_lights != 1) strategy = PT_LIGHTS_SPATIAL;
    ctx->grid_lazy = false;
    ctx->trace_far_choice = -1;          // a new scene: the trial runs again
    if (sc.n_lights > 0) {
        if (strategy == PT_LIGHTS_SPATIAL) {
            const uint32_t max_voxels = 64;
            float diag[3] = {sc.wb_max[0] - sc.wb_min[0], sc.wb_max[1] - sc.wb_min[1], sc.wb_max[2] - sc.wb_min[2]};
            int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);
            float bmax = diag[ext];
            size_t nvox = 1;
            for (int i = 0; i < 3; i++) {
                float c = std::ceil(diag[i] / bmax * (float)max_voxels);
                uint32_t v = c > 0.0f ? (c >= 4294967296.0f ? 0xffffffffu : (uint32_t)c) : 0u;
                v = std::min(std::max(v, 1u), max_voxels);
                g.voxels[i] = v;
                nvox *= v;
            }
            g.single = 0;
            size_t bytes = nvox * g.stride * sizeof(float);
            // Dense while it is small (every voxel's tables made here, at upload: 0.6 ms for RT1M's 262 144 voxels x 2 lights); from
            // PBRTGPU_LIGHT_GRID_DENSE_MAX bytes on (default 2 GiB: about a thousand lights) the grid is filled as the reference fills its hash table
            // (spatial.rs:199-260), voxel by voxel on first touch -- render_tiles lists the voxels each bounce's hits fall in (k_grid_mark) and makes
            // their rows before the bounce is shaded.  A mesh light of ten thousand triangles is 80 KB per voxel: 21 GB dense, a few hundred MB for
            // the voxels a frame actually visits.
            size_t dense_max = (size_t)2 << 30;
            if (const char* e = std::getenv("PBRTGPU_LIGHT_GRID_DENSE_MAX")) dense_max = std::strtoull(e, nullptr, 10);
            ctx->grid_lazy = bytes > dense_max;
            ctx->grid_nvox = nvox;
            if (!ctx->grid_lazy) {
                PT_HIP(ctx->d_grid.alloc(bytes + 64));
                g.data = ctx->d_grid.as<float>();
                PT_HIP(ptk_light_grid(ctx->stream, sc, ctx->d_grid.as<float>(), (uint32_t)nvox));
                PT_HIP(hipStreamSynchronize(ctx->stream));
            } else {
                const size_t row_bytes = (size_t)g.stride * sizeof(float);
                ctx->grid_rows_cap = std::min<size_t>(nvox, std::max<size_t>(256, ((size_t)256 << 20) / row_bytes));
                ctx->grid_rows_used = 0;
                PT_HIP(ctx->d_grid.alloc(ctx->grid_rows_cap * row_bytes + 64));
                PT_HIP(ctx->d_grid_rows.alloc(nvox * 4));
                PT_HIP(ctx->d_grid_todo.alloc(nvox * 4 + 64));          // the voxel list, then its counter
                PT_HIP(hipMemsetAsync(ctx->d_grid_rows.p, 0xff, nvox * 4, ctx->stream));
                PT_HIP(hipMemsetAsync(ctx->d_grid_todo.as<uint32_t>() + nvox, 0, 64, ctx->stream));
                g.data = ctx->d_grid.as<float>();
                g.row_of = ctx->d_grid_rows.as<int32_t>();
            }
        } else {
            // uniform / power: one Distribution1D (lightdistrib/uniform.rs, power.rs:9-17)
            g.voxels[0] = g.voxels[1] = g.voxels[2] = 1;
            g.single = 1;
            uint32_t nl = sc.n_lights;
            std::vector<float> tab(g.stride + 16);
            for (uint32_t i = 0; i < nl; i++) {
                if (strategy == PT_LIGHTS_UNIFORM) tab[i] = 1.0f;
                else {
                    float n = lights[i].two_sided ? 2.0f : 1.0f;
                    float s = n * lights[i].area * 3.14159265358979323846f;
                    float r = lights[i].L[0] * s, gg = lights[i].L[1] * s, b = lights[i].L[2] * s;
                    tab[i] = 0.212671f * r + 0.715160f * gg + 0.072169f * b;
                }
            }
            float* cdf = tab.data() + nl;
            cdf[0] = 0.0f;
            for (uint32_t i = 1; i < nl + 1; i++) cdf[i] = cdf[i - 1] + tab[i - 1] / (float)nl;
            float func_int = cdf[nl];
            if (func_int == 0.0f) for (uint32_t i = 1; i < nl + 1; i++) cdf[i] = (float)i / (float)nl;
            else for (uint32_t i = 1; i < nl + 1; i++) cdf[i] /= func_int;
            cdf[nl + 1] = func_int;
            if ((st = upload(ctx, ctx->d_grid, tab.data(), tab.size())) != PT_OK) return st;
            g.data = ctx->d_grid.as<float>();
        }
    }

    // ---- film + scratch buffers ---------------------------------------------------
    size_t npx = (size_t)ctx->film_w * ctx->film_h;
    PT_HIP(ctx->d_own.alloc(npx * 16));
    PT_HIP(ctx->d_spillfilm.alloc(npx * 16));
    PT_HIP(ctx->d_xyzw.alloc(npx * 16));
    PT_HIP(ctx->d_rgb.alloc(npx * 12));
    // every kernel runs on ctx->stream (non-blocking: no implicit ordering with the null stream), so do these
    PT_HIP(hipMemsetAsync(ctx->d_own.p, 0, npx * 16, ctx->stream));
    PT_HIP(hipMemsetAsync(ctx->d_spillfilm.p, 0, npx * 16, ctx->stream));
    PT_HIP(hipMemsetAsync(ctx->d_xyzw.p, 0, npx * 16, ctx->stream));
    ctx->xyzw_committed = false;
    if (!ctx->d_counters.p) {
        PT_HIP(ctx->d_counters.alloc(sizeof(PtCounters)));
        PT_HIP(hipMemsetAsync(ctx->d_counters.p, 0, sizeof(PtCounters), ctx->stream));
        if (!ctx->d_err.p) {
            PT_HIP(ctx->d_err.alloc(16));
            PT_HIP(hipMemsetAsync(ctx->d_err.p, 0, 16, ctx->stream));
        }
        PT_HIP(ctx->d_ticket.alloc(16));
        PT_HIP(ctx->d_counts.alloc(PT_COUNTS_WORDS * 4));
        PT_HIP(hipMemsetAsync(ctx->d_counts.p, 0, PT_COUNTS_WORDS * 4, ctx->stream));
    }
    PT_HIP(hipStreamSynchronize(ctx->stream));
    if ((st = ensure_traversal_scratch(ctx)) != PT_OK) return st;
    double t2 = now_ms();

    pt_scene_info& inf = ctx->info;
    std::memset(&inf, 0, sizeof(inf));
    for (int i = 0; i < 4; i++) { inf.sample_bounds[i] = fm.sample_bounds[i]; inf.cropped_bounds[i] = fm.crop[i]; }
    inf.spp = (int32_t)sb.spp;
    inf.n_lights = sc.n_lights;
    inf.n_nodes = (uint32_t)up_n_nodes;
    inf.n_leaves = up_n_leaves;
    std::memcpy(inf.world_bound, sc.wb_min, 12);
    std::memcpy(inf.world_bound + 3, sc.wb_max, 12);
    inf.bvh_build_ms = t1 - t0;
    inf.upload_ms = t2 - t1;
    inf.bvh_on_device = dev_build.used ? 1 : 0;
    inf.reserved = 0;
    ctx->have_scene = true;
    return PT_OK;
}

pt_status pt_scene_info_get(const pt_context* ctx, pt_scene_info* out) {
    if (!ctx || !out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return PT_ERR_NO_SCENE;
    *out = ctx->info;
    return PT_OK;
}

pt_status pt_film_clear(pt_context* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    size_t npx = (size_t)ctx->film_w * ctx->film_h;
    PT_HIP(hipMemsetAsync(ctx->d_own.p, 0, npx * 16, ctx->stream));
    PT_HIP(hipMemsetAsync(ctx->d_spillfilm.p, 0, npx * 16, ctx->stream));
    PT_HIP(hipMemsetAsync(ctx->d_xyzw.p, 0, npx * 16, ctx->stream));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->xyzw_committed = false;
    return PT_OK;
}

// Renders the listed tiles; radiance_out (device, optional) receives per-sample radiance.
// The lazily filled light grid, before a bounce is shaded: the voxels this bounce's hits fall in and nobody has asked for yet get their rows.
// One counter read-back per bounce (scenes with thousands of lights only: the dense grid needs none of this).
static pt_status fill_light_grid(pt_context* ctx, const PtQueues& Q) {
    PtScene& sc = ctx->sc;
    uint32_t* todo = ctx->d_grid_todo.as<uint32_t>();
    uint32_t* todo_count = todo + ctx->grid_nvox;
    PT_HIP(ptk_grid_mark(ctx->stream, ctx->grid_wide, sc, ctx->paths, Q, ctx->d_grid_rows.as<int32_t>(), todo, todo_count));
    uint32_t n = 0;
    PT_HIP(hipMemcpyAsync(&n, todo_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    if (n == 0) return PT_OK;
    const size_t row_bytes = (size_t)sc.grid.stride * sizeof(float);
    if (ctx->grid_rows_used + n > ctx->grid_rows_cap) {          // grow the row store (doubling, never past one row per voxel) and move the rows made so far
        size_t cap = ctx->grid_rows_cap;
        while (cap < ctx->grid_rows_used + n) cap *= 2;
        cap = std::min(cap, ctx->grid_nvox);
        void* bigger = nullptr;
        if (hipMalloc(&bigger, cap * row_bytes + 64) != hipSuccess)
            return ctx->fail(PT_ERR_DEVICE, "light grid: no memory for the tables of the voxels this frame visits (" + std::to_string(cap * row_bytes >> 20) + " MiB)");
        PT_HIP(hipMemcpyAsync(bigger, ctx->d_grid.p, ctx->grid_rows_used * row_bytes, hipMemcpyDeviceToDevice, ctx->stream));
        PT_HIP(hipStreamSynchronize(ctx->stream));
        ctx->d_grid.release();
        ctx->d_grid.p = bigger; ctx->d_grid.bytes = cap * row_bytes + 64;
        ctx->grid_rows_cap = cap;
        sc.grid.data = ctx->d_grid.as<float>();
    }
    float* rows = ctx->d_grid.as<float>() + ctx->grid_rows_used * (size_t)sc.grid.stride;
    PT_HIP(ptk_light_grid(ctx->stream, sc, rows, n, todo));
    PT_HIP(ptk_grid_assign(ctx->stream, ctx->d_grid_rows.as<int32_t>(), todo, n, (uint32_t)ctx->grid_rows_used, todo_count));
    ctx->grid_rows_used += n;
    if (std::getenv("PBRTGPU_BUILD_TRACE")) std::fprintf(stderr, "[light grid] %u voxels filled on first touch (%zu of %zu so far, %zu MiB of rows)\n", n, ctx->grid_rows_used, ctx->grid_nvox,
                                                         ctx->grid_rows_used * row_bytes >> 20);
    return PT_OK;
}

static pt_status render_tiles(pt_context* ctx, const pt_tile* tiles, uint32_t n_tiles, float* d_radiance_out) {
    const PtScene& sc = ctx->sc;
    const int32_t* sbnd = sc.film.sample_bounds;
    std::vector<pt_tile> all;
    if (n_tiles == 0 && tiles == nullptr) {      // the reference's 16x16 decomposition (sampler.rs:266-289)
        for (int32_t y = sbnd[1]; y < sbnd[3]; y += 16)
            for (int32_t x = sbnd[0]; x < sbnd[2]; x += 16) {
                pt_tile t = {x, y, std::min(x + 16, sbnd[2]), std::min(y + 16, sbnd[3])};
                all.push_back(t);
            }
        tiles = all.data();
        n_tiles = (uint32_t)all.size();
    }
    // The pass's pixel list is expanded from the tiles on the device (k_expand_tiles); the host only checks the tiles and
    // prefix-sums their sizes.
    std::vector<uint32_t> tile_off(n_tiles);
    size_t n_pixels_total = 0;
    for (uint32_t i = 0; i < n_tiles; i++) {
        const pt_tile& t = tiles[i];
        if (t.x0 < sbnd[0] || t.y0 < sbnd[1] || t.x1 > sbnd[2] || t.y1 > sbnd[3] || t.x1 < t.x0 || t.y1 < t.y0)
            return ctx->fail(PT_ERR_INVALID_ARGUMENT, "tile outside the sample bounds");
        tile_off[i] = (uint32_t)n_pixels_total;
        n_pixels_total += (size_t)(t.x1 - t.x0) * (size_t)(t.y1 - t.y0);
        if (n_pixels_total > (size_t)(sbnd[2] - sbnd[0]) * (size_t)(sbnd[3] - sbnd[1])) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "tiles overlap (more pixels than the sample bounds hold)");
    }
    if (n_pixels_total == 0) return PT_OK;
    const uint32_t spp = sc.sobol.spp;

    // Paths in flight per pass.  Every k_trace launch ends with a drain tail (ray lengths are heavy-tailed and a
    // lane sees only a handful of rays per launch), so big launches pay: 4 M paths -> 662 Mrays/s, 64 M -> 850 on
    // RT1M, 192 M (the 256-spp frame in two passes instead of five) 1 037, 288 M (one pass: nine launches of up to 140 M rays) 1 048:
    // 277 M slots are 64 GB of path state, 6.6 GB of queues and 3.3 GB of sort buffers -- HBM3E is 288 GB, and a pool never takes more
    // than half of what is free.
    size_t pool_target = (size_t)288 << 20;
    if (const char* e = std::getenv("PBRTGPU_POOL_PATHS")) pool_target = std::max<size_t>(65536, std::strtoull(e, nullptr, 10));
    pool_target = std::min<size_t>(pool_target, (size_t)1 << 30);      // 32-bit path ids and queue counters with room to spare
    const bool ao = sc.integrator == PT_INTEGRATOR_AO;
    const bool rec = sc.integrator == PT_INTEGRATOR_DIRECTLIGHTING || sc.integrator == PT_INTEGRATOR_WHITTED;
    // next-event entries per path: Whitted one per light, DirectLighting "one" a single one, "all" one per light sample
    const uint32_t rec_epp = !rec ? 0u : (sc.integrator == PT_INTEGRATOR_DIRECTLIGHTING && sc.direct_strategy == PT_DIRECT_ONE) ? 1u
                             : sc.integrator == PT_INTEGRATOR_WHITTED ? std::max(1u, sc.n_lights) : std::max(1u, ctx->light_samples_total);
    const uint32_t rec_depth = (uint32_t)std::max(1, sc.max_depth);
    const size_t rec_per_path = 64 + (size_t)rec_depth * PT_REC_FRAME_F4 * 16 + (size_t)rec_epp * (6 * 16 + 1 + 4 + 4 + 8);
    {   // never more than half of what the device has free (what this render still has to allocate: a pool or frame store that exists is not counted twice)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            // per path: the pool's eleven float4 and its words, five queues + the sort's lists; textured scenes: the evaluated parameters; large scenes: the
            // continuation sort's lists; the recursive integrators: their frames and next-event entries
            size_t per_path = 11 * 16 + 8 + 8 + 6 * 4 + 4 + 6 * 4 + 4 * 4;
            if (sc.textured && !sc.n_instances) per_path += PT_TEX_RES_F4 * 16;
            if (ctx->sort_cont != 0) per_path += 4 * 4;
            if (rec) {
                // The frame store used to be held under 6 GB whatever the device had: a 64-spp directlighting frame of RT1M then ran as 22 passes of
                // 3 M camera samples, each ending in the traversal kernel's drain tail (5.9 ms launches: 687 Mrays/s where the same kernel does
                // 1 070 on launches of 150 ms).  Now it takes what the pool takes: up to half of the free memory together.
                const size_t have = ctx->pool_paths * per_path + ctx->rec_paths * rec_per_path;      // already allocated: comes back to the budget
                const size_t budget = (free_b + have) / 2;
                pool_target = std::min(pool_target, std::max<size_t>(65536, budget / (per_path + rec_per_path)));
            } else if (ctx->pool_paths < pool_target) {
                pool_target = std::min(pool_target, std::max<size_t>(ctx->pool_paths, std::max<size_t>(1u << 20, (free_b / 2) / per_path)));
            }
        }
    }
    if (rec) pool_target = std::min<size_t>(pool_target, ((size_t)1 << 31) / std::max(1u, rec_epp));      // next-event entries are numbered path x entries-per-path in 32 bits
    if (rec) if (const char* e = std::getenv("PBRTGPU_REC_POOL_BYTES")) pool_target = std::max<size_t>(65536, std::min<size_t>(pool_target, std::strtoull(e, nullptr, 10) / rec_per_path));
    if (ao) pool_target = std::max<size_t>(65536, std::min<size_t>(pool_target, ((size_t)128 << 20) / (size_t)sc.ao_samples));      // <= 128 M occlusion rays (4.4 GB) per pass: big launches amortise the drain tail
    size_t chunk_pix = std::min(n_pixels_total, pool_target);
    uint32_t S = (uint32_t)std::max<size_t>(1, std::min<size_t>(spp, pool_target / chunk_pix));
    {   // equal passes: 256 spp at 63 spp per pass would leave a 4-spp runt
        const uint32_t passes = (spp + S - 1) / S;
        S = (spp + passes - 1) / passes;
    }
    pt_status st;
    if ((st = ensure_pool(ctx, chunk_pix * S)) != PT_OK) return st;
    if (ctx->pixels_cap < n_pixels_total) {
        PT_HIP(ctx->d_pixels.alloc(n_pixels_total * 4));
        ctx->pixels_cap = n_pixels_total;
    }
    {
        const size_t sb_w = (size_t)(sbnd[2] - sbnd[0]), sb_h = (size_t)(sbnd[3] - sbnd[1]);
        const size_t bm_bytes = ((sb_w * sb_h + 31) / 32) * 4;
        if (ctx->d_tilebits.bytes < bm_bytes) PT_HIP(ctx->d_tilebits.alloc(bm_bytes));
        if (ctx->d_tiles.bytes < (size_t)n_tiles * 20) PT_HIP(ctx->d_tiles.alloc((size_t)n_tiles * 20));
        uint32_t* d_off = reinterpret_cast<uint32_t*>(ctx->d_tiles.as<char>() + (size_t)n_tiles * 16);
        PT_HIP(hipMemsetAsync(ctx->d_tilebits.p, 0, bm_bytes, ctx->stream));
        PT_HIP(hipMemsetAsync(ctx->d_err.p, 0, 4, ctx->stream));          // this call's errors only (a sampler hook may have left the Halton bit behind)
        PT_HIP(hipMemcpyAsync(ctx->d_tiles.p, tiles, (size_t)n_tiles * 16, hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(hipMemcpyAsync(d_off, tile_off.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(ptk_expand_tiles(ctx->stream, ctx->d_tiles.as<int4>(), d_off, n_tiles, sbnd[0], sbnd[1], (uint32_t)sb_w, ctx->d_pixels.as<uint32_t>(),
                                ctx->d_tilebits.as<uint32_t>(), ctx->d_err.as<uint32_t>()));
        uint32_t herr0 = 0;      // pageable source buffers above: the copies have been staged when the calls return; one sync covers the check
        PT_HIP(hipMemcpyAsync(&herr0, ctx->d_err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(hipStreamSynchronize(ctx->stream));
        if (herr0 & 2u) {
            PT_HIP(hipMemsetAsync(ctx->d_err.p, 0, 4, ctx->stream));
            PT_HIP(hipStreamSynchronize(ctx->stream));
            return ctx->fail(PT_ERR_INVALID_ARGUMENT, "tiles overlap: the tiles of one pt_render call must be disjoint");
        }
    }

    PtQueues Q;
    Q.nee = ctx->d_qnee.as<uint32_t>();
    Q.counts = ctx->d_counts.as<uint32_t>();
    Q.sorted = ctx->d_qsorted.as<uint32_t>();
    Q.shadow = ctx->d_qshadow.as<uint32_t>();
    Q.probe = ctx->d_qprobe.as<uint32_t>();
    Q.bin = nullptr;
    if (sc.general_materials) {          // the material sort keeps each entry's bin between its two kernels
        if (ctx->d_qbin.bytes < ctx->pool_paths * 2) PT_HIP(ctx->d_qbin.alloc(ctx->pool_paths * 2));
        Q.bin = ctx->d_qbin.as<uint16_t>();
    }
    Q.shadow_key = nullptr;
    // pt_raysort.hip: a key per shadow ray beside the list (k_shade writes it; k_rec_nee_lists for the recursive integrators, whose lists hold
    // next-event ENTRIES, rec_epp per camera sample), each launch's list is sorted by it
    const bool rec_i = sc.integrator == PT_INTEGRATOR_DIRECTLIGHTING || sc.integrator == PT_INTEGRATOR_WHITTED;
    if (ctx->sort_shadow_min > 0 && (sc.integrator == PT_INTEGRATOR_PATH || rec_i)) {
        const size_t want_cap = rec_i ? chunk_pix * (size_t)S * std::max(1u, rec_epp) : ctx->pool_paths;
        if (ctx->sort_cap < want_cap) {
            const size_t tb = ptk_sort_rays_temp_bytes((uint32_t)want_cap);
            if (tb == 0) return ctx->fail(PT_ERR_DEVICE, "radix sort scratch size query failed");
            PT_HIP(ctx->d_sort_ids.alloc(want_cap * 4));
            PT_HIP(ctx->d_sort_keys[0].alloc(want_cap * 4));
            PT_HIP(ctx->d_sort_keys[1].alloc(want_cap * 4));
            PT_HIP(ctx->d_sort_temp.alloc(tb));
            ctx->sort_cap = want_cap;
        }
        Q.shadow_key = ctx->d_sort_keys[0].as<uint32_t>();
    }
    // Continuation rays in origin order as well, where it pays: once nodes + leaf records outgrow the 256 MiB Infinity Cache the traversal
    // kernel waits on HBM, and rays that start in one cell miss the caches together instead of one by one.
    int sort_cont = ctx->sort_cont;
    if (sort_cont < 0) sort_cont = (ctx->n_nodes_up * sizeof(PtNode) + ctx->n_tris_up * sizeof(PtTri) > ((size_t)256 << 20)) ? 1 : 0;
    if (sc.integrator != PT_INTEGRATOR_PATH) sort_cont = 0;
    if (sort_cont) {
        if (ctx->csort_cap < ctx->pool_paths) {
            const size_t tb = ptk_sort_rays_keep_temp_bytes((uint32_t)ctx->pool_paths);
            if (tb == 0) return ctx->fail(PT_ERR_DEVICE, "radix sort scratch size query failed");
            PT_HIP(ctx->d_csort_ids.alloc(ctx->pool_paths * 4));
            PT_HIP(ctx->d_csort_keys[0].alloc(ctx->pool_paths * 4));
            PT_HIP(ctx->d_csort_keys[1].alloc(ctx->pool_paths * 4));
            PT_HIP(ctx->d_csort_temp.alloc(tb));
            ctx->csort_cap = ctx->pool_paths;
        }
    }
    PtCounters* cnt = ctx->d_counters.as<PtCounters>();
    uint32_t* err = ctx->d_err.as<uint32_t>();
    size_t ev_i = 0;
    std::vector<std::pair<size_t, int>> spans;   // (event index, kind) kind 0 = trace, 1 = shade
    const bool bounce_log = std::getenv("PBRTGPU_BOUNCE_LOG") != nullptr;
    std::vector<std::array<uint32_t, 3>> bounce_counts;
    const bool no_events = std::getenv("PBRTGPU_NO_EVENTS") != nullptr;   // experiment: cost of the per-bounce event records

    hipEvent_t ev_begin = get_event(ctx, ev_i++), ev_end = get_event(ctx, ev_i++);
    PT_HIP(hipEventRecord(ev_begin, ctx->stream));

    for (size_t c0 = 0; c0 < n_pixels_total; c0 += chunk_pix) {
        uint32_t n_pix = (uint32_t)std::min(chunk_pix, n_pixels_total - c0);
        const uint32_t* d_pix = ctx->d_pixels.as<uint32_t>() + c0;
        for (uint32_t s0 = 0; s0 < spp; s0 += S) {
            uint32_t ns = std::min(S, spp - s0);
            uint32_t* qa = ctx->d_qa.as<uint32_t>();
            uint32_t* qb = ctx->d_qb.as<uint32_t>();
            Q.cur = qa; Q.next = qb;
            PT_HIP(ptk_gen(ctx->stream, ctx->grid_wide, sc, ctx->paths, Q, d_pix, n_pix, s0, ns, cnt));
            if (ao) {
                // camera rays -> closest hits -> ao_samples occlusion rays per hit (compacted) -> any-hit -> ordered sum
                const uint32_t n_paths = n_pix * ns;
                const size_t cap = (size_t)n_paths * (size_t)sc.ao_samples;
                if (ctx->ao_rays_cap < cap) {      // per ray: origin + t_max, direction (the record of a shadow work item), weight, item id, occlusion flag
                    PT_HIP(ctx->d_ao.alloc(cap * 41 + 256));
                    ctx->ao_rays_cap = cap;
                    PT_HIP(ptk_iota(ctx->stream, ctx->grid_wide, reinterpret_cast<uint32_t*>(ctx->d_ao.as<char>() + ctx->ao_rays_cap * 36), (uint32_t)cap));
                }
                float4* ao_o = ctx->d_ao.as<float4>();
                float4* ao_d = ao_o + ctx->ao_rays_cap;
                float* ao_w = reinterpret_cast<float*>(ao_d + ctx->ao_rays_cap);
                uint32_t* ao_ids = reinterpret_cast<uint32_t*>(ao_w + ctx->ao_rays_cap);
                uint8_t* ao_occ = reinterpret_cast<uint8_t*>(ao_ids + ctx->ao_rays_cap);
                uint32_t* ao_count = ctx->d_ticket.as<uint32_t>() + 2;
                hipEvent_t a = get_event(ctx, ev_i), b = get_event(ctx, ev_i + 1), c = get_event(ctx, ev_i + 2);
                if (!a || !b || !c) return ctx->fail(PT_ERR_DEVICE, "hipEventCreate failed");
                spans.push_back({ev_i, 0});
                ev_i += 3;
                PT_HIP(hipMemsetAsync(ctx->d_ticket.p, 0, 16, ctx->stream));
                PT_HIP(ptk_ao_tag(ctx->stream, ctx->grid_wide, ctx->paths, n_pix, n_paths, s0));
                PT_HIP(hipEventRecord(a, ctx->stream));
                PT_HIP(ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, sc, ctx->paths, Q, cnt, ctx->d_spill.as<uint32_t>(), ctx->spill_depth, err));
                ctx->trace_launches++;
                PT_HIP(ptk_ao_rays(ctx->stream, ctx->grid_wide, sc, ctx->paths, n_paths, ao_o, ao_d, ao_w, ao_count, cnt));
                {   // the occlusion rays as shadow work items of the wavefront traversal kernel itself (any hit -> occlusion flag);
                    // the item count is read on the device, so the pass never waits for the host
                    PtPaths AP = ctx->paths;
                    AP.sh_o = ao_o; AP.sh_d = ao_d; AP.occluded = ao_occ;
                    PtQueues AQ = Q;
                    AQ.shadow = ao_ids;
                    PT_HIP(ptk_ao_queue(ctx->stream, AQ, ao_count, (uint32_t)sc.ao_samples));
                    PT_HIP(ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, sc, AP, AQ, cnt, ctx->d_spill.as<uint32_t>(), ctx->spill_depth, err));
                    ctx->trace_launches++;
                }
                PT_HIP(hipEventRecord(b, ctx->stream));
                PT_HIP(ptk_ao_resolve(ctx->stream, ctx->grid_wide, sc, ctx->paths, n_paths, ao_w, ao_occ));
                PT_HIP(hipEventRecord(c, ctx->stream));
            } else if (rec) {
                // DirectLighting / Whitted: depth-first walk over the specular trees, two traversal launches per tree level
                const uint32_t n_paths = n_pix * ns;
                // sized by THIS render's largest pass (chunk_pix * S, which the 6 GB bound on pool_target above limits), not by the path
                // pool: the pool never shrinks, and a context that has rendered a 288 M-path frame before would ask for 230 GB here
                const size_t np = chunk_pix * (size_t)S, ne = np * rec_epp;
                if (ctx->rec_paths < np || ctx->rec_epp != rec_epp || ctx->rec_depth != rec_depth || !ctx->d_rec.p) {
                    PT_HIP(ctx->d_rec.alloc(np * rec_per_path + 65536));
                    ctx->rec_paths = np; ctx->rec_epp = rec_epp; ctx->rec_depth = rec_depth;
                    if (!ctx->d_counts2.p) PT_HIP(ctx->d_counts2.alloc(PT_COUNTS_WORDS * 4));
                }
                PT_HIP(hipMemsetAsync(ctx->d_counts2.p, 0, PT_COUNTS_WORDS * 4, ctx->stream));
                PtRec R;
                char* rb = ctx->d_rec.as<char>();
                auto take = [&](size_t bytes) { char* q = rb; rb += (bytes + 255) & ~(size_t)255; return q; };
                R.diff = (float4*)take(np * 64);
                R.frames = (float4*)take(np * (size_t)rec_depth * PT_REC_FRAME_F4 * 16);
                R.sh_o = (float4*)take(ne * 16); R.sh_d = (float4*)take(ne * 16); R.pr_o = (float4*)take(ne * 16); R.pr_d = (float4*)take(ne * 16);
                R.A = (float4*)take(ne * 16); R.B = (float4*)take(ne * 16);
                R.panic = err;
                R.prec = (int32_t*)take(ne * 4); R.flags = (uint32_t*)take(ne * 4);
                uint32_t* nl_shadow = (uint32_t*)take(ne * 4);
                uint32_t* nl_probe = (uint32_t*)take(ne * 4);
                R.occ = (uint8_t*)take(ne);
                R.n_paths = (uint32_t)np; R.max_depth = rec_depth; R.epp = rec_epp;
                R.n_arrays = (sc.integrator == PT_INTEGRATOR_DIRECTLIGHTING && sc.direct_strategy == PT_DIRECT_ALL) ? 2u * sc.n_lights * (uint32_t)std::max(sc.max_depth, 0) : 0u;
                R.s0 = s0; R.n_pix = n_pix;
                if (5u + 2u * R.n_arrays > 60000u) return ctx->fail(PT_ERR_UNSUPPORTED, "directlighting \"all\": too many lights x maxdepth for the sampler's array dimensions");
                PtPaths NP = ctx->paths;             // the node's next-event rays as shadow / probe work items
                NP.sh_o = R.sh_o; NP.sh_d = R.sh_d; NP.pr_o = R.pr_o; NP.pr_d = R.pr_d; NP.occluded = R.occ; NP.probe_rec = R.prec;
                PtQueues Qn = Q;
                Qn.counts = ctx->d_counts2.as<uint32_t>(); Qn.shadow = nl_shadow; Qn.probe = nl_probe;
                PT_HIP(ptk_rec_init(ctx->stream, ctx->grid_wide, sc, ctx->paths, R, n_paths));
                // The walk ends when no camera sample is live.  That count comes back through page-locked memory ONE LEVEL BEHIND: level i + 1 is
                // queued before the host looks at level i's count, so the device never waits for the host (a level launched after the last one
                // finds empty lists and does nothing).
                if (!ctx->h_live) PT_HIP(hipHostMalloc((void**)&ctx->h_live, 64));
                hipEvent_t rb_ev[2] = {get_event(ctx, ev_i), get_event(ctx, ev_i + 1)};
                if (!rb_ev[0] || !rb_ev[1]) return ctx->fail(PT_ERR_DEVICE, "hipEventCreate failed");
                ev_i += 2;
                for (uint32_t iter = 0; iter < 100000u; iter++) {
                    hipEvent_t a = get_event(ctx, ev_i), b = get_event(ctx, ev_i + 1), c = get_event(ctx, ev_i + 2);
                    if (!a || !b || !c) return ctx->fail(PT_ERR_DEVICE, "hipEventCreate failed");
                    if (ev_i < 3000) { spans.push_back({ev_i, 0}); ev_i += 3; }
                    Qn.cur = Q.cur;
                    PT_HIP(hipEventRecord(a, ctx->stream));
                    PT_HIP(ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, sc, ctx->paths, Q, cnt, ctx->d_spill.as<uint32_t>(), ctx->spill_depth, err));
                    PT_HIP(ptk_prep(ctx->stream, Qn, 0));
                    PT_HIP(ptk_rec_enter(ctx->stream, ctx->grid_shade, sc, ctx->paths, Q, Qn, R, cnt, rec_epp));
                    PtQueues Qt = Qn;
                    if (Qn.shadow_key) {
                        // The node's shadow rays start at the hit points of this level -- scattered through a scene of small triangles -- and head for the
                        // lights: ordered by origin cell and direction octant like the path integrator's (pt_raysort.hip).  Costs this level one
                        // counter read-back; only the work list moves (results are written per entry).
                        uint32_t n_sh = 0;
                        PT_HIP(hipMemcpyAsync(&n_sh, Qn.counts + PT_Q_SHADOW, 4, hipMemcpyDeviceToHost, ctx->stream));
                        PT_HIP(hipStreamSynchronize(ctx->stream));
                        if (n_sh >= (uint32_t)ctx->sort_shadow_min) {
                            uint32_t* sorted = nullptr;
                            PT_HIP(ptk_sort_shadow_rays(ctx->stream, Qn.shadow, ctx->d_sort_ids.as<uint32_t>(), ctx->d_sort_keys[0].as<uint32_t>(),
                                                        ctx->d_sort_keys[1].as<uint32_t>(), ctx->d_sort_temp.p, ctx->d_sort_temp.bytes, n_sh, &sorted));
                            Qt.shadow = sorted;
                        }
                    }
                    PT_HIP(ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, sc, NP, Qt, cnt, ctx->d_spill.as<uint32_t>(), ctx->spill_depth, err));
                    ctx->trace_launches += 2;
                    PT_HIP(hipEventRecord(b, ctx->stream));
                    PT_HIP(ptk_rec_next(ctx->stream, ctx->grid_shade, sc, ctx->paths, Q, R));
                    PT_HIP(ptk_prep(ctx->stream, Q, 1));
                    PT_HIP(hipEventRecord(c, ctx->stream));
                    std::swap(Q.cur, Q.next);
                    PT_HIP(hipMemcpyAsync(&ctx->h_live[iter & 1u], Q.counts + PT_Q_CUR, 4, hipMemcpyDeviceToHost, ctx->stream));
                    PT_HIP(hipEventRecord(rb_ev[iter & 1u], ctx->stream));
                    if (iter >= 1u) {
                        PT_HIP(hipEventSynchronize(rb_ev[(iter - 1u) & 1u]));
                        if (ctx->h_live[(iter - 1u) & 1u] == 0) { ctx->trace_launches -= 2; break; }      // the level just queued is an empty one
                    }
                }
            } else if (sc.n_lights > 0) {          // no lights: li() returns zero immediately (path.rs:71-74)
                uint32_t* shadow_sorted = nullptr;          // the ordered shadow list for the next traversal launch, if one was made
                uint32_t* cont_sorted = nullptr;            // the ordered copy of cur for the next traversal launch (sort_cont == 1)
                uint32_t* cont_spare = ctx->d_csort_ids.as<uint32_t>();
                // Which traversal kernel: k_trace_far (nodes fetched by lane pairs) only pays where rays miss the caches -- 16 M sparse triangles +15 %, but
                // 8 M triangles whose rays end early -12 %, RT1M -4 % -- and neither the scene's size nor its depth tells the two apart.  So a scene larger than
                // the Infinity Cache decides by trial: its first incoherent bounce after an upload (bounce 1 of the first pass: the largest one) is
                // traced twice, by each kernel, on the same lists; results are identical (the second launch rewrites them), the counters are put back.
                const char* far_min_env = std::getenv("PBRTGPU_TRACE_FAR_MIN_BYTES");          // (tests: the trial on small scenes)
                const size_t far_min_bytes = far_min_env ? (size_t)std::strtoull(far_min_env, nullptr, 10) : ((size_t)256 << 20);
                const bool far_able = ptk_trace_has_far(sc) && ctx->n_nodes_up * sizeof(PtNode) + ctx->n_tris_up * sizeof(PtTri) > far_min_bytes;
                if (ctx->trace_far >= 0 || !far_able) ctx->trace_far_choice = (ctx->trace_far > 0 && ptk_trace_has_far(sc)) ? 1 : 0;
                int bounce_i = 0;
                PtQueues Q_last_trace = Q;
                auto trace = [&]() -> hipError_t {
                    PtQueues Qt = Q;
                    if (shadow_sorted) Qt.shadow = shadow_sorted;
                    shadow_sorted = nullptr;
                    if (cont_sorted) Qt.cur = cont_sorted;
                    cont_sorted = nullptr;
                    Q_last_trace = Qt;
                    return ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, sc, ctx->paths, Qt, cnt, ctx->d_spill.as<uint32_t>(), ctx->spill_depth, err,
                                     ctx->trace_far_choice > 0 ? 1 : 0);
                };
                auto far_trial = [&](hipEvent_t near_a, hipEvent_t near_b) -> pt_status {      // right after bounce 1's launch by k_trace (timed by near_a .. near_b)
                    hipEvent_t fa = get_event(ctx, ev_i), fb = get_event(ctx, ev_i + 1);
                    if (!fa || !fb) return ctx->fail(PT_ERR_DEVICE, "hipEventCreate failed");
                    ev_i += 2;
                    if (!ctx->d_cnt_save.p) PT_HIP(ctx->d_cnt_save.alloc(sizeof(PtCounters)));
                    PT_HIP(hipMemcpyAsync(ctx->d_cnt_save.p, cnt, sizeof(PtCounters), hipMemcpyDeviceToDevice, ctx->stream));
                    PT_HIP(hipMemsetAsync(Q.counts + PT_Q_SEG_TICKET0, 0, 8u * 32u * 4u, ctx->stream));          // the launch's work tickets
                    PT_HIP(hipEventRecord(fa, ctx->stream));
                    PT_HIP(ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, sc, ctx->paths, Q_last_trace, cnt, ctx->d_spill.as<uint32_t>(), ctx->spill_depth, err, 1));
                    PT_HIP(hipEventRecord(fb, ctx->stream));
                    PT_HIP(hipMemcpyAsync(cnt, ctx->d_cnt_save.p, sizeof(PtCounters), hipMemcpyDeviceToDevice, ctx->stream));
                    PT_HIP(hipStreamSynchronize(ctx->stream));
                    float near_ms = 0, far_ms = 0;
                    PT_HIP(hipEventElapsedTime(&near_ms, near_a, near_b));
                    PT_HIP(hipEventElapsedTime(&far_ms, fa, fb));
                    ctx->trace_far_choice = far_ms < 0.97f * near_ms ? 1 : 0;          // (a tie stays with the kernel everybody else runs)
                    if (std::getenv("PBRTGPU_BUILD_TRACE")) std::fprintf(stderr, "[trace] trial on bounce 1: k_trace %.2f ms, k_trace_far %.2f ms -> %s\n", near_ms, far_ms, ctx->trace_far_choice ? "k_trace_far" : "k_trace");
                    return PT_OK;
                };
                // after a bounce's shading: order its shadow rays by where they start (pt_raysort.hip); costs one counter read-back
                auto sort_shadow = [&]() -> pt_status {
                    if (!Q.shadow_key && !sort_cont) return PT_OK;
                    uint32_t qc[PT_Q_SHADOW + 1];             // one read-back: the next bounce's continuation rays (prep has moved next to cur) and its shadow rays
                    qc[PT_Q_SHADOW] = 0;
                    PT_HIP(hipMemcpyAsync(qc, Q.counts, Q.shadow_key ? sizeof(qc) : 4 * (PT_Q_CUR + 1u), hipMemcpyDeviceToHost, ctx->stream));      // (the shadow list unsorted: the first word alone)
                    PT_HIP(hipStreamSynchronize(ctx->stream));
                    const uint32_t n_sh = qc[PT_Q_SHADOW], n_next = qc[PT_Q_CUR];
                    if (sort_cont && n_next >= (uint32_t)ctx->sort_cont_min) {
                        // Q.next holds the list k_shade has just written: its rays' keys, then the ordered copy into the spare list
                        PT_HIP(ptk_cont_keys(ctx->stream, ctx->grid_wide, sc, ctx->paths, Q.next, n_next, ctx->d_csort_keys[0].as<uint32_t>()));
                        PT_HIP(ptk_sort_rays_keep(ctx->stream, Q.next, cont_spare, ctx->d_csort_keys[0].as<uint32_t>(), ctx->d_csort_keys[1].as<uint32_t>(),
                                                  ctx->d_csort_temp.p, ctx->d_csort_temp.bytes, n_next));
                        if (sort_cont == 2) std::swap(Q.next, cont_spare);      // shading walks the ordered list too; the old list is the spare now
                        else cont_sorted = cont_spare;
                    }
                    if (!Q.shadow_key || n_sh < (uint32_t)ctx->sort_shadow_min) return PT_OK;
                    PT_HIP(ptk_sort_shadow_rays(ctx->stream, Q.shadow, ctx->d_sort_ids.as<uint32_t>(), ctx->d_sort_keys[0].as<uint32_t>(),
                                                ctx->d_sort_keys[1].as<uint32_t>(), ctx->d_sort_temp.p, ctx->d_sort_temp.bytes, n_sh, &shadow_sorted));
                    return PT_OK;
                };
                auto bounce = [&]() -> pt_status {
                    const bool timed = !no_events;
                    hipEvent_t a = nullptr, b = nullptr, c = nullptr;
                    if (bounce_log) {          // diagnostic (PBRTGPU_BOUNCE_LOG): queue sizes going into this bounce; costs a sync per bounce
                        uint32_t qc[PT_Q_PROBE + 1];
                        PT_HIP(hipMemcpyAsync(qc, Q.counts, sizeof(qc), hipMemcpyDeviceToHost, ctx->stream));
                        PT_HIP(hipStreamSynchronize(ctx->stream));
                        bounce_counts.push_back({qc[PT_Q_CUR], qc[PT_Q_SHADOW], qc[PT_Q_PROBE]});
                    }
                    if (timed) {
                        a = get_event(ctx, ev_i); b = get_event(ctx, ev_i + 1); c = get_event(ctx, ev_i + 2);
                        if (!a || !b || !c) return ctx->fail(PT_ERR_DEVICE, "hipEventCreate failed");
                        spans.push_back({ev_i, 0});
                        ev_i += 3;
                        PT_HIP(hipEventRecord(a, ctx->stream));
                    }
                    const bool trial = ctx->trace_far_choice < 0 && bounce_i == 1;
                    hipEvent_t ta = a, tb = b;
                    if (trial && !timed) {
                        ta = get_event(ctx, ev_i); tb = get_event(ctx, ev_i + 1);
                        if (!ta || !tb) return ctx->fail(PT_ERR_DEVICE, "hipEventCreate failed");
                        ev_i += 2;
                        PT_HIP(hipEventRecord(ta, ctx->stream));
                    }
                    PT_HIP(trace());
                    if (timed || trial) PT_HIP(hipEventRecord(tb, ctx->stream));
                    if (trial) {
                        const pt_status ts = far_trial(ta, tb);
                        if (ts != PT_OK) return ts;
                    }
                    bounce_i++;
                    PT_HIP(ptk_nee_resolve(ctx->stream, ctx->grid_wide, sc, ctx->paths, Q));
                    ctx->trace_launches++;
                    PT_HIP(ptk_prep(ctx->stream, Q, 0));
                    if (ctx->grid_lazy) {
                        const pt_status gs = fill_light_grid(ctx, Q);
                        if (gs != PT_OK) return gs;
                    }
                    PT_HIP(ptk_shade(ctx->stream, ctx->grid_shade, ctx->sc, ctx->paths, Q, cnt, ctx->nee_split, ctx->shade_local < 0 ? ((ctx->scene_has_lobe_materials && !ctx->sc.textured) ? 1 : 0) : ctx->shade_local));
                    PT_HIP(ptk_prep(ctx->stream, Q, 1));
                    {
                        const pt_status ss = sort_shadow();
                        if (ss != PT_OK) return ss;
                    }
                    if (timed) PT_HIP(hipEventRecord(c, ctx->stream));
                    std::swap(Q.cur, Q.next);
                    return PT_OK;
                };
                for (int b = 0; b <= sc.max_depth; b++)
                    if ((st = bounce()) != PT_OK) return st;
                // paths can outlive max_depth+1 iterations only by passing through material-less
                // surfaces; the last SHADE may also have queued NEE work.  One counter read per pass.
                for (;;) {
                    uint32_t counts[PT_Q_NEE + 1];
                    PT_HIP(hipMemcpyAsync(counts, Q.counts, sizeof(counts), hipMemcpyDeviceToHost, ctx->stream));
                    PT_HIP(hipStreamSynchronize(ctx->stream));
                    if (counts[PT_Q_CUR] == 0 && counts[PT_Q_NEE] == 0) break;
                    if (counts[PT_Q_CUR] == 0) {     // only NEE resolves left
                        PT_HIP(trace());
                        PT_HIP(ptk_nee_resolve(ctx->stream, ctx->grid_wide, sc, ctx->paths, Q));
                        ctx->trace_launches++;
                        PT_HIP(ptk_prep(ctx->stream, Q, 0));
                    } else if ((st = bounce()) != PT_OK) return st;
                }
            }
            PT_HIP(ptk_film(ctx->stream, ctx->grid_wide, sc, ctx->paths, d_pix, n_pix, ns, ctx->d_own.as<float4>(), ctx->d_spillfilm.as<float4>(),
                            d_radiance_out ? d_radiance_out + (size_t)c0 * spp * 3 : nullptr, s0, spp));
        }
    }
    PT_HIP(hipEventRecord(ev_end, ctx->stream));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t herr = 0;
    PT_HIP(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    if (herr) {
        PT_HIP(hipMemsetAsync(err, 0, 4, ctx->stream));
        PT_HIP(hipStreamSynchronize(ctx->stream));
        if (herr & 1u) return ctx->fail(PT_ERR_DEVICE, "traversal stack overflow (BVH deeper than the computed bound)");
        return ctx->fail(PT_ERR_UNSUPPORTED, "a path asked the Halton sampler for more than its 1000 dimensions: the reference panics there (halton.rs:103-107); "
                                             "lower maxdepth or use the Sobol' sampler");
    }
#ifdef PT_STACK_HIST
    {   // diagnostic build: histogram of the traversal stack depth at node visits (k_trace's lean visit)
        uint32_t hist[64];
        PT_HIP(hipMemcpy(hist, ctx->d_spill.as<uint32_t>() + 1024, sizeof(hist), hipMemcpyDeviceToHost));
        PT_HIP(hipMemsetAsync(ctx->d_spill.as<uint32_t>() + 1024, 0, sizeof(hist), ctx->stream)); PT_HIP(hipStreamSynchronize(ctx->stream));
        double tot = 0, cum = 0;
        for (uint32_t v : hist) tot += v;
        std::fprintf(stderr, "[stack depth at node visits]");
        for (int k = 0; k < 64; k++) { cum += hist[k]; if (hist[k]) std::fprintf(stderr, " %d:%.4f", k, cum / tot); }
        std::fprintf(stderr, "\n");
    }
#endif
#ifdef PT_PROFILE_PHASES
    {   // diagnostic build (tools/tune_trace.sh "prof:-DPT_PROFILE_PHASES:3"): where a traversal wave's clocks go
        unsigned long long pr[16];
        PT_HIP(hipMemcpy(pr, ctx->d_spill.p, sizeof(pr), hipMemcpyDeviceToHost));
        PT_HIP(hipMemsetAsync(ctx->d_spill.p, 0, sizeof(pr), ctx->stream)); PT_HIP(hipStreamSynchronize(ctx->stream));
        const double tot = (double)pr[11];
        std::fprintf(stderr, "[phases] wave clocks %.3e | node rounds %llu (%.1f lanes): issue %.1f%% wait %.1f%% finish %.1f%% = %.0f clk/round | leaf rounds %llu (%.1f items): issue %.1f%% wait %.1f%% finish %.1f%% = %.0f clk/round | general visits %llu: %.1f%% | service %.1f%%\n",
                     tot, pr[3], pr[3] + pr[13] ? (double)pr[4] / (double)(pr[3] + pr[13]) : 0.0, 100.0 * pr[0] / tot, 100.0 * pr[1] / tot, 100.0 * pr[2] / tot,
                     pr[3] ? (double)(pr[0] + pr[1] + pr[2]) / (double)pr[3] : 0.0, pr[8], pr[8] ? (double)pr[9] / (double)pr[8] : 0.0, 100.0 * pr[5] / tot, 100.0 * pr[6] / tot,
                     100.0 * pr[7] / tot, pr[8] ? (double)(pr[5] + pr[6] + pr[7]) / (double)pr[8] : 0.0, pr[13], 100.0 * pr[12] / tot, 100.0 * pr[10] / tot);
        std::fprintf(stderr, "[phases] service in parts: prefetch state machine %.1f%%, retiring + handing out rays %.1f%% (%.0f + %.0f clocks per iteration)\n", 100.0 * pr[14] / tot,
                     100.0 * pr[15] / tot, (double)pr[14] / (double)(pr[3] + pr[8] + pr[13] + 1), (double)pr[15] / (double)(pr[3] + pr[8] + pr[13] + 1));
    }
#endif
    {   // diagnostic build -DPT_PROFILE_SHADE: where a shading wave's clocks go
        unsigned long long sp[16];
        if (ptk_shade_prof_read(sp)) {
            double tot = 0;
            for (int k = 0; k < 16; k++) if (k != 12) tot += (double)sp[k];
            static const char* names[16] = {"surface", "emit+bsdf setup", "grid+pick light", "u_light,u_scat", "light sample+f", "bsdf MIS+probe", "pend stores",
                                            "u continuation", "continuation", "compaction", "sample tables", "ticket", "", "wait list[item]", "wait path state", "wait record+index"};
            std::fprintf(stderr, "[shade phases] %llu iterations of 64 paths, %.0f clocks each:", sp[12], sp[12] ? tot / (double)sp[12] : 0.0);
            for (int k = 0; k < 16; k++) if (k != 12 && sp[k]) std::fprintf(stderr, " %s %.1f%%", names[k], 100.0 * (double)sp[k] / tot);
            std::fprintf(stderr, "\n");
        }
    }
    float ms = 0;
    PT_HIP(hipEventElapsedTime(&ms, ev_begin, ev_end));
    ctx->render_ms += ms;
    size_t span_i = 0;
    for (auto& sp : spans) {
        float t_ms = 0, s_ms = 0;
        PT_HIP(hipEventElapsedTime(&t_ms, ctx->ev[sp.first], ctx->ev[sp.first + 1]));
        PT_HIP(hipEventElapsedTime(&s_ms, ctx->ev[sp.first + 1], ctx->ev[sp.first + 2]));
        ctx->trace_ms += t_ms;
        ctx->shade_ms += s_ms;
        if (bounce_log && span_i < bounce_counts.size()) {
            const auto& q = bounce_counts[span_i];
            std::fprintf(stderr, "[bounce %zu] rays: %u continuation + %u shadow + %u probe | trace %.3f ms (%.1f Mrays/s) | resolve + shade %.3f ms (%.2f ns per shaded path)\n", span_i, q[0],
                         q[1], q[2], t_ms, ((double)q[0] + q[1] + q[2]) / (t_ms * 1e3), s_ms, q[0] ? s_ms * 1e6 / q[0] : 0.0);
        }
        span_i++;
    }
    ctx->xyzw_committed = false;
    return PT_OK;
}

pt_status pt_render(pt_context* ctx, const pt_tile* tiles, uint32_t n_tiles) {
    if (!ctx || (n_tiles > 0 && !tiles)) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    return render_tiles(ctx, tiles, n_tiles, nullptr);
}

static pt_status film_to_xyzw(pt_context* ctx) {
    if (ctx->xyzw_committed) return PT_OK;
    uint32_t n = ctx->film_w * ctx->film_h;
    PT_HIP(ptk_film_xyzw(ctx->stream, ctx->d_own.as<float4>(), ctx->d_spillfilm.as<float4>(), ctx->d_xyzw.as<float4>(), n));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

pt_status pt_film_download_xyzw(pt_context* ctx, float* out) {
    if (!ctx || !out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    pt_status st = film_to_xyzw(ctx);
    if (st != PT_OK) return st;
    PT_HIP(hipMemcpy(out, ctx->d_xyzw.p, (size_t)ctx->film_w * ctx->film_h * 16, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_film_device_xyzw(pt_context* ctx, void** dev_ptr, size_t* n_floats) {
    if (!ctx || !dev_ptr || !n_floats) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    pt_status st = film_to_xyzw(ctx);
    if (st != PT_OK) return st;
    *dev_ptr = ctx->d_xyzw.p;
    *n_floats = (size_t)ctx->film_w * ctx->film_h * 4;
    return PT_OK;
}

pt_status pt_film_commit_xyzw(pt_context* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    ctx->xyzw_committed = true;
    return PT_OK;
}

// The exchange step without a collective: add another rank's film, handed over in host memory (a host whose ranks talk through MPI or
// shared memory rather than RCCL; two contexts on one device, which RCCL refuses to put in one communicator).
pt_status pt_film_add_xyzw(pt_context* ctx, const float* xyzw) {
    if (!ctx || !xyzw) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    pt_status st = film_to_xyzw(ctx);
    if (st != PT_OK) return st;
    const uint32_t n = ctx->film_w * ctx->film_h;
    DevBuf other;
    PT_HIP(other.alloc((size_t)n * 16));
    PT_HIP(hipMemcpyAsync(other.p, xyzw, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ptk_film_add(ctx->stream, ctx->d_xyzw.as<float4>(), other.as<float4>(), n));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->xyzw_committed = true;
    return PT_OK;
}

// The path's one exchange step, inside the library: sum the per-rank XYZW films over an RCCL communicator the caller owns
// (Film::merge_film_tile across GPUs, film.rs:219-241).  librccl is looked up at run time -- first the copy already mapped into
// the process (the communicator must come from that very copy: a torch process carries its own), else the system's.
}  // extern "C"
namespace {
struct RcclApi {
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
};
// Resolved once, by whichever thread asks first: `pbrt_gpu --gpus N` calls pt_film_allreduce from N host threads at the same time, and a
// thread that saw a half-filled table would skip the collective its peers are already blocked in.  (A function-local static's
// initialiser runs under the language's own once-guard.)
const RcclApi& rccl_api() {
    static const RcclApi api = [] {
        RcclApi a;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(h, "ncclAllReduce"));
        a.reduce = reinterpret_cast<decltype(a.reduce)>(dlsym(h, "ncclReduce"));
        a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
        return a;
    }();
    return api;
}
}  // namespace
extern "C" {

pt_status pt_film_allreduce(pt_context* ctx, void* nccl_comm, int root) {
    if (!ctx || !nccl_comm) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    const RcclApi& api = rccl_api();
    if (!api.all_reduce || !api.reduce) return ctx->fail(PT_ERR_UNSUPPORTED, "librccl.so.1 (ncclAllReduce / ncclReduce) not found");
    pt_status st = film_to_xyzw(ctx);
    if (st != PT_OK) return st;
    const size_t n = (size_t)ctx->film_w * ctx->film_h * 4;
    const int kFloat = 7, kSum = 0;          // ncclFloat32, ncclSum (rccl.h)
    int rc = root < 0 ? api.all_reduce(ctx->d_xyzw.p, ctx->d_xyzw.p, n, kFloat, kSum, nccl_comm, ctx->stream)
                      : api.reduce(ctx->d_xyzw.p, ctx->d_xyzw.p, n, kFloat, kSum, root, nccl_comm, ctx->stream);
    if (rc != 0) return ctx->fail(PT_ERR_DEVICE, std::string("RCCL: ") + (api.error_string ? api.error_string(rc) : "error"));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->xyzw_committed = true;
    return PT_OK;
}

pt_status pt_film_resolve_rgb(pt_context* ctx, float* rgb_out) {
    if (!ctx || !rgb_out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    pt_status st = film_to_xyzw(ctx);
    if (st != PT_OK) return st;
    uint32_t n = ctx->film_w * ctx->film_h;
    PT_HIP(ptk_film_rgb(ctx->stream, ctx->d_xyzw.as<float4>(), ctx->d_rgb.as<float>(), n, ctx->sc.film.scale));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    PT_HIP(hipMemcpy(rgb_out, ctx->d_rgb.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    return PT_OK;
}

static pt_status trace_batch(pt_context* ctx, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out, uint8_t* occ, int any_hit) {
    if (!ctx || !o || !d || !tmax || (!any_hit && !out) || (any_hit && !occ)) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    DevBuf d_o, d_d, d_t, d_out;
    PT_HIP(d_o.alloc((size_t)n * 12)); PT_HIP(d_d.alloc((size_t)n * 12)); PT_HIP(d_t.alloc((size_t)n * 4));
    PT_HIP(d_out.alloc((size_t)n * (any_hit ? 1 : sizeof(pt_hit))));
    PT_HIP(hipMemcpy(d_o.p, o, (size_t)n * 12, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_d.p, d, (size_t)n * 12, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_t.p, tmax, (size_t)n * 4, hipMemcpyHostToDevice));
    PT_HIP(hipMemsetAsync(ctx->d_ticket.p, 0, 16, ctx->stream));
    hipEvent_t a = get_event(ctx, 0), b = get_event(ctx, 1);
    PT_HIP(hipEventRecord(a, ctx->stream));
    PT_HIP(ptk_trace_batch(ctx->stream, ctx->grid_trace, ctx->sc, n, d_o.as<float>(), d_d.as<float>(), d_t.as<float>(), any_hit ? nullptr : d_out.as<pt_hit>(),
                           any_hit ? d_out.as<uint8_t>() : nullptr, any_hit, ctx->d_ticket.as<uint32_t>(), ctx->d_counters.as<PtCounters>(),
                           ctx->d_spill.as<uint32_t>(), ctx->spill_depth, ctx->d_err.as<uint32_t>()));
    PT_HIP(hipEventRecord(b, ctx->stream));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    PT_HIP(hipEventElapsedTime(&ms, a, b));
    ctx->trace_ms += ms;
    ctx->trace_launches++;
    uint32_t herr = 0;
    PT_HIP(hipMemcpy(&herr, ctx->d_err.p, 4, hipMemcpyDeviceToHost));
    if (herr) { PT_HIP(hipMemsetAsync(ctx->d_err.p, 0, 4, ctx->stream)); PT_HIP(hipStreamSynchronize(ctx->stream)); return ctx->fail(PT_ERR_DEVICE, "traversal stack overflow"); }
    if (any_hit) PT_HIP(hipMemcpy(occ, d_out.p, n, hipMemcpyDeviceToHost));
    else PT_HIP(hipMemcpy(out, d_out.p, (size_t)n * sizeof(pt_hit), hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_trace_closest(pt_context* ctx, uint32_t n, const float* o, const float* d, const float* tmax, pt_hit* out) {
    return trace_batch(ctx, n, o, d, tmax, out, nullptr, 0);
}
pt_status pt_trace_any(pt_context* ctx, uint32_t n, const float* o, const float* d, const float* tmax, uint8_t* occluded_out) {
    return trace_batch(ctx, n, o, d, tmax, nullptr, occluded_out, 1);
}

// Caller rays through the kernel the renderer itself traces with (see k_wavefront_results): ray i becomes path slot i; the
// continuation / shadow / probe queues list the rays of each kind in caller order, exactly the mix one bounce launches.
pt_status pt_trace_wavefront(pt_context* ctx, uint32_t n, const float* o, const float* d, const float* tmax, const uint8_t* kind, pt_hit* out,
                             uint8_t* occluded_out) {
    if (!ctx || !o || !d || !tmax || !kind || !out || !occluded_out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    std::vector<float> o4((size_t)n * 4), d4((size_t)n * 4);
    std::vector<uint32_t> ids[3];
    for (uint32_t i = 0; i < n; i++) {
        if (kind[i] < 1 || kind[i] > 3) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "ray kind must be 1 (continuation), 2 (shadow) or 3 (probe)");
        ids[kind[i] - 1].push_back(i);
        for (int k = 0; k < 3; k++) { o4[4 * (size_t)i + k] = o[3 * (size_t)i + k]; d4[4 * (size_t)i + k] = d[3 * (size_t)i + k]; }
        o4[4 * (size_t)i + 3] = tmax[i];
        d4[4 * (size_t)i + 3] = 0.0f;
    }
    pt_status st;
    if ((st = ensure_pool(ctx, std::max<size_t>(n, 65536))) != PT_OK) return st;
    const PtPaths& P = ctx->paths;
    float4* const dst_o[3] = {P.ray_o, P.sh_o, P.pr_o};
    float4* const dst_d[3] = {P.ray_d, P.sh_d, P.pr_d};
    for (int k = 0; k < 3; k++) {
        PT_HIP(hipMemcpyAsync(dst_o[k], o4.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(hipMemcpyAsync(dst_d[k], d4.data(), (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    }
    PtQueues Q;
    Q.cur = ctx->d_qa.as<uint32_t>(); Q.next = ctx->d_qb.as<uint32_t>();
    Q.nee = ctx->d_qnee.as<uint32_t>(); Q.counts = ctx->d_counts.as<uint32_t>(); Q.sorted = ctx->d_qsorted.as<uint32_t>();
    Q.shadow = ctx->d_qshadow.as<uint32_t>(); Q.probe = ctx->d_qprobe.as<uint32_t>(); Q.shadow_key = nullptr; Q.bin = nullptr;
    uint32_t* const dst_q[3] = {Q.cur, Q.shadow, Q.probe};
    for (int k = 0; k < 3; k++)
        if (!ids[k].empty()) PT_HIP(hipMemcpyAsync(dst_q[k], ids[k].data(), ids[k].size() * 4, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint32_t> counts(PT_COUNTS_WORDS, 0u);
    counts[PT_Q_CUR] = (uint32_t)ids[0].size(); counts[PT_Q_SHADOW] = (uint32_t)ids[1].size(); counts[PT_Q_PROBE] = (uint32_t)ids[2].size();
    PT_HIP(hipMemcpyAsync(Q.counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    DevBuf d_kind, d_out, d_occ;
    PT_HIP(d_kind.alloc(n)); PT_HIP(d_out.alloc((size_t)n * sizeof(pt_hit))); PT_HIP(d_occ.alloc(n));
    PT_HIP(hipMemcpyAsync(d_kind.p, kind, n, hipMemcpyHostToDevice, ctx->stream));
    hipEvent_t a = get_event(ctx, 0), b = get_event(ctx, 1);
    PT_HIP(hipEventRecord(a, ctx->stream));
    PT_HIP(ptk_trace(ctx->stream, ctx->grid_trace, ctx->grid_trace_dist, ctx->sc, P, Q, ctx->d_counters.as<PtCounters>(), ctx->d_spill.as<uint32_t>(), ctx->spill_depth,
                     ctx->d_err.as<uint32_t>()));
    PT_HIP(hipEventRecord(b, ctx->stream));
    PT_HIP(ptk_wavefront_results(ctx->stream, ctx->grid_wide, ctx->sc, P, n, d_kind.as<uint8_t>(), d_out.as<pt_hit>(), d_occ.as<uint8_t>()));
    PT_HIP(hipMemsetAsync(Q.counts, 0, PT_COUNTS_WORDS * 4, ctx->stream));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    PT_HIP(hipEventElapsedTime(&ms, a, b));
    ctx->trace_ms += ms;
    ctx->trace_launches++;
    uint32_t herr = 0;
    PT_HIP(hipMemcpy(&herr, ctx->d_err.p, 4, hipMemcpyDeviceToHost));
    if (herr) { PT_HIP(hipMemsetAsync(ctx->d_err.p, 0, 4, ctx->stream)); PT_HIP(hipStreamSynchronize(ctx->stream)); return ctx->fail(PT_ERR_DEVICE, "traversal stack overflow"); }
    PT_HIP(hipMemcpy(out, d_out.p, (size_t)n * sizeof(pt_hit), hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(occluded_out, d_occ.p, n, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_generate_camera_rays(pt_context* ctx, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, float* out_o, float* out_d,
                                  float* out_pfilm) {
    if (!ctx || !pixel_xy || !sample_index || !out_o || !out_d || !out_pfilm) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    DevBuf d_px, d_si, d_o, d_d, d_pf;
    PT_HIP(d_px.alloc((size_t)n * 8)); PT_HIP(d_si.alloc((size_t)n * 4));
    PT_HIP(d_o.alloc((size_t)n * 12)); PT_HIP(d_d.alloc((size_t)n * 12)); PT_HIP(d_pf.alloc((size_t)n * 8));
    PT_HIP(hipMemcpy(d_px.p, pixel_xy, (size_t)n * 8, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_si.p, sample_index, (size_t)n * 4, hipMemcpyHostToDevice));
    PT_HIP(ptk_camera_rays(ctx->stream, ctx->sc, n, d_px.as<int32_t>(), d_si.as<uint32_t>(), d_o.as<float>(), d_d.as<float>(), d_pf.as<float>()));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    PT_HIP(hipMemcpy(out_o, d_o.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(out_d, d_d.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(out_pfilm, d_pf.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_bsdf_eval(pt_context* ctx, uint32_t material, uint32_t n, const float* wo, const float* wi, uint32_t flags, float* f_out, float* pdf_out) {
    if (!ctx || !wo || !wi || !f_out || !pdf_out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    if (material >= ctx->n_materials) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "material index out of range");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    DevBuf d_wo, d_wi, d_f, d_pdf;
    PT_HIP(d_wo.alloc((size_t)n * 12)); PT_HIP(d_wi.alloc((size_t)n * 12)); PT_HIP(d_f.alloc((size_t)n * 12)); PT_HIP(d_pdf.alloc((size_t)n * 4));
    PT_HIP(hipMemcpy(d_wo.p, wo, (size_t)n * 12, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_wi.p, wi, (size_t)n * 12, hipMemcpyHostToDevice));
    PT_HIP(ptk_bsdf_eval(ctx->stream, ctx->sc, material, n, d_wo.as<float>(), d_wi.as<float>(), flags, d_f.as<float>(), d_pdf.as<float>()));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    PT_HIP(hipMemcpy(f_out, d_f.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(pdf_out, d_pdf.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_bsdf_sample(pt_context* ctx, uint32_t material, uint32_t n, const float* wo, const float* u, uint32_t flags, float* f_out, float* wi_out,
                         float* pdf_out, uint32_t* type_out) {
    if (!ctx || !wo || !u || !f_out || !wi_out || !pdf_out || !type_out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    if (material >= ctx->n_materials) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "material index out of range");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    DevBuf d_wo, d_u, d_f, d_wi, d_pdf, d_t;
    PT_HIP(d_wo.alloc((size_t)n * 12)); PT_HIP(d_u.alloc((size_t)n * 8)); PT_HIP(d_f.alloc((size_t)n * 12)); PT_HIP(d_wi.alloc((size_t)n * 12));
    PT_HIP(d_pdf.alloc((size_t)n * 4)); PT_HIP(d_t.alloc((size_t)n * 4));
    PT_HIP(hipMemcpy(d_wo.p, wo, (size_t)n * 12, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_u.p, u, (size_t)n * 8, hipMemcpyHostToDevice));
    PT_HIP(ptk_bsdf_sample(ctx->stream, ctx->sc, material, n, d_wo.as<float>(), d_u.as<float>(), flags, d_f.as<float>(), d_wi.as<float>(), d_pdf.as<float>(),
                           d_t.as<uint32_t>()));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    PT_HIP(hipMemcpy(f_out, d_f.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(wi_out, d_wi.p, (size_t)n * 12, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(pdf_out, d_pdf.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    PT_HIP(hipMemcpy(type_out, d_t.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_sobol_samples(pt_context* ctx, uint32_t n, const int32_t* pixel_xy, const uint32_t* sample_index, const uint32_t* dim, float* out) {
    if (!ctx || !pixel_xy || !sample_index || !dim || !out) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    if (n == 0) return PT_OK;
    (void)hipSetDevice(ctx->device);
    DevBuf d_px, d_si, d_dim, d_out;
    PT_HIP(d_px.alloc((size_t)n * 8)); PT_HIP(d_si.alloc((size_t)n * 4)); PT_HIP(d_dim.alloc((size_t)n * 4)); PT_HIP(d_out.alloc((size_t)n * 4));
    PT_HIP(hipMemcpy(d_px.p, pixel_xy, (size_t)n * 8, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_si.p, sample_index, (size_t)n * 4, hipMemcpyHostToDevice));
    PT_HIP(hipMemcpy(d_dim.p, dim, (size_t)n * 4, hipMemcpyHostToDevice));
    PT_HIP(ptk_sobol_samples(ctx->stream, ctx->sc, n, d_px.as<int32_t>(), d_si.as<uint32_t>(), d_dim.as<uint32_t>(), d_out.as<float>()));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    PT_HIP(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_radiance_samples(pt_context* ctx, const pt_tile* tile, float* out_rgb) {
    if (!ctx || !tile || !out_rgb) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    size_t npx = (size_t)std::max(0, tile->x1 - tile->x0) * (size_t)std::max(0, tile->y1 - tile->y0);
    if (npx == 0) return PT_OK;
    size_t n = npx * ctx->sc.sobol.spp * 3;
    DevBuf d_rad;
    PT_HIP(d_rad.alloc(n * 4));
    // the film is left untouched: render into scratch copies
    size_t fpx = (size_t)ctx->film_w * ctx->film_h;
    DevBuf keep_own, keep_spill;
    PT_HIP(keep_own.alloc(fpx * 16)); PT_HIP(keep_spill.alloc(fpx * 16));
    PT_HIP(hipMemcpyAsync(keep_own.p, ctx->d_own.p, fpx * 16, hipMemcpyDeviceToDevice, ctx->stream));
    PT_HIP(hipMemcpyAsync(keep_spill.p, ctx->d_spillfilm.p, fpx * 16, hipMemcpyDeviceToDevice, ctx->stream));
    pt_status st = render_tiles(ctx, tile, 1, d_rad.as<float>());
    PT_HIP(hipMemcpyAsync(ctx->d_own.p, keep_own.p, fpx * 16, hipMemcpyDeviceToDevice, ctx->stream));
    PT_HIP(hipMemcpyAsync(ctx->d_spillfilm.p, keep_spill.p, fpx * 16, hipMemcpyDeviceToDevice, ctx->stream));
    PT_HIP(hipStreamSynchronize(ctx->stream));
    if (st != PT_OK) return st;
    PT_HIP(hipMemcpy(out_rgb, d_rad.p, n * 4, hipMemcpyDeviceToHost));
    return PT_OK;
}

pt_status pt_bvh_leaf_order(const pt_scene_desc* d, uint32_t* order_out, uint32_t* n_nodes, uint32_t* n_leaves, uint32_t* max_stack) {
    if (!d || !order_out || (d->n_triangles == 0 && d->n_spheres == 0)) return PT_ERR_INVALID_ARGUMENT;
    if (d->n_triangles > 0 && (!d->P || !d->indices || !d->tri_mesh || !d->meshes)) return PT_ERR_INVALID_ARGUMENT;
    if (d->n_spheres > 0 && !d->spheres) return PT_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> tri_flags(d->n_triangles, 0);
    std::vector<PtSphere> sph;
    std::vector<ptbvh::SpherePrim> sprims;
    build_spheres(d, sph, sprims);
    for (ptbvh::SpherePrim& sp : sprims) sp.flags = PT_TRI_SPHERE;
    ptbvh::Result bvh;
    if (!ptbvh::build(d->P, d->indices, tri_flags.data(), d->n_triangles, sprims.data(), d->n_spheres, d->split_method,
                      d->max_node_prims > 0 ? d->max_node_prims : 4, &bvh))
        return PT_ERR_UNSUPPORTED;
    for (uint32_t r = 0; r < d->n_triangles + d->n_spheres; r++) order_out[r] = bvh.tris[r].prim;
    if (n_nodes) *n_nodes = (uint32_t)bvh.nodes.size();
    if (n_leaves) *n_leaves = bvh.n_leaves;
    if (max_stack) *max_stack = bvh.max_stack;
    return PT_OK;
}

pt_status pt_set_bvh_build(pt_context* ctx, int where) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (where != PT_BVH_BUILD_AUTO && where != PT_BVH_BUILD_HOST && where != PT_BVH_BUILD_DEVICE) return ctx->fail(PT_ERR_INVALID_ARGUMENT, "unknown BVH build location");
    ctx->bvh_build_where = where;
    return PT_OK;
}

pt_status pt_scene_bvh_digest(pt_context* ctx, uint64_t* nodes_digest, uint64_t* records_digest) {
    if (!ctx || !nodes_digest || !records_digest) return PT_ERR_INVALID_ARGUMENT;
    if (!ctx->have_scene) return ctx->fail(PT_ERR_NO_SCENE, "no scene uploaded");
    (void)hipSetDevice(ctx->device);
    auto digest = [&](const DevBuf& b, size_t bytes, uint64_t* out) -> pt_status {
        std::vector<unsigned char> h(bytes);
        if (bytes) PT_HIP(hipMemcpy(h.data(), b.p, bytes, hipMemcpyDeviceToHost));
        uint64_t x = 1469598103934665603ull;
        for (size_t i = 0; i < bytes; i++) { x ^= h[i]; x *= 1099511628211ull; }
        *out = x;
        return PT_OK;
    };
    pt_status st = digest(ctx->d_nodes, ctx->n_nodes_up * sizeof(PtNode), nodes_digest);
    if (st != PT_OK) return st;
    return digest(ctx->d_tris, ctx->n_tris_up * sizeof(PtTri), records_digest);
}

pt_status pt_get_counters(pt_context* ctx, pt_counters* out) {
    if (!ctx || !out) return PT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    std::memset(out, 0, sizeof(*out));
    if (ctx->d_counters.p) {
        PtCounters c;
        PT_HIP(hipMemcpy(&c, ctx->d_counters.p, sizeof(c), hipMemcpyDeviceToHost));
        out->camera_rays = c.camera_rays; out->regular_rays = c.regular_rays; out->shadow_rays = c.shadow_rays;
        out->nodes_visited = c.nodes; out->tris_tested = c.tris; out->path_vertices = c.vertices;
        out->nodes_from_lds = c.nodes_lds;
    }
    out->trace_launches = ctx->trace_launches;
    out->trace_ms = ctx->trace_ms; out->shade_ms = ctx->shade_ms; out->render_ms = ctx->render_ms;
    return PT_OK;
}

pt_status pt_reset_counters(pt_context* ctx) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_counters.p) {
        PT_HIP(hipMemsetAsync(ctx->d_counters.p, 0, sizeof(PtCounters), ctx->stream));
        PT_HIP(hipStreamSynchronize(ctx->stream));
    }
    ctx->trace_launches = 0;
    ctx->trace_ms = ctx->shade_ms = ctx->render_ms = 0;
    return PT_OK;
}

}  // extern "C"
