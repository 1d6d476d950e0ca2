/* c_abi_smoke.c -- the C ABI driven from plain C: no HIP headers, no C++, no ctypes.
 *
 * This is the call sequence of INTEGRATION.md section 3 (what a Rust `impl Integrator for GpuPathIntegrator` does behind
 * src/core/integrator/integrator.rs:6-9), written against include/pbrtgpu.h alone and compiled with gcc: fill a pt_scene_desc by
 * hand (the Cornell box of BASELINE.md config 1, the values SceneContext::pbrt_shape / make_scene would flatten,
 * src/core/api/scene_context/scene_context.rs:1201-1318), check the ABI version, upload, render every tile, read the film back.
 * tests/test_gpu_c_abi.py compiles it, runs it on the GPU and compares the film with the ctypes path on the same scene.
 *
 *   c_abi_smoke RES SPP OUT.xyzw      writes RES*RES*4 floats {X, Y, Z, weight}; prints the ray counters
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pbrtgpu.h"

#define MAX_V 256
#define MAX_T 128
static float P[3 * MAX_V];
static uint32_t idx[3 * MAX_T], tri_mesh[MAX_T];
static pt_mesh meshes[64];
static uint32_t n_v, n_t, n_m;

/* Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [p0 p1 p2 p3] under the identity CTM */
static void quad(int material, int area_light, const float* p0, const float* p1, const float* p2, const float* p3) {
    const float* p[4] = {p0, p1, p2, p3};
    for (int i = 0; i < 4; i++) memcpy(&P[3 * (n_v + i)], p[i], 12);
    const uint32_t q[6] = {0, 1, 2, 0, 2, 3};
    for (int i = 0; i < 6; i++) idx[3 * n_t + i] = n_v + q[i];
    tri_mesh[n_t] = tri_mesh[n_t + 1] = n_m;
    meshes[n_m].flags = PT_MESH_TWO_SIDED;       /* "twosided" defaults to true (triangle.rs:707); no N / S / uv given */
    meshes[n_m].material = material;
    meshes[n_m].area_light = area_light;
    meshes[n_m].object = 0;
    n_v += 4; n_t += 2; n_m += 1;
}
static void quad9(int material, int area_light, float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz, float dx, float dy,
                  float dz) {
    const float a[3] = {ax, ay, az}, b[3] = {bx, by, bz}, c[3] = {cx, cy, cz}, d[3] = {dx, dy, dz};
    quad(material, area_light, a, b, c, d);
}
/* six quads from the four top corners and a height */
static void cuboid(int material, const float top[4][3], float height) {
    float bt[4][3];
    for (int i = 0; i < 4; i++) { bt[i][0] = top[i][0]; bt[i][1] = top[i][1] - height; bt[i][2] = top[i][2]; }
    quad(material, -1, top[0], top[1], top[2], top[3]);
    quad(material, -1, bt[3], bt[2], bt[1], bt[0]);
    for (int i = 0; i < 4; i++) {
        const int j = (i + 1) % 4;
        quad(material, -1, top[i], bt[i], bt[j], top[j]);
    }
}

static int fail(pt_context* ctx, const char* what, int st) {
    fprintf(stderr, "c_abi_smoke: %s failed with status %d: %s\n", what, st, ctx ? pt_last_error(ctx) : "(no context)");
    if (ctx) pt_context_destroy(ctx);
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: c_abi_smoke RES SPP OUT.xyzw\n"); return 2; }
    const int res = atoi(argv[1]), spp = atoi(argv[2]);
    if (pt_abi_version() != PT_ABI_VERSION) { fprintf(stderr, "c_abi_smoke: header is ABI %d, library is ABI %d\n", PT_ABI_VERSION, pt_abi_version()); return 1; }

    pt_material mats[3];
    memset(mats, 0, sizeof(mats));                /* zero = constant parameters, no textures */
    const float kd[3][3] = {{0.73f, 0.73f, 0.73f}, {0.12f, 0.45f, 0.15f}, {0.65f, 0.05f, 0.05f}};     /* white, green, red */
    for (int i = 0; i < 3; i++) { mats[i].type = PT_MATERIAL_MATTE; memcpy(mats[i].kd, kd[i], 12); mats[i].sigma = 0.0f; }
    pt_area_light light;
    memset(&light, 0, sizeof(light));
    light.L[0] = 17.0f; light.L[1] = 12.0f; light.L[2] = 4.0f; light.two_sided = 0; light.n_samples = 1;

    quad9(0, -1, 552.8f, 0, 0, 0, 0, 0, 0, 0, 559.2f, 549.6f, 0, 559.2f);                         /* floor */
    quad9(0, -1, 556, 548.8f, 0, 556, 548.8f, 559.2f, 0, 548.8f, 559.2f, 0, 548.8f, 0);           /* ceiling */
    quad9(0, -1, 549.6f, 0, 559.2f, 0, 0, 559.2f, 0, 548.8f, 559.2f, 556, 548.8f, 559.2f);        /* back */
    quad9(1, -1, 0, 0, 559.2f, 0, 0, 0, 0, 548.8f, 0, 0, 548.8f, 559.2f);                         /* right, green */
    quad9(2, -1, 552.8f, 0, 0, 549.6f, 0, 559.2f, 556, 548.8f, 559.2f, 556, 548.8f, 0);           /* left, red */
    const float short_top[4][3] = {{130, 165, 65}, {82, 165, 225}, {240, 165, 272}, {290, 165, 114}};
    const float tall_top[4][3] = {{423, 330, 247}, {265, 330, 296}, {314, 330, 456}, {472, 330, 406}};
    cuboid(0, short_top, 165.0f);
    cuboid(0, tall_top, 330.0f);
    quad9(0, 0, 343, 548.75f, 227, 343, 548.75f, 332, 213, 548.75f, 332, 213, 548.75f, 227);      /* the light, facing down */

    pt_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.n_vertices = n_v; d.P = P;
    d.n_triangles = n_t; d.indices = idx; d.tri_mesh = tri_mesh;
    d.n_meshes = n_m; d.meshes = meshes;
    d.n_materials = 3; d.materials = mats;
    d.n_area_lights = 1; d.area_lights = &light;
    d.split_method = PT_SPLIT_SAH; d.max_node_prims = 4;
    /* LookAt 278 273 -800  278 273 0  0 1 0: the camera looks down +z with +y up, so camera-to-world is a translation */
    const float c2w[16] = {1, 0, 0, 278, 0, 1, 0, 273, 0, 0, 1, -800, 0, 0, 0, 1};
    memcpy(d.camera_to_world, c2w, sizeof(c2w));
    d.fov = 39.3f;
    d.screen_window[0] = -1; d.screen_window[1] = 1; d.screen_window[2] = -1; d.screen_window[3] = 1;     /* square film */
    d.lens_radius = 0.0f; d.focal_distance = 1e6f; d.shutter_open = 0.0f; d.shutter_close = 1.0f;
    d.xres = res; d.yres = res;
    d.crop_window[0] = 0; d.crop_window[1] = 1; d.crop_window[2] = 0; d.crop_window[3] = 1;
    d.filter_radius[0] = d.filter_radius[1] = 0.5f;
    for (int i = 0; i < 256; i++) d.filter_table[i] = 1.0f;                                                 /* BoxFilter::evaluate == 1 */
    d.film_scale = 1.0f; d.max_sample_luminance = INFINITY;
    d.sampler = PT_SAMPLER_SOBOL; d.spp = spp; d.max_depth = 5; d.rr_threshold = 1.0f; d.light_strategy = PT_LIGHTS_SPATIAL;
    d.integrator = PT_INTEGRATOR_PATH;

    pt_context* ctx = NULL;
    int st = pt_context_create(0, &ctx);
    if (st != PT_OK) { fprintf(stderr, "c_abi_smoke: no HIP device (status %d); the library has no CPU fallback\n", st); return 3; }
    if ((st = pt_scene_upload(ctx, &d)) != PT_OK) return fail(ctx, "pt_scene_upload", st);
    pt_scene_info info;
    if ((st = pt_scene_info_get(ctx, &info)) != PT_OK) return fail(ctx, "pt_scene_info_get", st);
    if ((st = pt_film_clear(ctx)) != PT_OK) return fail(ctx, "pt_film_clear", st);
    if ((st = pt_render(ctx, NULL, 0)) != PT_OK) return fail(ctx, "pt_render", st);       /* every 16x16 tile, all spp */
    const int w = info.cropped_bounds[2] - info.cropped_bounds[0], h = info.cropped_bounds[3] - info.cropped_bounds[1];
    float* xyzw = (float*)malloc((size_t)w * h * 16);
    float* rgb = (float*)malloc((size_t)w * h * 12);
    if ((st = pt_film_download_xyzw(ctx, xyzw)) != PT_OK) return fail(ctx, "pt_film_download_xyzw", st);
    if ((st = pt_film_resolve_rgb(ctx, rgb)) != PT_OK) return fail(ctx, "pt_film_resolve_rgb", st);
    pt_counters c;
    if ((st = pt_get_counters(ctx, &c)) != PT_OK) return fail(ctx, "pt_get_counters", st);
    /* an error path: a tile outside the sample bounds is refused with a message, and the context stays usable */
    pt_tile bad = {info.sample_bounds[2], info.sample_bounds[3], info.sample_bounds[2] + 16, info.sample_bounds[3] + 16};
    st = pt_render(ctx, &bad, 1);
    if (st != PT_ERR_INVALID_ARGUMENT || !pt_last_error(ctx)[0]) return fail(ctx, "pt_render(tile outside the film) should be PT_ERR_INVALID_ARGUMENT", st);
    FILE* f = fopen(argv[3], "wb");
    if (!f || fwrite(xyzw, 16, (size_t)w * h, f) != (size_t)w * h) { fprintf(stderr, "c_abi_smoke: cannot write %s\n", argv[3]); return 1; }
    fclose(f);
    double mean = 0;
    for (int i = 0; i < 3 * w * h; i++) mean += rgb[i];
    printf("film %d %d spp %d lights %u nodes %u camera_rays %llu regular_rays %llu shadow_rays %llu mean_rgb %.6f\n", w, h, info.spp, info.n_lights, info.n_nodes,
           (unsigned long long)c.camera_rays, (unsigned long long)c.regular_rays, (unsigned long long)c.shadow_rays, mean / (3.0 * w * h));
    free(xyzw); free(rgb);
    pt_context_destroy(ctx);
    return 0;
}
