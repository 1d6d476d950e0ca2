/*
 * pbrtgpu_host.h -- host-side front end of libpbrtgpu.so: the .pbrt scene-description
 * parser and the scene context that flattens it into a pt_scene_desc.
 *
 * It stands in for (paths relative to the reference tree)
 *   pbrt_parse_file / pbrt_parse_string     src/core/parser/parser.rs:60-135
 *   trait ParseContext                      src/core/api/parse_context.rs:5-66
 *   SceneContext (the ParseContext that builds the scene)
 *                                           src/core/api/scene_context/scene_context.rs:817-1396
 * for the subset of the format the accelerated path renders:
 *   shapes      "trianglemesh", "plymesh" (ASCII / binary / gzip PLY), "sphere" (full or clipped by zmin / zmax /
 *               phimax, under any affine CTM; as a scene object or as an area light)
 *   materials   matte, plastic, mirror, glass, metal, uber, substrate with constant parameters; named
 *               materials; colours as rgb, .spd "spectrum" files or "blackbody" (metal defaults to the
 *               measured copper spectrum); Texture "constant" / "scale" / "mix" (folded when constant), "checkerboard"
 *               (2-D with uv / spherical / cylindrical / planar mapping, closed-form or point-sampled; 3-D), "uv",
 *               "bilerp", "dots", "fbm", "wrinkled", "windy", "marble" on colour parameters and Matte's sigma
 *               and "imagemap" (PNG / TGA / PFM / EXR files; MIP pyramid, EWA or trilinear) on colour parameters and Matte's
 *               sigma (evaluated per hit with ray differentials); "bumpmap" with any of those float textures or a number
 *   lights      diffuse area lights
 *   camera      perspective;  filters box / gaussian / mitchell / sinc / triangle
 *   samplers    halton (the default), sobol;  integrators path, ao;  accelerator bvh (sah, hlbvh, middle, equal)
 *   the full transform / attribute stack, named coordinate systems, Include, ObjectBegin / ObjectEnd / ObjectInstance
 * Anything else is reported as PT_ERR_UNSUPPORTED with a message naming the directive -- never silently
 * approximated.
 *
 * The C++ class interface (ParseContext with the reference's method names) is in
 * pbrt-r3_amd/csrc/host/pth_parse_context.h; this header is the C ABI over it.
 */
#ifndef PBRTGPU_HOST_H
#define PBRTGPU_HOST_H

#include "pbrtgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

/* Parse a .pbrt file (Include resolved relative to the including file).  On failure *out is NULL
 * and err (if non-NULL) receives a message. */
pt_status pth_parse_file(const char* filename, pth_scene** out, char* err, size_t err_cap);
/* Command-line options of the reference that act while the scene is being assembled (src/bin/pbrt.rs:360-366,
 * :234-238): --quick (pixelsamples 1, a quarter of the film resolution per axis), --quick_full_resolution,
 * --pixelsamples (0 = keep the scene's). */
typedef struct {
    int32_t quick;
    int32_t quick_full_resolution;
    int32_t pixelsamples;
    int32_t reserved;
} pth_options;
pt_status pth_parse_file_opts(const char* filename, const pth_options* opts, pth_scene** out, char* err, size_t err_cap);
/* Parse scene text; work_dir is the base for Include (may be NULL). */
pt_status pth_parse_string(const char* text, const char* work_dir, pth_scene** out, char* err, size_t err_cap);
/* The flattened scene; pointers stay valid until pth_scene_free. */
const pt_scene_desc* pth_scene_get_desc(const pth_scene* s);
/* Film "filename" parameter (default "pbrt.exr"). */
const char* pth_scene_output_filename(const pth_scene* s);
/* Command-line overrides of the reference CLI (src/bin/pbrt.rs:234-244). */
void pth_scene_set_pixelsamples(pth_scene* s, int spp);
/* Non-fatal diagnostics collected while parsing (one per line). */
const char* pth_scene_warnings(const pth_scene* s);
void pth_scene_free(pth_scene* s);

/* Parser in isolation: runs `text` through a context that logs every ParseContext callback, one
 * directive per line (cf. the reference's PrintContext, --cat).  On a syntax error out holds the message. */
pt_status pth_parse_to_log(const char* text, const char* work_dir, char* out, size_t cap);

/* Planck's law as the front end evaluates it for "blackbody" spectra: out[i] = blackbody(lambda_nm[i], t_kelvin) in W / (m^2 sr m)
 * (src/core/spectrum/blackbody.rs:3-22; zero for t <= 0).  Exported so that the reference's own known-answer test
 * (tests/spectrum.rs:10-40) runs against this restatement. */
void pth_blackbody(const double* lambda_nm, int n, double t_kelvin, double* out);

/* Linear-RGB float image as PFM (the smallest float format the reference can also write,
 * src/core/imageio/write_image.rs:59-76). */
pt_status pth_write_pfm(const char* path, const float* rgb, int width, int height);
/* Film::write_image (film.rs:440-484 -> imageio/write_image.rs:59-76) by file extension: ".exr" (32-bit float scanline
 * OpenEXR, uncompressed; data window = the cropped pixel bounds at (x0, y0) inside a full_w x full_h display window),
 * ".png" (8-bit, to_byte = clamp(255 * gamma_correct(v)) as write_image.rs:16-18), ".pfm".  Anything else:
 * PT_ERR_UNSUPPORTED. */
pt_status pth_write_image(const char* path, const float* rgb, int width, int height, int x0, int y0, int full_w, int full_h);

/* ---- live display: the reference's `--display-server host:port` (TevDisplay, src/displays/tev/; Film::render_start /
 * update_display / render_end, src/core/film/film.rs:278-360, :424-438).  The packets are tev's IPC CreateImage / UpdateImage as
 * IPCGen writes them (src/displays/tev/display.rs:147-235); an update is cut into 128 x 128 tiles (display.rs:266-345). */
typedef struct pth_display pth_display;
pt_status pth_display_connect(const char* host_port, pth_display** out, char* err, size_t err_cap);
/* Film::render_start: create the image (the film's full resolution, channels R G B) under `title` (the film's filename). */
pt_status pth_display_start(pth_display* d, const char* title, uint32_t full_width, uint32_t full_height);
/* Film::update_display: a width x height block of linear RGB (3 floats per pixel, rows first) whose top-left pixel is (x, y). */
pt_status pth_display_update(pth_display* d, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const float* rgb);
void pth_display_close(pth_display* d);
/* The two packets by themselves (tests): return the packet size, write it when cap suffices. */
size_t pth_tev_create_packet(const char* name, uint32_t width, uint32_t height, unsigned char* out, size_t cap);
size_t pth_tev_update_packet(const char* name, uint32_t x, uint32_t y, uint32_t width, uint32_t height, const float* rgb, unsigned char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
