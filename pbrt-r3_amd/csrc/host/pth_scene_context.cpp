// pth_scene_context.cpp -- the ParseContext that turns a scene description into the flattened
// pt_scene_desc the device library consumes.  It restates, for the accelerated subset, what the
// reference's SceneContext does between the parser callbacks and Integrator::render:
//   transform / attribute stacks, named coordinate systems   scene_context.rs:821-943, :1037-1082
//   options (Film, Camera, Sampler, ...) + WorldEnd factories  scene_context.rs:953-1016, :606-729
//   pbrt_shape for "trianglemesh"                              scene_context.rs:1201-1318,
//                                                             shapes/triangle.rs:696-868
//   material lookup with per-shape overrides                   scene_context.rs:224-296, materials/matte.rs:56-61
//   diffuse area lights                                        lights/diffuse.rs:165-194
// All arithmetic that reaches the renderer (CTM products, vertex transforms, filter tables) is f32 in
// the reference's operation order.  Unsupported directives set an error naming them.
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <sstream>

#include "../../../include/pbrtgpu_host.h"
#include "../pt_host_math.h"
#include "pth_parse_context.h"
#include <memory>
#include <set>
#include "pth_ply.h"
#include "pth_texture_image.h"
#include "pth_spectrum.h"

namespace pth {

namespace {

const float kPi = 3.14159265358979323846f;

struct V3f { float x, y, z; };
inline V3f sub(V3f a, V3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3f crossf(V3f a, V3f b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
inline float lenf(V3f a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3f normf(V3f a) { float l = lenf(a); return {a.x / l, a.y / l, a.z / l}; }

// Matrix4x4::rotate (matrix4x4.rs:107-131) with transpose as inverse
Xf xf_rotate(float theta, float x, float y, float z) {
    V3f a = normf({x, y, z});
    float r = theta * (kPi / 180.0f);
    float s = std::sin(r), c = std::cos(r);
    Xf t;
    t.m = m_identity();
    t.m.a[0] = a.x * a.x + (1.0f - a.x * a.x) * c;
    t.m.a[1] = a.x * a.y * (1.0f - c) - a.z * s;
    t.m.a[2] = a.x * a.z * (1.0f - c) + a.y * s;
    t.m.a[4] = a.x * a.y * (1.0f - c) + a.z * s;
    t.m.a[5] = a.y * a.y + (1.0f - a.y * a.y) * c;
    t.m.a[6] = a.y * a.z * (1.0f - c) - a.x * s;
    t.m.a[8] = a.x * a.z * (1.0f - c) - a.y * s;
    t.m.a[9] = a.y * a.z * (1.0f - c) + a.x * s;
    t.m.a[10] = a.z * a.z + (1.0f - a.z * a.z) * c;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t.inv.a[4 * i + j] = t.m.a[4 * j + i];
    return t;
}
// Transform::look_at (transform.rs:68-83): m = world->camera (inverse of the built matrix), minv = camera->world
bool xf_look_at(const float e[3], const float l[3], const float u[3], Xf* out) {
    V3f pos = {e[0], e[1], e[2]}, look = {l[0], l[1], l[2]};
    V3f up = normf({u[0], u[1], u[2]});
    V3f dir = normf(sub(look, pos));
    V3f right = crossf(up, dir);
    if (lenf(right) == 0.0f) return false;
    right = normf(right);
    V3f nup = normf(crossf(dir, right));
    M44 c2w = m_identity();
    c2w.a[0] = right.x; c2w.a[1] = nup.x; c2w.a[2] = dir.x; c2w.a[3] = pos.x;
    c2w.a[4] = right.y; c2w.a[5] = nup.y; c2w.a[6] = dir.y; c2w.a[7] = pos.y;
    c2w.a[8] = right.z; c2w.a[9] = nup.z; c2w.a[10] = dir.z; c2w.a[11] = pos.z;
    M44 w2c;
    if (!m_inverse(c2w, &w2c)) return false;
    out->m = w2c;
    out->inv = c2w;
    return true;
}
inline void xf_point(const M44& m, const float* p, float* o) {     // matrix4x4.rs:311-324
    float x = p[0], y = p[1], z = p[2];
    float xp = m.a[0] * x + m.a[1] * y + m.a[2] * z + m.a[3];
    float yp = m.a[4] * x + m.a[5] * y + m.a[6] * z + m.a[7];
    float zp = m.a[8] * x + m.a[9] * y + m.a[10] * z + m.a[11];
    float wp = m.a[12] * x + m.a[13] * y + m.a[14] * z + m.a[15];
    if (wp == 1.0f) { o[0] = xp; o[1] = yp; o[2] = zp; }
    else { o[0] = xp / wp; o[1] = yp / wp; o[2] = zp / wp; }
}
inline void xf_vector(const M44& m, const float* v, float* o) {
    float x = v[0], y = v[1], z = v[2];
    o[0] = m.a[0] * x + m.a[1] * y + m.a[2] * z;
    o[1] = m.a[4] * x + m.a[5] * y + m.a[6] * z;
    o[2] = m.a[8] * x + m.a[9] * y + m.a[10] * z;
}
inline void xf_normal(const M44& minv, const float* n, float* o) {  // Transform::transform_normal: transpose of the inverse
    float x = n[0], y = n[1], z = n[2];
    o[0] = minv.a[0] * x + minv.a[4] * y + minv.a[8] * z;
    o[1] = minv.a[1] * x + minv.a[5] * y + minv.a[9] * z;
    o[2] = minv.a[2] * x + minv.a[6] * y + minv.a[10] * z;
}
inline bool swaps_handedness(const M44& m) {
    float det = m.a[0] * (m.a[5] * m.a[10] - m.a[6] * m.a[9]) - m.a[1] * (m.a[4] * m.a[10] - m.a[6] * m.a[8]) + m.a[2] * (m.a[4] * m.a[9] - m.a[5] * m.a[8]);
    return det < 0.0f;
}

struct TransformSet { Xf t[2]; };
const unsigned kAllBits = 3, kStartBit = 1, kEndBit = 2;

struct MaterialInstance { std::string name; ParamSet params; bool none = false; };
struct GraphicsState {
    MaterialInstance material;                                   // default: matte, no params (graphics_state.rs:61-91)
    std::map<std::string, MaterialInstance> named_materials;
    std::string area_light_name;
    ParamSet area_light_params;
    bool reverse_orientation = false;
};

}  // namespace

class GpuSceneContext : public ParseContext {
public:
    // ---- flattened output
    std::vector<float> P, N, S, UV;
    std::vector<uint32_t> indices, tri_mesh;
    std::vector<pt_mesh> meshes;
    std::vector<pt_sphere> spheres;                                     // Shape "sphere", spliced into the primitive order by before_triangle
    std::vector<pt_instance> instances;
    std::map<std::string, uint32_t> object_ids;
    std::set<uint32_t> object_redefined;
    uint32_t cur_object = 0;                                            // 0: world; k: inside ObjectBegin of object k - 1
    uint32_t n_extra = 0;                                               // creation counter of spheres and instances
    bool quick_render = false, quick_full_resolution = false;          // PbrtOptions (--quick, --quick_full_resolution)
    std::vector<pt_material> materials;
    std::map<std::string, float> float_textures;                        // constant-folded named textures (global, see pbrt_texture)
    std::map<std::string, std::array<float, 3>> spectrum_textures;
    std::map<std::string, std::string> unsupported_textures;          // name -> class of textures that are not on the path
    std::vector<std::unique_ptr<Pyramid>> pyramids;                     // MIP pyramids of the imagemap textures (stable addresses)
    std::vector<pt_image> images;
    std::vector<pt_texture> textures;                                   // textures that depend on the hit: evaluated on the device
    std::map<std::string, uint32_t> node_float_textures, node_spectrum_textures;   // name -> index into textures
    std::vector<pt_area_light> area_lights;
    bool any_n = false, any_s = false, any_uv = false;
    pt_scene_desc desc;
    std::string out_filename = "pbrt.exr";
    std::string error, warnings;
    bool world_ended = false;

    GpuSceneContext() {
        TransformSet id;
        id.t[0].m = id.t[0].inv = id.t[1].m = id.t[1].inv = m_identity();
        transforms.push_back(id);
        bits.push_back(kAllBits);
        gstates.push_back(GraphicsState());
        gstates.back().material.name = "matte";
        std::memset(&desc, 0, sizeof(desc));
    }

    void fail(const std::string& m) { if (error.empty()) error = m; }
    void warn(const std::string& m) { warnings += m + "\n"; }

    // ---- transforms (scene_context.rs:821-943)
    void mul(const Xf& t) { for (int i = 0; i < 2; i++) if (bits.back() & (1u << i)) transforms.back().t[i] = xf_mul(transforms.back().t[i], t); }
    void set(const Xf& t) { for (int i = 0; i < 2; i++) if (bits.back() & (1u << i)) transforms.back().t[i] = t; }
    void pbrt_identity() override { Xf t; t.m = t.inv = m_identity(); set(t); }
    void pbrt_translate(float dx, float dy, float dz) override { mul(xf_translate(dx, dy, dz)); }
    void pbrt_rotate(float angle, float ax, float ay, float az) override { mul(xf_rotate(angle, ax, ay, az)); }
    void pbrt_scale(float sx, float sy, float sz) override { mul(xf_scale(sx, sy, sz)); }
    void pbrt_look_at(float ex, float ey, float ez, float lx, float ly, float lz, float ux, float uy, float uz) override {
        float e[3] = {ex, ey, ez}, l[3] = {lx, ly, lz}, u[3] = {ux, uy, uz};
        Xf t;
        if (!xf_look_at(e, l, u, &t)) { fail("LookAt: up vector and viewing direction are parallel"); return; }
        mul(t);
    }
    bool from16(const std::vector<float>& t, Xf* out) {
        if (t.size() != 16) return false;
        M44 m;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m.a[4 * i + j] = t[4 * j + i];    // file order is column-major
        out->m = m;
        return m_inverse(m, &out->inv);
    }
    void pbrt_concat_transform(const std::vector<float>& t) override { Xf x; if (from16(t, &x)) mul(x); else warn("Singular matrix in ConcatTransform"); }
    void pbrt_transform(const std::vector<float>& t) override { Xf x; if (from16(t, &x)) set(x); else warn("Singular matrix in Transform"); }
    void pbrt_coordinate_system(const std::string& name) override { named_cs[name] = transforms.back(); }
    void pbrt_coord_sys_transform(const std::string& name) override {
        auto it = named_cs.find(name);
        if (it == named_cs.end()) { warn("Couldn't find named coordinate system \"" + name + "\""); return; }
        for (int i = 0; i < 2; i++) if (bits.back() & (1u << i)) transforms.back().t[i] = it->second.t[i];
    }
    void pbrt_active_transform_all() override { bits.back() = kAllBits; }
    void pbrt_active_transform_end_time() override { bits.back() = kEndBit; }
    void pbrt_active_transform_start_time() override { bits.back() = kStartBit; }
    void pbrt_transform_times(float, float) override {}

    // ---- options
    void pbrt_pixel_filter(const std::string& name, const ParamSet& p) override { filter_name = name; filter_params = p; }
    void pbrt_film(const std::string& name, const ParamSet& p) override { film_name = name; film_params = p; }
    void pbrt_sampler(const std::string& name, const ParamSet& p) override { sampler_name = name; sampler_params = p; }
    void pbrt_accelerator(const std::string& name, const ParamSet& p) override { accel_name = name; accel_params = p; }
    void pbrt_integrator(const std::string& name, const ParamSet& p) override { integrator_name = name; integrator_params = p; }
    void pbrt_camera(const std::string& name, const ParamSet& p) override {
        camera_name = name; camera_params = p;
        TransformSet inv;                                          // "camera" = inverse of the CTM (scene_context.rs:1001-1005)
        for (int i = 0; i < 2; i++) inv.t[i] = xf_inverse(transforms.back().t[i]);
        named_cs["camera"] = inv;
        have_camera = true;
    }
    void pbrt_make_named_medium(const std::string&, const ParamSet&) override { warn("MakeNamedMedium ignored: PathIntegrator does not handle media (path.rs:128)"); }
    void pbrt_medium_interface(const std::string&, const std::string&) override { warn("MediumInterface ignored: PathIntegrator does not handle media"); }

    // ---- world block
    void pbrt_world_begin() override {
        Xf id; id.m = id.inv = m_identity();
        transforms.back().t[0] = transforms.back().t[1] = id;
        bits.back() = kAllBits;
        named_cs["world"] = transforms.back();
    }
    void pbrt_attribute_begin() override { gstates.push_back(gstates.back()); transforms.push_back(transforms.back()); bits.push_back(bits.back()); }
    void pbrt_attribute_end() override {
        if (gstates.size() <= 1) { warn("Unmatched AttributeEnd"); return; }
        gstates.pop_back(); transforms.pop_back(); bits.pop_back();
    }
    void pbrt_transform_begin() override { transforms.push_back(transforms.back()); bits.push_back(bits.back()); }
    void pbrt_transform_end() override {
        if (transforms.size() <= 1) { warn("Unmatched TransformEnd"); return; }
        transforms.pop_back(); bits.pop_back();
    }
    // Texture directive (scene_context.rs:1077-1114).  "constant", and "scale" / "mix" of constants, fold to a value here;
    // "checkerboard" (2-D with any mapping, 3-D), "uv", "bilerp", and scale / mix over those, become pt_texture nodes that
    // the device evaluates at each hit; image maps and the noise textures are reported when something uses them.  As in
    // the reference the texture tables are NOT scoped by AttributeBegin/End (pbrt_texture writes through the shared map
    // without the copy-on-write that named materials get).
    struct Child { bool is_node = false; uint32_t node = 0; float v[3] = {0, 0, 0}; };
    Child child_float(const ParamSet& p, const char* n, float dflt) {            // tp.get_float_texture(n, dflt)
        Child c; uint32_t t = 0; float v = dflt;
        lookup_float(p, p, n, &v, &t);
        if (t) { c.is_node = true; c.node = t - 1; } else c.v[0] = c.v[1] = c.v[2] = v;
        return c;
    }
    Child child_rgb(const ParamSet& p, const char* n, float dflt) {              // tp.get_spectrum_texture(n, Spectrum(dflt))
        Child c; uint32_t t = 0; float v[3] = {dflt, dflt, dflt};
        lookup_rgb(p, p, n, v, &t);
        if (t) { c.is_node = true; c.node = t - 1; } else { c.v[0] = v[0]; c.v[1] = v[1]; c.v[2] = v[2]; }
        return c;
    }
    static void set_child(pt_texture& t, int k, const Child& c) {
        t.tex[k] = c.is_node ? (int32_t)c.node : -1;
        if (!c.is_node) { t.value[k][0] = c.v[0]; t.value[k][1] = c.v[1]; t.value[k][2] = c.v[2]; }
    }
    // create_texture_mapping2d (mapping2d.rs:178-214)
    bool set_mapping2d(const ParamSet& p, pt_texture& t) {
        const std::string mapping = p.find_one_string("mapping", "uv");
        const Xf& t2w = transforms.back().t[0];
        std::memcpy(t.world_to_texture, t2w.inv.a, 64);
        t.su = 1.0f; t.sv = 1.0f;
        t.v1[0] = 1.0f; t.v2[1] = 1.0f;
        if (mapping == "uv") {
            t.mapping = PT_MAPPING_UV;
            t.su = p.find_one_float("uscale", 1.0f); t.sv = p.find_one_float("vscale", 1.0f);
            t.du = p.find_one_float("udelta", 0.0f); t.dv = p.find_one_float("vdelta", 0.0f);
        } else if (mapping == "spherical") t.mapping = PT_MAPPING_SPHERICAL;
        else if (mapping == "cylindrical") t.mapping = PT_MAPPING_CYLINDRICAL;
        else if (mapping == "planar") {
            t.mapping = PT_MAPPING_PLANAR;
            if (const std::vector<float>* v = p.get_vectors("v1")) if (v->size() >= 3) { t.v1[0] = (*v)[0]; t.v1[1] = (*v)[1]; t.v1[2] = (*v)[2]; }
            if (const std::vector<float>* v = p.get_vectors("v2")) if (v->size() >= 3) { t.v2[0] = (*v)[0]; t.v2[1] = (*v)[1]; t.v2[2] = (*v)[2]; }
            t.du = p.find_one_float("udelta", 0.0f); t.dv = p.find_one_float("vdelta", 0.0f);
        } else { warn("2D texture mapping \"" + mapping + "\" unknown"); return false; }
        return true;
    }
    void pbrt_texture(const std::string& name, const std::string& type, const std::string& tex_class, const ParamSet& p) override {
        if (!error.empty()) return;
        const bool is_float = type == "float", is_spec = type == "color" || type == "rgb" || type == "spectrum";
        if (!is_float && !is_spec) { warn("Texture type \"" + type + "\" unknown."); return; }
        auto forget = [&]() {
            float_textures.erase(name); spectrum_textures.erase(name); unsupported_textures.erase(name);
            (is_float ? node_float_textures : node_spectrum_textures).erase(name);
        };
        auto child = [&](const char* n, float dflt) { return is_float ? child_float(p, n, dflt) : child_rgb(p, n, dflt); };
        auto add_node = [&](const pt_texture& t) {
            forget();
            textures.push_back(t);
            (is_float ? node_float_textures : node_spectrum_textures)[name] = (uint32_t)textures.size() - 1;
        };
        pt_texture t;
        std::memset(&t, 0, sizeof(t));
        t.tex[0] = t.tex[1] = t.tex[2] = -1;
        if (tex_class == "constant") {
            float c[3] = {1.0f, 1.0f, 1.0f};
            if (is_float) c[0] = c[1] = c[2] = p.find_one_float("value", 1.0f); else spectrum_from(p, "value", c);
            if (!error.empty()) return;
            forget();
            if (is_float) float_textures[name] = c[0]; else spectrum_textures[name] = {c[0], c[1], c[2]};
        } else if (tex_class == "scale" || tex_class == "mix") {
            const bool mix = tex_class == "mix";
            Child t1 = child("tex1", 1.0f), t2 = child("tex2", 1.0f), amt;
            if (mix) amt = child_float(p, "amount", 0.5f);
            if (!error.empty()) return;
            if (!t1.is_node && !t2.is_node && !amt.is_node) {                 // folds (scale.rs:20-24, mix.rs:27-32)
                float v[3];
                for (int i = 0; i < 3; i++) v[i] = mix ? t1.v[i] * (1.0f - amt.v[0]) + t2.v[i] * amt.v[0] : t1.v[i] * t2.v[i];
                forget();
                if (is_float) float_textures[name] = v[0]; else spectrum_textures[name] = {v[0], v[1], v[2]};
                return;
            }
            t.type = mix ? PT_TEX_MIX : PT_TEX_SCALE;
            set_child(t, 0, t1); set_child(t, 1, t2);
            if (mix) set_child(t, 2, amt);
            add_node(t);
        } else if (tex_class == "checkerboard") {                             // checkerboard.rs:134-185
            const int dim = p.find_one_int("dimension", 2);
            if (dim != 2 && dim != 3) { warn(std::to_string(dim) + " dimensional checkerboard texture not supported"); return; }
            Child t1 = child("tex1", 1.0f), t2 = child("tex2", 0.0f);
            if (!error.empty()) return;
            set_child(t, 0, t1); set_child(t, 1, t2);
            if (dim == 2) {
                t.type = PT_TEX_CHECKERBOARD_2D;
                if (!set_mapping2d(p, t)) return;
                const std::string aa = p.find_one_string("aamode", "closedform");
                if (aa == "none") t.aa_none = 1;
                else if (aa != "closedform") { warn("Antialiasing mode \"" + aa + "\" not understood by \"Checkerboard2DTexture\""); return; }
            } else {
                t.type = PT_TEX_CHECKERBOARD_3D;
                std::memcpy(t.world_to_texture, transforms.back().t[0].m.a, 64);   // IdentityMapping3D::new(tex2world), as the reference passes it
            }
            add_node(t);
        } else if (tex_class == "imagemap") {                                 // imagemap.rs:84-229
            t.type = PT_TEX_IMAGEMAP;
            if (!set_mapping2d(p, t)) return;
            std::string file = p.find_one_string("filename", "");
            if (file.empty()) { fail("Texture \"" + name + "\": imagemap without a filename"); return; }
            if (file[0] != '/' && !p.base_dir.empty()) file = p.base_dir + "/" + file;
            auto wrap_mode = [](const std::string& m) { return m == "black" ? PT_WRAP_BLACK : (m == "clamp" ? PT_WRAP_CLAMP : PT_WRAP_REPEAT); };
            const std::string wrap = p.find_one_string("wrap", "repeat");
            t.swrap = wrap_mode(p.find_one_string("swrap", wrap));
            t.twrap = wrap_mode(p.find_one_string("twrap", wrap));
            t.max_anisotropy = p.find_one_float("maxanisotropy", 8.0f);
            t.trilinear = p.find_one_bool("trilinear", false) ? 1 : 0;
            const float scale = p.find_one_float("scale", 1.0f);
            auto has_ext = [&](const char* e) { size_t n = std::strlen(e); return file.size() > n && file.compare(file.size() - n, n, e) == 0 && file[file.size() - n - 1] == '.'; };
            bool gamma = !has_ext("exr");
            if (file.find("_alpha") != std::string::npos || file.find("_bump") != std::string::npos) gamma = false;      // "Fixed in pbrt-r3" (:118-122)
            gamma = p.find_one_bool("gamma", gamma);
            RgbImage img;
            std::string ierr;
            if (!read_image_file(file, &img, &ierr)) { fail("Texture \"" + name + "\": " + ierr); return; }
            pyramids.emplace_back(new Pyramid());
            build_pyramid(img, is_float ? 1 : 3, scale, gamma, t.swrap, t.twrap, pyramids.back().get());
            t.image = (int32_t)pyramids.size() - 1;
            add_node(t);
        } else if (tex_class == "dots") {                                     // dots.rs:45-62
            t.type = PT_TEX_DOTS;
            if (!set_mapping2d(p, t)) return;
            Child t1 = child("tex1", 1.0f), t2 = child("tex2", 0.0f);
            if (!error.empty()) return;
            set_child(t, 0, t1); set_child(t, 1, t2);
            add_node(t);
        } else if (tex_class == "fbm" || tex_class == "wrinkled" || tex_class == "windy" || (tex_class == "marble" && is_spec)) {
            // fbm.rs:35-56, wrinkled.rs:35-56, windy.rs:31-48, marble.rs:82-95 (no float marble: create_texture.rs:52-54)
            t.type = tex_class == "fbm" ? PT_TEX_FBM : (tex_class == "wrinkled" ? PT_TEX_WRINKLED : (tex_class == "windy" ? PT_TEX_WINDY : PT_TEX_MARBLE));
            std::memcpy(t.world_to_texture, transforms.back().t[0].m.a, 64);       // IdentityMapping3D::new(tex2world)
            if (t.type != PT_TEX_WINDY) {
                t.octaves = p.find_one_int("octaves", 8);
                t.omega = p.find_one_float("roughness", 0.5f);
            }
            if (t.type == PT_TEX_MARBLE) {
                t.scale = p.find_one_float("scale", 1.0f);
                t.variation = p.find_one_float("variation", 0.2f);
            }
            add_node(t);
        } else if (tex_class == "uv" && is_spec) {                            // uv.rs:27-33
            t.type = PT_TEX_UV;
            if (!set_mapping2d(p, t)) return;
            add_node(t);
        } else if (tex_class == "bilerp") {                                   // bilerp.rs:34-63
            t.type = PT_TEX_BILERP;
            if (!set_mapping2d(p, t)) return;
            const char* names[4] = {"v00", "v01", "v10", "v11"};
            const float dflt[4] = {0.0f, 1.0f, 0.0f, 1.0f};
            for (int k = 0; k < 4; k++) {
                float c[3] = {dflt[k], dflt[k], dflt[k]};
                if (is_float) c[0] = c[1] = c[2] = p.find_one_float(names[k], dflt[k]); else spectrum_from(p, names[k], c);
                t.value[k][0] = c[0]; t.value[k][1] = c[1]; t.value[k][2] = c[2];
            }
            if (!error.empty()) return;
            add_node(t);
        } else if (tex_class == "normal" && is_spec) {
            forget();
            unsupported_textures[name] = tex_class;       // known to the reference, not on the path: fails only if something uses it
        } else {                                           // create_texture.rs:57-60, :106-109: not created
            warn(std::string(is_float ? "Float" : "Spectrum") + " texture \"" + tex_class + "\" unknown.");
        }
    }
    void pbrt_material(const std::string& name, const ParamSet& p) override {
        MaterialInstance mi;
        mi.name = name; mi.params = p;
        mi.none = name.empty() || name == "none";
        gstates.back().material = mi;
    }
    void pbrt_make_named_material(const std::string& name, const ParamSet& p) override {
        MaterialInstance mi;
        mi.name = p.find_one_string("type", "");
        if (mi.name.empty()) warn("No parameter string \"type\" found in MakeNamedMaterial");
        mi.params = p;
        mi.none = mi.name == "none";
        gstates.back().named_materials[name] = mi;
    }
    void pbrt_named_material(const std::string& name) override {
        auto it = gstates.back().named_materials.find(name);
        if (it == gstates.back().named_materials.end()) { warn("NamedMaterial \"" + name + "\" unknown."); return; }
        gstates.back().material = it->second;
    }
    void pbrt_light_source(const std::string& name, const ParamSet&) override {
        fail("LightSource \"" + name + "\": only diffuse area lights are on the accelerated path");
    }
    void pbrt_area_light_source(const std::string& name, const ParamSet& p) override {
        gstates.back().area_light_name = name;
        gstates.back().area_light_params = p;
    }
    void pbrt_reverse_orientation() override { gstates.back().reverse_orientation = !gstates.back().reverse_orientation; }
    // Object instancing (scene_context.rs:1327-1391): shapes between ObjectBegin / ObjectEnd go to the named object (tagged in
    // pt_mesh.object / pt_sphere.object), ObjectInstance records the CTM and its stored inverse.
    void pbrt_object_begin(const std::string& name) override {
        pbrt_attribute_begin();
        auto it = object_ids.find(name);
        if (it == object_ids.end()) { object_ids[name] = (uint32_t)object_ids.size(); it = object_ids.find(name); }
        else object_redefined.insert(it->second);           // instances.insert(name, Vec::new()) replaces the list (:1333)
        cur_object = it->second + 1;
        if (object_redefined.count(it->second)) fail("ObjectBegin \"" + name + "\" defined twice: redefinition is outside the accelerated path");
    }
    void pbrt_object_end() override {
        cur_object = 0;
        pbrt_attribute_end();
    }
    void pbrt_object_instance(const std::string& name) override {
        if (!error.empty() || cur_object) return;                    // ignored inside an object definition (:1352-1357)
        auto it = object_ids.find(name);
        if (it == object_ids.end()) return;                          // unknown name: nothing happens (:1365)
        const TransformSet& ts = transforms.back();
        if (std::memcmp(&ts.t[0].m, &ts.t[1].m, sizeof(M44)) != 0) { fail("animated instance transforms are outside the accelerated path"); return; }
        const Xf& c = ts.t[0];
        if (c.m.a[12] != 0.0f || c.m.a[13] != 0.0f || c.m.a[14] != 0.0f || c.m.a[15] != 1.0f) { fail("ObjectInstance under a projective transform is outside the accelerated path"); return; }
        pt_instance in;
        std::memset(&in, 0, sizeof(in));
        std::memcpy(in.instance_to_world, c.m.a, 64);
        std::memcpy(in.world_to_instance, c.inv.a, 64);
        in.object = it->second;
        in.before_triangle = (uint32_t)tri_mesh.size();
        in.order = n_extra++;
        instances.push_back(in);
    }

    // TextureParams (core/param_set/texture_params.rs:36-105): a texture binding is looked up in the shape's
    // parameters first, but constant float / spectrum VALUES come from the material's parameters first and
    // from the shape's only when the material does not give them (the reverse of pbrt-v3; reproduced as is).
    // get_spectrum_from (texture_params.rs:56-73): an rgb value, else a "spectrum" file converted to RGB
    bool spectrum_from(const ParamSet& ps, const std::string& n, float out[3]) {
        if (ps.find_one_rgb(n, out)) return true;
        if (ps.is_unsupported(n)) { fail("parameter \"" + n + "\" given as xyz / inline spectrum: outside the accelerated path"); return false; }
        auto f = ps.spectrum_files.find(n);
        auto b = ps.blackbodies.find(n);
        if (f == ps.spectrum_files.end() && b == ps.blackbodies.end()) return false;
        std::string err;
        const SpectrumTables* T = spectrum_tables(&err);
        if (!T) { fail("spectrum \"" + n + "\": " + err); return false; }
        if (f != ps.spectrum_files.end()) {
            if (!rgb_from_spd_file(*T, f->second, out, &err)) { fail("spectrum \"" + n + "\": " + err); return false; }
        } else {
            rgb_from_blackbody(*T, b->second, out);
        }
        return true;
    }
    // get_texture (texture_params.rs:85-105): the texture NAME bound to a parameter, shape first
    static const std::string* bound_texture(const ParamSet& geom, const ParamSet& mat, const std::string& n) {
        auto g = geom.textures.find(n);
        if (g != geom.textures.end() && !g->second.empty()) return &g->second[0];
        auto m = mat.textures.find(n);
        if (m != mat.textures.end() && !m->second.empty()) return &m->second[0];
        return nullptr;
    }
    // get_spectrum_texture_or_null (texture_params.rs:126-139) for textures that fold to a constant
    // tex_out: receives node index + 1 when the bound texture is evaluated on the device; a parameter that cannot be
    // texture-driven there (tex_out == nullptr) reports it.
    bool lookup_rgb(const ParamSet& geom, const ParamSet& mat, const std::string& n, float out[3], uint32_t* tex_out = nullptr) {
        if (const std::string* tex = bound_texture(geom, mat, n)) {
            auto nd = node_spectrum_textures.find(*tex);
            if (nd != node_spectrum_textures.end()) {
                if (!tex_out) { fail("parameter \"" + n + "\" uses texture \"" + *tex + "\", which varies over the surface: not supported for this parameter"); return false; }
                *tex_out = nd->second + 1;
                return true;
            }
            auto it = spectrum_textures.find(*tex);
            if (it == spectrum_textures.end()) {
                if (float_textures.count(*tex)) warn("Couldn't find spectrum texture named \"" + *tex + "\" for parameter \"" + n + "\"");
                else if (unsupported_textures.count(*tex)) fail("parameter \"" + n + "\" uses texture \"" + *tex + "\" (" + unsupported_textures[*tex] + "): only textures that fold to a constant are on the accelerated path");
                else warn("Couldn't find spectrum texture named \"" + *tex + "\" for parameter \"" + n + "\"");
                return false;
            }
            out[0] = it->second[0]; out[1] = it->second[1]; out[2] = it->second[2];
            return true;
        }
        if (spectrum_from(mat, n, out)) return true;
        return spectrum_from(geom, n, out);
    }
    // TextureParams::get_float_texture(_or_null) for constant values (texture_params.rs:36-54, :107-124): material first, then shape
    bool lookup_float(const ParamSet& geom, const ParamSet& mat, const std::string& n, float* out, uint32_t* tex_out = nullptr) {
        if (const std::string* tex = bound_texture(geom, mat, n)) {
            auto nd = node_float_textures.find(*tex);
            if (nd != node_float_textures.end()) {
                if (!tex_out) { fail("parameter \"" + n + "\" uses texture \"" + *tex + "\", which varies over the surface: not supported for this parameter"); return false; }
                *tex_out = nd->second + 1;
                return true;
            }
            auto it = float_textures.find(*tex);
            if (it == float_textures.end()) {
                if (unsupported_textures.count(*tex)) fail("parameter \"" + n + "\" uses texture \"" + *tex + "\" (" + unsupported_textures[*tex] + "): only textures that fold to a constant are on the accelerated path");
                else warn("Couldn't find float texture named \"" + *tex + "\" for parameter \"" + n + "\"");
                return false;
            }
            *out = it->second;
            return true;
        }
        if (mat.floats.count(n)) { *out = mat.find_one_float(n, *out); return true; }
        if (geom.floats.count(n)) { *out = geom.find_one_float(n, *out); return true; }
        return false;
    }
    int material_for_shape(const ParamSet& geom) {
        const MaterialInstance& mi = gstates.back().material;
        if (mi.none) return -1;
        const ParamSet& mp = mi.params;
        pt_material m;
        std::memset(&m, 0, sizeof(m));
        auto set3 = [](float* c, float v) { c[0] = c[1] = c[2] = v; };
        set3(m.opacity, 1.0f);
        m.eta = 1.5f;
        m.uroughness = m.vroughness = PT_ROUGHNESS_UNSET;
        m.remap_roughness = geom.find_one_bool("remaproughness", mp.find_one_bool("remaproughness", true)) ? 1 : 0;
        auto eta_or_index = [&]() {                              // get_float_texture_helper(&["eta", "index"], 1.5)
            if (!lookup_float(geom, mp, "eta", &m.eta, &m.tex_eta)) lookup_float(geom, mp, "index", &m.eta, &m.tex_eta);
        };
        if (mi.name == "matte") {                                   // matte.rs:56-61
            m.type = PT_MATERIAL_MATTE;
            set3(m.kd, 0.5f);
            lookup_rgb(geom, mp, "Kd", m.kd, &m.tex_kd);
            lookup_float(geom, mp, "sigma", &m.sigma, &m.tex_sigma);
        } else if (mi.name == "plastic") {                          // plastic.rs:73-86
            m.type = PT_MATERIAL_PLASTIC;
            set3(m.kd, 0.25f); set3(m.ks, 0.25f); m.roughness = 0.1f;
            lookup_rgb(geom, mp, "Kd", m.kd, &m.tex_kd); lookup_rgb(geom, mp, "Ks", m.ks, &m.tex_ks);
            lookup_float(geom, mp, "roughness", &m.roughness, &m.tex_roughness);
        } else if (mi.name == "mirror") {                           // mirror.rs:43-47
            m.type = PT_MATERIAL_MIRROR;
            set3(m.kr, 0.9f);
            lookup_rgb(geom, mp, "Kr", m.kr, &m.tex_kr);
        } else if (mi.name == "glass") {                            // glass.rs:124-143
            m.type = PT_MATERIAL_GLASS;
            set3(m.kr, 1.0f); set3(m.kt, 1.0f); m.uroughness = m.vroughness = 0.0f;
            lookup_rgb(geom, mp, "Kr", m.kr, &m.tex_kr); lookup_rgb(geom, mp, "Kt", m.kt, &m.tex_kt);
            lookup_float(geom, mp, "uroughness", &m.uroughness, &m.tex_uroughness); lookup_float(geom, mp, "vroughness", &m.vroughness, &m.tex_vroughness);
            eta_or_index();
        } else if (mi.name == "metal") {                            // metal.rs:127-149
            m.type = PT_MATERIAL_METAL;
            m.roughness = 0.01f;
            bool have_eta = lookup_rgb(geom, mp, "eta", m.metal_eta, &m.tex_metal_eta), have_k = lookup_rgb(geom, mp, "k", m.metal_k, &m.tex_metal_k);
            if (!have_eta || !have_k) {                 // defaults: the measured copper spectrum (metal.rs:87-132)
                std::string err;
                const SpectrumTables* T = spectrum_tables(&err);
                if (!T) { fail("Material \"metal\" defaults: " + err); return -1; }
                if (!have_eta) rgb_from_sampled(*T, T->cu_lambda, T->cu_n, m.metal_eta);
                if (!have_k) rgb_from_sampled(*T, T->cu_lambda, T->cu_k, m.metal_k);
            }
            lookup_float(geom, mp, "roughness", &m.roughness, &m.tex_roughness);
            lookup_float(geom, mp, "uroughness", &m.uroughness, &m.tex_uroughness); lookup_float(geom, mp, "vroughness", &m.vroughness, &m.tex_vroughness);
        } else if (mi.name == "uber") {                             // uber.rs:142-168
            m.type = PT_MATERIAL_UBER;
            set3(m.kd, 0.25f); set3(m.ks, 0.25f); m.roughness = 0.1f;
            lookup_rgb(geom, mp, "Kd", m.kd, &m.tex_kd); lookup_rgb(geom, mp, "Ks", m.ks, &m.tex_ks);
            lookup_rgb(geom, mp, "Kr", m.kr, &m.tex_kr); lookup_rgb(geom, mp, "Kt", m.kt, &m.tex_kt);
            lookup_rgb(geom, mp, "opacity", m.opacity, &m.tex_opacity);
            lookup_float(geom, mp, "roughness", &m.roughness, &m.tex_roughness);
            lookup_float(geom, mp, "uroughness", &m.uroughness, &m.tex_uroughness); lookup_float(geom, mp, "vroughness", &m.vroughness, &m.tex_vroughness);
            eta_or_index();
        } else if (mi.name == "substrate") {                        // substrate.rs:70-86
            m.type = PT_MATERIAL_SUBSTRATE;
            set3(m.kd, 0.5f); set3(m.ks, 0.5f); m.uroughness = m.vroughness = 0.1f;
            lookup_rgb(geom, mp, "Kd", m.kd, &m.tex_kd); lookup_rgb(geom, mp, "Ks", m.ks, &m.tex_ks);
            lookup_float(geom, mp, "uroughness", &m.uroughness, &m.tex_uroughness); lookup_float(geom, mp, "vroughness", &m.vroughness, &m.tex_vroughness);
        } else {
            fail("Material \"" + mi.name + "\": outside the accelerated path (matte, plastic, mirror, glass, metal, uber, substrate are supported)");
            return -1;
        }
        {   // "bumpmap": get_float_texture_or_null (texture_params.rs:107-116) -- a named float texture, else a number
            // (a constant texture, which still tilts the frame where dndu != 0), else none
            float bv = 0.0f;
            uint32_t bt = 0;
            if (lookup_float(geom, mp, "bumpmap", &bv, &bt)) {
                if (!bt) {
                    pt_texture t;
                    std::memset(&t, 0, sizeof(t));
                    t.type = PT_TEX_CONSTANT;
                    t.tex[0] = t.tex[1] = t.tex[2] = -1;
                    t.value[0][0] = t.value[0][1] = t.value[0][2] = bv;
                    textures.push_back(t);
                    bt = (uint32_t)textures.size();
                }
                m.tex_bump = bt;
            }
        }
        if (!error.empty()) return -1;
        for (size_t i = 0; i < materials.size(); i++)
            if (std::memcmp(&materials[i], &m, sizeof(m)) == 0) return (int)i;
        materials.push_back(m);
        return (int)materials.size() - 1;
    }
    int area_light_for_shape() {
        const GraphicsState& gs = gstates.back();
        if (gs.area_light_name.empty()) return -1;
        if (gs.area_light_name != "diffuse" && gs.area_light_name != "area") { fail("AreaLightSource \"" + gs.area_light_name + "\" unknown"); return -1; }
        float L[3] = {1.0f, 1.0f, 1.0f}, sc[3] = {1.0f, 1.0f, 1.0f};
        spectrum_from(gs.area_light_params, "L", L);          // ParamSet::find_one_spectrum: rgb, spectrum file or blackbody
        spectrum_from(gs.area_light_params, "scale", sc);
        if (!error.empty()) return -1;
        pt_area_light al;
        al.L[0] = L[0] * sc[0]; al.L[1] = L[1] * sc[1]; al.L[2] = L[2] * sc[2];
        al.two_sided = gs.area_light_params.find_one_bool("twosided", false) ? 1 : 0;
        {   // "nsamples" (diffuse.rs:177-189) sizes the sample arrays of DirectLighting's "all" strategy
            int ns = gs.area_light_params.find_one_int("nsamples", 1);
            if (quick_render) ns = std::max(ns / 4, 1);
            if (ns < 1 || ns > 4096) { fail("AreaLightSource \"nsamples\" outside [1, 4096] (the reference divides by it)"); return -1; }
            al.n_samples = ns;
        }
        area_lights.push_back(al);
        return (int)area_lights.size() - 1;
    }
    static bool is_fillable_uv(const std::vector<uint32_t>& vi, size_t nv) {     // triangle.rs:733-752
        std::vector<int> check(nv, 0);
        for (size_t f = 0; f + 2 < vi.size(); f += 3)
            for (int j = 0; j < 3; j++) {
                uint32_t v = vi[f + j];
                if (check[v] == 0 || check[v] == j) check[v] = j;
                else return false;
            }
        return true;
    }

    void pbrt_shape(const std::string& name, const ParamSet& p) override {
        if (!error.empty()) return;
        if (name != "trianglemesh" && name != "plymesh" && name != "sphere") { fail("Shape \"" + name + "\": only trianglemesh, plymesh and sphere are on the accelerated path"); return; }
        const TransformSet& ts = transforms.back();
        if (std::memcmp(&ts.t[0].m, &ts.t[1].m, sizeof(M44)) != 0) { fail("animated transforms are outside the accelerated path"); return; }
        if (p.has("alpha") || p.has("shadowalpha")) { fail("alpha-masked shapes are outside the accelerated path"); return; }
        if (name == "sphere") {                                     // create_sphere_shape (shapes/sphere.rs:401-420)
            const Xf& o2w = ts.t[0];
            if (o2w.m.a[12] != 0.0f || o2w.m.a[13] != 0.0f || o2w.m.a[14] != 0.0f || o2w.m.a[15] != 1.0f) { fail("sphere under a projective transform is outside the accelerated path"); return; }
            pt_sphere sp;
            std::memset(&sp, 0, sizeof(sp));
            std::memcpy(sp.object_to_world, o2w.m.a, 64);
            std::memcpy(sp.world_to_object, o2w.inv.a, 64);       // create_shapes passes object2world.inverse(): the stored m_inv, not a re-inversion
            sp.radius = p.find_one_float("radius", 1.0f);
            sp.zmin = p.find_one_float("zmin", -sp.radius);
            sp.zmax = p.find_one_float("zmax", sp.radius);
            sp.phimax = p.find_one_float("phimax", 360.0f);
            if (!(sp.radius > 0.0f)) { fail("sphere radius must be positive"); return; }
            sp.flags = gstates.back().reverse_orientation ? PT_SPHERE_REVERSE_ORIENTATION : 0u;
            sp.material = material_for_shape(p);
            sp.area_light = area_light_for_shape();
            if (!error.empty()) return;
            sp.before_triangle = (uint32_t)tri_mesh.size();
            sp.object = cur_object;
            sp.order = n_extra++;
            if (cur_object) sp.area_light = -1;                     // "Area lights not supported with object instancing" (:1302-1304)
            spheres.push_back(sp);
            return;
        }
        if (name == "plymesh") {                                    // shapes/plymesh.rs:251-380
            std::string file = p.find_one_string("filename", "");
            if (!file.empty() && file[0] != '/' && !p.base_dir.empty()) file = p.base_dir + "/" + file;
            PlyMesh ply;
            std::string perr;
            if (!read_ply(file, &ply, &perr)) { fail("plymesh: " + perr); return; }
            if (ply.indices.empty() || ply.P.empty()) { warn("Invalid mesh"); return; }
            emit_mesh(p, ply.indices, ply.P, ply.N.empty() ? nullptr : &ply.N, nullptr, ply.UV);
            return;
        }
        const std::vector<int>* vi = p.get_ints("indices");
        const std::vector<float>* ps = p.get_points("P");
        if (!vi || !ps || vi->empty() || ps->empty()) { warn("Invalid mesh"); return; }
        size_t nv = ps->size() / 3;
        std::vector<uint32_t> idx(vi->size());
        for (size_t i = 0; i < vi->size(); i++) {
            if ((*vi)[i] < 0 || (size_t)(*vi)[i] >= nv) { fail("trianglemesh has out-of-bounds vertex index"); return; }
            idx[i] = (uint32_t)(*vi)[i];
        }
        std::vector<float> uv;
        const std::vector<float>* fuv = p.get_floats("uv");
        if (!fuv) fuv = p.get_floats("st");
        if (fuv) uv = *fuv;
        else if (is_fillable_uv(idx, nv)) {                         // triangle.rs:796-821
            uv.assign(2 * nv, 0.0f);
            const float tri_uv[3][2] = {{0, 0}, {1, 0}, {1, 1}};
            for (size_t f = 0; f + 2 < idx.size(); f += 3)
                for (int j = 0; j < 3; j++) {
                    uint32_t v = idx[f + j];
                    if (uv[2 * v] == 0.0f && uv[2 * v + 1] == 0.0f) { uv[2 * v] = tri_uv[j][0]; uv[2 * v + 1] = tri_uv[j][1]; }
                }
        }
        const std::vector<float>* sv = p.get_points("S");
        if (!sv) sv = p.get_vectors("S");
        const std::vector<float>* nn = p.get_points("N");
        if (!nn) nn = p.get_normals("N");
        emit_mesh(p, idx, *ps, nn, sv, uv);
    }

    // create_triangle_mesh (triangle.rs:696-731) on object-space arrays: pre-transform, flags, material / light, degenerate filter
    void emit_mesh(const ParamSet& p, const std::vector<uint32_t>& idx, const std::vector<float>& obj_p, const std::vector<float>* nn,
                   const std::vector<float>* sv, const std::vector<float>& uv) {
        const TransformSet& ts = transforms.back();
        const std::vector<float>* ps = &obj_p;
        size_t nv = obj_p.size() / 3;
        for (uint32_t i : idx)
            if (i >= nv) { fail("mesh has out-of-bounds vertex index"); return; }
        if (!uv.empty() && uv.size() / 2 < nv) { fail("mesh uv count does not match P"); return; }
        if ((sv && sv->size() / 3 < nv) || (nn && nn->size() / 3 < nv)) { fail("mesh S / N count does not match P"); return; }

        const Xf& o2w = ts.t[0];
        std::vector<float> wp(3 * nv), wn, ws;                      // TriangleMesh::new pre-transforms (triangle.rs:49-66)
        for (size_t i = 0; i < nv; i++) xf_point(o2w.m, &(*ps)[3 * i], &wp[3 * i]);
        if (nn) { wn.resize(3 * nv); for (size_t i = 0; i < nv; i++) xf_normal(o2w.inv, &(*nn)[3 * i], &wn[3 * i]); }
        if (sv) { ws.resize(3 * nv); for (size_t i = 0; i < nv; i++) xf_vector(o2w.m, &(*sv)[3 * i], &ws[3 * i]); }

        pt_mesh mesh;
        mesh.flags = 0;
        if (p.find_one_bool("twosided", true)) mesh.flags |= PT_MESH_TWO_SIDED;
        if (gstates.back().reverse_orientation) mesh.flags |= PT_MESH_REVERSE_ORIENTATION;
        if (swaps_handedness(o2w.m)) mesh.flags |= PT_MESH_SWAPS_HANDEDNESS;
        if (nn) mesh.flags |= PT_MESH_HAS_N;
        if (sv) mesh.flags |= PT_MESH_HAS_S;
        if (!uv.empty()) mesh.flags |= PT_MESH_HAS_UV;
        mesh.material = material_for_shape(p);
        mesh.area_light = area_light_for_shape();
        mesh.object = cur_object;
        if (cur_object) mesh.area_light = -1;                       // "Area lights not supported with object instancing" (:1302-1304)
        if (!error.empty()) return;

        uint32_t base = (uint32_t)(P.size() / 3);
        uint32_t mesh_id = (uint32_t)meshes.size();
        size_t kept = 0;
        for (size_t f = 0; f + 2 < idx.size(); f += 3) {            // drop triangles with area <= 1e-16 (triangle.rs:726)
            const float* a = &wp[3 * idx[f]]; const float* b = &wp[3 * idx[f + 1]]; const float* c = &wp[3 * idx[f + 2]];
            V3f e1 = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2 = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            float area = 0.5f * lenf(crossf(e1, e2));
            if (!(area > 1e-16f)) continue;
            indices.push_back(base + idx[f]); indices.push_back(base + idx[f + 1]); indices.push_back(base + idx[f + 2]);
            tri_mesh.push_back(mesh_id);
            kept++;
        }
        (void)kept;
        meshes.push_back(mesh);
        size_t old_nv = P.size() / 3;
        P.insert(P.end(), wp.begin(), wp.end());
        auto grow = [&](std::vector<float>& dst, const std::vector<float>& src, int width, bool& any) {
            if (!src.empty() && !any) { dst.assign(old_nv * width, 0.0f); any = true; }
            if (any) { if (src.empty()) dst.insert(dst.end(), nv * width, 0.0f); else dst.insert(dst.end(), src.begin(), src.begin() + nv * width); }
        };
        grow(N, wn, 3, any_n);
        grow(S, ws, 3, any_s);
        grow(UV, uv, 2, any_uv);
    }

    // ---- WorldEnd: the factories of scene_context.rs:606-729 reduced to parameter capture
    float gaussian1(float d, float alpha, float expv) { return std::fmax(0.0f, std::exp(-alpha * d * d) - expv); }
    float mitchell1(float x, float b, float c) {
        x = std::fabs(2.0f * x);
        if (x > 1.0f) return ((-b - 6.0f * c) * x * x * x + (6.0f * b + 30.0f * c) * x * x + (-12.0f * b - 48.0f * c) * x + (8.0f * b + 24.0f * c)) * (1.0f / 6.0f);
        return ((12.0f - 9.0f * b - 6.0f * c) * x * x * x + (-18.0f + 12.0f * b + 6.0f * c) * x * x + (6.0f - 2.0f * b)) * (1.0f / 6.0f);
    }
    static float sinc1(float x) {                                   // filters/sinc.rs:24-31
        x = std::fabs(x);
        if (x < 1e-5f) return 1.0f;
        const float pi = 3.14159265358979323846f;
        return std::sin(pi * x) / (pi * x);
    }
    static float windowed_sinc(float x, float radius, float tau) {  // filters/sinc.rs:33-41
        x = std::fabs(x);
        if (x > radius) return 0.0f;
        float lanczos = sinc1(x / tau);
        return sinc1(x) * lanczos;
    }
    bool make_filter() {
        float rx, ry;
        const ParamSet& fp = filter_params;
        int kind;
        if (filter_name == "box") { kind = 0; rx = fp.find_one_float("xwidth", 0.5f); ry = fp.find_one_float("ywidth", 0.5f); }
        else if (filter_name == "gaussian") { kind = 1; rx = fp.find_one_float("xwidth", 2.0f); ry = fp.find_one_float("ywidth", 2.0f); }
        else if (filter_name == "mitchell") { kind = 2; rx = fp.find_one_float("xwidth", 2.0f); ry = fp.find_one_float("ywidth", 2.0f); }
        else if (filter_name == "triangle") { kind = 3; rx = fp.find_one_float("xwidth", 2.0f); ry = fp.find_one_float("ywidth", 2.0f); }
        else if (filter_name == "sinc") { kind = 4; rx = fp.find_one_float("xwidth", 4.0f); ry = fp.find_one_float("ywidth", 4.0f); }
        else { fail("PixelFilter \"" + filter_name + "\" is not supported"); return false; }
        const float tau = fp.find_one_float("tau", 3.0f);
        float alpha = fp.find_one_float("alpha", 2.0f), B = fp.find_one_float("B", 1.0f / 3.0f), C = fp.find_one_float("C", 1.0f / 3.0f);
        float ex = std::exp(-alpha * rx * rx), ey = std::exp(-alpha * ry * ry);
        desc.filter_radius[0] = rx; desc.filter_radius[1] = ry;
        for (int y = 0; y < 16; y++)                                 // Film::new's table (film.rs:102-120)
            for (int x = 0; x < 16; x++) {
                float xx = (float)x * (rx / 15.0f), yy = (float)y * (ry / 15.0f), v = 1.0f;
                if (kind == 1) v = gaussian1(xx, alpha, ex) * gaussian1(yy, alpha, ey);
                else if (kind == 2) v = mitchell1(xx * (1.0f / rx), B, C) * mitchell1(yy * (1.0f / ry), B, C);
                else if (kind == 3) v = std::fmax(0.0f, rx - std::fabs(xx)) * std::fmax(0.0f, ry - std::fabs(yy));
                else if (kind == 4) v = windowed_sinc(xx, rx, tau) * windowed_sinc(yy, ry, tau);
                desc.filter_table[y * 16 + x] = v;
            }
        return true;
    }

    void pbrt_world_end() override {
        world_ended = true;
        if (!error.empty()) return;
        if (indices.empty() && spheres.empty()) { fail("scene has no primitives on the accelerated path"); return; }
        // film (film.rs:520-580)
        if (film_name != "image") warn("Film \"" + film_name + "\" treated as \"image\"");
        desc.xres = film_params.find_one_int("xresolution", 1280);
        desc.yres = film_params.find_one_int("yresolution", 720);
        if (quick_render && !quick_full_resolution) {               // create_film (film.rs:548-554)
            desc.xres = std::max(1, desc.xres / 4);
            desc.yres = std::max(1, desc.yres / 4);
        }
        desc.crop_window[0] = 0.0f; desc.crop_window[1] = 1.0f; desc.crop_window[2] = 0.0f; desc.crop_window[3] = 1.0f;
        if (const std::vector<float>* cw = film_params.get_floats("cropwindow")) {
            if (cw->size() != 4) { fail("\"cropwindow\" expects 4 values"); return; }
            for (int i = 0; i < 4; i++) desc.crop_window[i] = (*cw)[i];
        }
        desc.film_scale = film_params.find_one_float("scale", 1.0f);
        desc.max_sample_luminance = film_params.find_one_float("maxsampleluminance", std::numeric_limits<float>::infinity());
        out_filename = film_params.find_one_string("filename", "pbrt.exr");
        if (!make_filter()) return;
        // camera (cameras/perspective.rs:337-394)
        if (camera_name != "perspective") { fail("Camera \"" + camera_name + "\": only perspective is on the accelerated path"); return; }
        TransformSet c2w;
        if (have_camera) c2w = named_cs["camera"];
        else { c2w.t[0].m = c2w.t[0].inv = c2w.t[1].m = c2w.t[1].inv = m_identity(); }
        if (std::memcmp(&c2w.t[0].m, &c2w.t[1].m, sizeof(M44)) != 0) { fail("animated camera transforms are outside the accelerated path"); return; }
        std::memcpy(desc.camera_to_world, c2w.t[0].m.a, 64);
        float so = camera_params.find_one_float("shutteropen", 0.0f), sc = camera_params.find_one_float("shutterclose", 1.0f);
        if (sc < so) std::swap(so, sc);
        desc.shutter_open = so; desc.shutter_close = sc;
        desc.lens_radius = camera_params.find_one_float("lensradius", 0.0f);
        desc.focal_distance = camera_params.find_one_float("focaldistance", 1e6f);
        float aspect = (float)desc.xres / (float)desc.yres;
        float frame = camera_params.find_one_float("frameaspectratio", aspect);
        if (const std::vector<float>* sw = camera_params.get_floats("screenwindow")) {
            if (sw->size() < 4) { fail("\"screenwindow\" should have four values"); return; }
            for (int i = 0; i < 4; i++) desc.screen_window[i] = (*sw)[i];
        } else if (frame > 1.0f) {
            desc.screen_window[0] = -frame; desc.screen_window[1] = frame; desc.screen_window[2] = -1.0f; desc.screen_window[3] = 1.0f;
        } else {
            desc.screen_window[0] = -1.0f; desc.screen_window[1] = 1.0f; desc.screen_window[2] = -1.0f / frame; desc.screen_window[3] = 1.0f / frame;
        }
        desc.fov = camera_params.find_one_float("fov", 90.0f);
        float halffov = camera_params.find_one_float("halffov", -1.0f);
        if (halffov > 0.0f) desc.fov = 2.0f * halffov;
        // sampler (render_options.rs:71: the default is halton)
        if (sampler_name == "sobol") {
            desc.sampler = PT_SAMPLER_SOBOL;
        } else if (sampler_name == "halton") {          // samplers/halton.rs:275-299
            desc.sampler = PT_SAMPLER_HALTON;
            desc.halton_sample_at_center = sampler_params.find_one_bool("samplepixelcenter", false) ? 1 : 0;
        } else {
            fail("Sampler \"" + sampler_name + "\": only the index-addressed samplers (halton, sobol) are reproducible on a wavefront");
            return;
        }
        desc.spp = sampler_params.find_one_int("pixelsamples", 16);
        if (quick_render) desc.spp = 1;                              // create_halton_sampler / create_sobol_sampler
        // integrator (integrators/path.rs:252-271)
        if (integrator_name != "path" && integrator_name != "ao" && integrator_name != "directlighting" && integrator_name != "whitted") {
            fail("Integrator \"" + integrator_name + "\": only path, ao, directlighting and whitted are on the accelerated path");
            return;
        }
        desc.integrator = integrator_name == "ao" ? PT_INTEGRATOR_AO : (integrator_name == "directlighting" ? PT_INTEGRATOR_DIRECTLIGHTING
                          : (integrator_name == "whitted" ? PT_INTEGRATOR_WHITTED : PT_INTEGRATOR_PATH));
        if (desc.integrator == PT_INTEGRATOR_DIRECTLIGHTING || desc.integrator == PT_INTEGRATOR_WHITTED) {
            // create_direct_lighting_integrator (directlighting.rs:137-158), create_whitted_integrator (whitted.rs:112-135)
            desc.direct_strategy = integrator_params.find_one_string("strategy", "all") == "one" ? PT_DIRECT_ONE : PT_DIRECT_ALL;
            if (integrator_params.ints.count("pixelbounds")) { fail("Integrator \"" + integrator_name + "\": \"pixelbounds\" is outside the accelerated path"); return; }
        }
        desc.ao_cos_sample = integrator_params.find_one_bool("cossample", true) ? 1 : 0;      // create_ao_integrator (ao.rs:118-138)
        desc.ao_samples = integrator_params.find_one_int("nsamples", 64);
        if (quick_render) desc.ao_samples = 1;
        if (desc.integrator == PT_INTEGRATOR_AO && desc.ao_samples < 1) { fail("Integrator \"ao\": nsamples must be positive"); return; }
        desc.max_depth = integrator_params.find_one_int("maxdepth", 5);
        desc.rr_threshold = integrator_params.find_one_float("rrthreshold", 1.0f);
        std::string ls = integrator_params.find_one_string("lightsamplestrategy", "spatial");
        desc.light_strategy = ls == "uniform" ? PT_LIGHTS_UNIFORM : (ls == "power" ? PT_LIGHTS_POWER : PT_LIGHTS_SPATIAL);
        // accelerator (accelerators/bvh/create_bvh_accelerator.rs:13-27)
        if (accel_name != "bvh") { fail("Accelerator \"" + accel_name + "\": only bvh is on the accelerated path"); return; }
        std::string sm = accel_params.find_one_string("splitmethod", "sah");
        desc.split_method = sm == "hlbvh" ? PT_SPLIT_HLBVH : (sm == "middle" ? PT_SPLIT_MIDDLE : (sm == "equal" ? PT_SPLIT_EQUAL_COUNTS : PT_SPLIT_SAH));
        desc.max_node_prims = accel_params.find_one_int("maxnodeprims", 4);
        // geometry
        desc.n_vertices = (uint32_t)(P.size() / 3);
        desc.P = P.data();
        desc.N = any_n ? N.data() : nullptr;
        desc.S = any_s ? S.data() : nullptr;
        desc.UV = any_uv ? UV.data() : nullptr;
        desc.n_triangles = (uint32_t)(indices.size() / 3);
        desc.indices = indices.data();
        desc.tri_mesh = tri_mesh.data();
        desc.n_meshes = (uint32_t)meshes.size();
        desc.meshes = meshes.data();
        desc.n_spheres = (uint32_t)spheres.size();
        desc.spheres = spheres.empty() ? nullptr : spheres.data();
        desc.n_instances = (uint32_t)instances.size();
        desc.instances = instances.empty() ? nullptr : instances.data();
        desc.n_textures = (uint32_t)textures.size();
        desc.textures = textures.empty() ? nullptr : textures.data();
        images.clear();
        for (auto& py : pyramids) { py->desc.texels = py->texels.data(); images.push_back(py->desc); }
        desc.n_images = (uint32_t)images.size();
        desc.images = images.empty() ? nullptr : images.data();
        desc.n_materials = (uint32_t)materials.size();
        desc.materials = materials.data();
        desc.n_area_lights = (uint32_t)area_lights.size();
        desc.area_lights = area_lights.data();
    }

private:
    std::vector<TransformSet> transforms;
    std::vector<unsigned> bits;
    std::vector<GraphicsState> gstates;
    std::map<std::string, TransformSet> named_cs;
    std::string filter_name = "box", film_name = "image", sampler_name = "halton", accel_name = "bvh", integrator_name = "path", camera_name = "perspective";
    ParamSet filter_params, film_params, sampler_params, accel_params, integrator_params, camera_params;
    bool have_camera = false;
};

}  // namespace pth

struct pth_scene {
    pth::GpuSceneContext ctx;
};

static pt_status finish(pth_scene* s, bool parsed, const std::string& perr, pth_scene** out, char* err, size_t cap) {
    std::string msg;
    pt_status st = PT_OK;
    if (!parsed) { msg = perr; st = PT_ERR_INVALID_ARGUMENT; }
    else if (!s->ctx.error.empty()) { msg = s->ctx.error; st = PT_ERR_UNSUPPORTED; }
    else if (!s->ctx.world_ended) { msg = "scene description has no WorldEnd"; st = PT_ERR_INVALID_ARGUMENT; }
    if (st != PT_OK) {
        if (err && cap) { std::snprintf(err, cap, "%s", msg.c_str()); }
        delete s;
        *out = nullptr;
        return st;
    }
    *out = s;
    return PT_OK;
}

extern "C" {

pt_status pth_parse_file(const char* filename, pth_scene** out, char* err, size_t err_cap) {
    return pth_parse_file_opts(filename, nullptr, out, err, err_cap);
}
pt_status pth_parse_file_opts(const char* filename, const pth_options* opts, pth_scene** out, char* err, size_t err_cap) {
    if (!filename || !out) return PT_ERR_INVALID_ARGUMENT;
    pth_scene* s = new pth_scene;
    if (opts) {
        s->ctx.quick_render = opts->quick != 0 || opts->quick_full_resolution != 0;     // bin/pbrt.rs:360-366
        s->ctx.quick_full_resolution = opts->quick_full_resolution != 0;
    }
    std::string perr;
    bool ok = pth::pbrt_parse_file(filename, s->ctx, &perr);
    pt_status st = finish(s, ok, perr, out, err, err_cap);
    if (st == PT_OK && opts && opts->pixelsamples > 0) (*out)->ctx.desc.spp = opts->pixelsamples;   // bin/pbrt.rs:234-238
    return st;
}
pt_status pth_parse_string(const char* text, const char* work_dir, pth_scene** out, char* err, size_t err_cap) {
    if (!text || !out) return PT_ERR_INVALID_ARGUMENT;
    pth_scene* s = new pth_scene;
    std::string perr;
    bool ok = pth::pbrt_parse_string(text, work_dir ? work_dir : ".", s->ctx, &perr);
    return finish(s, ok, perr, out, err, err_cap);
}
const pt_scene_desc* pth_scene_get_desc(const pth_scene* s) { return s ? &s->ctx.desc : nullptr; }
const char* pth_scene_output_filename(const pth_scene* s) { return s ? s->ctx.out_filename.c_str() : ""; }
void pth_scene_set_pixelsamples(pth_scene* s, int spp) { if (s && spp > 0) s->ctx.desc.spp = spp; }
const char* pth_scene_warnings(const pth_scene* s) { return s ? s->ctx.warnings.c_str() : ""; }
void pth_scene_free(pth_scene* s) { delete s; }

}  // extern "C"
