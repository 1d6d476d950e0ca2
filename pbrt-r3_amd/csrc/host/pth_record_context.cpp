// pth_record_context.cpp -- a ParseContext that writes every callback it receives to a text log,
// one line per directive, in the spirit of the reference's PrintContext (--cat,
// src/core/api/print_context.rs).  Used to test the parser in isolation.
#include <cstdio>
#include <sstream>

#include "../../../include/pbrtgpu_host.h"
#include "pth_parse_context.h"

namespace pth {

class RecordContext : public ParseContext {
public:
    std::ostringstream log;
    static void put(std::ostringstream& o, const ParamSet& p) {
        auto nums = [&](const char* ty, const std::map<std::string, std::vector<float>>& m) {
            for (auto& kv : m) { o << " \"" << ty << " " << kv.first << "\" ["; for (float f : kv.second) { char b[64]; std::snprintf(b, sizeof b, " %.9g", f); o << b; } o << " ]"; }
        };
        nums("float", p.floats); nums("point", p.points); nums("vector", p.vectors); nums("normal", p.normals); nums("rgb", p.rgbs);
        for (auto& kv : p.ints) { o << " \"integer " << kv.first << "\" ["; for (int v : kv.second) o << " " << v; o << " ]"; }
        for (auto& kv : p.bools) { o << " \"bool " << kv.first << "\" ["; for (bool v : kv.second) o << (v ? " true" : " false"); o << " ]"; }
        for (auto& kv : p.strings) { o << " \"string " << kv.first << "\" ["; for (auto& v : kv.second) o << " \"" << v << "\""; o << " ]"; }
        for (auto& kv : p.textures) { o << " \"texture " << kv.first << "\" ["; for (auto& v : kv.second) o << " \"" << v << "\""; o << " ]"; }
        for (auto& u : p.unsupported) o << " \"" << u << "\" [unsupported]";
    }
    void line(const char* n) { log << n << "\n"; }
    void linef(const char* n, std::initializer_list<float> v) { log << n; for (float f : v) { char b[64]; std::snprintf(b, sizeof b, " %.9g", f); log << b; } log << "\n"; }
    void linev(const char* n, const std::vector<float>& v) { log << n << " ["; for (float f : v) { char b[64]; std::snprintf(b, sizeof b, " %.9g", f); log << b; } log << " ]\n"; }
    void lines(const char* n, const std::string& s) { log << n << " \"" << s << "\"\n"; }
    void linep(const char* n, const std::string& s, const ParamSet& p) { log << n << " \"" << s << "\""; put(log, p); log << "\n"; }

    void pbrt_identity() override { line("Identity"); }
    void pbrt_translate(float x, float y, float z) override { linef("Translate", {x, y, z}); }
    void pbrt_rotate(float a, float x, float y, float z) override { linef("Rotate", {a, x, y, z}); }
    void pbrt_scale(float x, float y, float z) override { linef("Scale", {x, y, z}); }
    void pbrt_look_at(float a, float b, float c, float d, float e, float f, float g, float h, float i) override { linef("LookAt", {a, b, c, d, e, f, g, h, i}); }
    void pbrt_concat_transform(const std::vector<float>& t) override { linev("ConcatTransform", t); }
    void pbrt_transform(const std::vector<float>& t) override { linev("Transform", t); }
    void pbrt_coordinate_system(const std::string& n) override { lines("CoordinateSystem", n); }
    void pbrt_coord_sys_transform(const std::string& n) override { lines("CoordSysTransform", n); }
    void pbrt_active_transform_all() override { line("ActiveTransform All"); }
    void pbrt_active_transform_end_time() override { line("ActiveTransform EndTime"); }
    void pbrt_active_transform_start_time() override { line("ActiveTransform StartTime"); }
    void pbrt_transform_times(float a, float b) override { linef("TransformTimes", {a, b}); }
    void pbrt_pixel_filter(const std::string& n, const ParamSet& p) override { linep("PixelFilter", n, p); }
    void pbrt_film(const std::string& n, const ParamSet& p) override { linep("Film", n, p); }
    void pbrt_sampler(const std::string& n, const ParamSet& p) override { linep("Sampler", n, p); }
    void pbrt_accelerator(const std::string& n, const ParamSet& p) override { linep("Accelerator", n, p); }
    void pbrt_integrator(const std::string& n, const ParamSet& p) override { linep("Integrator", n, p); }
    void pbrt_camera(const std::string& n, const ParamSet& p) override { linep("Camera", n, p); }
    void pbrt_make_named_medium(const std::string& n, const ParamSet& p) override { linep("MakeNamedMedium", n, p); }
    void pbrt_medium_interface(const std::string& a, const std::string& b) override { log << "MediumInterface \"" << a << "\" \"" << b << "\"\n"; }
    void pbrt_world_begin() override { line("WorldBegin"); }
    void pbrt_attribute_begin() override { line("AttributeBegin"); }
    void pbrt_attribute_end() override { line("AttributeEnd"); }
    void pbrt_transform_begin() override { line("TransformBegin"); }
    void pbrt_transform_end() override { line("TransformEnd"); }
    void pbrt_texture(const std::string& n, const std::string& t, const std::string& c, const ParamSet& p) override {
        log << "Texture \"" << n << "\" \"" << t << "\" \"" << c << "\""; put(log, p); log << "\n";
    }
    void pbrt_material(const std::string& n, const ParamSet& p) override { linep("Material", n, p); }
    void pbrt_make_named_material(const std::string& n, const ParamSet& p) override { linep("MakeNamedMaterial", n, p); }
    void pbrt_named_material(const std::string& n) override { lines("NamedMaterial", n); }
    void pbrt_light_source(const std::string& n, const ParamSet& p) override { linep("LightSource", n, p); }
    void pbrt_area_light_source(const std::string& n, const ParamSet& p) override { linep("AreaLightSource", n, p); }
    void pbrt_shape(const std::string& n, const ParamSet& p) override { linep("Shape", n, p); }
    void pbrt_reverse_orientation() override { line("ReverseOrientation"); }
    void pbrt_object_begin(const std::string& n) override { lines("ObjectBegin", n); }
    void pbrt_object_end() override { line("ObjectEnd"); }
    void pbrt_object_instance(const std::string& n) override { lines("ObjectInstance", n); }
    void pbrt_world_end() override { line("WorldEnd"); }
};

}  // namespace pth

// Parses `text` with a recording context; writes the callback log (or the error message) to out.
// Returns PT_OK, or PT_ERR_INVALID_ARGUMENT on a syntax error.
extern "C" pt_status pth_parse_to_log(const char* text, const char* work_dir, char* out, size_t cap) {
    if (!text || !out || cap == 0) return PT_ERR_INVALID_ARGUMENT;
    pth::RecordContext rc;
    std::string err;
    bool ok = pth::pbrt_parse_string(text, work_dir ? work_dir : ".", rc, &err);
    std::string s = ok ? rc.log.str() : err;
    std::snprintf(out, cap, "%s", s.c_str());
    return ok ? PT_OK : PT_ERR_INVALID_ARGUMENT;
}
