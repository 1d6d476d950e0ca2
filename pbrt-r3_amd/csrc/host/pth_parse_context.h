// pth_parse_context.h -- C++ mirror of the reference's scene-description callback interface
// (trait ParseContext, src/core/api/parse_context.rs:5-66): same method names, same argument
// meaning, so a consumer written against the reference reads the same here.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace pth {

// src/core/param_set/param_set.rs: typed key -> value list.  Spectra are kept as RGB triples
// (the reference is built with RGBSpectrum by default, base/types.rs + spectrum/rgb.rs).
class ParamSet {
public:
    std::map<std::string, std::vector<float>> floats, points, vectors, normals, rgbs;
    std::map<std::string, std::vector<int>> ints;
    std::map<std::string, std::vector<bool>> bools;
    std::map<std::string, std::vector<std::string>> strings, textures;
    std::vector<std::string> unsupported;      // "blackbody"/"xyz"/inline "spectrum" entries: named ("type name"), not converted
    std::map<std::string, std::vector<float>> blackbodies;   // "blackbody name" [T scale ...]
    std::string base_dir;                      // directory of the file the directive came from (relative "filename" parameters)
    std::map<std::string, std::string> spectrum_files;   // "spectrum name" "file.spd": path resolved against the including file
    bool is_unsupported(const std::string& n) const {
        for (auto& u : unsupported) { size_t k = u.find(' '); if (k != std::string::npos && u.substr(k + 1) == n) return true; }
        return false;
    }

    float find_one_float(const std::string& n, float d) const { auto i = floats.find(n); return (i != floats.end() && i->second.size() == 1) ? i->second[0] : d; }
    int find_one_int(const std::string& n, int d) const { auto i = ints.find(n); return (i != ints.end() && i->second.size() == 1) ? i->second[0] : d; }
    bool find_one_bool(const std::string& n, bool d) const { auto i = bools.find(n); return (i != bools.end() && i->second.size() == 1) ? (bool)i->second[0] : d; }
    std::string find_one_string(const std::string& n, const std::string& d) const { auto i = strings.find(n); return (i != strings.end() && i->second.size() == 1) ? i->second[0] : d; }
    bool find_one_rgb(const std::string& n, float out[3]) const {
        auto i = rgbs.find(n);
        if (i == rgbs.end() || i->second.size() != 3) return false;
        out[0] = i->second[0]; out[1] = i->second[1]; out[2] = i->second[2];
        return true;
    }
    const std::vector<float>* get_floats(const std::string& n) const { auto i = floats.find(n); return i == floats.end() ? nullptr : &i->second; }
    const std::vector<float>* get_points(const std::string& n) const { auto i = points.find(n); return i == points.end() ? nullptr : &i->second; }
    const std::vector<float>* get_vectors(const std::string& n) const { auto i = vectors.find(n); return i == vectors.end() ? nullptr : &i->second; }
    const std::vector<float>* get_normals(const std::string& n) const { auto i = normals.find(n); return i == normals.end() ? nullptr : &i->second; }
    const std::vector<int>* get_ints(const std::string& n) const { auto i = ints.find(n); return i == ints.end() ? nullptr : &i->second; }
    bool has(const std::string& n) const {
        return floats.count(n) || points.count(n) || vectors.count(n) || normals.count(n) || rgbs.count(n) || ints.count(n) || bools.count(n) ||
               strings.count(n) || textures.count(n);
    }
};

class ParseContext {
public:
    virtual ~ParseContext() {}
    virtual void pbrt_identity() = 0;
    virtual void pbrt_translate(float dx, float dy, float dz) = 0;
    virtual void pbrt_rotate(float angle, float ax, float ay, float az) = 0;
    virtual void pbrt_scale(float sx, float sy, float sz) = 0;
    virtual void pbrt_look_at(float ex, float ey, float ez, float lx, float ly, float lz, float ux, float uy, float uz) = 0;
    virtual void pbrt_concat_transform(const std::vector<float>& t) = 0;
    virtual void pbrt_transform(const std::vector<float>& t) = 0;
    virtual void pbrt_coordinate_system(const std::string& name) = 0;
    virtual void pbrt_coord_sys_transform(const std::string& name) = 0;
    virtual void pbrt_active_transform_all() = 0;
    virtual void pbrt_active_transform_end_time() = 0;
    virtual void pbrt_active_transform_start_time() = 0;
    virtual void pbrt_transform_times(float start, float end) = 0;
    virtual void pbrt_pixel_filter(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_film(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_sampler(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_accelerator(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_integrator(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_camera(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_make_named_medium(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_medium_interface(const std::string& inside_name, const std::string& outside_name) = 0;
    virtual void pbrt_world_begin() = 0;
    virtual void pbrt_attribute_begin() = 0;
    virtual void pbrt_attribute_end() = 0;
    virtual void pbrt_transform_begin() = 0;
    virtual void pbrt_transform_end() = 0;
    virtual void pbrt_texture(const std::string& name, const std::string& type, const std::string& tex_name, const ParamSet& params) = 0;
    virtual void pbrt_material(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_make_named_material(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_named_material(const std::string& name) = 0;
    virtual void pbrt_light_source(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_area_light_source(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_shape(const std::string& name, const ParamSet& params) = 0;
    virtual void pbrt_reverse_orientation() = 0;
    virtual void pbrt_object_begin(const std::string& name) = 0;
    virtual void pbrt_object_end() = 0;
    virtual void pbrt_object_instance(const std::string& name) = 0;
    virtual void pbrt_world_end() = 0;
};

// src/core/parser/parser.rs:60-135.  Returns false and fills err on a syntax error.
bool pbrt_parse_file(const std::string& filename, ParseContext& ctx, std::string* err);
bool pbrt_parse_string(const std::string& text, const std::string& work_dir, ParseContext& ctx, std::string* err);

}  // namespace pth
