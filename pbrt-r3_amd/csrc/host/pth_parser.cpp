// pth_parser.cpp -- .pbrt scene-description parser: tokens -> ParseContext callbacks.
// Behaviour follows src/core/parser/parser.rs:132-423 (directive table and argument counts)
// and src/core/parser/read_file.rs:14-37 (Include relative to the including file); the
// implementation is a hand-written recursive tokenizer, not the reference's nom combinators.
#include "pth_parse_context.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pth {

namespace {

struct Tok {
    enum Kind { END, IDENT, STRING, NUMBER, LBRACKET, RBRACKET } kind = END;
    std::string text;
    double num = 0.0;
    int line = 0;
};

class Lexer {
public:
    Lexer(const std::string& s) : s_(s) {}
    Tok next() {
        skip_space();
        Tok t;
        t.line = line_;
        if (pos_ >= s_.size()) return t;
        char c = s_[pos_];
        if (c == '[') { pos_++; t.kind = Tok::LBRACKET; return t; }
        if (c == ']') { pos_++; t.kind = Tok::RBRACKET; return t; }
        if (c == '"') {
            size_t e = s_.find('"', pos_ + 1);
            if (e == std::string::npos) { t.kind = Tok::END; err_ = "unterminated string"; return t; }
            t.kind = Tok::STRING;
            t.text = s_.substr(pos_ + 1, e - pos_ - 1);
            for (char ch : t.text) if (ch == '\n') line_++;
            pos_ = e + 1;
            return t;
        }
        if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
            char* endp = nullptr;
            double v = (double)std::strtof(s_.c_str() + pos_, &endp);     // decimal -> f32 directly: no double rounding
            size_t n = (size_t)(endp - (s_.c_str() + pos_));
            if (n > 0) { t.kind = Tok::NUMBER; t.num = v; t.text = s_.substr(pos_, n); pos_ += n; return t; }
        }
        size_t b = pos_;
        while (pos_ < s_.size() && !std::isspace((unsigned char)s_[pos_]) && s_[pos_] != '"' && s_[pos_] != '[' && s_[pos_] != ']' && s_[pos_] != '#') pos_++;
        t.kind = Tok::IDENT;
        t.text = s_.substr(b, pos_ - b);
        if (t.text.empty()) { pos_++; err_ = "unexpected character"; t.kind = Tok::END; }
        return t;
    }
    Tok peek() { size_t p = pos_; int l = line_; Tok t = next(); pos_ = p; line_ = l; return t; }
    const std::string& error() const { return err_; }
    int line() const { return line_; }

private:
    void skip_space() {
        for (;;) {
            while (pos_ < s_.size() && std::isspace((unsigned char)s_[pos_])) { if (s_[pos_] == '\n') line_++; pos_++; }
            if (pos_ < s_.size() && s_[pos_] == '#') { while (pos_ < s_.size() && s_[pos_] != '\n') pos_++; continue; }
            break;
        }
    }
    const std::string& s_;
    size_t pos_ = 0;
    int line_ = 1;
    std::string err_;
};

struct Parser {
    ParseContext& ctx;
    std::string* err;
    int depth = 0;

    bool fail(const std::string& m, int line) {
        if (err) { std::ostringstream o; o << "parse error (line " << line << "): " << m; *err = o.str(); }
        return false;
    }

    bool read_numbers(Lexer& lx, size_t n, std::vector<float>& out, const char* what) {
        out.clear();
        bool bracket = lx.peek().kind == Tok::LBRACKET;
        if (bracket) lx.next();
        for (;;) {
            Tok t = lx.peek();
            if (t.kind != Tok::NUMBER) break;
            lx.next();
            out.push_back((float)t.num);
            if (!bracket && out.size() == n) break;
        }
        if (bracket) { Tok t = lx.next(); if (t.kind != Tok::RBRACKET) return fail("expected ']'", t.line); }
        if (n && out.size() != n) { std::ostringstream o; o << what << " required " << n << " arguments"; return fail(o.str(), lx.line()); }
        return true;
    }
    bool read_string(Lexer& lx, std::string& out) {
        Tok t = lx.next();
        if (t.kind != Tok::STRING) return fail("expected a quoted string", t.line);
        out = t.text;
        return true;
    }

    // "type name" value | [ values ]   repeated until the next directive
    bool read_params(Lexer& lx, ParamSet& ps) {
        ps.base_dir = cur_dir;
        for (;;) {
            Tok t = lx.peek();
            if (t.kind != Tok::STRING) return true;
            lx.next();
            std::istringstream is(t.text);
            std::string type, name;
            is >> type >> name;
            if (type.empty() || name.empty()) return fail("parameter declaration must be \"type name\": '" + t.text + "'", t.line);
            std::vector<float> nums;
            std::vector<long long> inums;       // integers are read from the token text (exact beyond 2^24)
            std::vector<std::string> strs;
            Tok v = lx.peek();
            bool bracket = v.kind == Tok::LBRACKET;
            if (bracket) lx.next();
            for (;;) {
                Tok e = lx.peek();
                if (e.kind == Tok::NUMBER) {
                    lx.next();
                    nums.push_back((float)e.num);
                    char* ep = nullptr;
                    long long iv = std::strtoll(e.text.c_str(), &ep, 10);
                    inums.push_back((ep && *ep == 0) ? iv : (long long)e.num);
                }
                else if (e.kind == Tok::STRING && (bracket || (nums.empty() && strs.empty()))) { lx.next(); strs.push_back(e.text); }
                else if (e.kind == Tok::IDENT && (e.text == "true" || e.text == "false") && (bracket || (nums.empty() && strs.empty()))) { lx.next(); strs.push_back(e.text); }
                else break;
                if (!bracket) break;
            }
            if (bracket) { Tok e = lx.next(); if (e.kind != Tok::RBRACKET) return fail("expected ']' after values of '" + t.text + "'", e.line); }
            if (type == "float") ps.floats[name] = nums;
            else if (type == "integer") { std::vector<int> iv; for (long long f : inums) iv.push_back((int)f); ps.ints[name] = iv; }
            else if (type == "bool") { std::vector<bool> bv; for (auto& s : strs) bv.push_back(s == "true"); ps.bools[name] = bv; }
            else if (type == "string") ps.strings[name] = strs;
            else if (type == "texture") ps.textures[name] = strs;
            else if (type == "point" || type == "point3" || type == "point2") ps.points[name] = nums;
            else if (type == "vector" || type == "vector3" || type == "vector2") ps.vectors[name] = nums;
            else if (type == "normal" || type == "normal3") ps.normals[name] = nums;
            else if (type == "rgb" || type == "color") ps.rgbs[name] = nums;
            else if (type == "spectrum" && !strs.empty() && nums.empty()) {      // parser/common.rs:108-111: spectrum values are file names
                const std::string& f = strs[0];
                ps.spectrum_files[name] = (!f.empty() && f[0] == '/') ? f : (cur_dir.empty() ? f : cur_dir + "/" + f);
            }
            else if (type == "blackbody" && !nums.empty()) ps.blackbodies[name] = nums;
            else if (type == "spectrum" || type == "blackbody" || type == "xyz") ps.unsupported.push_back(type + " " + name);
            else return fail("unknown parameter type '" + type + "'", t.line);
        }
    }

    std::string cur_dir;        // directory of the text being parsed: base of Include and of .spd file names
    bool parse_text(const std::string& text, const std::string& work_dir) {
        struct DirScope { std::string& d; std::string saved; ~DirScope() { d = saved; } } scope{cur_dir, cur_dir};
        cur_dir = work_dir;
        if (++depth > 32) return fail("Include nesting too deep", 0);
        Lexer lx(text);
        for (;;) {
            Tok t = lx.next();
            if (t.kind == Tok::END) { if (!lx.error().empty()) return fail(lx.error(), t.line); break; }
            if (t.kind != Tok::IDENT) return fail("expected a directive, got '" + t.text + "'", t.line);
            const std::string& op = t.text;
            std::vector<float> v;
            std::string name, a2, a3;
            ParamSet ps;
            if (op == "Identity") ctx.pbrt_identity();
            else if (op == "Translate") { if (!read_numbers(lx, 3, v, "Translate")) return false; ctx.pbrt_translate(v[0], v[1], v[2]); }
            else if (op == "Rotate") { if (!read_numbers(lx, 4, v, "Rotate")) return false; ctx.pbrt_rotate(v[0], v[1], v[2], v[3]); }
            else if (op == "Scale") { if (!read_numbers(lx, 3, v, "Scale")) return false; ctx.pbrt_scale(v[0], v[1], v[2]); }
            else if (op == "LookAt") { if (!read_numbers(lx, 9, v, "LookAt")) return false; ctx.pbrt_look_at(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]); }
            else if (op == "ConcatTransform") { if (!read_numbers(lx, 16, v, "ConcatTransform")) return false; ctx.pbrt_concat_transform(v); }
            else if (op == "Transform") { if (!read_numbers(lx, 16, v, "Transform")) return false; ctx.pbrt_transform(v); }
            else if (op == "CoordinateSystem") { if (!read_string(lx, name)) return false; ctx.pbrt_coordinate_system(name); }
            else if (op == "CoordSysTransform") { if (!read_string(lx, name)) return false; ctx.pbrt_coord_sys_transform(name); }
            else if (op == "ActiveTransform") {
                Tok w = lx.next();
                if (w.text == "All") ctx.pbrt_active_transform_all();
                else if (w.text == "EndTime") ctx.pbrt_active_transform_end_time();
                else if (w.text == "StartTime") ctx.pbrt_active_transform_start_time();
                else return fail("ActiveTransform expects All, StartTime or EndTime", w.line);
            }
            else if (op == "ActiveTransformAll") ctx.pbrt_active_transform_all();
            else if (op == "ActiveTransformEndTime") ctx.pbrt_active_transform_end_time();
            else if (op == "ActiveTransformStartTime") ctx.pbrt_active_transform_start_time();
            else if (op == "TransformTimes") { if (!read_numbers(lx, 2, v, "TransformTimes")) return false; ctx.pbrt_transform_times(v[0], v[1]); }
            else if (op == "PixelFilter" || op == "Film" || op == "Sampler" || op == "Accelerator" || op == "Integrator" || op == "Camera" ||
                     op == "MakeNamedMedium" || op == "Material" || op == "MakeNamedMaterial" || op == "LightSource" || op == "AreaLightSource" ||
                     op == "Shape") {
                if (!read_string(lx, name) || !read_params(lx, ps)) return false;
                if (op == "PixelFilter") ctx.pbrt_pixel_filter(name, ps);
                else if (op == "Film") ctx.pbrt_film(name, ps);
                else if (op == "Sampler") ctx.pbrt_sampler(name, ps);
                else if (op == "Accelerator") ctx.pbrt_accelerator(name, ps);
                else if (op == "Integrator") ctx.pbrt_integrator(name, ps);
                else if (op == "Camera") ctx.pbrt_camera(name, ps);
                else if (op == "MakeNamedMedium") ctx.pbrt_make_named_medium(name, ps);
                else if (op == "Material") ctx.pbrt_material(name, ps);
                else if (op == "MakeNamedMaterial") ctx.pbrt_make_named_material(name, ps);
                else if (op == "LightSource") ctx.pbrt_light_source(name, ps);
                else if (op == "AreaLightSource") ctx.pbrt_area_light_source(name, ps);
                else ctx.pbrt_shape(name, ps);
            }
            else if (op == "MediumInterface") {
                if (!read_string(lx, name)) return false;
                a2 = name;
                if (lx.peek().kind == Tok::STRING && !read_string(lx, a2)) return false;
                ctx.pbrt_medium_interface(name, a2);
            }
            else if (op == "Texture") {
                if (!read_string(lx, name) || !read_string(lx, a2) || !read_string(lx, a3) || !read_params(lx, ps)) return false;
                ctx.pbrt_texture(name, a2, a3, ps);
            }
            else if (op == "NamedMaterial") { if (!read_string(lx, name)) return false; ctx.pbrt_named_material(name); }
            else if (op == "WorldBegin") ctx.pbrt_world_begin();
            else if (op == "WorldEnd") ctx.pbrt_world_end();
            else if (op == "AttributeBegin") ctx.pbrt_attribute_begin();
            else if (op == "AttributeEnd") ctx.pbrt_attribute_end();
            else if (op == "TransformBegin") ctx.pbrt_transform_begin();
            else if (op == "TransformEnd") ctx.pbrt_transform_end();
            else if (op == "ReverseOrientation") ctx.pbrt_reverse_orientation();
            else if (op == "ObjectBegin") { if (!read_string(lx, name)) return false; ctx.pbrt_object_begin(name); }
            else if (op == "ObjectEnd") ctx.pbrt_object_end();
            else if (op == "ObjectInstance") { if (!read_string(lx, name)) return false; ctx.pbrt_object_instance(name); }
            else if (op == "Include") {
                if (!read_string(lx, name)) return false;
                std::string path = (!name.empty() && name[0] == '/') ? name : (work_dir.empty() ? name : work_dir + "/" + name);
                std::ifstream f(path, std::ios::binary);
                if (!f) return fail("cannot open included file '" + path + "'", t.line);
                std::stringstream ss;
                ss << f.rdbuf();
                size_t slash = path.find_last_of('/');
                if (!parse_text(ss.str(), slash == std::string::npos ? std::string(".") : path.substr(0, slash))) return false;
            }
            else return fail("unknown directive '" + op + "'", t.line);
        }
        depth--;
        return true;
    }
};

}  // namespace

bool pbrt_parse_string(const std::string& text, const std::string& work_dir, ParseContext& ctx, std::string* err) {
    Parser p{ctx, err};
    return p.parse_text(text, work_dir);
}

bool pbrt_parse_file(const std::string& filename, ParseContext& ctx, std::string* err) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) { if (err) *err = "cannot open '" + filename + "'"; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    size_t slash = filename.find_last_of('/');
    return pbrt_parse_string(ss.str(), slash == std::string::npos ? std::string(".") : filename.substr(0, slash), ctx, err);
}

}  // namespace pth
