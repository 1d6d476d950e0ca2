// pth_ply.cpp -- see pth_ply.h
#include "pth_ply.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <sstream>

namespace pth {

namespace {

enum Type { T_NONE = 0, T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64 };
Type type_of(const std::string& s) {
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_NONE;
}
int size_of(Type t) { return (t == T_I8 || t == T_U8) ? 1 : (t == T_I16 || t == T_U16) ? 2 : (t == T_F64) ? 8 : 4; }

struct Property { std::string name; Type type = T_NONE; bool list = false; Type count_type = T_NONE; };
struct Element { std::string name; size_t count = 0; std::vector<Property> props; };

struct Reader {
    gzFile f = nullptr;      // zlib reads plain files transparently, so one path serves .ply and .ply.gz
    bool swap = false, ascii = false;
    ~Reader() { if (f) gzclose(f); }
    bool line(std::string* out) {
        out->clear();
        char buf[4096];
        for (;;) {
            if (!gzgets(f, buf, sizeof(buf))) return !out->empty();
            out->append(buf);
            if (!out->empty() && out->back() == '\n') break;
        }
        while (!out->empty() && (out->back() == '\n' || out->back() == '\r')) out->pop_back();
        return true;
    }
    bool bytes(void* dst, int n) { return gzread(f, dst, (unsigned)n) == n; }
    // next whitespace-separated token of an ASCII payload
    bool token(std::string* out) {
        out->clear();
        int c;
        while ((c = gzgetc(f)) != -1 && (c == ' ' || c == '\t' || c == '\n' || c == '\r')) {}
        if (c == -1) return false;
        do { out->push_back((char)c); } while ((c = gzgetc(f)) != -1 && !(c == ' ' || c == '\t' || c == '\n' || c == '\r'));
        return true;
    }
    bool scalar(Type t, double* out) {
        if (ascii) {
            std::string tok;
            if (!token(&tok)) return false;
            char* e = nullptr;
            *out = std::strtod(tok.c_str(), &e);
            return e != tok.c_str();
        }
        unsigned char b[8];
        int n = size_of(t);
        if (!bytes(b, n)) return false;
        if (swap) for (int i = 0; i < n / 2; i++) std::swap(b[i], b[n - 1 - i]);
        switch (t) {
            case T_I8: *out = (double)(int8_t)b[0]; break;
            case T_U8: *out = (double)b[0]; break;
            case T_I16: { int16_t v; std::memcpy(&v, b, 2); *out = v; break; }
            case T_U16: { uint16_t v; std::memcpy(&v, b, 2); *out = v; break; }
            case T_I32: { int32_t v; std::memcpy(&v, b, 4); *out = v; break; }
            case T_U32: { uint32_t v; std::memcpy(&v, b, 4); *out = v; break; }
            case T_F32: { float v; std::memcpy(&v, b, 4); *out = v; break; }
            case T_F64: { double v; std::memcpy(&v, b, 8); *out = v; break; }
            default: return false;
        }
        return true;
    }
};

int vertex_slot(const std::string& n) {     // plymesh.rs:54-118
    if (n == "x") return 0; if (n == "y") return 1; if (n == "z") return 2;
    if (n == "nx") return 3; if (n == "ny") return 4; if (n == "nz") return 5;
    if (n == "u" || n == "s" || n == "texture_u" || n == "texture_s") return 6;
    if (n == "v" || n == "t" || n == "texture_v" || n == "texture_t") return 7;
    return -1;
}

}  // namespace

bool read_ply(const std::string& path, PlyMesh* out, std::string* err) {
    auto fail = [&](const std::string& m) { if (err) *err = m + " (\"" + path + "\")"; return false; };
    Reader r;
    r.f = gzopen(path.c_str(), "rb");
    if (!r.f) return fail("Unable to open file");
    std::string ln;
    if (!r.line(&ln) || ln != "ply") return fail("not a PLY file");
    std::vector<Element> elems;
    bool have_format = false;
    for (;;) {
        if (!r.line(&ln)) return fail("unexpected end of header");
        std::istringstream is(ln);
        std::string kw;
        is >> kw;
        if (kw == "end_header") break;
        if (kw == "comment" || kw == "obj_info" || kw.empty()) continue;
        if (kw == "format") {
            std::string fmt;
            is >> fmt;
            uint16_t one = 1;
            bool host_le = *reinterpret_cast<unsigned char*>(&one) == 1;
            if (fmt == "ascii") r.ascii = true;
            else if (fmt == "binary_little_endian") r.swap = !host_le;
            else if (fmt == "binary_big_endian") r.swap = host_le;
            else return fail("unknown PLY format '" + fmt + "'");
            have_format = true;
        } else if (kw == "element") {
            Element e;
            is >> e.name >> e.count;
            elems.push_back(e);
        } else if (kw == "property") {
            if (elems.empty()) return fail("property before any element");
            Property p;
            std::string t;
            is >> t;
            if (t == "list") {
                std::string ct, it;
                is >> ct >> it >> p.name;
                p.list = true; p.count_type = type_of(ct); p.type = type_of(it);
                if (p.count_type == T_NONE) return fail("unknown PLY type '" + ct + "'");
            } else {
                is >> p.name;
                p.type = type_of(t);
            }
            if (p.type == T_NONE) return fail("unknown PLY property type in '" + ln + "'");
            elems.back().props.push_back(p);
        } else {
            return fail("unexpected header line '" + ln + "'");
        }
    }
    if (!have_format) return fail("PLY header has no format line");

    out->P.clear(); out->N.clear(); out->UV.clear(); out->indices.clear();
    for (const Element& e : elems) {
        if (e.name == "vertex") {
            bool has[8] = {false, false, false, false, false, false, false, false};
            for (const Property& p : e.props) {
                int s = p.list ? -1 : vertex_slot(p.name);
                if (s >= 0) {
                    if (p.type != T_F32) return fail("vertex property '" + p.name + "' must be float (the reference accepts nothing else)");
                    has[s] = true;
                }
            }
            const bool has_p = has[0] || has[1] || has[2], has_n = has[3] || has[4] || has[5], has_uv = has[6] || has[7];
            if (has_p) out->P.reserve(3 * e.count);
            for (size_t i = 0; i < e.count; i++) {
                float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (const Property& p : e.props) {
                    if (p.list) {
                        double n;
                        if (!r.scalar(p.count_type, &n)) return fail("truncated vertex data");
                        for (long k = 0; k < (long)n; k++) { double d; if (!r.scalar(p.type, &d)) return fail("truncated vertex data"); }
                        continue;
                    }
                    double d;
                    if (!r.scalar(p.type, &d)) return fail("truncated vertex data");
                    int s = vertex_slot(p.name);
                    if (s >= 0) v[s] = (float)d;
                }
                if (has_p) { out->P.push_back(v[0]); out->P.push_back(v[1]); out->P.push_back(v[2]); }
                if (has_n) { out->N.push_back(v[3]); out->N.push_back(v[4]); out->N.push_back(v[5]); }
                if (has_uv) { out->UV.push_back(v[6]); out->UV.push_back(v[7]); }
            }
        } else {
            const bool is_face = e.name == "face";
            if (is_face) out->indices.reserve(3 * e.count);
            for (size_t i = 0; i < e.count; i++) {
                for (const Property& p : e.props) {
                    if (!p.list) { double d; if (!r.scalar(p.type, &d)) return fail("truncated " + e.name + " data"); continue; }
                    double n;
                    if (!r.scalar(p.count_type, &n)) return fail("truncated " + e.name + " data");
                    const bool idx_list = is_face && (p.name == "vertex_indices" || p.name == "vertex_index");
                    if (idx_list && p.type != T_I32 && p.type != T_U32) return fail("face indices must be int or uint lists (the reference accepts nothing else)");
                    long cnt = (long)n;
                    uint32_t vi[4] = {0, 0, 0, 0};
                    for (long k = 0; k < cnt; k++) {
                        double d;
                        if (!r.scalar(p.type, &d)) return fail("truncated " + e.name + " data");
                        if (idx_list && k < 4) vi[k] = (uint32_t)(int32_t)d;
                    }
                    if (!idx_list) continue;
                    if (cnt == 3) { out->indices.push_back(vi[0]); out->indices.push_back(vi[1]); out->indices.push_back(vi[2]); }
                    else if (cnt == 4) {
                        out->indices.push_back(vi[0]); out->indices.push_back(vi[1]); out->indices.push_back(vi[2]);
                        out->indices.push_back(vi[3]); out->indices.push_back(vi[0]); out->indices.push_back(vi[2]);
                    } else {
                        std::ostringstream m;
                        m << "Ignoring face with " << cnt << " vertices (only triangles and quads are supported!)";
                        return fail(m.str());
                    }
                }
            }
        }
    }
    return true;
}

}  // namespace pth
