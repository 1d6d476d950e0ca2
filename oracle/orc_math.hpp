// ORACLE -- test infrastructure only.  CPU restatement of pbrt-r3's arithmetic;
// never linked into or called by the product (pbrt-r3_amd/).  Build with
// -ffp-contract=off -fno-fast-math: Rust never contracts a*b+c into an FMA.
//
// orc_math.hpp: Float=f32 vector/bounds/matrix helpers.
//   follows src/core/geometry/vector3.rs, bounds3.rs, misc.rs, intersect.rs,
//           src/core/misc/float.rs, src/core/transform/{matrix4x4,transform}.rs,
//           src/core/base/constants.rs
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

typedef float Float;

// core/base/constants.rs:16-36
static const Float kPi = 3.14159265358979323846f;
static const Float kInvPi = 0.31830988618379067154f;
static const Float kInv2Pi = 0.15915494309189533577f;
static const Float kPiOver2 = kPi / 2.0f;
static const Float kPiOver4 = kPi / 4.0f;
static const Float kOneMinusEpsilon = 0.99999994f;
static const Float kShadowEpsilon = 0.0001f;
static const Float kInfinity = std::numeric_limits<Float>::infinity();
static const Float kMachineEpsilon = std::numeric_limits<Float>::epsilon() * 0.5f;

// Rust's f32::max / f32::min ignore a NaN operand.
inline Float fmax_(Float a, Float b) { return std::fmax(a, b); }
inline Float fmin_(Float a, Float b) { return std::fmin(a, b); }
// Float::clamp(x, lo, hi): NaN stays NaN, else max(lo) then min(hi).
inline Float clampf(Float x, Float lo, Float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
inline Float lerpf(Float t, Float a, Float b) { return (1.0f - t) * a + t * b; }  // core/base/functions.rs lerp

struct V3 {
    Float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(Float x_, Float y_, Float z_) : x(x_), y(y_), z(z_) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, Float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(Float s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, Float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3 vabs(V3 a) { return V3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
// vector3.rs:104-106
inline Float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float abs_dot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
inline Float length_squared(V3 a) { return dot(a, a); }
inline Float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// vector3.rs:119-127: three divisions
inline V3 normalize(V3 a) { Float l = length(a); return V3(a.x / l, a.y / l, a.z / l); }
inline Float distance_squared(V3 a, V3 b) { V3 v = a - b; return dot(v, v); }
// vector3.rs:134-142: plain f32 (quirk Q5)
inline V3 cross(V3 a, V3 b) {
    return V3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
// geometry/misc.rs:25-32
inline V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0.0f ? n * -1.0f : n; }
// geometry/misc.rs:35-50
inline int max_dimension(V3 v) {
    if (v.x > v.y) return v.x > v.z ? 0 : 2;
    return v.y > v.z ? 1 : 2;
}
inline V3 permute(V3 v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }
inline Float max_component(V3 v) { return fmax_(v.x, fmax_(v.y, v.z)); }
// geometry/misc.rs:62-70
inline void coordinate_system(V3 v1, V3* v2, V3* v3) {
    if (std::fabs(v1.x) > std::fabs(v1.y))
        *v2 = V3(-v1.z, 0.0f, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3(0.0f, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = normalize(cross(v1, *v2));
}

struct V2 {
    Float x, y;
    V2() : x(0), y(0) {}
    V2(Float x_, Float y_) : x(x_), y(y_) {}
    Float operator[](int i) const { return i == 0 ? x : y; }
};
inline V2 operator+(V2 a, V2 b) { return V2(a.x + b.x, a.y + b.y); }
inline V2 operator-(V2 a, V2 b) { return V2(a.x - b.x, a.y - b.y); }
inline V2 operator*(V2 a, Float s) { return V2(a.x * s, a.y * s); }
inline V2 operator*(Float s, V2 a) { return V2(s * a.x, s * a.y); }

// core/misc/float.rs:23-56
inline uint32_t f2b(Float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline Float b2f(uint32_t u) { Float f; std::memcpy(&f, &u, 4); return f; }
inline Float next_float_down(Float v) {
    if (std::isinf(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t ui = f2b(v);
    if (v > 0.0f) ui = ui == 0 ? 0 : ui - 1;            // saturating_sub
    else ui = ui == 0xffffffffu ? ui : ui + 1;           // saturating_add
    return b2f(ui);
}
inline Float next_float_up(Float v) {
    if (std::isinf(v) && v > 0.0f) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t ui = f2b(v);
    if (v >= 0.0f) ui = ui == 0xffffffffu ? ui : ui + 1;
    else ui = ui == 0 ? 0 : ui - 1;
    return b2f(ui);
}
// geometry/misc.rs:5-23
inline V3 offset_ray_origin(V3 p, V3 p_error, V3 n, V3 w) {
    Float d = dot(vabs(n), p_error);
    V3 offset = d * n;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = p + offset;
    for (int i = 0; i < 3; i++) {
        if (offset[i] > 0.0f) po[i] = next_float_up(po[i]);
        else if (offset[i] < 0.0f) po[i] = next_float_down(po[i]);
    }
    return po;
}

// geometry/bounds3.rs
struct Bounds3 {
    V3 min, max;
    // Default: inverted (bounds3.rs:269-276; Float::MIN is the lowest finite f32)
    Bounds3() : min(std::numeric_limits<Float>::max(), std::numeric_limits<Float>::max(), std::numeric_limits<Float>::max()),
                max(std::numeric_limits<Float>::lowest(), std::numeric_limits<Float>::lowest(), std::numeric_limits<Float>::lowest()) {}
    Bounds3(V3 a, V3 b) {
        min = V3(a.x <= b.x ? a.x : b.x, a.y <= b.y ? a.y : b.y, a.z <= b.z ? a.z : b.z);
        max = V3(a.x >= b.x ? a.x : b.x, a.y >= b.y ? a.y : b.y, a.z >= b.z ? a.z : b.z);
    }
    V3 diagonal() const { return max - min; }
    int maximum_extent() const {
        V3 d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        if (d.y > d.z) return 1;
        return 2;
    }
    V3 offset(V3 p) const {
        V3 o = p - min;
        if (max.x > min.x) o.x = o.x / (max.x - min.x);
        if (max.y > min.y) o.y = o.y / (max.y - min.y);
        if (max.z > min.z) o.z = o.z / (max.z - min.z);
        return o;
    }
    Float surface_area() const {
        V3 d = diagonal();
        return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    V3 lerp(V3 t) const { return V3(lerpf(t.x, min.x, max.x), lerpf(t.y, min.y, max.y), lerpf(t.z, min.z, max.z)); }
    // bounds3.rs:184-195
    Float distance_squared(V3 p) const {
        Float d = 0.0f;
        for (int i = 0; i < 3; i++) {
            Float a = min[i] - p[i], b = p[i] - max[i];
            Float m = a >= b ? a : b;
            Float delta = 0.0f >= m ? 0.0f : m;
            d += delta * delta;
        }
        return d;
    }
    Float distance(V3 p) const { return std::sqrt(distance_squared(p)); }
};
inline Float min_le(Float a, Float b) { return a <= b ? a : b; }   // bounds3.rs:40-46
inline Float max_ge(Float a, Float b) { return a >= b ? a : b; }
inline Bounds3 bunion(const Bounds3& a, const Bounds3& b) {
    Bounds3 r;
    r.min = V3(min_le(a.min.x, b.min.x), min_le(a.min.y, b.min.y), min_le(a.min.z, b.min.z));
    r.max = V3(max_ge(a.max.x, b.max.x), max_ge(a.max.y, b.max.y), max_ge(a.max.z, b.max.z));
    return r;
}
inline Bounds3 bunion_p(const Bounds3& a, V3 p) {
    Bounds3 r;
    r.min = V3(min_le(a.min.x, p.x), min_le(a.min.y, p.y), min_le(a.min.z, p.z));
    r.max = V3(max_ge(a.max.x, p.x), max_ge(a.max.y, p.y), max_ge(a.max.z, p.z));
    return r;
}

// core/transform/matrix4x4.rs (row-major m[16])
struct Mat4 {
    Float m[16];
    static Mat4 identity() {
        Mat4 r;
        for (int i = 0; i < 16; i++) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return r;
    }
    static Mat4 translate(Float x, Float y, Float z) {
        Mat4 r = identity();
        r.m[3] = x; r.m[7] = y; r.m[11] = z;
        return r;
    }
    static Mat4 scale(Float x, Float y, Float z) {
        Mat4 r = identity();
        r.m[0] = x; r.m[5] = y; r.m[10] = z;
        return r;
    }
    // matrix4x4.rs:231-309 (Gauss-Jordan with full pivoting)
    bool inverse(Mat4* out) const {
        int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
        Float minv[16];
        std::memcpy(minv, m, sizeof(minv));
        for (int i = 0; i < 4; i++) {
            int irow = 0, icol = 0;
            Float big = 0.0f;
            for (int j = 0; j < 4; j++) {
                if (ipiv[j] != 1) {
                    for (int k = 0; k < 4; k++) {
                        if (ipiv[k] == 0) {
                            if (std::fabs(minv[4 * j + k]) >= big) {
                                big = std::fabs(minv[4 * j + k]);
                                irow = j;
                                icol = k;
                            }
                        } else if (ipiv[k] > 1) {
                            return false;
                        }
                    }
                }
            }
            ipiv[icol] += 1;
            if (irow != icol)
                for (int k = 0; k < 4; k++) { Float t = minv[4 * irow + k]; minv[4 * irow + k] = minv[4 * icol + k]; minv[4 * icol + k] = t; }
            indxr[i] = irow;
            indxc[i] = icol;
            if (minv[4 * icol + icol] == 0.0f) return false;
            Float pivinv = 1.0f / minv[4 * icol + icol];
            minv[4 * icol + icol] = 1.0f;
            for (int j = 0; j < 4; j++) minv[4 * icol + j] *= pivinv;
            for (int j = 0; j < 4; j++) {
                if (j != icol) {
                    Float save = minv[4 * j + icol];
                    minv[4 * j + icol] = 0.0f;
                    for (int k = 0; k < 4; k++) minv[4 * j + k] -= minv[4 * icol + k] * save;
                }
            }
        }
        for (int j = 3; j >= 0; j--) {
            if (indxr[j] != indxc[j])
                for (int k = 0; k < 4; k++) { Float t = minv[4 * k + indxr[j]]; minv[4 * k + indxr[j]] = minv[4 * k + indxc[j]]; minv[4 * k + indxc[j]] = t; }
        }
        std::memcpy(out->m, minv, sizeof(minv));
        return true;
    }
    // matrix4x4.rs:311-324
    V3 transform_point(V3 p) const {
        Float x = p.x, y = p.y, z = p.z;
        Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
        Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
        Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
        Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
        if (wp == 1.0f) return V3(xp, yp, zp);
        return V3(xp / wp, yp / wp, zp / wp);
    }
    V3 transform_vector(V3 p) const {
        Float x = p.x, y = p.y, z = p.z;
        return V3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
    }
};
// matrix4x4.rs:351-380
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[4 * i + j] = a.m[4 * i + 0] * b.m[0 + j] + a.m[4 * i + 1] * b.m[4 + j] + a.m[4 * i + 2] * b.m[8 + j] + a.m[4 * i + 3] * b.m[12 + j];
    return r;
}

// core/transform/transform.rs
struct Transform {
    Mat4 m, minv;
    static Transform identity() { Transform t; t.m = Mat4::identity(); t.minv = Mat4::identity(); return t; }
    static Transform translate(Float x, Float y, Float z) { Transform t; t.m = Mat4::translate(x, y, z); t.minv = Mat4::translate(-x, -y, -z); return t; }
    static Transform scale(Float x, Float y, Float z) { Transform t; t.m = Mat4::scale(x, y, z); t.minv = Mat4::scale(1.0f / x, 1.0f / y, 1.0f / z); return t; }
    static Transform from_matrix(const Mat4& mm) { Transform t; t.m = mm; mm.inverse(&t.minv); return t; }
    Transform inverse() const { Transform t; t.m = minv; t.minv = m; return t; }
    // transform.rs:89-99
    static Transform perspective(Float fov, Float n, Float f);
    V3 transform_point(V3 p) const { return m.transform_point(p); }
    V3 transform_vector(V3 v) const { return m.transform_vector(v); }
};
inline Transform operator*(const Transform& a, const Transform& b) {
    Transform t;
    t.m = a.m * b.m;
    t.minv = b.minv * a.minv;
    return t;
}
inline Float radians(Float x) { return x * (kPi / 180.0f); }
inline Transform Transform::perspective(Float fov, Float n, Float f) {
    Mat4 persp = Mat4::identity();
    persp.m[10] = f / (f - n);
    persp.m[11] = -f * n / (f - n);
    persp.m[14] = 1.0f;
    persp.m[15] = 0.0f;
    Float inv_tan_ang = 1.0f / std::tan(radians(fov) / 2.0f);
    return Transform::scale(inv_tan_ang, inv_tan_ang, 1.0f) * Transform::from_matrix(persp);
}

// core/spectrum/rgb.rs, convert.rs
struct RGB {
    Float c[3];
    RGB() { c[0] = c[1] = c[2] = 0.0f; }
    RGB(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    explicit RGB(Float v) { c[0] = c[1] = c[2] = v; }
    bool is_black() const { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
    Float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
    Float max_component_value() const { return fmax_(fmax_(c[0], c[1]), c[2]); }
    bool is_valid() const { return std::isfinite(c[0]) && std::isfinite(c[1]) && std::isfinite(c[2]); }
};
inline RGB operator+(RGB a, RGB b) { return RGB(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
inline RGB operator*(RGB a, RGB b) { return RGB(a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2]); }
inline RGB operator*(RGB a, Float s) { return RGB(a.c[0] * s, a.c[1] * s, a.c[2] * s); }
inline RGB operator*(Float s, RGB a) { return RGB(s * a.c[0], s * a.c[1], s * a.c[2]); }
inline RGB operator/(RGB a, Float s) { return RGB(a.c[0] / s, a.c[1] / s, a.c[2] / s); }
inline RGB& operator+=(RGB& a, RGB b) { a = a + b; return a; }
inline void rgb_to_xyz(const Float rgb[3], Float xyz[3]) {
    xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
    xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
    xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}
inline void xyz_to_rgb(const Float xyz[3], Float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

}  // namespace orc
