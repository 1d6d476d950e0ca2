// pt_host_math.h -- the few host-side f32 matrix routines scene upload needs to turn
// camera parameters into the raster->camera matrix exactly as pbrt-r3 does
// (src/core/camera/projective.rs:23-53, src/cameras/perspective.rs:28-44,
//  src/core/transform/{transform.rs:35-48,:89-99,:418-426, matrix4x4.rs:231-309,:351-380}).
// A transform is kept as the pair (m, inverse m) like the reference's Transform, because
// products multiply the inverses separately instead of re-inverting.
#pragma once
#include <cmath>
#include <cstring>

namespace pth {

struct M44 { float a[16]; };

inline M44 m_identity() { M44 r; for (int i = 0; i < 16; i++) r.a[i] = (i % 5 == 0) ? 1.0f : 0.0f; return r; }
inline M44 m_mul(const M44& x, const M44& y) {
    M44 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.a[4 * i + j] = x.a[4 * i] * y.a[j] + x.a[4 * i + 1] * y.a[4 + j] + x.a[4 * i + 2] * y.a[8 + j] + x.a[4 * i + 3] * y.a[12 + j];
    return r;
}
// Gauss-Jordan elimination with full pivoting, same pivot order and operation order as the reference.
inline bool m_inverse(const M44& src, M44* dst) {
    int col_of[4], row_of[4], used[4] = {0, 0, 0, 0};
    float w[16];
    std::memcpy(w, src.a, sizeof(w));
    for (int step = 0; step < 4; step++) {
        int prow = 0, pcol = 0;
        float best = 0.0f;
        for (int r = 0; r < 4; r++) {
            if (used[r] == 1) continue;
            for (int c = 0; c < 4; c++) {
                if (used[c] == 0) {
                    float v = std::fabs(w[4 * r + c]);
                    if (v >= best) { best = v; prow = r; pcol = c; }
                } else if (used[c] > 1) {
                    return false;
                }
            }
        }
        used[pcol]++;
        if (prow != pcol)
            for (int k = 0; k < 4; k++) std::swap(w[4 * prow + k], w[4 * pcol + k]);
        row_of[step] = prow;
        col_of[step] = pcol;
        if (w[4 * pcol + pcol] == 0.0f) return false;
        float inv = 1.0f / w[4 * pcol + pcol];
        w[4 * pcol + pcol] = 1.0f;
        for (int k = 0; k < 4; k++) w[4 * pcol + k] *= inv;
        for (int r = 0; r < 4; r++) {
            if (r == pcol) continue;
            float f = w[4 * r + pcol];
            w[4 * r + pcol] = 0.0f;
            for (int k = 0; k < 4; k++) w[4 * r + k] -= w[4 * pcol + k] * f;
        }
    }
    for (int step = 3; step >= 0; step--)
        if (row_of[step] != col_of[step])
            for (int k = 0; k < 4; k++) std::swap(w[4 * k + row_of[step]], w[4 * k + col_of[step]]);
    std::memcpy(dst->a, w, sizeof(w));
    return true;
}

struct Xf { M44 m, inv; };
inline Xf xf_scale(float x, float y, float z) {
    Xf t; t.m = m_identity(); t.inv = m_identity();
    t.m.a[0] = x; t.m.a[5] = y; t.m.a[10] = z;
    t.inv.a[0] = 1.0f / x; t.inv.a[5] = 1.0f / y; t.inv.a[10] = 1.0f / z;
    return t;
}
inline Xf xf_translate(float x, float y, float z) {
    Xf t; t.m = m_identity(); t.inv = m_identity();
    t.m.a[3] = x; t.m.a[7] = y; t.m.a[11] = z;
    t.inv.a[3] = -x; t.inv.a[7] = -y; t.inv.a[11] = -z;
    return t;
}
inline Xf xf_mul(const Xf& p, const Xf& q) { Xf t; t.m = m_mul(p.m, q.m); t.inv = m_mul(q.inv, p.inv); return t; }
inline Xf xf_inverse(const Xf& p) { Xf t; t.m = p.inv; t.inv = p.m; return t; }
inline bool xf_perspective(float fov_deg, float n, float f, Xf* out) {
    M44 persp = m_identity();
    persp.a[10] = f / (f - n);
    persp.a[11] = -f * n / (f - n);
    persp.a[14] = 1.0f;
    persp.a[15] = 0.0f;
    Xf p;
    p.m = persp;
    if (!m_inverse(persp, &p.inv)) return false;
    const float pi = 3.14159265358979323846f;
    float radians = fov_deg * (pi / 180.0f);
    float inv_tan = 1.0f / std::tan(radians / 2.0f);
    *out = xf_mul(xf_scale(inv_tan, inv_tan, 1.0f), p);
    return true;
}
// raster -> camera for a perspective camera with the given screen window and film resolution
inline bool raster_to_camera(float fov_deg, const float sw[4] /* x0 x1 y0 y1 */, int xres, int yres, M44* out) {
    Xf cam_to_screen;
    if (!xf_perspective(fov_deg, 1e-2f, 1000.0f, &cam_to_screen)) return false;
    Xf screen_to_raster = xf_mul(xf_mul(xf_scale((float)xres, (float)yres, 1.0f), xf_scale(1.0f / (sw[1] - sw[0]), 1.0f / (sw[2] - sw[3]), 1.0f)),
                                 xf_translate(-sw[0], -sw[3], 0.0f));
    Xf r2c = xf_mul(xf_inverse(cam_to_screen), xf_inverse(screen_to_raster));
    *out = r2c.m;
    return true;
}

}  // namespace pth
