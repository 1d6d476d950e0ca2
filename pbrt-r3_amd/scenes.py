"""Scene assembly: the host-side mirror of what pbrt-r3's SceneContext does between
the parser callbacks and Integrator::render, flattened into a pt_scene_desc.

  SceneBuilder.look_at / camera / film / sampler / integrator / accelerator
      <- pbrt_look_at, pbrt_camera, pbrt_film, pbrt_sampler, pbrt_integrator,
         pbrt_accelerator (src/core/api/parse_context.rs:5-66) with the defaults of
         render_options.rs:60-88 where this path supports them
  SceneBuilder.material / area_light_source / shape_trianglemesh
      <- pbrt_material, pbrt_area_light_source, pbrt_shape("trianglemesh")
         (scene_context.rs:1201-1318, shapes/triangle.rs:696-868)

All arithmetic that feeds the renderer is done in float32 in the reference's
operation order (the reference is f32 throughout, base/types.rs:3-6).

cornell_box() and rt1m() are the two synthetic configs of BASELINE.md.
"""
import ctypes as C
import numpy as np

from . import capi

f32 = np.float32

_libm = C.CDLL("libm.so.6")
_libm.expf.argtypes = [C.c_float]
_libm.expf.restype = C.c_float


def _expf(x):
    return f32(_libm.expf(float(x)))


_libm.sinf.argtypes = [C.c_float]
_libm.sinf.restype = C.c_float


def _sinf(x):
    return f32(_libm.sinf(float(x)))


_libm.cosf.argtypes = [C.c_float]
_libm.cosf.restype = C.c_float


def _cosf(x):
    return f32(_libm.cosf(float(x)))


# --------------------------------------------------------------------------
# PCG32 (src/core/rng.rs:8-67), vectorised: the LCG state after k steps is
# a^k * s0 + c * (a^(k-1) + ... + 1)  (mod 2^64), so all states come from two
# wrapping cumulative ops.
def pcg32_uint32(n, sequence=None):
    M = np.uint64(0x5851f42d4c957f2d)
    with np.errstate(over="ignore"):
        if sequence is None:
            state0 = np.uint64(0x853c49e6748fea9b)
            inc = np.uint64(0xda3e39cb94b95bdb)
        else:
            inc = np.uint64(((int(sequence) << 1) | 1) & 0xFFFFFFFFFFFFFFFF)
            s = np.uint64(0)
            s = s * M + inc                      # first uniform_uint32()
            s = s + np.uint64(0x853c49e6748fea9b)
            s = s * M + inc                      # second
            state0 = s
        pw = np.full(n, M, np.uint64)
        pw[0] = np.uint64(1)
        pw = np.cumprod(pw, dtype=np.uint64)             # a^k, k = 0..n-1
        geo = np.cumsum(pw, dtype=np.uint64)             # 1 + a + ... + a^k
        old = np.empty(n, np.uint64)
        old[0] = state0
        if n > 1:
            old[1:] = pw[1:] * state0 + inc * geo[:-1]
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xorshifted >> rot) | (xorshifted << ((~rot + np.uint32(1)) & np.uint32(31)))


def pcg32_uniform_float(n, sequence=None):
    u = pcg32_uint32(n, sequence)
    f = u.astype(np.float32) * f32(2.3283064365386963e-10)
    return np.minimum(f32(0.99999994), f)


# --------------------------------------------------------------------------
# f32 helpers in the reference's operation order
def _v(x, y, z):
    return np.array([x, y, z], np.float32)


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b):
    return _v(f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0]))


def _length(a):
    return np.sqrt(f32(f32(f32(a[0] * a[0]) + f32(a[1] * a[1])) + f32(a[2] * a[2])))


def _normalize(a):
    l = _length(a)
    return _v(a[0] / l, a[1] / l, a[2] / l)


def camera_to_world_look_at(eye, look, up):
    """Matrix4x4::camera_to_world (src/core/transform/matrix4x4.rs:169-217)."""
    pos = _v(*eye)
    lk = _v(*look)
    upn = _normalize(_v(*up))
    d = _normalize(lk - pos)
    right = _cross(upn, d)
    assert _length(right) != 0
    right = _normalize(right)
    new_up = _normalize(_cross(d, right))
    return np.array([right[0], new_up[0], d[0], pos[0],
                     right[1], new_up[1], d[1], pos[1],
                     right[2], new_up[2], d[2], pos[2],
                     0, 0, 0, 1], np.float32)


def _transform_points(m, P):
    """Matrix4x4::transform_point (matrix4x4.rs:311-324), vectorised, f32."""
    m = np.asarray(m, np.float32)
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    xp = ((m[0] * x + m[1] * y) + m[2] * z) + m[3]
    yp = ((m[4] * x + m[5] * y) + m[6] * z) + m[7]
    zp = ((m[8] * x + m[9] * y) + m[10] * z) + m[11]
    wp = ((m[12] * x + m[13] * y) + m[14] * z) + m[15]
    out = np.stack([xp, yp, zp], 1).astype(np.float32)
    nz = wp != f32(1.0)
    if np.any(nz):
        out[nz] = (out[nz] / wp[nz, None]).astype(np.float32)
    return out


def transform_translate(x, y, z):
    """Transform::translate (transform.rs:33-37): (m, m_inv) row-major."""
    m, mi = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
    m[:3, 3] = [x, y, z]
    mi[:3, 3] = [-x, -y, -z]
    return m.reshape(-1), mi.reshape(-1)


def transform_scale(x, y, z):
    """Transform::scale (transform.rs:39-43)."""
    m, mi = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = x, y, z
    mi[0, 0], mi[1, 1], mi[2, 2] = f32(1.0) / f32(x), f32(1.0) / f32(y), f32(1.0) / f32(z)
    return m.reshape(-1), mi.reshape(-1)


def transform_rotate_x(theta_deg):
    """Transform::rotate_x (transform.rs:45-49, matrix4x4.rs:67-75); sin/cos through the libm port above."""
    r = f32(theta_deg) * (f32(np.pi) / f32(180.0))
    sn, cs = _sinf(r), _cosf(r)
    m = np.eye(4, dtype=np.float32)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = cs, -sn, sn, cs
    return m.reshape(-1), m.T.copy().reshape(-1)


def _mul4(a, b):
    """Matrix4x4 * Matrix4x4 (matrix4x4.rs:320-348): row . column, left to right, f32."""
    a, b = np.asarray(a, np.float32).reshape(4, 4), np.asarray(b, np.float32).reshape(4, 4)
    out = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            acc = a[i, 0] * b[0, j]
            for k in range(1, 4):
                acc = f32(acc + a[i, k] * b[k, j])
            out[i, j] = acc
    return out.reshape(-1)


def transform_mul(t1, t2):
    """Transform * Transform (transform.rs:276-281): m = m1*m2, m_inv = m2_inv*m1_inv."""
    return _mul4(t1[0], t2[0]), _mul4(t2[1], t1[1])


def _swaps_handedness(m):
    m = np.asarray(m, np.float32)
    det = (m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8])) + m[2] * (m[4] * m[9] - m[5] * m[8])
    return bool(det < 0)


def _tri_areas(P, idx):
    p0, p1, p2 = P[idx[:, 0]], P[idx[:, 1]], P[idx[:, 2]]
    a, b = (p1 - p0).astype(np.float32), (p2 - p0).astype(np.float32)
    cx = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    cy = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    cz = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return f32(0.5) * np.sqrt((cx * cx + cy * cy) + cz * cz)


def _is_fillable_uv(idx, n_vertices):
    """triangle.rs:733-752, sequential exactly as the reference (slot 0 doubles as 'unset')."""
    check = [0] * n_vertices
    for face in idx.tolist():
        for j in range(3):
            v = face[j]
            if check[v] == 0 or check[v] == j:
                check[v] = j
            else:
                return False
    return True


class SceneDesc:
    """A filled pt_scene_desc plus the numpy buffers it points into."""

    def __init__(self):
        self.desc = capi.pt_scene_desc()
        self.buffers = {}

    def _set(self, name, arr, ctype):
        self.buffers[name] = arr
        setattr(self.desc, name, arr.ctypes.data_as(C.POINTER(ctype)) if arr is not None and arr.size else None)


class Tex:
    """Handle of a texture added to a SceneBuilder."""
    def __init__(self, index):
        self.index = index


class SceneBuilder:
    def __init__(self):
        self.P, self.N, self.S, self.UV = [], [], [], []
        self.idx, self.tri_mesh = [], []
        self.meshes, self.materials, self.area_lights = [], [], []
        self.spheres = []
        self.textures = []
        self.images = []            # (pt_image, texel buffer)
        self.instances = []
        self.objects = {}           # name -> object index
        self.cur_object = 0         # 0 = world, k = inside ObjectBegin of object k - 1
        self.n_extra = 0            # creation counter of spheres and instances (ties at one before_triangle)
        self.n_vertices = 0
        self.cur_material = self._add_material(capi.PT_MATERIAL_MATTE, (0.5, 0.5, 0.5), 0.0)   # default matte
        self.cur_area_light = -1
        self.reverse_orientation = False
        # render options (render_options.rs:60-88 where supported)
        self.camera_to_world = np.eye(4, dtype=np.float32).reshape(-1)
        self.fov, self.screen_window = 90.0, None
        self.lens_radius, self.focal_distance = 0.0, 1e6
        self.shutter = (0.0, 1.0)
        self.xres, self.yres, self.crop = 1280, 720, (0.0, 1.0, 0.0, 1.0)
        self.filter_radius, self.filter_table = (0.5, 0.5), np.ones(256, np.float32)
        self.film_scale, self.max_sample_luminance = 1.0, float("inf")
        self.spp, self.max_depth, self.rr_threshold = 16, 5, 1.0
        self.integrator, self.ao_samples, self.ao_cos_sample = capi.PT_INTEGRATOR_PATH, 64, True
        self.direct_strategy = capi.PT_DIRECT_ALL
        self.light_strategy = capi.PT_LIGHTS_SPATIAL
        self.split_method, self.max_node_prims = capi.PT_SPLIT_SAH, 4

    # ---- options
    def look_at(self, eye, look, up):
        self.camera_to_world = camera_to_world_look_at(eye, look, up)

    def camera_perspective(self, fov=90.0, lensradius=0.0, focaldistance=1e6, screenwindow=None, frameaspectratio=None,
                           shutteropen=0.0, shutterclose=1.0):
        self.fov, self.lens_radius, self.focal_distance = fov, lensradius, focaldistance
        self.screen_window, self.frameaspectratio = screenwindow, frameaspectratio
        self.shutter = (min(shutteropen, shutterclose), max(shutteropen, shutterclose))

    def film(self, xresolution=1280, yresolution=720, cropwindow=(0.0, 1.0, 0.0, 1.0), scale=1.0, maxsampleluminance=float("inf")):
        self.xres, self.yres, self.crop = int(xresolution), int(yresolution), tuple(cropwindow)
        self.film_scale, self.max_sample_luminance = scale, maxsampleluminance

    def pixel_filter_box(self, xwidth=0.5, ywidth=0.5):
        """filters/box_filter.rs + Film::new's 16x16 table (film.rs:102-120): BoxFilter::evaluate == 1."""
        self.filter_radius, self.filter_table = (xwidth, ywidth), np.ones(256, np.float32)

    def _filter_table(self, radius, evaluate):
        """Film::new (film.rs:102-120): table[y*16+x] = filter.evaluate(x*rx/15, y*ry/15)."""
        rx, ry = f32(radius[0]), f32(radius[1])
        tab = np.empty(256, np.float32)
        for y in range(16):
            for x in range(16):
                tab[y * 16 + x] = evaluate(f32(x) * f32(rx / f32(15)), f32(y) * f32(ry / f32(15)))
        self.filter_radius, self.filter_table = (float(rx), float(ry)), tab

    def pixel_filter_gaussian(self, xwidth=2.0, ywidth=2.0, alpha=2.0):
        """filters/gaussian.rs:12-31 (expf through libm, as the reference)."""
        a = f32(alpha)
        ex, ey = _expf(-a * f32(xwidth) * f32(xwidth)), _expf(-a * f32(ywidth) * f32(ywidth))
        g = lambda d, e: max(f32(0.0), f32(_expf(f32(f32(-a * d) * d)) - e))
        self._filter_table((xwidth, ywidth), lambda x, y: f32(g(x, ex) * g(y, ey)))

    def pixel_filter_triangle(self, xwidth=2.0, ywidth=2.0):
        """filters/triangle.rs:17-21."""
        rx, ry = f32(xwidth), f32(ywidth)
        self._filter_table((xwidth, ywidth), lambda x, y: f32(max(f32(0.0), f32(rx - abs(x))) * max(f32(0.0), f32(ry - abs(y)))))

    def pixel_filter_mitchell(self, xwidth=2.0, ywidth=2.0, B=1.0 / 3.0, C=1.0 / 3.0):
        """filters/mitchell.rs:20-47."""
        b, c = f32(B), f32(C)

        def m1(x):
            x = abs(f32(f32(2.0) * x))
            if x > 1.0:
                v = f32(f32(f32(f32(f32(f32(-b - f32(6.0) * c) * x) * x) * x + f32(f32(f32(f32(6.0) * b + f32(30.0) * c) * x) * x)) +
                            f32(f32(f32(-12.0) * b - f32(48.0) * c) * x)) + f32(f32(8.0) * b + f32(24.0) * c))
            else:
                v = f32(f32(f32(f32(f32(f32(12.0) - f32(9.0) * b - f32(6.0) * c) * x) * x) * x +
                            f32(f32(f32(f32(-18.0) + f32(12.0) * b + f32(6.0) * c) * x) * x)) + f32(f32(6.0) - f32(2.0) * b))
            return f32(v * f32(f32(1.0) / f32(6.0)))
        irx, iry = f32(1.0) / f32(xwidth), f32(1.0) / f32(ywidth)
        self._filter_table((xwidth, ywidth), lambda x, y: f32(m1(f32(x * irx)) * m1(f32(y * iry))))

    def pixel_filter_sinc(self, xwidth=4.0, ywidth=4.0, tau=3.0):
        """filters/sinc.rs:15-57 (Lanczos-windowed sinc; sinf through libm, as the reference)."""
        pi, t = f32(np.pi), f32(tau)

        def sinc(x):
            x = abs(f32(x))
            if x < 1e-5:
                return f32(1.0)
            return f32(_sinf(f32(pi * x)) / f32(pi * x))

        def wsinc(x, radius):
            x = abs(f32(x))
            if x > f32(radius):
                return f32(0.0)
            lanczos = sinc(f32(x / t))
            return f32(sinc(x) * lanczos)
        self._filter_table((xwidth, ywidth), lambda x, y: f32(wsinc(x, xwidth) * wsinc(y, ywidth)))

    def sampler_sobol(self, pixelsamples=16):
        self.spp, self.sampler = int(pixelsamples), capi.PT_SAMPLER_SOBOL

    def sampler_halton(self, pixelsamples=16, samplepixelcenter=False):
        """samplers/halton.rs:275-299 (the reference's default sampler)."""
        self.spp, self.sampler, self.halton_center = int(pixelsamples), capi.PT_SAMPLER_HALTON, bool(samplepixelcenter)

    def integrator_path(self, maxdepth=5, rrthreshold=1.0, lightsamplestrategy="spatial"):
        self.max_depth, self.rr_threshold = int(maxdepth), float(rrthreshold)
        self.light_strategy = {"uniform": capi.PT_LIGHTS_UNIFORM, "power": capi.PT_LIGHTS_POWER}.get(lightsamplestrategy, capi.PT_LIGHTS_SPATIAL)

    def integrator_ao(self, nsamples=64, cossample=True):
        """Integrator "ao" (integrators/ao.rs:118-138)."""
        self.integrator, self.ao_samples, self.ao_cos_sample = capi.PT_INTEGRATOR_AO, int(nsamples), bool(cossample)

    def integrator_directlighting(self, maxdepth=5, strategy="all"):
        """Integrator "directlighting" (integrators/directlighting.rs:137-158)."""
        self.integrator, self.max_depth = capi.PT_INTEGRATOR_DIRECTLIGHTING, int(maxdepth)
        self.direct_strategy = capi.PT_DIRECT_ONE if strategy == "one" else capi.PT_DIRECT_ALL

    def integrator_whitted(self, maxdepth=5):
        """Integrator "whitted" (integrators/whitted.rs:112-135)."""
        self.integrator, self.max_depth = capi.PT_INTEGRATOR_WHITTED, int(maxdepth)

    def accelerator_bvh(self, splitmethod="sah", maxnodeprims=4):
        self.split_method = {"sah": 0, "hlbvh": 1, "middle": 2, "equal": 3}.get(splitmethod, 0)
        self.max_node_prims = int(maxnodeprims)

    # ---- graphics state
    def _add_material(self, typ, kd=(0.0, 0.0, 0.0), sigma=0.0, **kw):
        """Colour parameters (and Matte's sigma) may be a Tex handle from the texture_* methods: the parameter is then
        Texture::evaluate(si) at each hit (the reference's get_spectrum_texture / get_float_texture binding)."""
        m = capi.pt_material()
        m.type = typ
        m.opacity[:] = [1.0, 1.0, 1.0]
        m.eta, m.remap_roughness = 1.5, 1
        m.uroughness = m.vroughness = capi.PT_ROUGHNESS_UNSET
        kw = dict(kw, kd=kd, sigma=sigma)
        bump = kw.pop("bump", None)             # "bumpmap": a float texture, or a number (a constant texture, texture_params.rs:107-116)
        if bump is not None:
            m.tex_bump = (bump if isinstance(bump, Tex) else self.texture_constant(float(bump))).index + 1
        for k, v in kw.items():
            if isinstance(v, Tex):
                setattr(m, "tex_" + k, v.index + 1)
            elif isinstance(v, (tuple, list)):
                getattr(m, k)[:] = [float(x) for x in v]
            else:
                setattr(m, k, v)
        self.materials.append(m)
        return len(self.materials) - 1

    # ---- textures (src/textures/): each returns a Tex handle usable as a material parameter or as a child texture.
    # `to_world` = (m, m_inv) of the CTM at the Texture directive (spherical / cylindrical / 3-D mappings use its inverse).
    def _texture(self, typ, children=(), values=(), mapping="uv", uscale=1.0, vscale=1.0, udelta=0.0, vdelta=0.0, v1=(1, 0, 0), v2=(0, 1, 0),
                 to_world=None, aamode="closedform", octaves=0, omega=0.0, scale=0.0, variation=0.0):
        t = capi.pt_texture()
        t.type = typ
        for i in range(3):
            t.tex[i] = -1
        for i, c in enumerate(children):
            if isinstance(c, Tex):
                assert c.index < len(self.textures)
                t.tex[i] = c.index
            else:
                t.value[i][:] = [float(x) for x in (c if isinstance(c, (tuple, list)) else (c, c, c))]
        for i, v in enumerate(values):
            t.value[i][:] = [float(x) for x in (v if isinstance(v, (tuple, list)) else (v, v, v))]
        t.octaves, t.omega, t.scale, t.variation = int(octaves), float(omega), float(scale), float(variation)
        if typ in (capi.PT_TEX_CHECKERBOARD_2D, capi.PT_TEX_UV, capi.PT_TEX_BILERP, capi.PT_TEX_DOTS, capi.PT_TEX_IMAGEMAP):       # create_texture_mapping2d
            t.mapping = {"uv": capi.PT_MAPPING_UV, "spherical": capi.PT_MAPPING_SPHERICAL, "cylindrical": capi.PT_MAPPING_CYLINDRICAL,
                         "planar": capi.PT_MAPPING_PLANAR}[mapping]
            t.aa_none = 1 if (aamode == "none" and typ == capi.PT_TEX_CHECKERBOARD_2D) else 0
            t.su, t.sv = (uscale, vscale) if mapping == "uv" else (1.0, 1.0)
            t.du, t.dv = (udelta, vdelta) if mapping in ("uv", "planar") else (0.0, 0.0)
            t.v1[:] = [float(x) for x in (v1 if mapping == "planar" else (1, 0, 0))]
            t.v2[:] = [float(x) for x in (v2 if mapping == "planar" else (0, 1, 0))]
            w2t = np.eye(4, dtype=np.float32).reshape(-1) if to_world is None else np.asarray(to_world[1], np.float32).reshape(-1)
            t.world_to_texture[:] = [float(x) for x in w2t]
        elif typ in (capi.PT_TEX_CHECKERBOARD_3D, capi.PT_TEX_FBM, capi.PT_TEX_WRINKLED, capi.PT_TEX_WINDY, capi.PT_TEX_MARBLE):
            w2t = np.eye(4, dtype=np.float32).reshape(-1) if to_world is None else np.asarray(to_world[0], np.float32).reshape(-1)
            t.world_to_texture[:] = [float(x) for x in w2t]
        self.textures.append(t)
        return Tex(len(self.textures) - 1)

    def texture_constant(self, value):
        return self._texture(capi.PT_TEX_CONSTANT, values=[value])

    def texture_scale(self, tex1=1.0, tex2=1.0):
        return self._texture(capi.PT_TEX_SCALE, children=[tex1, tex2])

    def texture_mix(self, tex1=1.0, tex2=1.0, amount=0.5):
        return self._texture(capi.PT_TEX_MIX, children=[tex1, tex2, amount])

    def texture_checkerboard(self, tex1=1.0, tex2=0.0, dimension=2, **kw):
        if dimension == 3:      # IdentityMapping3D is handed tex2world itself, not its inverse (checkerboard.rs:159, mapping3d.rs:19-24)
            return self._texture(capi.PT_TEX_CHECKERBOARD_3D, children=[tex1, tex2], **kw)
        return self._texture(capi.PT_TEX_CHECKERBOARD_2D, children=[tex1, tex2], **kw)

    def image_pyramid(self, base):
        """MIPMap::new's pyramid (core/texture/mipmap.rs:406-441) over a power-of-two image `base` (H, W) or (H, W, 3), already in
        texture orientation (row 0 = t 0): each level halves the dimensions still > 1 with a*0.5 + b*0.5.  Returns the image index."""
        a = np.ascontiguousarray(base, np.float32)
        if a.ndim == 2:
            a = a[:, :, None]
        h, w, c = a.shape
        assert c in (1, 3) and (w & (w - 1)) == 0 and (h & (h - 1)) == 0
        levels = [a]
        while levels[-1].shape[0] * levels[-1].shape[1] != 1:
            cur = levels[-1]
            if cur.shape[1] > 1:
                cur = (cur[:, 0::2] * f32(0.5) + cur[:, 1::2] * f32(0.5)).astype(np.float32)
            if cur.shape[0] > 1:
                cur = (cur[0::2] * f32(0.5) + cur[1::2] * f32(0.5)).astype(np.float32)
            levels.append(cur)
        buf = np.ascontiguousarray(np.concatenate([l.reshape(-1) for l in levels]), np.float32)
        im = capi.pt_image()
        im.width, im.height, im.channels, im.n_levels = w, h, c, len(levels)
        im.texels = buf.ctypes.data_as(C.POINTER(C.c_float))
        self.images.append((im, buf))
        return len(self.images) - 1

    def texture_imagemap(self, image, trilinear=False, maxanisotropy=8.0, wrap="repeat", swrap=None, twrap=None, **kw):
        """textures/imagemap.rs over an image_pyramid() index."""
        t = self._texture(capi.PT_TEX_IMAGEMAP, **kw)
        tx = self.textures[t.index]
        modes = {"repeat": capi.PT_WRAP_REPEAT, "black": capi.PT_WRAP_BLACK, "clamp": capi.PT_WRAP_CLAMP}
        tx.image, tx.trilinear, tx.max_anisotropy = int(image), 1 if trilinear else 0, float(maxanisotropy)
        tx.swrap, tx.twrap = modes[swrap or wrap], modes[twrap or wrap]
        return t

    def texture_dots(self, tex1=1.0, tex2=0.0, **kw):
        return self._texture(capi.PT_TEX_DOTS, children=[tex1, tex2], **kw)

    def texture_fbm(self, octaves=8, roughness=0.5, to_world=None):
        return self._texture(capi.PT_TEX_FBM, octaves=octaves, omega=roughness, to_world=to_world)

    def texture_wrinkled(self, octaves=8, roughness=0.5, to_world=None):
        return self._texture(capi.PT_TEX_WRINKLED, octaves=octaves, omega=roughness, to_world=to_world)

    def texture_windy(self, to_world=None):
        return self._texture(capi.PT_TEX_WINDY, to_world=to_world)

    def texture_marble(self, octaves=8, roughness=0.5, scale=1.0, variation=0.2, to_world=None):
        return self._texture(capi.PT_TEX_MARBLE, octaves=octaves, omega=roughness, scale=scale, variation=variation, to_world=to_world)

    def texture_uv(self, **kw):
        return self._texture(capi.PT_TEX_UV, **kw)

    def texture_bilerp(self, v00=0.0, v01=1.0, v10=0.0, v11=1.0, **kw):
        return self._texture(capi.PT_TEX_BILERP, values=[v00, v01, v10, v11], **kw)

    def material_matte(self, Kd=(0.5, 0.5, 0.5), sigma=0.0, bumpmap=None):
        self.cur_material = self._add_material(capi.PT_MATERIAL_MATTE, Kd, sigma, bump=bumpmap)

    # defaults below are the reference's create_*_material defaults
    @staticmethod
    def _ft(v, unset=False):
        """A float parameter: a number, a Tex handle (float texture evaluated per hit), or None = not given."""
        if isinstance(v, Tex):
            return v
        if v is None and unset:
            return capi.PT_ROUGHNESS_UNSET
        return float(v)

    def material_plastic(self, Kd=(0.25,) * 3, Ks=(0.25,) * 3, roughness=0.1, remaproughness=True, bumpmap=None):
        """materials/plastic.rs:73-86."""
        self.cur_material = self._add_material(capi.PT_MATERIAL_PLASTIC, Kd, ks=Ks, roughness=self._ft(roughness), remap_roughness=int(remaproughness), bump=bumpmap)

    def material_mirror(self, Kr=(0.9,) * 3, bumpmap=None):
        """materials/mirror.rs:43-47."""
        self.cur_material = self._add_material(capi.PT_MATERIAL_MIRROR, kr=Kr, bump=bumpmap)

    def material_glass(self, Kr=(1.0,) * 3, Kt=(1.0,) * 3, eta=1.5, uroughness=0.0, vroughness=0.0, remaproughness=True):
        """materials/glass.rs:124-143."""
        self.cur_material = self._add_material(capi.PT_MATERIAL_GLASS, kr=Kr, kt=Kt, eta=self._ft(eta), uroughness=self._ft(uroughness),
                                               vroughness=self._ft(vroughness), remap_roughness=int(remaproughness))

    def material_metal(self, eta, k, roughness=0.01, uroughness=None, vroughness=None, remaproughness=True):
        """materials/metal.rs:127-149; eta and k as RGB (the reference's default is the copper SPD converted to RGB)."""
        self.cur_material = self._add_material(
            capi.PT_MATERIAL_METAL, metal_eta=eta, metal_k=k, roughness=self._ft(roughness), remap_roughness=int(remaproughness),
            uroughness=self._ft(uroughness, True), vroughness=self._ft(vroughness, True))

    def material_uber(self, Kd=(0.25,) * 3, Ks=(0.25,) * 3, Kr=(0.0,) * 3, Kt=(0.0,) * 3, opacity=(1.0,) * 3, eta=1.5, roughness=0.1,
                      uroughness=None, vroughness=None, remaproughness=True):
        """materials/uber.rs:142-168."""
        self.cur_material = self._add_material(
            capi.PT_MATERIAL_UBER, Kd, ks=Ks, kr=Kr, kt=Kt, opacity=opacity, eta=self._ft(eta), roughness=self._ft(roughness),
            remap_roughness=int(remaproughness), uroughness=self._ft(uroughness, True), vroughness=self._ft(vroughness, True))

    def material_substrate(self, Kd=(0.5,) * 3, Ks=(0.5,) * 3, uroughness=0.1, vroughness=0.1, remaproughness=True):
        """materials/substrate.rs:70-86."""
        self.cur_material = self._add_material(capi.PT_MATERIAL_SUBSTRATE, Kd, ks=Ks, uroughness=self._ft(uroughness),
                                               vroughness=self._ft(vroughness), remap_roughness=int(remaproughness))

    def material_none(self):
        self.cur_material = -1

    def area_light_source_diffuse(self, L=(1.0, 1.0, 1.0), scale=(1.0, 1.0, 1.0), twosided=False, nsamples=1):
        al = capi.pt_area_light()
        al.L[:] = [float(f32(l) * f32(s)) for l, s in zip(L, scale)]
        al.two_sided = 1 if twosided else 0
        al.n_samples = int(nsamples)
        self.area_lights.append(al)
        self.cur_area_light = len(self.area_lights) - 1

    def no_area_light(self):
        self.cur_area_light = -1

    # ---- shapes
    def shape_trianglemesh(self, P, indices, N=None, S=None, uv=None, twosided=True, object_to_world=None):
        P = np.asarray(P, np.float32).reshape(-1, 3)
        idx = np.asarray(indices, np.int64).reshape(-1, 3)
        nv = len(P)
        swaps = False
        if object_to_world is not None:
            m = np.asarray(object_to_world, np.float32).reshape(-1)
            P = _transform_points(m, P)
            swaps = _swaps_handedness(m)
            if N is not None or S is not None:
                raise NotImplementedError("transformed N/S: pass world-space attributes")
        if uv is not None:
            UV = np.asarray(uv, np.float32).reshape(-1, 2)
        elif _is_fillable_uv(idx, nv):            # triangle.rs:796-821
            UV = np.zeros((nv, 2), np.float32)
            tri_uv = np.array([[0, 0], [1, 0], [1, 1]], np.float32)
            for i in range(len(idx)):             # sequential, as the reference
                for j in range(3):
                    v = idx[i, j]
                    if UV[v, 0] == 0.0 and UV[v, 1] == 0.0:
                        UV[v] = tri_uv[j]
        else:
            UV = None
        return self._append_mesh(P, idx, N, S, UV, twosided, swaps)

    def _append_mesh(self, P, idx, N, S, UV, twosided, swaps):
        nv = len(P)
        keep = _tri_areas(P, idx) > f32(1e-16)      # triangle.rs:726
        idx = idx[keep]
        flags = 0
        if twosided:
            flags |= capi.PT_MESH_TWO_SIDED
        if self.reverse_orientation:
            flags |= capi.PT_MESH_REVERSE_ORIENTATION
        if swaps:
            flags |= capi.PT_MESH_SWAPS_HANDEDNESS
        if N is not None:
            flags |= capi.PT_MESH_HAS_N
        if S is not None:
            flags |= capi.PT_MESH_HAS_S
        if UV is not None:
            flags |= capi.PT_MESH_HAS_UV
        mesh = capi.pt_mesh(flags, self.cur_material, self.cur_area_light, self.cur_object)
        self.meshes.append(mesh)
        mid = len(self.meshes) - 1
        self.P.append(P)
        self.N.append(np.asarray(N, np.float32).reshape(-1, 3) if N is not None else None)
        self.S.append(np.asarray(S, np.float32).reshape(-1, 3) if S is not None else None)
        self.UV.append(UV)
        self.idx.append((idx + self.n_vertices).astype(np.uint32))
        self.tri_mesh.append(np.full(len(idx), mid, np.uint32))
        self.n_vertices += nv
        return mid

    def shape_sphere(self, radius=1.0, zmin=None, zmax=None, phimax=360.0, object_to_world=None, world_to_object=None):
        """Shape "sphere" (shapes/sphere.rs:401-420) under the given CTM (m and m_inv, row-major; identity by
        default).  A translate/scale/rotate product and its inverse are built by `transform_*` below."""
        sp = capi.pt_sphere()
        m = np.eye(4, dtype=np.float32).reshape(-1) if object_to_world is None else np.asarray(object_to_world, np.float32).reshape(-1)
        if world_to_object is None:
            assert object_to_world is None, "pass the inverse too: the reference never re-inverts a CTM"
            mi = m.copy()
        else:
            mi = np.asarray(world_to_object, np.float32).reshape(-1)
        sp.object_to_world[:] = [float(v) for v in m]
        sp.world_to_object[:] = [float(v) for v in mi]
        sp.radius = radius
        sp.zmin = -radius if zmin is None else zmin
        sp.zmax = radius if zmax is None else zmax
        sp.phimax = phimax
        sp.flags = capi.PT_SPHERE_REVERSE_ORIENTATION if self.reverse_orientation else 0
        sp.material, sp.area_light = self.cur_material, self.cur_area_light
        sp.before_triangle = sum(len(i) for i in self.idx)
        sp.object, sp.order = self.cur_object, self.n_extra
        self.n_extra += 1
        self.spheres.append(sp)
        return len(self.spheres) - 1

    # ---- object instancing (scene_context.rs:1327-1391)
    def object_begin(self, name):
        """ObjectBegin opens an attribute scope (scene_context.rs:1330): material / area light / orientation set inside do not leak."""
        self._saved_gstate = (self.cur_material, self.cur_area_light, self.reverse_orientation)
        self.objects[name] = len(self.objects)
        self.cur_object = self.objects[name] + 1

    def object_end(self):
        self.cur_object = 0
        self.cur_material, self.cur_area_light, self.reverse_orientation = self._saved_gstate

    def object_instance(self, name, to_world=None):
        """ObjectInstance under the CTM `to_world` = (m, m_inv)."""
        if self.cur_object:
            return                          # ignored inside an object definition (:1352-1357)
        it = capi.pt_instance()
        m = np.eye(4, dtype=np.float32).reshape(-1) if to_world is None else np.asarray(to_world[0], np.float32).reshape(-1)
        mi = np.eye(4, dtype=np.float32).reshape(-1) if to_world is None else np.asarray(to_world[1], np.float32).reshape(-1)
        it.instance_to_world[:] = [float(v) for v in m]
        it.world_to_instance[:] = [float(v) for v in mi]
        it.object = self.objects[name]
        it.before_triangle = sum(len(i) for i in self.idx)
        it.order = self.n_extra
        self.n_extra += 1
        self.instances.append(it)

    def shape_trianglemesh_fast(self, P, indices, twosided=True):
        """Bulk path for meshes whose triangles do not share vertices (indices == arange):
        identical result to shape_trianglemesh (uv fill = (0,0),(1,0),(1,1) per triangle)."""
        P = np.asarray(P, np.float32).reshape(-1, 3)
        idx = np.asarray(indices, np.int64).reshape(-1, 3)
        assert np.array_equal(idx.reshape(-1), np.arange(len(P)))
        UV = np.tile(np.array([[0, 0], [1, 0], [1, 1]], np.float32), (len(idx), 1))
        return self._append_mesh(P, idx, None, None, UV, twosided, False)

    # ---- finish
    def build(self):
        sd = SceneDesc()
        d = sd.desc
        P = np.ascontiguousarray(np.concatenate(self.P), np.float32) if self.P else np.zeros((0, 3), np.float32)

        def cat(parts, width):
            if all(p is None for p in parts):
                return None
            return np.ascontiguousarray(np.concatenate([p if p is not None else np.zeros((len(pp), width), np.float32)
                                                        for p, pp in zip(parts, self.P)]), np.float32)
        N, S, UV = cat(self.N, 3), cat(self.S, 3), cat(self.UV, 2)
        idx = np.ascontiguousarray(np.concatenate(self.idx), np.uint32) if self.idx else np.zeros((0, 3), np.uint32)
        tm = np.ascontiguousarray(np.concatenate(self.tri_mesh), np.uint32) if self.tri_mesh else np.zeros(0, np.uint32)
        d.n_vertices = len(P)
        sd._set("P", P, C.c_float)
        sd._set("N", N, C.c_float) if N is not None else None
        sd._set("S", S, C.c_float) if S is not None else None
        sd._set("UV", UV, C.c_float) if UV is not None else None
        d.n_triangles = len(idx)
        sd._set("indices", idx, C.c_uint32)
        sd._set("tri_mesh", tm, C.c_uint32)
        meshes = (capi.pt_mesh * max(1, len(self.meshes)))(*self.meshes)
        mats = (capi.pt_material * max(1, len(self.materials)))(*self.materials)
        als = (capi.pt_area_light * max(1, len(self.area_lights)))(*self.area_lights)
        sd.buffers["meshes"], sd.buffers["materials"], sd.buffers["area_lights"] = meshes, mats, als
        d.n_meshes, d.meshes = len(self.meshes), meshes
        d.n_materials, d.materials = len(self.materials), mats
        d.n_area_lights, d.area_lights = len(self.area_lights), als
        if self.textures:
            tex = (capi.pt_texture * len(self.textures))(*self.textures)
            sd.buffers["textures"] = tex
            d.n_textures, d.textures = len(self.textures), tex
        if self.instances:
            ins = (capi.pt_instance * len(self.instances))(*self.instances)
            sd.buffers["instances"] = ins
            d.n_instances, d.instances = len(self.instances), ins
        if self.images:
            ims = (capi.pt_image * len(self.images))(*[im for im, _ in self.images])
            sd.buffers["images"] = ims
            sd.buffers["image_texels"] = [buf for _, buf in self.images]
            d.n_images, d.images = len(self.images), ims
        if self.spheres:
            sph = (capi.pt_sphere * len(self.spheres))(*self.spheres)
            sd.buffers["spheres"] = sph
            d.n_spheres, d.spheres = len(self.spheres), sph
        d.split_method, d.max_node_prims = self.split_method, self.max_node_prims
        d.camera_to_world[:] = [float(v) for v in self.camera_to_world]
        d.fov = self.fov
        # create_perspective_camera (cameras/perspective.rs:337-394)
        aspect = f32(self.xres) / f32(self.yres)
        frame = f32(getattr(self, "frameaspectratio", None) or aspect)
        if self.screen_window is not None:
            sw = [f32(v) for v in self.screen_window]
        elif frame > 1.0:
            sw = [-frame, frame, f32(-1.0), f32(1.0)]
        else:
            sw = [f32(-1.0), f32(1.0), f32(-1.0) / frame, f32(1.0) / frame]
        d.screen_window[:] = [float(v) for v in sw]
        d.lens_radius, d.focal_distance = self.lens_radius, self.focal_distance
        d.shutter_open, d.shutter_close = self.shutter
        d.xres, d.yres = self.xres, self.yres
        d.crop_window[:] = self.crop
        d.filter_radius[:] = self.filter_radius
        d.filter_table[:] = [float(v) for v in self.filter_table]
        d.film_scale, d.max_sample_luminance = self.film_scale, self.max_sample_luminance
        d.sampler, d.spp = getattr(self, "sampler", capi.PT_SAMPLER_SOBOL), self.spp
        d.halton_sample_at_center = 1 if getattr(self, "halton_center", False) else 0
        d.max_depth, d.rr_threshold, d.light_strategy = self.max_depth, self.rr_threshold, self.light_strategy
        d.integrator, d.ao_samples, d.ao_cos_sample = self.integrator, self.ao_samples, int(self.ao_cos_sample)
        d.direct_strategy = self.direct_strategy
        return sd


# --------------------------------------------------------------------------
def _quad(b, p0, p1, p2, p3, **kw):
    b.shape_trianglemesh([p0, p1, p2, p3], [0, 1, 2, 0, 2, 3], **kw)


def _cuboid(b, top4, height):
    """Six quads (12 triangles) from the four top corners and a height."""
    t = [np.array(p, np.float32) for p in top4]
    bt = [np.array([p[0], p[1] - height, p[2]], np.float32) for p in t]
    _quad(b, t[0], t[1], t[2], t[3])
    _quad(b, bt[3], bt[2], bt[1], bt[0])
    for i in range(4):
        j = (i + 1) % 4
        _quad(b, t[i], bt[i], bt[j], t[j])


def cornell_box(res=512, spp=64, max_depth=5, light_strategy="spatial", sampler="sobol"):
    """BASELINE config 1: 36 triangles, matte walls, one-sided ceiling quad light L=(17,12,4)."""
    b = SceneBuilder()
    b.look_at((278, 273, -800), (278, 273, 0), (0, 1, 0))
    b.camera_perspective(fov=39.3)
    b.film(xresolution=res, yresolution=res)
    b.pixel_filter_box()
    b.sampler_sobol(spp) if sampler == "sobol" else b.sampler_halton(spp)
    b.integrator_path(maxdepth=max_depth, lightsamplestrategy=light_strategy)
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    b.material_matte(white)
    _quad(b, (552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2))                    # floor
    _quad(b, (556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0))        # ceiling
    _quad(b, (549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2))      # back
    b.material_matte(green)
    _quad(b, (0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2))                    # right
    b.material_matte(red)
    _quad(b, (552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0))        # left
    b.material_matte(white)
    _cuboid(b, [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)], 165.0)   # short block
    _cuboid(b, [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)], 330.0)  # tall block
    b.area_light_source_diffuse(L=(17, 12, 4))
    _quad(b, (343, 548.75, 227), (343, 548.75, 332), (213, 548.75, 332), (213, 548.75, 227))  # light, faces -y
    return b.build()


def rt1m(n_triangles=1000000, res=1024, spp=256, max_depth=8, s=0.005, seed_sequence=1, sampler="sobol", materials="matte", light="quad", instances=0):
    """BASELINE config 2 ("RT1M"): 12-triangle enclosure + light, the rest random matte triangles.

    Filler triangle k draws, in order, cx cy cz then v0x..v2z as lerp(uniform_float(), lo, hi)
    from PCG32 RNG::new_sequence(1) (src/core/rng.rs:21-33, :59-66)."""
    b = SceneBuilder()
    b.look_at((0, 0, -3.4), (0, 0, 0), (0, 1, 0))
    b.camera_perspective(fov=40.0)
    b.film(xresolution=res, yresolution=res)
    b.pixel_filter_box()
    b.sampler_sobol(spp) if sampler == "sobol" else b.sampler_halton(spp)
    b.integrator_path(maxdepth=max_depth, rrthreshold=1.0, lightsamplestrategy="spatial")
    b.accelerator_bvh("sah", 4)
    b.material_matte((0.5, 0.5, 0.5))
    _quad(b, (1, -1, -1), (-1, -1, -1), (-1, -1, 1), (1, -1, 1))      # floor
    _quad(b, (1, 1, -1), (1, 1, 1), (-1, 1, 1), (-1, 1, -1))          # ceiling
    _quad(b, (1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, 1))          # back
    _quad(b, (-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1))      # right
    _quad(b, (1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1))          # left
    n_fill = max(0, int(n_triangles) - 12)
    if instances > 0 and n_fill:
        # secondary workload (bench.py --instances K): the filler triangles become ONE object, instanced K times on a g x g x g grid of cells
        # (g = ceil(cbrt(K)), each copy scaled by 1 / g) -- K x n_fill triangles on screen for n_fill in memory; k_trace_inst / k_shade_general_inst*
        b.object_begin("fill")
    if n_fill:
        u = pcg32_uniform_float(12 * n_fill, seed_sequence).reshape(n_fill, 12)
        one = f32(1.0)
        c = (one - u[:, 0:3]) * f32(-0.9) + u[:, 0:3] * f32(0.9)                       # lerp(t, lo, hi)
        off = (one - u[:, 3:12]) * f32(-s) + u[:, 3:12] * f32(s)
        verts = (np.repeat(c, 3, axis=0).reshape(n_fill, 9) + off).astype(np.float32).reshape(-1, 3)
        if materials == "matte":
            b.shape_trianglemesh_fast(verts, np.arange(3 * n_fill), twosided=True)
        else:
            # "killeroo-class" stand-in for BASELINE config 4: the same geometry cut into six shapes with
            # matte / plastic / metal / glass / mirror / substrate materials (40/20/15/15/5/5 %)
            cuts = (np.cumsum([0.0, 0.40, 0.20, 0.15, 0.15, 0.05, 0.05]) * n_fill).astype(np.int64)
            setters = [lambda: b.material_matte((0.5, 0.5, 0.5)),
                       lambda: b.material_plastic(Kd=(0.4, 0.2, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.1),
                       lambda: b.material_metal(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.05),
                       lambda: b.material_glass(eta=1.5),
                       lambda: b.material_mirror(),
                       lambda: b.material_substrate(Kd=(0.2, 0.3, 0.5), Ks=(0.4, 0.4, 0.4))]
            if materials == "textured":
                # "crown-class" stand-in for BASELINE config 5: the mixed split with texture-driven parameters -- an image map
                # (EWA), a closed-form checkerboard, fbm through a mix, and bump maps on two of the shapes
                rng = np.random.default_rng(17)
                img = b.image_pyramid(rng.random((256, 256, 3), dtype=np.float32))
                gray = b.image_pyramid(rng.random((128, 128), dtype=np.float32))
                t_img = b.texture_imagemap(img, uscale=4.0, vscale=4.0)
                t_chk = b.texture_checkerboard((0.8, 0.7, 0.2), (0.2, 0.2, 0.25), uscale=8.0, vscale=8.0)
                t_fbm = b.texture_mix((0.2, 0.3, 0.6), (0.9, 0.9, 0.9), amount=b.texture_fbm(octaves=6, to_world=transform_scale(8.0, 8.0, 8.0)))
                t_bump = b.texture_scale(0.01, b.texture_imagemap(gray, trilinear=True, uscale=6.0, vscale=6.0))
                t_wr = b.texture_scale(0.01, b.texture_wrinkled(octaves=4, to_world=transform_scale(10.0, 10.0, 10.0)))
                setters = [lambda: b.material_matte(t_img, bumpmap=t_bump),
                           lambda: b.material_plastic(Kd=t_chk, Ks=(0.3, 0.3, 0.3), roughness=0.1),
                           lambda: b.material_metal(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.05),
                           lambda: b.material_glass(eta=1.5),
                           lambda: b.material_mirror(bumpmap=t_wr),
                           lambda: b.material_substrate(Kd=t_fbm, Ks=(0.4, 0.4, 0.4))]
            for g, setm in enumerate(setters):
                lo, hi = int(cuts[g]), int(cuts[g + 1])
                if hi > lo:
                    setm()
                    b.shape_trianglemesh_fast(verts[3 * lo:3 * hi], np.arange(3 * (hi - lo)), twosided=True)
            b.material_matte((0.5, 0.5, 0.5))
    if instances > 0 and n_fill:
        b.object_end()
        g = int(np.ceil(instances ** (1.0 / 3.0) - 1e-9))
        k = 0
        for iz in range(g):
            for iy in range(g):
                for ix in range(g):
                    if k < instances:
                        c = [(-1.0 + (2 * i + 1) / g) * 0.9 for i in (ix, iy, iz)]
                        b.object_instance("fill", transform_mul(transform_translate(c[0], c[1], c[2]), transform_scale(1.0 / g, 1.0 / g, 1.0 / g)))
                        k += 1
        b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=(17, 12, 4))
    if light == "sphere":       # secondary workload: an analytic sphere light (the sphere-capable kernel instantiations run)
        t = transform_translate(0.0, 0.85, 0.0)
        b.shape_sphere(radius=0.1, object_to_world=t[0], world_to_object=t[1])
    else:
        _quad(b, (0.25, 0.999, -0.25), (0.25, 0.999, 0.25), (-0.25, 0.999, 0.25), (-0.25, 0.999, -0.25))  # faces -y
    return b.build()


def all_tiles(info, tile=16):
    """The reference's 16x16 decomposition of the sample bounds (sampler.rs:266-289)."""
    x0, y0, x1, y1 = list(info.sample_bounds)
    out = []
    for y in range(y0, y1, tile):
        for x in range(x0, x1, tile):
            out.append((x, y, min(x + tile, x1), min(y + tile, y1)))
    return out
