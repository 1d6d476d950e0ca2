"""ctypes loader for oracle/liboracle.so -- the CPU restatement used as the checker.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
LIB_NATIVE = os.path.join(ORACLE_DIR, "liboracle_native.so")      # bench.py's CPU timing leg only: built on the host that times it (oracle/Makefile `native`)
PORTABLE_FLAGS = "g++ -O2 -ffp-contract=off -fno-fast-math"
NATIVE_FLAGS = "g++ -O3 -march=native -ffp-contract=off -fno-fast-math"
KAT = os.path.join(ORACLE_DIR, "orc_kat")

pkg = importlib.import_module("pbrt-r3_amd")
capi = pkg.capi
DATA_DIR = capi.DATA_DIR


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"])


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, u32 = C.c_void_p, C.c_uint32
        lib.orc_scene_create.argtypes = [C.POINTER(capi.pt_scene_desc), C.c_char_p, C.POINTER(vp)]
        lib.orc_scene_destroy.argtypes = [vp]
        lib.orc_scene_destroy.restype = None
        lib.orc_scene_info.argtypes = [vp, C.POINTER(capi.pt_scene_info)]
        lib.orc_scene_info.restype = None
        lib.orc_bvh_ordered_prims.argtypes = [vp, vp, u32]
        lib.orc_bvh_ordered_prims.restype = u32
        lib.orc_render.argtypes = [vp, C.POINTER(capi.pt_tile), u32, C.c_int, vp, C.POINTER(capi.pt_counters)]
        lib.orc_render.restype = C.c_double
        lib.orc_resolve_rgb.argtypes = [vp, C.c_uint64, C.c_float, vp]
        lib.orc_resolve_rgb.restype = None
        lib.orc_radiance_samples.argtypes = [vp, C.POINTER(capi.pt_tile), vp]
        lib.orc_radiance_samples.restype = None
        lib.orc_reference_panics.argtypes = [C.c_int]
        lib.orc_reference_panics.restype = u32
        for name in ("orc_trace_closest", "orc_trace_any"):
            getattr(lib, name).argtypes = [vp, u32, vp, vp, vp, vp, C.POINTER(capi.pt_counters)]
            getattr(lib, name).restype = None
        lib.orc_trace_exhaustive.argtypes = [vp, u32, vp, vp, vp, vp]
        lib.orc_trace_exhaustive.restype = None
        lib.orc_generate_camera_rays.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        lib.orc_generate_camera_rays.restype = None
        lib.orc_sobol_samples.argtypes = [vp, u32, vp, vp, vp, vp]
        lib.orc_sobol_samples.restype = None
        lib.orc_bsdf_eval.argtypes = [vp, u32, u32, vp, vp, u32, vp, vp]
        lib.orc_bsdf_eval.restype = None
        lib.orc_bsdf_sample.argtypes = [vp, u32, u32, vp, vp, u32, vp, vp, vp, vp]
        lib.orc_bsdf_sample.restype = None
        lib.orc_roughness_to_alpha.argtypes = [C.c_float]
        lib.orc_roughness_to_alpha.restype = C.c_float
        lib.orc_light_distribution.argtypes = [vp, vp, vp, vp]
        lib.orc_light_distribution.restype = u32
        lib.orc_light_voxels.argtypes = [vp, vp]
        lib.orc_light_voxels.restype = None
        lib.orc_order_entry.argtypes = [u32, u32]
        lib.orc_order_entry.restype = u32
        lib.orc_radical_inverse.argtypes = [u32, C.c_uint64]
        lib.orc_radical_inverse.restype = C.c_float
        lib.orc_rng_floats.argtypes = [C.c_uint64, C.c_int, u32, vp, vp]
        lib.orc_rng_floats.restype = None
        lib.orc_next_float_up.argtypes = [C.c_float]
        lib.orc_next_float_up.restype = C.c_float
        lib.orc_next_float_down.argtypes = [C.c_float]
        lib.orc_next_float_down.restype = C.c_float
        lib.orc_dist1d_sample_discrete.argtypes = [vp, u32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.orc_dist1d_sample_discrete.restype = u32
        lib.orc_dist1d_sample_continuous.argtypes = [vp, u32, C.c_float, C.POINTER(C.c_float), C.POINTER(u32)]
        lib.orc_dist1d_sample_continuous.restype = C.c_float
        lib.orc_dist1d_discrete_pdf.argtypes = [vp, u32, u32]
        lib.orc_dist1d_discrete_pdf.restype = C.c_float
        lib.orc_cosine_sample_hemisphere.argtypes = [C.c_float, C.c_float, vp]
        lib.orc_cosine_sample_hemisphere.restype = None
        lib.orc_sobol_load.argtypes = [C.c_char_p]
        lib.orc_sobol_sample_float.argtypes = [C.c_int64, u32]
        lib.orc_sobol_sample_float.restype = C.c_float
        lib.orc_sobol_interval_to_index.argtypes = [u32, C.c_uint64, C.c_int32, C.c_int32]
        lib.orc_sobol_interval_to_index.restype = C.c_uint64
        assert lib.orc_sobol_load(DATA_DIR.encode()) == 0

    def reference_panics(self, reset=True):
        """Bits noted since the last reset where the reference would have panicked and the restatement went on (1: a Halton dimension
        past the prime tables, halton.rs:103-108)."""
        return int(self.lib.orc_reference_panics(1 if reset else 0))

    def scene(self, scene_desc):
        return OracleScene(self, scene_desc)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, orc, sd):
        self.orc, self.lib, self.sd = orc, orc.lib, sd
        self.h = C.c_void_p()
        rc = self.lib.orc_scene_create(C.byref(sd.desc), DATA_DIR.encode(), C.byref(self.h))
        assert rc == 0, "orc_scene_create failed"
        self.info = capi.pt_scene_info()
        self.lib.orc_scene_info(self.h, C.byref(self.info))

    def close(self):
        if self.h:
            self.lib.orc_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def film_shape(self):
        cb = self.info.cropped_bounds
        return (cb[3] - cb[1], cb[2] - cb[0])

    def ordered_prims(self):
        n = self.sd.desc.n_triangles + self.sd.desc.n_spheres
        out = np.empty(n, np.uint32)
        m = self.lib.orc_bvh_ordered_prims(self.h, _p(out), n)
        return out[:m]

    def render(self, tiles=None, threads=8, want_image=True):
        h, w = self.film_shape
        xyzw = np.zeros((h, w, 4), np.float32) if want_image else None
        cnt = capi.pt_counters()
        if tiles is None:
            secs = self.lib.orc_render(self.h, None, 0, threads, _p(xyzw) if want_image else None, C.byref(cnt))
        else:
            arr = capi.tiles_array(tiles)
            secs = self.lib.orc_render(self.h, arr, len(tiles), threads, _p(xyzw) if want_image else None, C.byref(cnt))
        return xyzw, cnt.as_dict(), secs

    def resolve_rgb(self, xyzw):
        xyzw = np.ascontiguousarray(xyzw, np.float32)
        rgb = np.empty(xyzw.shape[:-1] + (3,), np.float32)
        self.lib.orc_resolve_rgb(_p(xyzw), xyzw.size // 4, self.sd.desc.film_scale, _p(rgb))
        return rgb

    def radiance_samples(self, tile):
        t = capi.pt_tile(*[int(v) for v in tile])
        npx = (t.x1 - t.x0) * (t.y1 - t.y0)
        out = np.empty((npx, self.info.spp, 3), np.float32)
        self.lib.orc_radiance_samples(self.h, C.byref(t), _p(out))
        return out

    def trace_closest(self, o, d, tmax, exhaustive=False):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.empty(len(tmax), capi.HIT_DTYPE)
        cnt = capi.pt_counters()
        if exhaustive:
            self.lib.orc_trace_exhaustive(self.h, len(tmax), _p(o), _p(d), _p(tmax), _p(out))
        else:
            self.lib.orc_trace_closest(self.h, len(tmax), _p(o), _p(d), _p(tmax), _p(out), C.byref(cnt))
        return out, cnt.as_dict()

    def trace_any(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.empty(len(tmax), np.uint8)
        cnt = capi.pt_counters()
        self.lib.orc_trace_any(self.h, len(tmax), _p(o), _p(d), _p(tmax), _p(out), C.byref(cnt))
        return out, cnt.as_dict()

    def generate_camera_rays(self, pixel_xy, sample_index):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.int32); sample_index = np.ascontiguousarray(sample_index, np.uint32)
        n = len(sample_index)
        o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32); pf = np.empty((n, 2), np.float32)
        self.lib.orc_generate_camera_rays(self.h, n, _p(pixel_xy), _p(sample_index), _p(o), _p(d), _p(pf))
        return o, d, pf

    def sobol_samples(self, pixel_xy, sample_index, dim):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.int32); sample_index = np.ascontiguousarray(sample_index, np.uint32)
        dim = np.ascontiguousarray(dim, np.uint32)
        out = np.empty(len(dim), np.float32)
        self.lib.orc_sobol_samples(self.h, len(dim), _p(pixel_xy), _p(sample_index), _p(dim), _p(out))
        return out

    def bsdf_eval(self, material, wo, wi, flags=31):
        wo = np.ascontiguousarray(wo, np.float32); wi = np.ascontiguousarray(wi, np.float32)
        n = len(wo)
        f = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32)
        self.lib.orc_bsdf_eval(self.h, C.c_uint32(material), C.c_uint32(n), _p(wo), _p(wi), C.c_uint32(flags), _p(f), _p(pdf))
        return f, pdf

    def bsdf_sample(self, material, wo, u, flags=31):
        wo = np.ascontiguousarray(wo, np.float32); u = np.ascontiguousarray(u, np.float32)
        n = len(wo)
        f = np.empty((n, 3), np.float32); wi = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); t = np.empty(n, np.uint32)
        self.lib.orc_bsdf_sample(self.h, C.c_uint32(material), C.c_uint32(n), _p(wo), _p(u), C.c_uint32(flags), _p(f), _p(wi), _p(pdf), _p(t))
        return f, wi, pdf, t

    def light_distribution(self, p):
        n = self.info.n_lights
        func = np.empty(n, np.float32); cdf = np.empty(n + 1, np.float32)
        pp = np.asarray(p, np.float32)
        self.lib.orc_light_distribution(self.h, _p(pp), _p(func), _p(cdf))
        return func, cdf


_oracle = None
_native = None


def load_native():
    """The same restatement compiled for THIS host's cores (-O3 -march=native, still without fused multiply-add): always rebuilt here, a
    library built elsewhere with -march=native may not run on this CPU."""
    global _native
    if _native is None:
        subprocess.check_call(["make", "-s", "-B", "-C", ORACLE_DIR, "native"])
        _native = Oracle(C.CDLL(LIB_NATIVE))
    return _native


def load():
    global _oracle
    if _oracle is None:
        if not os.path.exists(LIB):
            build()
        _oracle = Oracle(C.CDLL(LIB))
    return _oracle
