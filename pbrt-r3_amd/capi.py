"""ctypes bindings for include/pbrtgpu.h (libpbrtgpu.so).

Every binding goes through the C ABI; there is no Python or CPU re-implementation
of any kernel here.  If the shared library is absent, load_library() raises.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBRTGPU_LIB") or os.path.join(HERE, "csrc", "libpbrtgpu.so")   # override: tuning builds only
DATA_DIR = os.path.join(HERE, "data")

PT_MATERIAL_NONE, PT_MATERIAL_MATTE, PT_MATERIAL_PLASTIC, PT_MATERIAL_MIRROR = 0, 1, 2, 3
PT_MATERIAL_GLASS, PT_MATERIAL_METAL, PT_MATERIAL_UBER, PT_MATERIAL_SUBSTRATE = 4, 5, 6, 7
PT_ROUGHNESS_UNSET = -1.0
PT_MESH_TWO_SIDED, PT_MESH_REVERSE_ORIENTATION, PT_MESH_SWAPS_HANDEDNESS = 1, 2, 4
PT_MESH_HAS_N, PT_MESH_HAS_S, PT_MESH_HAS_UV = 8, 16, 32
PT_SPLIT_SAH, PT_SPLIT_HLBVH, PT_SPLIT_MIDDLE, PT_SPLIT_EQUAL_COUNTS = 0, 1, 2, 3
PT_LIGHTS_UNIFORM, PT_LIGHTS_POWER, PT_LIGHTS_SPATIAL = 0, 1, 2
PT_SAMPLER_SOBOL, PT_SAMPLER_HALTON = 0, 1

STATUS_NAMES = {0: "PT_OK", 1: "PT_ERR_INVALID_ARGUMENT", 2: "PT_ERR_NO_DEVICE", 3: "PT_ERR_DEVICE",
                4: "PT_ERR_UNSUPPORTED", 5: "PT_ERR_NO_SCENE", 6: "PT_ERR_OUT_OF_MEMORY"}


class PtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), msg))
        self.status = status


class pt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("kd", C.c_float * 3), ("sigma", C.c_float), ("ks", C.c_float * 3), ("kr", C.c_float * 3),
                ("kt", C.c_float * 3), ("opacity", C.c_float * 3), ("eta", C.c_float), ("roughness", C.c_float),
                ("uroughness", C.c_float), ("vroughness", C.c_float), ("remap_roughness", C.c_int32),
                ("metal_eta", C.c_float * 3), ("metal_k", C.c_float * 3),
                ("tex_kd", C.c_uint32), ("tex_ks", C.c_uint32), ("tex_kr", C.c_uint32), ("tex_kt", C.c_uint32),
                ("tex_opacity", C.c_uint32), ("tex_sigma", C.c_uint32), ("tex_metal_eta", C.c_uint32), ("tex_metal_k", C.c_uint32),
                ("tex_bump", C.c_uint32), ("tex_roughness", C.c_uint32), ("tex_uroughness", C.c_uint32), ("tex_vroughness", C.c_uint32), ("tex_eta", C.c_uint32)]


class pt_texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("tex", C.c_int32 * 3), ("value", (C.c_float * 3) * 4), ("mapping", C.c_int32),
                ("aa_none", C.c_int32), ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float),
                ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("world_to_texture", C.c_float * 16),
                ("octaves", C.c_int32), ("omega", C.c_float), ("scale", C.c_float), ("variation", C.c_float),
                ("image", C.c_int32), ("trilinear", C.c_int32), ("max_anisotropy", C.c_float), ("swrap", C.c_int32), ("twrap", C.c_int32),
                ("reserved", C.c_int32 * 3)]


class pt_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("n_levels", C.c_uint32),
                ("texels", C.POINTER(C.c_float))]


PT_WRAP_REPEAT, PT_WRAP_BLACK, PT_WRAP_CLAMP = range(3)


(PT_TEX_CONSTANT, PT_TEX_SCALE, PT_TEX_MIX, PT_TEX_CHECKERBOARD_2D, PT_TEX_CHECKERBOARD_3D, PT_TEX_UV, PT_TEX_BILERP, PT_TEX_DOTS, PT_TEX_FBM,
 PT_TEX_WRINKLED, PT_TEX_WINDY, PT_TEX_MARBLE, PT_TEX_IMAGEMAP) = range(13)
PT_MAPPING_UV, PT_MAPPING_SPHERICAL, PT_MAPPING_CYLINDRICAL, PT_MAPPING_PLANAR = range(4)


class pt_area_light(C.Structure):
    _fields_ = [("L", C.c_float * 3), ("two_sided", C.c_int32), ("n_samples", C.c_int32)]


class pt_mesh(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("material", C.c_int32), ("area_light", C.c_int32), ("object", C.c_uint32)]


class pt_sphere(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16),
                ("radius", C.c_float), ("zmin", C.c_float), ("zmax", C.c_float), ("phimax", C.c_float),
                ("flags", C.c_uint32), ("material", C.c_int32), ("area_light", C.c_int32),
                ("before_triangle", C.c_uint32), ("object", C.c_uint32), ("order", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class pt_instance(C.Structure):
    _fields_ = [("instance_to_world", C.c_float * 16), ("world_to_instance", C.c_float * 16), ("object", C.c_uint32),
                ("before_triangle", C.c_uint32), ("order", C.c_uint32), ("reserved", C.c_uint32)]


PT_SPHERE_REVERSE_ORIENTATION = 1


class pt_scene_desc(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_uint32), ("P", C.POINTER(C.c_float)), ("N", C.POINTER(C.c_float)),
        ("S", C.POINTER(C.c_float)), ("UV", C.POINTER(C.c_float)),
        ("n_triangles", C.c_uint32), ("indices", C.POINTER(C.c_uint32)), ("tri_mesh", C.POINTER(C.c_uint32)),
        ("n_meshes", C.c_uint32), ("meshes", C.POINTER(pt_mesh)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(pt_material)),
        ("n_area_lights", C.c_uint32), ("area_lights", C.POINTER(pt_area_light)),
        ("split_method", C.c_int32), ("max_node_prims", C.c_int32),
        ("camera_to_world", C.c_float * 16), ("fov", C.c_float), ("screen_window", C.c_float * 4),
        ("lens_radius", C.c_float), ("focal_distance", C.c_float),
        ("shutter_open", C.c_float), ("shutter_close", C.c_float),
        ("xres", C.c_int32), ("yres", C.c_int32), ("crop_window", C.c_float * 4),
        ("filter_radius", C.c_float * 2), ("filter_table", C.c_float * 256),
        ("film_scale", C.c_float), ("max_sample_luminance", C.c_float),
        ("sampler", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
        ("rr_threshold", C.c_float), ("light_strategy", C.c_int32),
        ("halton_sample_at_center", C.c_int32),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(pt_sphere)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(pt_texture)),
        ("n_images", C.c_uint32), ("images", C.POINTER(pt_image)),
        ("n_instances", C.c_uint32), ("instances", C.POINTER(pt_instance)), ("integrator", C.c_int32), ("ao_samples", C.c_int32), ("ao_cos_sample", C.c_int32), ("direct_strategy", C.c_int32),
    ]


class pt_tile(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class pt_hit(C.Structure):
    _fields_ = [("t", C.c_float), ("prim", C.c_int32), ("b0", C.c_float), ("b1", C.c_float)]


class pt_counters(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("regular_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("path_vertices", C.c_uint64),
                ("trace_launches", C.c_uint64), ("trace_ms", C.c_double), ("shade_ms", C.c_double),
                ("render_ms", C.c_double), ("nodes_from_lds", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class pt_scene_info(C.Structure):
    _fields_ = [("sample_bounds", C.c_int32 * 4), ("cropped_bounds", C.c_int32 * 4), ("spp", C.c_int32),
                ("n_lights", C.c_uint32), ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32),
                ("world_bound", C.c_float * 6), ("bvh_build_ms", C.c_double), ("upload_ms", C.c_double),
                ("bvh_on_device", C.c_int32), ("reserved", C.c_int32)]


BVH_BUILD_AUTO, BVH_BUILD_HOST, BVH_BUILD_DEVICE = 0, 1, 2
PT_INTEGRATOR_PATH, PT_INTEGRATOR_AO, PT_INTEGRATOR_DIRECTLIGHTING, PT_INTEGRATOR_WHITTED = 0, 1, 2, 3
PT_DIRECT_ALL, PT_DIRECT_ONE = 0, 1


HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<i4"), ("b0", "<f4"), ("b1", "<f4")])

# Every symbol include/pbrtgpu.h declares (tests check the library exports them all).
SYMBOLS = [
    "pt_context_create", "pt_context_destroy", "pt_last_error", "pt_abi_version", "pt_set_data_dir",
    "pt_scene_upload", "pt_scene_info_get", "pt_film_clear", "pt_render", "pt_film_download_xyzw",
    "pt_film_device_xyzw", "pt_film_commit_xyzw", "pt_film_allreduce", "pt_film_add_xyzw", "pt_film_resolve_rgb", "pt_trace_closest", "pt_trace_any", "pt_trace_wavefront",
    "pt_generate_camera_rays", "pt_sobol_samples", "pt_radiance_samples", "pt_get_counters", "pt_reset_counters",
    "pt_bvh_leaf_order", "pt_bsdf_eval", "pt_bsdf_sample", "pt_set_bvh_build", "pt_scene_bvh_digest",
]

_lib = None


def load_library(path=None):
    """dlopen libpbrtgpu.so.  Raises (never substitutes another implementation) if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError("libpbrtgpu.so not found at %s -- build it with __graft_entry__.build(); "
                      "there is no CPU fallback" % p)
    lib = C.CDLL(p)
    vp, u32, fp = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
    lib.pt_context_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.pt_context_destroy.argtypes = [vp]
    lib.pt_context_destroy.restype = None
    lib.pt_last_error.argtypes = [vp]
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_set_data_dir.argtypes = [vp, C.c_char_p]
    lib.pt_scene_upload.argtypes = [vp, C.POINTER(pt_scene_desc)]
    lib.pt_scene_info_get.argtypes = [vp, C.POINTER(pt_scene_info)]
    lib.pt_film_clear.argtypes = [vp]
    lib.pt_render.argtypes = [vp, C.POINTER(pt_tile), u32]
    lib.pt_film_download_xyzw.argtypes = [vp, vp]
    lib.pt_film_device_xyzw.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.pt_film_commit_xyzw.argtypes = [vp]
    lib.pt_film_allreduce.argtypes = [vp, vp, C.c_int]
    lib.pt_film_add_xyzw.argtypes = [vp, fp]
    lib.pt_film_resolve_rgb.argtypes = [vp, vp]
    lib.pt_trace_closest.argtypes = [vp, u32, vp, vp, vp, vp]
    lib.pt_trace_any.argtypes = [vp, u32, vp, vp, vp, vp]
    lib.pt_trace_wavefront.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    lib.pt_generate_camera_rays.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    lib.pt_sobol_samples.argtypes = [vp, u32, vp, vp, vp, vp]
    lib.pt_radiance_samples.argtypes = [vp, C.POINTER(pt_tile), vp]
    lib.pt_bsdf_eval.argtypes = [vp, u32, u32, vp, vp, u32, vp, vp]
    lib.pt_bsdf_sample.argtypes = [vp, u32, u32, vp, vp, u32, vp, vp, vp, vp]
    lib.pt_get_counters.argtypes = [vp, C.POINTER(pt_counters)]
    lib.pt_reset_counters.argtypes = [vp]
    lib.pt_set_bvh_build.argtypes = [vp, C.c_int]
    lib.pt_scene_bvh_digest.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pt_bvh_leaf_order.argtypes = [C.POINTER(pt_scene_desc), vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def tiles_array(tiles):
    arr = (pt_tile * len(tiles))()
    for i, t in enumerate(tiles):
        arr[i] = pt_tile(*[int(v) for v in t])
    return arr


HOST_SYMBOLS = ["pth_parse_file", "pth_parse_file_opts", "pth_parse_string", "pth_scene_get_desc", "pth_scene_output_filename",
                "pth_scene_set_pixelsamples", "pth_scene_warnings", "pth_scene_free", "pth_write_pfm", "pth_write_image", "pth_parse_to_log",
                "pth_display_connect", "pth_display_start", "pth_display_update", "pth_display_close", "pth_tev_create_packet", "pth_tev_update_packet", "pth_blackbody"]


class ParsedScene:
    """A scene parsed from .pbrt text by the C++ front end (include/pbrtgpu_host.h).  Quacks like
    scenes.SceneDesc (has .desc), so it can be uploaded or handed to the oracle."""

    def __init__(self, text=None, filename=None, work_dir=None, lib=None):
        self.lib = lib or load_library()
        L = self.lib
        L.pth_parse_file.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.pth_parse_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        L.pth_scene_get_desc.argtypes = [C.c_void_p]
        L.pth_scene_get_desc.restype = C.POINTER(pt_scene_desc)
        L.pth_scene_output_filename.argtypes = [C.c_void_p]
        L.pth_scene_output_filename.restype = C.c_char_p
        L.pth_scene_warnings.argtypes = [C.c_void_p]
        L.pth_scene_warnings.restype = C.c_char_p
        L.pth_scene_set_pixelsamples.argtypes = [C.c_void_p, C.c_int]
        L.pth_scene_set_pixelsamples.restype = None
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_scene_free.restype = None
        self.h = C.c_void_p()
        err = C.create_string_buffer(2048)
        if filename is not None:
            st = L.pth_parse_file(filename.encode(), C.byref(self.h), err, 2048)
        else:
            st = L.pth_parse_string(text.encode(), (work_dir or ".").encode(), C.byref(self.h), err, 2048)
        if st != 0:
            raise PtError(st, err.value.decode())
        self.desc = L.pth_scene_get_desc(self.h).contents
        self.buffers = {}

    @property
    def output_filename(self):
        return self.lib.pth_scene_output_filename(self.h).decode()

    @property
    def warnings(self):
        return self.lib.pth_scene_warnings(self.h).decode()

    def set_pixelsamples(self, spp):
        self.lib.pth_scene_set_pixelsamples(self.h, int(spp))

    def __del__(self):
        try:
            if self.h:
                self.lib.pth_scene_free(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


def parse_to_log(text, lib=None):
    """Directive log of the parser alone (pth_parse_to_log); raises PtError on a syntax error."""
    lib = lib or load_library()
    lib.pth_parse_to_log.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    out = C.create_string_buffer(1 << 20)
    st = lib.pth_parse_to_log(text.encode(), b".", out, 1 << 20)
    if st != 0:
        raise PtError(st, out.value.decode())
    return out.value.decode().splitlines()


def bvh_leaf_order(scene, lib=None):
    """Host-only BVH build (no GPU): returns (order, n_nodes, n_leaves, max_stack)."""
    lib = lib or load_library()
    order = np.empty(scene.desc.n_triangles + scene.desc.n_spheres, np.uint32)
    nn, nl, ms = C.c_uint32(), C.c_uint32(), C.c_uint32()
    st = lib.pt_bvh_leaf_order(C.byref(scene.desc), _ptr(order), C.byref(nn), C.byref(nl), C.byref(ms))
    if st != 0:
        raise PtError(st, "pt_bvh_leaf_order failed")
    return order, nn.value, nl.value, ms.value


class Context:
    """One device context (pt_context).  Mirrors the call sequence of the reference's
    render_scene(): build scene -> integrator.render(scene) -> film.write_image()."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load_library()
        self.h = C.c_void_p()
        st = self.lib.pt_context_create(int(device), C.byref(self.h))
        if st != 0:
            raise PtError(st, "pt_context_create(device=%d) failed (no HIP device? there is no CPU fallback)" % device)
        self._check(self.lib.pt_set_data_dir(self.h, DATA_DIR.encode()))
        self.info = None
        self._keep = None

    def _check(self, st):
        if st != 0:
            msg = self.lib.pt_last_error(self.h)
            raise PtError(st, msg.decode() if msg else "")

    def close(self):
        if self.h:
            self.lib.pt_context_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene):
        """scene: scenes.SceneDesc (keeps its numpy buffers alive)."""
        self._keep = scene
        self._check(self.lib.pt_scene_upload(self.h, C.byref(scene.desc)))
        info = pt_scene_info()
        self._check(self.lib.pt_scene_info_get(self.h, C.byref(info)))
        self.info = info
        return info

    def set_bvh_build(self, where):
        """BVH_BUILD_AUTO / _HOST / _DEVICE: where the lower half of an HLBVH build runs at the next upload."""
        self._check(self.lib.pt_set_bvh_build(self.h, int(where)))

    def bvh_digest(self):
        """(nodes, records) FNV-1a digests of the uploaded tree."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.pt_scene_bvh_digest(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def film_shape(self):
        cb = self.info.cropped_bounds
        return (cb[3] - cb[1], cb[2] - cb[0])

    def film_clear(self):
        self._check(self.lib.pt_film_clear(self.h))

    def render(self, tiles=None):
        if tiles is None:
            self._check(self.lib.pt_render(self.h, None, 0))
        else:
            arr = tiles_array(tiles)
            self._check(self.lib.pt_render(self.h, arr, len(tiles)))

    def film_xyzw(self):
        h, w = self.film_shape
        out = np.empty((h, w, 4), np.float32)
        self._check(self.lib.pt_film_download_xyzw(self.h, _ptr(out)))
        return out

    def film_device_xyzw(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.lib.pt_film_device_xyzw(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def film_commit_xyzw(self):
        self._check(self.lib.pt_film_commit_xyzw(self.h))

    def film_add_xyzw(self, xyzw):
        """Add another rank's {X,Y,Z,weight} film (as film_xyzw() returns it) and make the sum authoritative: the exchange step staged
        through the host."""
        a = np.ascontiguousarray(xyzw, np.float32)
        assert a.size == 4 * self.film_shape[0] * self.film_shape[1]
        self._check(self.lib.pt_film_add_xyzw(self.h, a.ctypes.data_as(C.POINTER(C.c_float))))

    def film_allreduce(self, nccl_comm, root=-1):
        """Sum the XYZW film over an RCCL communicator (ncclComm_t as an integer / c_void_p) inside the library."""
        self._check(self.lib.pt_film_allreduce(self.h, C.c_void_p(nccl_comm), root))

    def film_rgb(self):
        h, w = self.film_shape
        out = np.empty((h, w, 3), np.float32)
        self._check(self.lib.pt_film_resolve_rgb(self.h, _ptr(out)))
        return out

    def trace_closest(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(tmax)
        out = np.empty(n, HIT_DTYPE)
        self._check(self.lib.pt_trace_closest(self.h, n, _ptr(o), _ptr(d), _ptr(tmax), _ptr(out)))
        return out

    def trace_any(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(tmax)
        out = np.empty(n, np.uint8)
        self._check(self.lib.pt_trace_any(self.h, n, _ptr(o), _ptr(d), _ptr(tmax), _ptr(out)))
        return out

    def trace_wavefront(self, o, d, tmax, kind):
        """Rays through the renderer's own traversal kernel; kind: 1 continuation, 2 shadow, 3 probe.  Returns (hits, occluded)."""
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32); kind = np.ascontiguousarray(kind, np.uint8)
        n = len(tmax)
        out = np.empty(n, HIT_DTYPE); occ = np.empty(n, np.uint8)
        self._check(self.lib.pt_trace_wavefront(self.h, n, _ptr(o), _ptr(d), _ptr(tmax), _ptr(kind), _ptr(out), _ptr(occ)))
        return out, occ

    def generate_camera_rays(self, pixel_xy, sample_index):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.int32); sample_index = np.ascontiguousarray(sample_index, np.uint32)
        n = len(sample_index)
        o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32); pf = np.empty((n, 2), np.float32)
        self._check(self.lib.pt_generate_camera_rays(self.h, n, _ptr(pixel_xy), _ptr(sample_index), _ptr(o), _ptr(d), _ptr(pf)))
        return o, d, pf

    def sobol_samples(self, pixel_xy, sample_index, dim):
        pixel_xy = np.ascontiguousarray(pixel_xy, np.int32); sample_index = np.ascontiguousarray(sample_index, np.uint32)
        dim = np.ascontiguousarray(dim, np.uint32)
        n = len(dim)
        out = np.empty(n, np.float32)
        self._check(self.lib.pt_sobol_samples(self.h, n, _ptr(pixel_xy), _ptr(sample_index), _ptr(dim), _ptr(out)))
        return out

    def bsdf_eval(self, material, wo, wi, flags=31):
        """BSDF::f / pdf of one material on the canonical frame (local == world)."""
        wo = np.ascontiguousarray(wo, np.float32); wi = np.ascontiguousarray(wi, np.float32)
        n = len(wo)
        f = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32)
        self._check(self.lib.pt_bsdf_eval(self.h, C.c_uint32(material), C.c_uint32(n), _ptr(wo), _ptr(wi), C.c_uint32(flags), _ptr(f), _ptr(pdf)))
        return f, pdf

    def bsdf_sample(self, material, wo, u, flags=31):
        """BSDF::sample_f; type 0 = None."""
        wo = np.ascontiguousarray(wo, np.float32); u = np.ascontiguousarray(u, np.float32)
        n = len(wo)
        f = np.empty((n, 3), np.float32); wi = np.empty((n, 3), np.float32); pdf = np.empty(n, np.float32); t = np.empty(n, np.uint32)
        self._check(self.lib.pt_bsdf_sample(self.h, C.c_uint32(material), C.c_uint32(n), _ptr(wo), _ptr(u), C.c_uint32(flags), _ptr(f), _ptr(wi),
                                            _ptr(pdf), _ptr(t)))
        return f, wi, pdf, t

    def radiance_samples(self, tile):
        t = pt_tile(*[int(v) for v in tile])
        npx = (t.x1 - t.x0) * (t.y1 - t.y0)
        out = np.empty((npx, self.info.spp, 3), np.float32)
        self._check(self.lib.pt_radiance_samples(self.h, C.byref(t), _ptr(out)))
        return out

    def counters(self):
        c = pt_counters()
        self._check(self.lib.pt_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._check(self.lib.pt_reset_counters(self.h))
