"""Randomised differential test of the whole accelerated path: scenes drawn from a seed -- geometry, materials, textures, lights, camera,
film, sampler, integrator, accelerator -- rendered by the HIP path and by the oracle, compared as the feature scenes are (hits and
occlusion of random rays, per-sample radiance bit for bit, the film, every counter).  The fixed feature scenes switch each branch on
once; this walks combinations nobody wrote down.  FUZZ_SEEDS=a:b widens the seed range for a long run (the log of one such run is kept
under profiles/)."""
import os

import numpy as np
import pytest

import feature_scenes as fs
from helpers import pkg, scenes
from test_gpu_features import _compare

pytestmark = pytest.mark.gpu
HOST, DEVICE, AUTO = pkg.capi.BVH_BUILD_HOST, pkg.capi.BVH_BUILD_DEVICE, pkg.capi.BVH_BUILD_AUTO
T = scenes


def _seeds():
    spec = os.environ.get("FUZZ_SEEDS", "0:32")
    a, b = spec.split(":")
    return list(range(int(a), int(b)))


def _rgb(rng, lo=0.05, hi=0.95):
    return tuple(float(x) for x in rng.uniform(lo, hi, 3))


def _xform(rng, center, spread=0.0, mirror_ok=True):
    """translate * rotate * non-uniform scale (sometimes mirrored: a handedness-swapping transform)."""
    c = np.asarray(center, np.float64) + rng.uniform(-spread, spread, 3)
    s = rng.uniform(0.6, 1.4, 3)
    if mirror_ok and rng.random() < 0.2:
        s[int(rng.integers(0, 3))] *= -1.0
    m = T.transform_translate(float(c[0]), float(c[1]), float(c[2]))
    if rng.random() < 0.7:
        m = T.transform_mul(m, T.transform_rotate_x(float(rng.uniform(-80, 80))))
    return T.transform_mul(m, T.transform_scale(float(s[0]), float(s[1]), float(s[2])))


def _float_texture(b, rng, depth=0):
    kind = int(rng.integers(0, 8 if depth < 2 else 5))
    tw = _xform(rng, (0, 0, 0), 1.0, mirror_ok=False) if rng.random() < 0.5 else None
    if kind == 0:
        return b.texture_fbm(octaves=int(rng.integers(1, 9)), roughness=float(rng.uniform(0.3, 0.7)), to_world=tw)
    if kind == 1:
        return b.texture_wrinkled(octaves=int(rng.integers(1, 9)), roughness=float(rng.uniform(0.3, 0.7)), to_world=tw)
    if kind == 2:
        return b.texture_windy(to_world=tw)
    if kind == 3:
        return b.texture_checkerboard(float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), uscale=float(rng.uniform(1, 9)), vscale=float(rng.uniform(1, 9)),
                                      aamode="none" if rng.random() < 0.3 else "closedform")
    if kind == 4:
        return b.texture_bilerp(*[float(x) for x in rng.uniform(0, 1, 4)], mapping=str(rng.choice(["uv", "spherical", "cylindrical", "planar"])))
    if kind == 5:
        return b.texture_scale(float(rng.uniform(0.2, 1.0)), _float_texture(b, rng, depth + 1))
    if kind == 6:
        return b.texture_mix(_float_texture(b, rng, depth + 1), float(rng.uniform(0, 1)), amount=_float_texture(b, rng, depth + 2))
    gray = b.image_pyramid(fs.test_image(int(2 ** rng.integers(0, 6)), int(2 ** rng.integers(0, 6)), 1, seed=int(rng.integers(0, 99)))[..., 0])
    return b.texture_imagemap(gray, trilinear=bool(rng.random() < 0.4), maxanisotropy=float(rng.choice([1.0, 2.0, 8.0, 16.0])),
                              wrap=str(rng.choice(["repeat", "black", "clamp"])), uscale=float(rng.uniform(0.5, 4)), vscale=float(rng.uniform(0.5, 4)),
                              udelta=float(rng.uniform(-1, 1)), vdelta=float(rng.uniform(-1, 1)))


def _colour(b, rng, depth=0):
    """An RGB parameter: mostly a constant, otherwise a spectrum texture."""
    if rng.random() < 0.55 or depth > 1:
        return _rgb(rng)
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return b.texture_checkerboard(_rgb(rng), _rgb(rng), uscale=float(rng.uniform(1, 8)), vscale=float(rng.uniform(1, 8)))
    if kind == 1:
        return b.texture_checkerboard(_rgb(rng), _rgb(rng), dimension=3, to_world=_xform(rng, (0, 0, 0), 0.5, mirror_ok=False))
    if kind == 2:
        return b.texture_dots(_rgb(rng), _rgb(rng), uscale=float(rng.uniform(2, 9)), vscale=float(rng.uniform(2, 9)))
    if kind == 3:
        return b.texture_marble(octaves=int(rng.integers(2, 9)), roughness=0.5, scale=float(rng.uniform(1, 5)), variation=float(rng.uniform(0.1, 0.5)),
                                to_world=_xform(rng, (0, 0, 0), 0.5, mirror_ok=False))
    if kind == 4:
        return b.texture_scale(_colour(b, rng, depth + 1), _float_texture(b, rng, 1))
    if kind == 5:
        return b.texture_mix(_colour(b, rng, depth + 1), _colour(b, rng, depth + 1), amount=_float_texture(b, rng, 1) if rng.random() < 0.5 else float(rng.uniform(0, 1)))
    w, h = int(2 ** rng.integers(0, 7)), int(2 ** rng.integers(0, 7))
    img = b.image_pyramid(fs.test_image(w, h, 3, seed=int(rng.integers(0, 99))))
    mapping = str(rng.choice(["uv", "spherical", "cylindrical", "planar"]))
    return b.texture_imagemap(img, trilinear=bool(rng.random() < 0.4), maxanisotropy=float(rng.choice([1.0, 4.0, 8.0])), swrap=str(rng.choice(["repeat", "black", "clamp"])),
                              twrap=str(rng.choice(["repeat", "black", "clamp"])), mapping=mapping, uscale=float(rng.uniform(0.5, 4)), vscale=float(rng.uniform(0.5, 4)),
                              udelta=float(rng.uniform(-1, 1)), vdelta=float(rng.uniform(-1, 1)), v1=tuple(float(x) for x in rng.uniform(-1, 1, 3)),
                              v2=tuple(float(x) for x in rng.uniform(-1, 1, 3)),
                              to_world=_xform(rng, (0, 0, 0), 0.5, mirror_ok=False) if mapping in ("spherical", "cylindrical") else None)


def _rough(b, rng):
    if rng.random() < 0.2:
        return b.texture_scale(float(rng.uniform(0.05, 0.5)), _float_texture(b, rng, 1))
    return float(rng.choice([0.0, 0.001, 0.01, 0.1, 0.3, 0.8]))


def _material(b, rng, ao=False):
    """`ao`: the ambient-occlusion integrator is refused with material-less surfaces (the reference panics there, ao.rs:66-78) and with
    bump maps (documented in DESIGN.md section 8), so such scenes draw neither."""
    kind = int(rng.integers(0, 9 if ao else 10))
    bump = None
    if rng.random() < 0.2 and not ao:
        bump = b.texture_scale(float(rng.uniform(0.01, 0.1)), _float_texture(b, rng, 1)) if rng.random() < 0.8 else float(rng.uniform(0, 0.2))
    remap = bool(rng.random() < 0.7)
    if kind <= 2:
        sigma = 0.0 if rng.random() < 0.5 else (float(rng.uniform(0, 120)) if rng.random() < 0.7 else b.texture_scale(90.0, _float_texture(b, rng, 1)))
        b.material_matte(_colour(b, rng) if rng.random() < 0.9 else (0.0, 0.0, 0.0), sigma=sigma, bumpmap=bump)
    elif kind == 3:
        b.material_plastic(Kd=_colour(b, rng), Ks=_colour(b, rng), roughness=_rough(b, rng), remaproughness=remap, bumpmap=bump)
    elif kind == 4:
        b.material_mirror(Kr=_colour(b, rng), bumpmap=bump)
    elif kind == 5:
        black = rng.random() < 0.15
        b.material_glass(Kr=(0, 0, 0) if black else _colour(b, rng), Kt=(0, 0, 0) if (black or rng.random() < 0.15) else _colour(b, rng),
                         eta=float(rng.uniform(1.0, 2.0)), uroughness=_rough(b, rng), vroughness=_rough(b, rng), remaproughness=remap)
    elif kind == 6:
        aniso = rng.random() < 0.5
        b.material_metal(eta=_rgb(rng, 0.1, 2.0), k=_rgb(rng, 1.0, 4.0), roughness=_rough(b, rng), uroughness=_rough(b, rng) if aniso else None,
                         vroughness=_rough(b, rng) if aniso else None, remaproughness=remap)
    elif kind == 7:
        aniso = rng.random() < 0.5
        b.material_uber(Kd=_colour(b, rng), Ks=_colour(b, rng), Kr=_rgb(rng, 0.0, 0.3) if rng.random() < 0.5 else (0, 0, 0),
                        Kt=_rgb(rng, 0.0, 0.5) if rng.random() < 0.5 else (0, 0, 0), opacity=_rgb(rng, 0.3, 1.0) if rng.random() < 0.5 else (1, 1, 1),
                        eta=float(rng.uniform(1.1, 1.9)), roughness=_rough(b, rng), uroughness=_rough(b, rng) if aniso else None,
                        vroughness=_rough(b, rng) if aniso else None, remaproughness=remap)
    elif kind == 8:
        b.material_substrate(Kd=_colour(b, rng), Ks=_colour(b, rng), uroughness=_rough(b, rng), vroughness=_rough(b, rng), remaproughness=remap)
    else:
        b.material_none()


def _soup(rng, n, center, radius, size):
    c = np.asarray(center, np.float32) + rng.normal(0.0, radius, (n, 3)).astype(np.float32)
    off = rng.uniform(-size, size, (n, 3, 3)).astype(np.float32)
    return (c[:, None, :] + off).reshape(-1, 3), np.arange(3 * n)


def _shape(b, rng, where):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        P, idx = _soup(rng, int(rng.integers(1, 600)), where, float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.02, 0.25)))
        b.shape_trianglemesh(P, idx, twosided=bool(rng.random() < 0.8))
    elif kind == 1:
        P, N, UV, idx = fs.uv_sphere(where, float(rng.uniform(0.2, 0.8)), int(rng.integers(2, 9)), int(rng.integers(3, 12)))
        tang = np.tile(np.asarray(rng.uniform(-1, 1, 3), np.float32), (len(P), 1)) if rng.random() < 0.3 else None
        b.shape_trianglemesh(P, idx, N=N if rng.random() < 0.7 else None, uv=UV if rng.random() < 0.7 else None, S=tang, twosided=bool(rng.random() < 0.8))
    elif kind == 2:
        m = _xform(rng, where)
        r = float(rng.uniform(0.2, 0.7))
        partial = rng.random() < 0.4
        b.shape_sphere(radius=r, zmin=float(rng.uniform(-r, 0)) if partial else None, zmax=float(rng.uniform(0, r)) if partial else None,
                       phimax=float(rng.uniform(60, 360)) if partial else 360.0, object_to_world=m[0], world_to_object=m[1])
    elif kind == 3:
        q = np.asarray(where, np.float32) + rng.uniform(-0.7, 0.7, (4, 3)).astype(np.float32)
        b.shape_trianglemesh(q, [0, 1, 2, 0, 2, 3], uv=[(0, 0), (1, 0), (1, 1), (0, 1)] if rng.random() < 0.5 else None, twosided=bool(rng.random() < 0.8))
    else:
        b.reverse_orientation = True
        q = np.asarray(where, np.float32) + rng.uniform(-0.6, 0.6, (3, 3)).astype(np.float32)
        b.shape_trianglemesh(q, [0, 1, 2], twosided=False)
        b.reverse_orientation = False


def random_scene(seed):
    rng = np.random.default_rng(1000 + seed)
    b = scenes.SceneBuilder()
    eye = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -6.5 + float(rng.uniform(-0.5, 0.5)))
    b.look_at(eye, (0, 0, 0), (0, 1, 0))
    lens = rng.random() < 0.25
    b.camera_perspective(fov=float(rng.uniform(30, 60)), lensradius=float(rng.uniform(0.02, 0.2)) if lens else 0.0, focaldistance=float(rng.uniform(4, 8)) if lens else 1e6)
    res = int(rng.integers(28, 48))             # (the comparison renders a 16 x 16 tile at the centre of the sample bounds)
    crop = (0.0, 1.0, 0.0, 1.0) if rng.random() < 0.7 else (float(rng.uniform(0, 0.1)), float(rng.uniform(0.9, 1)), float(rng.uniform(0, 0.1)), float(rng.uniform(0.9, 1)))
    b.film(xresolution=res, yresolution=int(res * rng.uniform(0.8, 1.0)), cropwindow=crop, scale=float(rng.choice([1.0, 1.0, 2.5])),
           maxsampleluminance=float(rng.choice([np.inf, np.inf, 4.0])))
    filt = int(rng.integers(0, 10))
    exact_film = filt < 6
    if exact_film:
        b.pixel_filter_box()
    elif filt == 6:
        b.pixel_filter_gaussian(float(rng.uniform(1, 2.5)), float(rng.uniform(1, 2.5)), float(rng.uniform(1, 3)))
    elif filt == 7:
        b.pixel_filter_triangle(float(rng.uniform(1, 2.5)), float(rng.uniform(1, 2.5)))
    elif filt == 8:
        b.pixel_filter_mitchell(float(rng.uniform(1, 2.5)), float(rng.uniform(1, 2.5)))
    else:
        b.pixel_filter_sinc(float(rng.uniform(2, 4)), float(rng.uniform(2, 4)), float(rng.uniform(2, 4)))
    spp = int(rng.choice([1, 2, 4, 8]))
    if rng.random() < 0.6:
        b.sampler_sobol(spp)
    else:
        b.sampler_halton(int(rng.choice([1, 3, 4, 7])), samplepixelcenter=bool(rng.random() < 0.3))
    integ = int(rng.integers(0, 10))
    ao = integ >= 9
    if integ < 6:
        b.integrator_path(maxdepth=int(rng.integers(1, 8)), rrthreshold=float(rng.choice([1.0, 0.5, 0.05])), lightsamplestrategy=str(rng.choice(["spatial", "power", "uniform"])))
    elif integ < 8:
        b.integrator_directlighting(maxdepth=int(rng.integers(1, 6)), strategy=str(rng.choice(["all", "one"])))
    elif integ == 8:
        b.integrator_whitted(maxdepth=int(rng.integers(1, 6)))
    else:
        b.integrator_ao(nsamples=int(rng.choice([4, 16, 64])), cossample=bool(rng.random() < 0.5))
    b.accelerator_bvh(splitmethod=str(rng.choice(["sah", "sah", "hlbvh", "middle", "equal"])), maxnodeprims=int(rng.choice([1, 2, 4, 4, 8, 13, 255])))
    # objects for instancing, declared first
    names = []
    for k in range(int(rng.integers(0, 3)) if rng.random() < 0.35 else 0):
        name = "obj%d" % k
        b.object_begin(name)
        for _ in range(int(rng.integers(1, 4))):
            _material(b, rng, ao)
            _shape(b, rng, (0.0, 0.0, 0.0))
        b.object_end()
        names.append(name)
    fs.room(b, light_L=_rgb(rng, 3.0, 12.0), two_sided_light=bool(rng.random() < 0.3))
    for _ in range(int(rng.integers(1, 6))):
        _material(b, rng, ao)
        _shape(b, rng, tuple(float(x) for x in rng.uniform(-1.4, 1.4, 3)))
    for name in names:
        for _ in range(int(rng.integers(1, 3))):
            b.object_instance(name, _xform(rng, tuple(float(x) for x in rng.uniform(-1.3, 1.3, 3))))
    # more lights: emissive quads, mesh lights, sphere lights, with sample counts for directlighting
    for _ in range(int(rng.integers(0, 4))):
        b.material_matte(_rgb(rng))
        b.area_light_source_diffuse(L=_rgb(rng, 0.5, 8.0), twosided=bool(rng.random() < 0.4), nsamples=int(rng.choice([1, 1, 2, 5])))
        c = tuple(float(x) for x in rng.uniform(-1.6, 1.6, 3))
        k = int(rng.integers(0, 3))
        if k == 0:
            q = np.asarray(c, np.float32) + rng.uniform(-0.3, 0.3, (4, 3)).astype(np.float32)
            b.shape_trianglemesh(q, [0, 1, 2, 0, 2, 3])
        elif k == 1:
            P, N, UV, idx = fs.uv_sphere(c, float(rng.uniform(0.1, 0.3)), 3, 5)
            b.shape_trianglemesh(P, idx, N=N if rng.random() < 0.5 else None)
        else:
            m = _xform(rng, c)
            b.shape_sphere(radius=float(rng.uniform(0.1, 0.3)), object_to_world=m[0], world_to_object=m[1])
        b.no_area_light()
    return b.build(), exact_film


@pytest.fixture(scope="module")
def fuzz_ctx(pkg):
    import torch  # noqa: F401  (see conftest.gpu_ctx)
    ctx = pkg.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="module")
def fuzz_ctx_far(pkg):
    """A context that traces with k_trace_far -- nodes fetched by lane pairs -- wherever the scene allows it (the switch is read when the context is made)."""
    import torch  # noqa: F401
    old = os.environ.get("PBRTGPU_TRACE_FAR")
    os.environ["PBRTGPU_TRACE_FAR"] = "1"
    try:
        ctx = pkg.Context(0)
    finally:
        if old is None:
            del os.environ["PBRTGPU_TRACE_FAR"]
        else:
            os.environ["PBRTGPU_TRACE_FAR"] = old
    yield ctx
    ctx.close()


@pytest.mark.parametrize("seed", _seeds()[:int(os.environ.get("FUZZ_FAR_N", "16"))])
def test_random_scene_pairwise_node_fetch(fuzz_ctx_far, oracle, seed):
    """The same scenes through k_trace_far: a lane pair fetches its two nodes together, half of the lanes idle or parked on leaves, nodes in
    LDS and in HBM side by side in one pair -- whatever the draw produces; everything stays the oracle's."""
    test_random_scene(fuzz_ctx_far, oracle, seed)


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene(fuzz_ctx, oracle, seed):
    sd, exact_film = random_scene(seed)
    fuzz_ctx.set_bvh_build(DEVICE if seed % 3 == 0 else (HOST if seed % 3 == 1 else AUTO))
    osc = oracle.scene(sd)
    oracle.reference_panics()          # cleared
    try:
        fuzz_ctx.upload(sd)
        d = sd.desc
        what = "%d triangles %d spheres %d instances %d lights, integrator %d split %d leaf %d" % (
            d.n_triangles, d.n_spheres, d.n_instances, fuzz_ctx.info.n_lights, d.integrator, d.split_method, d.max_node_prims)
        try:
            err, frac = _compare(fuzz_ctx, osc, exact_film, weight_tol=1e-5)
        except pkg.capi.PtError as e:
            # the one run-time refusal: a path past the Halton sampler's 1000 dimensions, where the reference panics -- the oracle must see it too
            if "1000 dimensions" not in str(e):
                raise
            osc.render(threads=8)
            assert oracle.reference_panics() & 1
            print("\n[fuzz %d] %s: the reference panics (Halton dimensions), reported by both" % (seed, what))
        else:
            assert oracle.reference_panics() == 0
            print("\n[fuzz %d] %s: image rel-L2 %.2e" % (seed, what, err))
    finally:
        osc.close()
        fuzz_ctx.set_bvh_build(AUTO)


def test_halton_dimension_overflow_is_reported(fuzz_ctx, oracle):
    """Whitted recursion between mirrors asks the Halton sampler for more than its 1000 dimensions: the reference panics
    (halton.rs:103-108, PRIME_SUMS has 1000 entries); pt_render says so instead of returning a film, and the oracle notes the same."""
    b = fs.base(res=16, spp=1)
    b.sampler_halton(1)
    b.integrator_whitted(maxdepth=16)
    s = 2.0
    b.material_mirror(Kr=(0.95, 0.95, 0.95))          # a hall of mirrors: every vertex reflects on and samples 48 lights
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    scenes._quad(b, (s, -s, -7.0), (-s, -s, -7.0), (-s, s, -7.0), (s, s, -7.0))     # behind the camera
    b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=(3, 3, 3))
    P, N, UV, idx = fs.uv_sphere((0.0, 1.2, 0.0), 0.3, 4, 6)            # a 48-triangle mesh light: 48 lights, two dimensions each at every vertex
    b.shape_trianglemesh(P, idx)
    b.no_area_light()
    sd = b.build()
    osc = oracle.scene(sd)
    oracle.reference_panics()
    try:
        fuzz_ctx.upload(sd)
        fuzz_ctx.film_clear()
        with pytest.raises(pkg.capi.PtError, match="1000 dimensions"):
            fuzz_ctx.render()
        osc.render(threads=4)
        assert oracle.reference_panics() & 1
        # the context is usable afterwards
        sd2 = fs.scene_attributes()
        o2 = oracle.scene(sd2)
        fuzz_ctx.upload(sd2)
        _compare(fuzz_ctx, o2, True)
        o2.close()
    finally:
        osc.close()


def test_long_halton_paths_are_refused_at_upload(fuzz_ctx):
    """path + Halton + maxdepth > 124: 5 dimensions for the camera and 8 per vertex run past the sampler's 1000 dimensions, where the reference
    panics; the shading kernels do not check per sample (it cost them registers they do not have), so the upload refuses the combination."""
    b = fs.base(res=16, spp=1)
    b.sampler_halton(1)
    fs.room(b)
    b.integrator_path(maxdepth=124)
    fuzz_ctx.upload(b.build())                  # 5 + 8 * 124 = 997: fine
    b.integrator_path(maxdepth=125)
    with pytest.raises(pkg.capi.PtError, match="1000"):
        fuzz_ctx.upload(b.build())
    b.sampler_sobol(1)
    fuzz_ctx.upload(b.build())                  # Sobol' wraps instead (sobol.rs:39-56): no limit
