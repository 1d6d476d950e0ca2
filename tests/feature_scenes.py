"""Small scenes that switch on every branch of the accelerated path that the two BASELINE
scenes leave cold (vertex normals / tangents / uvs, one-sided and reversed meshes, Oren-Nayar,
black and absent materials, several lights with each sampling strategy, thin lens, wide
reconstruction filters, crop window, luminance clamp, every BVH split method and leaf size)."""
import numpy as np

from helpers import pkg, scenes

f32 = np.float32


def uv_sphere(center, radius, nt=8, nphi=12):
    P, N, UV, idx = [], [], [], []
    for t in range(nt + 1):
        th = np.pi * t / nt
        for p in range(nphi):
            ph = 2 * np.pi * p / nphi
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            P.append(np.array(center) + radius * n); N.append(n); UV.append([p / nphi, t / nt])
    for t in range(nt):
        for p in range(nphi):
            a, b = t * nphi + p, t * nphi + (p + 1) % nphi
            c, d = a + nphi, b + nphi
            idx += [a, c, b, b, c, d]
    return np.array(P, np.float32), np.array(N, np.float32), np.array(UV, np.float32), idx


def room(b, size=2.0, light_L=(10, 9, 8), two_sided_light=False):
    s = size
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    b.material_matte((0.2, 0.6, 0.3))
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    b.material_matte((0.7, 0.2, 0.2))
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=light_L, twosided=two_sided_light)
    h = 0.999 * s
    scenes._quad(b, (0.5, h, -0.5), (0.5, h, 0.5), (-0.5, h, 0.5), (-0.5, h, -0.5))
    b.no_area_light()


def base(res=48, spp=8, depth=5):
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -6.5), (0, 0, 0), (0, 1, 0))
    b.camera_perspective(fov=40.0)
    b.film(xresolution=res, yresolution=res)
    b.pixel_filter_box()
    b.sampler_sobol(spp)
    b.integrator_path(maxdepth=depth)
    return b


def scene_attributes():
    """Smooth-shaded sphere (N + uv), a tangent-carrying quad (S), a one-sided quad, a reversed one."""
    b = base()
    room(b)
    b.material_matte((0.6, 0.6, 0.8))
    P, N, UV, idx = uv_sphere((-0.7, -1.2, 0.3), 0.8)
    b.shape_trianglemesh(P, idx, N=N, uv=UV)
    b.material_matte((0.8, 0.7, 0.3))
    q = [(0.3, -2 + 0.01, -0.5), (1.5, -2 + 0.01, -0.5), (1.5, -0.9, 0.6), (0.3, -0.9, 0.6)]
    S = [(1, 0.2, 0)] * 4
    b.shape_trianglemesh(q, [0, 1, 2, 0, 2, 3], S=S)
    b.material_matte((0.3, 0.3, 0.9))
    b.shape_trianglemesh([(-1.9, 0.2, 1.0), (-0.6, 0.2, 1.5), (-0.6, 1.5, 1.5), (-1.9, 1.5, 1.0)], [0, 1, 2, 0, 2, 3], twosided=False)
    b.reverse_orientation = True
    b.shape_trianglemesh([(0.6, 0.2, 1.5), (1.9, 0.2, 1.0), (1.9, 1.5, 1.0), (0.6, 1.5, 1.5)], [0, 1, 2, 0, 2, 3], twosided=False)
    b.reverse_orientation = False
    P2, N2, UV2, idx2 = uv_sphere((1.0, 1.0, -0.5), 0.4, 5, 8)
    b.shape_trianglemesh(P2, idx2, N=N2)            # normals, auto uv rule (not fillable -> default uvs)
    return b.build()


def scene_materials_lights(strategy="spatial"):
    """Oren-Nayar, a black matte, a material-less occluder, three lights (one two-sided, one mesh light)."""
    b = base(depth=6)
    b.integrator_path(maxdepth=6, lightsamplestrategy=strategy)
    room(b, light_L=(6, 6, 6))
    b.material_matte((0.7, 0.6, 0.5), sigma=35.0)
    P, N, UV, idx = uv_sphere((-0.8, -1.2, 0.0), 0.8, 6, 10)
    b.shape_trianglemesh(P, idx)
    b.material_matte((0.0, 0.0, 0.0))
    scenes._cuboid(b, [(0.4, -1.0, -0.4), (0.4, -1.0, 0.6), (1.4, -1.0, 0.6), (1.4, -1.0, -0.4)], 1.0)
    b.material_none()
    scenes._quad(b, (-1.5, -0.5, -1.5), (1.5, -0.5, -1.5), (1.5, 1.0, -1.5), (-1.5, 1.0, -1.5))   # invisible pane in front
    b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=(4, 1, 1), twosided=True)
    scenes._quad(b, (-1.99, 0.0, -0.3), (-1.99, 0.0, 0.3), (-1.99, 0.6, 0.3), (-1.99, 0.6, -0.3))
    b.area_light_source_diffuse(L=(0.5, 0.5, 3))
    P3, N3, UV3, idx3 = uv_sphere((1.2, 1.2, 0.8), 0.25, 3, 5)
    b.shape_trianglemesh(P3, idx3)                    # 30-triangle mesh light
    b.no_area_light()
    return b.build()


def scene_camera_film(filt="gaussian"):
    """Thin lens, wide reconstruction filter, crop window, film scale, luminance clamp."""
    b = base(res=40, spp=4)
    b.camera_perspective(fov=40.0, lensradius=0.15, focaldistance=6.0)
    b.film(xresolution=40, yresolution=32, cropwindow=(0.2, 0.85, 0.1, 0.9), scale=2.0, maxsampleluminance=3.0)
    if filt == "gaussian":
        b.pixel_filter_gaussian(2.0, 2.0, 2.0)
    elif filt == "mitchell":
        b.pixel_filter_mitchell(2.0, 1.5)
    elif filt == "sinc":
        b.pixel_filter_sinc(3.0, 2.5, 2.0)
    else:
        b.pixel_filter_triangle(1.5, 2.0)
    room(b)
    b.material_matte((0.6, 0.6, 0.6))
    P, N, UV, idx = uv_sphere((0.0, -1.0, 0.0), 1.0, 6, 10)
    b.shape_trianglemesh(P, idx, N=N)
    return b.build()


def scene_accel(method, leaf):
    sd = scenes.rt1m(4000, res=32, spp=4, max_depth=4)
    sd.desc.split_method = {"sah": 0, "hlbvh": 1, "middle": 2, "equal": 3}[method]
    sd.desc.max_node_prims = leaf
    return sd


# ---- materials beyond matte (SURVEY.md section 8, rows a10/a11 "later")
MATERIAL_KINDS = ["plastic", "mirror", "glass", "glass_rough", "glass_reflect_only", "glass_black", "metal", "metal_aniso",
                  "uber", "uber_translucent", "substrate", "substrate_aniso", "plastic_noremap"]


def apply_material(b, kind):
    if kind == "plastic":
        b.material_plastic(Kd=(0.3, 0.1, 0.1), Ks=(0.4, 0.4, 0.4), roughness=0.15)
    elif kind == "plastic_noremap":
        b.material_plastic(Kd=(0.0, 0.0, 0.0), Ks=(0.6, 0.5, 0.4), roughness=0.2, remaproughness=False)
    elif kind == "mirror":
        b.material_mirror(Kr=(0.9, 0.85, 0.8))
    elif kind == "glass":
        b.material_glass(eta=1.5)
    elif kind == "glass_rough":
        b.material_glass(Kr=(0.9, 0.9, 0.9), Kt=(0.8, 0.9, 0.8), eta=1.33, uroughness=0.1, vroughness=0.2)
    elif kind == "glass_reflect_only":
        b.material_glass(Kr=(1.0, 1.0, 1.0), Kt=(0.0, 0.0, 0.0), eta=1.5, uroughness=0.05, vroughness=0.05)
    elif kind == "glass_black":
        b.material_glass(Kr=(0.0, 0.0, 0.0), Kt=(0.0, 0.0, 0.0))
    elif kind == "metal":
        b.material_metal(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.05)
    elif kind == "metal_aniso":
        b.material_metal(eta=(0.14, 0.37, 1.44), k=(3.98, 2.38, 1.6), uroughness=0.02, vroughness=0.2)
    elif kind == "uber":
        b.material_uber(Kd=(0.3, 0.3, 0.5), Ks=(0.2, 0.2, 0.2), Kr=(0.1, 0.1, 0.1), Kt=(0.2, 0.2, 0.2), eta=1.4, roughness=0.1)
    elif kind == "uber_translucent":
        b.material_uber(Kd=(0.5, 0.4, 0.3), Ks=(0.3, 0.3, 0.3), opacity=(0.6, 0.7, 0.8), uroughness=0.05, vroughness=0.15)
    elif kind == "substrate":
        b.material_substrate(Kd=(0.4, 0.2, 0.1), Ks=(0.3, 0.3, 0.3), uroughness=0.1, vroughness=0.1)
    elif kind == "substrate_aniso":
        b.material_substrate(Kd=(0.1, 0.3, 0.4), Ks=(0.5, 0.5, 0.5), uroughness=0.02, vroughness=0.3)
    else:
        raise ValueError(kind)


def scene_material_palette():
    """One small triangle per material kind: the scene only carries the material table for the BSDF hooks."""
    b = base(res=16, spp=1)
    room(b)
    index = {}
    for k, kind in enumerate(MATERIAL_KINDS):
        apply_material(b, kind)
        index[kind] = b.cur_material
        x = -1.8 + 0.25 * k
        b.shape_trianglemesh([(x, -1.9, 0), (x + 0.2, -1.9, 0), (x, -1.7, 0)], [0, 1, 2])
    index["matte"] = 1
    b.material_matte((0.5, 0.4, 0.3), sigma=25.0)
    index["oren_nayar"] = b.cur_material
    b.shape_trianglemesh([(1.6, -1.9, 0), (1.8, -1.9, 0), (1.6, -1.7, 0)], [0, 1, 2])
    sd = b.build()
    sd.material_index = index
    return sd


def scene_materials_render(kinds, res=40, spp=8, depth=6, sampler="sobol"):
    """Room with a sphere and two slabs carrying the given materials (smooth-shaded sphere: shading != geometric normal)."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    room(b)
    apply_material(b, kinds[0])
    P, N, UV, idx = uv_sphere((-0.6, -1.1, 0.2), 0.9, 10, 16)
    b.shape_trianglemesh(P, idx, N=N, uv=UV)
    apply_material(b, kinds[1 % len(kinds)])
    b.shape_trianglemesh([(0.3, -1.99, -0.8), (1.7, -1.99, -0.8), (1.7, -0.6, 0.9), (0.3, -0.6, 0.9)], [0, 1, 2, 0, 2, 3])
    apply_material(b, kinds[2 % len(kinds)])
    b.shape_trianglemesh([(-1.9, 0.0, 1.2), (-0.5, 0.0, 1.6), (-0.5, 1.4, 1.6), (-1.9, 1.4, 1.2)], [0, 1, 2, 0, 2, 3])
    return b.build()


def scene_spheres(strategy="spatial", split="sah", res=40, spp=8, depth=6, sampler="sobol", lights_only=False):
    """Analytic spheres in the triangle BVH (shapes/sphere.rs): two sphere lights (one rotated and non-uniformly scaled, one
    two-sided and reversed), and as scene objects a matte full sphere, a glass sphere, a mirror sphere clipped in z and phi and a
    plastic sphere under a handedness-swapping transform.  The room's own quad light stays, so light sampling mixes shapes."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    b.integrator_path(maxdepth=depth, lightsamplestrategy=strategy)
    b.accelerator_bvh(splitmethod=split)
    T = scenes
    # a sphere light ahead of every triangle in the primitive list
    b.area_light_source_diffuse(L=(30, 26, 20))
    t = T.transform_mul(T.transform_translate(-1.2, 1.3, 0.4), T.transform_mul(T.transform_rotate_x(30.0), T.transform_scale(0.25, 0.15, 0.2)))
    b.shape_sphere(radius=1.0, object_to_world=t[0], world_to_object=t[1])
    b.no_area_light()
    room(b, light_L=(4, 4, 4))
    if not lights_only:
        b.material_matte((0.3, 0.4, 0.8), sigma=20.0)
        t = T.transform_translate(-0.9, -1.4, 0.6)
        b.shape_sphere(radius=0.6, object_to_world=t[0], world_to_object=t[1])
        b.material_glass()
        t = T.transform_translate(0.9, -1.3, -0.3)
        b.shape_sphere(radius=0.7, object_to_world=t[0], world_to_object=t[1])
        b.material_mirror()
        t = T.transform_mul(T.transform_translate(0.2, 0.1, 1.1), T.transform_rotate_x(-70.0))
        b.shape_sphere(radius=0.55, zmin=-0.3, zmax=0.45, phimax=250.0, object_to_world=t[0], world_to_object=t[1])
        b.material_plastic()
        t = T.transform_mul(T.transform_translate(-0.2, -0.6, -0.9), T.transform_scale(0.4, -0.4, 0.4))
        b.shape_sphere(radius=1.0, object_to_world=t[0], world_to_object=t[1])
    # a two-sided light sphere with reversed orientation, after the triangles
    b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=(6, 8, 12), twosided=True)
    b.reverse_orientation = True
    t = T.transform_translate(1.3, 0.9, 0.8)
    b.shape_sphere(radius=0.2, object_to_world=t[0], world_to_object=t[1])
    b.reverse_orientation = False
    b.no_area_light()
    return b.build()


def scene_textures(res=48, spp=8, depth=5, sampler="sobol", aamode="closedform", lens=False):
    """Procedural textures (src/textures/) driving material parameters: a closed-form filtered checkerboard floor (uv mapping
    with scale / offset) whose checks are themselves a mix and a uv texture, a 3-D checkerboard under a transform on a matte
    sphere mesh with Oren-Nayar sigma from a bilerp float texture, a planar-mapped checkerboard on plastic Kd with a scale
    texture on Ks, spherical / cylindrical mappings on two analytic spheres (uber Kd / opacity, metal eta)."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    if lens:        # thin lens: the offset rays start at the lens sample (perspective.rs:146-165)
        b.camera_perspective(fov=40.0, lensradius=0.08, focaldistance=6.0)
    T = scenes
    s = 2.0
    uvt = b.texture_uv(uscale=3.0, vscale=2.0)
    mixt = b.texture_mix((0.8, 0.2, 0.1), (0.1, 0.3, 0.8), amount=b.texture_bilerp(0.1, 0.9, 0.6, 0.3))
    floor = b.texture_checkerboard(mixt, uvt, uscale=6.0, vscale=6.0, udelta=0.25, vdelta=0.1, aamode=aamode)
    b.material_matte(floor)
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    planar = b.texture_checkerboard((0.9, 0.9, 0.2), (0.2, 0.2, 0.2), mapping="planar", v1=(1.5, 0.0, 0.0), v2=(0.0, 1.5, 0.3), udelta=0.2, vdelta=0.4)
    b.material_plastic(Kd=planar, Ks=b.texture_scale((0.5, 0.5, 0.5), b.texture_checkerboard(1.0, 0.2, uscale=10.0, vscale=10.0)), roughness=0.05)
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    b.material_matte((0.2, 0.6, 0.3))
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    b.material_matte((0.7, 0.2, 0.2))
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=(10, 9, 8))
    h = 0.999 * s
    scenes._quad(b, (0.5, h, -0.5), (0.5, h, 0.5), (-0.5, h, 0.5), (-0.5, h, -0.5))
    b.no_area_light()
    # 3-D checkerboard under a rotation + scale, sigma from a float texture
    t3 = T.transform_mul(T.transform_rotate_x(25.0), T.transform_scale(3.0, 3.0, 3.0))
    c3 = b.texture_checkerboard((0.9, 0.5, 0.1), (0.1, 0.1, 0.4), dimension=3, to_world=t3)
    b.material_matte(c3, sigma=b.texture_bilerp(0.0, 40.0, 10.0, 60.0))
    P, N, UV, idx = uv_sphere((-1.0, -1.2, 0.4), 0.7)
    b.shape_trianglemesh(P, idx, N=N, uv=UV)
    # analytic spheres with spherical / cylindrical mappings (their own world_to_texture)
    ts = T.transform_translate(0.9, -1.3, -0.2)
    sph_tex = b.texture_checkerboard((0.8, 0.8, 0.8), (0.15, 0.3, 0.15), mapping="spherical", uscale=1.0, to_world=ts, aamode=aamode)
    b.material_uber(Kd=b.texture_scale(sph_tex, (8.0, 8.0, 8.0)) if False else sph_tex, Ks=(0.2, 0.2, 0.2),
                    opacity=b.texture_checkerboard((1.0, 1.0, 1.0), (0.3, 0.3, 0.3), mapping="cylindrical", to_world=ts))
    b.shape_sphere(radius=0.65, object_to_world=ts[0], world_to_object=ts[1])
    tm = T.transform_translate(0.1, 0.2, 1.2)
    b.material_metal(eta=b.texture_checkerboard((0.2, 0.92, 1.1), (1.5, 0.9, 0.4), uscale=4.0, vscale=8.0), k=(3.9, 2.45, 2.14), roughness=0.1)
    b.shape_sphere(radius=0.5, object_to_world=tm[0], world_to_object=tm[1])
    return b.build()


def scene_roughness_textures(res=48, spp=8, depth=6, sampler="sobol"):
    """Float textures behind "roughness" / "uroughness" / "vroughness" / "eta" (ABI 6): evaluated at every hit, roughness through
    roughness_to_alpha's `ln` on the device (plastic.rs:57-62, glass.rs:57-81, metal.rs:58-69, uber.rs:66,96-104,
    substrate.rs:45-54, trowbridge_reitz.rs:113-121).  A plastic wall whose roughness is a checkerboard, a rough-glass sphere with
    bilerp u/v roughness and a checkerboard index (its smooth checks make the glass specular there), a metal sphere whose only
    "roughness" is a texture (uroughness / vroughness fall back to it) with remapping off, an uber mesh with a textured vroughness
    beside a constant uroughness, and a substrate floor with fbm roughness."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    T = scenes
    s = 2.0
    b.material_substrate(Kd=(0.4, 0.3, 0.2), Ks=(0.3, 0.3, 0.3), uroughness=b.texture_scale(b.texture_fbm(octaves=4, roughness=0.6), 0.4),
                         vroughness=b.texture_bilerp(0.02, 0.3, 0.3, 0.02))
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    b.material_plastic(Kd=(0.3, 0.4, 0.7), Ks=(0.5, 0.5, 0.5), roughness=b.texture_checkerboard(0.02, 0.4, uscale=5.0, vscale=5.0, aamode="none"))
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    b.material_matte((0.2, 0.6, 0.3))
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    b.material_matte((0.7, 0.2, 0.2))
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    b.material_matte((0.5, 0.5, 0.5))
    b.area_light_source_diffuse(L=(10, 9, 8))
    h = 0.999 * s
    scenes._quad(b, (0.5, h, -0.5), (0.5, h, 0.5), (-0.5, h, 0.5), (-0.5, h, -0.5))
    b.no_area_light()
    tg = T.transform_translate(0.9, -1.3, -0.2)
    b.material_glass(eta=b.texture_checkerboard(1.3, 1.7, uscale=3.0, vscale=3.0, aamode="none"),
                     uroughness=b.texture_checkerboard(0.0, 0.15, uscale=2.0, vscale=4.0, aamode="none"),
                     vroughness=b.texture_checkerboard(0.0, 0.05, uscale=2.0, vscale=4.0, aamode="none"))
    b.shape_sphere(radius=0.65, object_to_world=tg[0], world_to_object=tg[1])
    tm = T.transform_translate(0.1, 0.2, 1.2)
    b.material_metal(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=b.texture_bilerp(0.01, 0.2, 0.08, 0.3), remaproughness=False)
    b.shape_sphere(radius=0.5, object_to_world=tm[0], world_to_object=tm[1])
    b.material_uber(Kd=(0.3, 0.3, 0.1), Ks=(0.4, 0.4, 0.4), Kr=(0.1, 0.1, 0.1), uroughness=0.05, vroughness=b.texture_checkerboard(0.02, 0.3, uscale=6.0, vscale=6.0),
                    eta=b.texture_bilerp(1.2, 1.6, 1.4, 1.8))
    P, N, UV, idx = uv_sphere((-1.0, -1.2, 0.4), 0.7)
    b.shape_trianglemesh(P, idx, N=N, uv=UV)
    return b.build()


# The scene of test_host_frontend.py::test_texture_directives_equal_programmatic_scene as .pbrt text (also rendered on the
# device by test_gpu_features.py): every texture class on the path, bound through shape and material parameter lists.
TEXTURED_PBRT = '''
    LookAt 0 0 -6.5  0 0 0  0 1 0
    Camera "perspective" "float fov" 40
    Film "image" "integer xresolution" 32 "integer yresolution" 32
    Sampler "sobol" "integer pixelsamples" 4
    Integrator "path" "integer maxdepth" 4
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [10 9 8]
        Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [0.5 1.99 -0.5  0.5 1.99 0.5  -0.5 1.99 0.5  -0.5 1.99 -0.5]
      AttributeEnd
      Texture "uvt" "spectrum" "uv" "float uscale" 3 "float vscale" 2
      Texture "amt" "float" "bilerp" "float v00" 0.1 "float v01" 0.9 "float v10" 0.6 "float v11" 0.3
      Texture "mixt" "color" "mix" "rgb tex1" [0.8 0.2 0.1] "rgb tex2" [0.1 0.3 0.8] "texture amount" "amt"
      Texture "floor" "spectrum" "checkerboard" "texture tex1" "mixt" "texture tex2" "uvt" "float uscale" 6 "float vscale" 6
              "float udelta" 0.25 "float vdelta" 0.1
      Material "matte" "texture Kd" "floor"
      Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [2 -2 -2  -2 -2 -2  -2 -2 2  2 -2 2]
      Texture "planar" "spectrum" "checkerboard" "rgb tex1" [0.9 0.9 0.2] "rgb tex2" [0.2 0.2 0.2] "string mapping" "planar"
              "vector v1" [1.5 0 0] "vector v2" [0 1.5 0.3] "float udelta" 0.2 "float vdelta" 0.4 "string aamode" "none"
      Texture "fine" "spectrum" "checkerboard" "rgb tex1" [1 1 1] "rgb tex2" [0.2 0.2 0.2] "float uscale" 10 "float vscale" 10
      Texture "ks" "spectrum" "scale" "rgb tex1" [0.5 0.5 0.5] "texture tex2" "fine"
      Material "plastic" "texture Kd" "planar" "texture Ks" "ks" "float roughness" 0.05
      Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [2 -2 2  -2 -2 2  -2 2 2  2 2 2]
      AttributeBegin
        Rotate 25 1 0 0
        Scale 3 3 3
        Texture "c3" "spectrum" "checkerboard" "integer dimension" 3 "rgb tex1" [0.9 0.5 0.1] "rgb tex2" [0.1 0.1 0.4]
      AttributeEnd
      AttributeBegin
        Translate 0.9 -1.3 -0.2
        Texture "sph" "spectrum" "checkerboard" "string mapping" "spherical" "rgb tex1" [0.8 0.8 0.8] "rgb tex2" [0.15 0.3 0.15]
        Texture "sig" "float" "bilerp" "float v00" 0 "float v01" 40 "float v10" 10 "float v11" 60
        Material "matte" "texture Kd" "c3"
        Shape "sphere" "float radius" 0.65 "texture sigma" "sig"
      AttributeEnd
      Material "uber" "texture Kd" "sph" "rgb Ks" [0.2 0.2 0.2]
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-2 -2 2  -2 -2 -2  -2 2 0]
    WorldEnd
    '''


def scene_noise_textures(res=48, spp=8, depth=4, sampler="sobol"):
    """Perlin-noise textures (core/texture/noise.rs): dots on the floor, fbm / wrinkled / windy as Matte sigma and as colours
    through scale / mix, marble on a sphere; each under its own 3-D mapping transform (the octave count of fbm follows the
    camera ray's differentials at the first hit and is max_octaves afterwards)."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    T = scenes
    s = 2.0
    dots = b.texture_dots((0.8, 0.8, 0.7), (0.7, 0.1, 0.1), uscale=5.0, vscale=5.0)
    b.material_matte(dots)
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    t1 = T.transform_scale(2.0, 2.0, 2.0)
    wr = b.texture_wrinkled(octaves=6, roughness=0.6, to_world=t1)
    b.material_matte(b.texture_scale((0.9, 0.8, 0.5), wr))
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    t2 = T.transform_mul(T.transform_rotate_x(40.0), T.transform_scale(1.5, 1.5, 1.5))
    fb = b.texture_fbm(octaves=8, roughness=0.5, to_world=t2)
    b.material_plastic(Kd=b.texture_mix((0.1, 0.2, 0.7), (0.9, 0.9, 0.9), amount=fb), Ks=(0.2, 0.2, 0.2), roughness=0.1)
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    b.material_matte(b.texture_scale((0.4, 0.6, 0.9), b.texture_windy(to_world=T.transform_scale(3.0, 3.0, 3.0))), sigma=b.texture_scale(60.0, b.texture_fbm(octaves=3)))
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    b.area_light_source_diffuse(L=(10, 9, 8))
    h = 0.999 * s
    scenes._quad(b, (0.5, h, -0.5), (0.5, h, 0.5), (-0.5, h, 0.5), (-0.5, h, -0.5))
    b.no_area_light()
    tm = T.transform_translate(0.0, -1.0, 0.3)
    b.material_matte(b.texture_marble(octaves=8, roughness=0.5, scale=4.0, variation=0.4, to_world=tm))
    b.shape_sphere(radius=0.9, object_to_world=tm[0], world_to_object=tm[1])
    return b.build()


def scene_bump(res=48, spp=8, depth=4, lens=False):
    """Bump mapping (core/material.rs:31-72): displacement textures bending the shading frame of a plain quad (zero dndu),
    a smooth-shaded mesh (dndu / dndv from vertex normals, triangle.rs:405-437), an analytic sphere (Weingarten dndu / dndv)
    and a mirror; a constant bump map given as a number; displacement evaluated at the shifted points with the camera ray's
    differentials at the first hit and with the 0.0005 fallback afterwards."""
    b = base(res=res, spp=spp, depth=depth)
    if lens:
        b.camera_perspective(fov=40.0, lensradius=0.05, focaldistance=6.0)
    T = scenes
    s = 2.0
    wr = b.texture_scale(0.05, b.texture_wrinkled(octaves=5, roughness=0.5, to_world=T.transform_scale(4.0, 4.0, 4.0)))
    b.material_matte((0.6, 0.6, 0.6), bumpmap=wr)
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    chk = b.texture_checkerboard(0.03, 0.0, uscale=8.0, vscale=8.0)
    b.material_plastic(Kd=(0.7, 0.3, 0.2), Ks=(0.3, 0.3, 0.3), roughness=0.1, bumpmap=chk)
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    b.material_mirror(bumpmap=b.texture_scale(0.02, b.texture_dots(1.0, 0.0, uscale=6.0, vscale=6.0)))
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    b.material_matte((0.7, 0.2, 0.2), bumpmap=0.1)
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    b.area_light_source_diffuse(L=(10, 9, 8))
    h = 0.999 * s
    scenes._quad(b, (0.5, h, -0.5), (0.5, h, 0.5), (-0.5, h, 0.5), (-0.5, h, -0.5))
    b.no_area_light()
    fb = b.texture_scale(0.2, b.texture_fbm(octaves=6, to_world=T.transform_scale(5.0, 5.0, 5.0)))
    b.material_matte((0.3, 0.5, 0.8), sigma=30.0, bumpmap=fb)
    P, N, UV, idx = uv_sphere((-0.9, -1.2, 0.3), 0.75)
    b.shape_trianglemesh(P, idx, N=N, uv=UV)
    ts = T.transform_translate(0.9, -1.25, -0.3)
    b.material_plastic(Kd=(0.2, 0.6, 0.3), Ks=(0.4, 0.4, 0.4), roughness=0.05, bumpmap=b.texture_scale(0.1, b.texture_windy(to_world=T.transform_scale(6.0, 6.0, 6.0))))
    b.shape_sphere(radius=0.7, object_to_world=ts[0], world_to_object=ts[1])
    return b.build()


def test_image(w, h, channels=3, seed=3):
    """A deterministic image with structure at several scales (so MIP levels differ) in texture orientation."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 0.5 + 0.5 * np.sin(x * (2 * np.pi * 3 / w)) * np.cos(y * (2 * np.pi * 2 / h))
    stripes = ((x.astype(np.int32) // max(1, w // 16) + y.astype(np.int32) // max(1, h // 8)) % 2).astype(np.float32)
    noise = rng.random((h, w), dtype=np.float32)
    chans = [0.6 * base + 0.3 * stripes + 0.1 * noise, 0.3 * base + 0.5 * noise + 0.2 * stripes, 0.8 * stripes * base + 0.2 * noise]
    img = np.stack(chans[:channels], -1).astype(np.float32)
    return img


def scene_imagemaps(res=48, spp=8, depth=4, sampler="sobol", trilinear=False, lens=False):
    """Image-map textures (textures/imagemap.rs, core/texture/mipmap.rs) over caller-built pyramids: EWA (default) or trilinear
    lookups, repeat / clamp / black wrap modes, a non-square RGB image on the floor under a uv scale, a float image driving
    Matte sigma and a bump map, spherical mapping on an analytic sphere, a planar mapping on the back wall."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    if lens:
        b.camera_perspective(fov=40.0, lensradius=0.05, focaldistance=6.0)
    T = scenes
    s = 2.0
    rgb = b.image_pyramid(test_image(64, 32, 3))
    gray = b.image_pyramid(test_image(32, 32, 1, seed=7)[..., 0])
    small = b.image_pyramid(test_image(8, 16, 3, seed=11))
    b.material_matte(b.texture_imagemap(rgb, trilinear=trilinear, uscale=3.0, vscale=2.0, udelta=0.13, vdelta=0.4))
    scenes._quad(b, (s, -s, -s), (-s, -s, -s), (-s, -s, s), (s, -s, s))
    b.material_plastic(Kd=b.texture_imagemap(small, trilinear=trilinear, wrap="clamp", mapping="planar", v1=(0.4, 0.0, 0.0), v2=(0.0, 0.4, 0.0), udelta=0.5, vdelta=0.5),
                       Ks=(0.2, 0.2, 0.2), roughness=0.1)
    scenes._quad(b, (s, -s, s), (-s, -s, s), (-s, s, s), (s, s, s))
    b.material_matte(b.texture_scale((0.9, 0.6, 0.3), b.texture_imagemap(rgb, trilinear=trilinear, swrap="black", twrap="repeat", uscale=1.6, vscale=1.6, udelta=-0.3)),
                     sigma=b.texture_scale(60.0, b.texture_imagemap(gray, trilinear=trilinear, maxanisotropy=2.0)))
    scenes._quad(b, (-s, -s, s), (-s, -s, -s), (-s, s, -s), (-s, s, s))
    b.material_matte((0.7, 0.3, 0.3), bumpmap=b.texture_scale(0.08, b.texture_imagemap(gray, trilinear=trilinear, uscale=2.0, vscale=2.0)))
    scenes._quad(b, (s, -s, -s), (s, -s, s), (s, s, s), (s, s, -s))
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (s, s, -s), (s, s, s), (-s, s, s), (-s, s, -s))
    b.area_light_source_diffuse(L=(10, 9, 8))
    h = 0.999 * s
    scenes._quad(b, (0.5, h, -0.5), (0.5, h, 0.5), (-0.5, h, 0.5), (-0.5, h, -0.5))
    b.no_area_light()
    ts = T.transform_translate(0.3, -1.1, 0.2)
    b.material_uber(Kd=b.texture_imagemap(rgb, trilinear=trilinear, mapping="spherical", to_world=ts), Ks=(0.3, 0.3, 0.3), roughness=0.05)
    b.shape_sphere(radius=0.85, object_to_world=ts[0], world_to_object=ts[1])
    return b.build()


def scene_instances(split="sah", res=48, spp=8, depth=5, sampler="sobol"):
    """Object instancing (scene_context.rs:1327-1391, core/primitive/transformed_primitive.rs): an object of a smooth-shaded
    blob + an analytic glass sphere + a plastic quad instanced four times (translate, rotate + non-uniform scale, a mirrored
    = handedness-swapping transform, a tiny far copy), a one-triangle object (wrapped without an accelerator), instances
    interleaved with world triangles and spheres in the primitive order, a light inside an object (ignored)."""
    b = base(res=res, spp=spp, depth=depth)
    if sampler == "halton":
        b.sampler_halton(spp)
    b.accelerator_bvh(splitmethod=split)
    T = scenes
    b.object_begin("thing")
    b.material_matte((0.3, 0.5, 0.8), sigma=20.0)
    P, N, UV, idx = uv_sphere((0.0, 0.0, 0.0), 0.45, nt=6, nphi=8)
    b.shape_trianglemesh(P, idx, N=N, uv=UV)
    b.material_glass()
    ts = T.transform_translate(0.0, 0.75, 0.0)
    b.shape_sphere(radius=0.3, object_to_world=ts[0], world_to_object=ts[1])
    b.material_plastic(Kd=b.texture_checkerboard((0.9, 0.8, 0.2), (0.2, 0.2, 0.2), uscale=4.0, vscale=4.0), Ks=(0.3, 0.3, 0.3), roughness=0.1)
    b.area_light_source_diffuse(L=(50, 50, 50))             # "Area lights not supported with object instancing": ignored
    scenes._quad(b, (-0.6, -0.46, -0.6), (0.6, -0.46, -0.6), (0.6, -0.46, 0.6), (-0.6, -0.46, 0.6))
    b.no_area_light()
    b.object_end()
    b.object_begin("shard")
    b.material_mirror()
    b.shape_trianglemesh([(-0.5, 0.0, 0.0), (0.5, 0.0, 0.1), (0.0, 0.9, 0.05)], [0, 1, 2])
    b.object_end()
    # an instance ahead of every world triangle
    b.object_instance("thing", T.transform_translate(-1.1, -1.5, 0.5))
    room(b)
    b.object_instance("thing", T.transform_mul(T.transform_translate(1.0, -1.3, 0.2), T.transform_mul(T.transform_rotate_x(35.0), T.transform_scale(1.2, 0.7, 1.0))))
    b.material_matte((0.6, 0.6, 0.6))
    ts2 = T.transform_translate(0.0, 1.2, 1.2)
    b.shape_sphere(radius=0.25, object_to_world=ts2[0], world_to_object=ts2[1])
    b.object_instance("thing", T.transform_mul(T.transform_translate(-0.2, 0.3, 1.4), T.transform_scale(-0.8, 0.8, 0.8)))
    b.object_instance("shard", T.transform_translate(0.2, -1.9, -0.9))
    b.object_instance("shard", T.transform_mul(T.transform_translate(-1.5, -0.5, 1.6), T.transform_rotate_x(-20.0)))
    b.object_instance("thing", T.transform_mul(T.transform_translate(1.6, 1.5, 1.7), T.transform_scale(0.2, 0.2, 0.2)))
    return b.build()


def scene_hlbvh_cluster():
    """More than maxnodeprims primitives in one Morton cell (a tight cluster, 40 exact duplicates) next to a sparse shell:
    emit_lbvh runs out of code bits and falls back to centroid medians (hlbvh.rs:102-157, :183-192)."""
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -5), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0)
    b.film(xresolution=16, yresolution=16); b.pixel_filter_box(); b.sampler_sobol(1); b.integrator_path(maxdepth=1)
    b.accelerator_bvh("hlbvh", 4)
    b.material_matte((0.5, 0.5, 0.5))
    rng = np.random.default_rng(2)
    for k in range(60):
        c = np.array([0.3, 0.2, 0.1]) + rng.random(3) * 1e-5
        b.shape_trianglemesh([tuple(c), tuple(c + [1e-6, 0, 0]), tuple(c + [0, 1e-6, 0])], [0, 1, 2])
    for k in range(40):
        b.shape_trianglemesh([(-0.5, -0.5, 0.5), (-0.4, -0.5, 0.5), (-0.5, -0.4, 0.5)], [0, 1, 2])
    for k in range(200):
        c = rng.random(3) * 4 - 2
        b.shape_trianglemesh([tuple(c), tuple(c + [0.05, 0, 0]), tuple(c + [0, 0.05, 0.02])], [0, 1, 2])
    return b.build()


def scene_ao(sampler="sobol", cossample=True, nsamples=16, spp=4, res=40, kind="boxes"):
    """Integrator "ao": an enclosure open at the top (escaping occlusion rays) with random matte clutter; kind "spheres" adds an
    analytic sphere with a partial sweep, "instances" the instanced object of scene_instances."""
    if kind == "instances":
        sd = scene_instances(split="sah", res=res, spp=spp, sampler=sampler)
        sd.desc.integrator, sd.desc.ao_samples, sd.desc.ao_cos_sample = pkg.capi.PT_INTEGRATOR_AO, nsamples, int(cossample)
        return sd
    b = scenes.SceneBuilder()
    b.look_at((0, 0.4, -3.4), (0, -0.2, 0), (0, 1, 0))
    b.camera_perspective(fov=42.0)
    b.film(xresolution=res, yresolution=res - 8)
    b.pixel_filter_box()
    b.sampler_sobol(spp) if sampler == "sobol" else b.sampler_halton(spp)
    b.integrator_ao(nsamples=nsamples, cossample=cossample)
    b.accelerator_bvh("sah", 4)
    b.material_matte((0.5, 0.5, 0.5))
    b.shape_trianglemesh([(1, -1, -1), (-1, -1, -1), (-1, -1, 1), (1, -1, 1)], [0, 1, 2, 0, 2, 3])       # floor
    b.shape_trianglemesh([(1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, 1)], [0, 1, 2, 0, 2, 3])           # back wall
    b.shape_trianglemesh([(-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)], [0, 1, 2, 0, 2, 3],
                         N=[(1, 0, 0)] * 4, uv=[(0, 0), (1, 0), (1, 1), (0, 1)])                            # side wall with normals / uv
    rng = np.random.default_rng(17)
    c = rng.uniform(-0.8, 0.8, (300, 3)).astype(np.float32)
    c[:, 1] = rng.uniform(-1.0, -0.2, 300)
    off = rng.uniform(-0.12, 0.12, (300, 3, 3)).astype(np.float32)
    b.material_plastic(Kd=(0.3, 0.3, 0.3), Ks=(0.2, 0.2, 0.2), roughness=0.1)
    b.shape_trianglemesh_fast((c[:, None, :] + off).reshape(-1, 3), np.arange(900), twosided=True)
    if kind == "spheres":
        T = scenes
        t = T.transform_mul(T.transform_translate(0.3, -0.5, 0.1), T.transform_rotate_x(30.0))
        b.shape_sphere(radius=0.4, zmin=-0.3, zmax=0.35, phimax=300.0, object_to_world=t[0], world_to_object=t[1])
    return b.build()


# ---- committed golden fixtures of the other SamplerIntegrators (tools/make_golden.py writes them from the oracle; tests/test_oracle_kat.py
# holds the oracle to them, tests/test_gpu_features.py the device): name -> (scene, tile of per-sample radiance)
def _golden_integrator(kind):
    import importlib
    capi = importlib.import_module("pbrt-r3_amd").capi
    if kind == "directlighting":         # strategy "all", three samples per light: the 2-D sample arrays (and quirk Q23) are in play
        sd = scene_materials_render(["glass", "mirror", "plastic"], spp=4)
        sd.desc.integrator, sd.desc.direct_strategy, sd.desc.max_depth = capi.PT_INTEGRATOR_DIRECTLIGHTING, capi.PT_DIRECT_ALL, 4
        for i in range(sd.desc.n_area_lights):
            sd.desc.area_lights[i].n_samples = 3
    elif kind == "whitted":              # glass + mirror trees to depth 4
        sd = scene_materials_render(["glass", "mirror", "metal"], spp=4)
        sd.desc.integrator, sd.desc.max_depth = capi.PT_INTEGRATOR_WHITTED, 4
    else:                                # ao: 16 cosine-distributed occlusion rays per camera sample
        from helpers import scenes
        sd = scenes.cornell_box(res=32, spp=4)
        sd.desc.integrator, sd.desc.ao_samples, sd.desc.ao_cos_sample = capi.PT_INTEGRATOR_AO, 16, 1
    return sd


GOLDEN_INTEGRATORS = {
    "directlighting_all_ns3_40x40_4spp": lambda: _golden_integrator("directlighting"),
    "whitted_depth4_40x40_4spp": lambda: _golden_integrator("whitted"),
    "ao_16cos_cornell_32x32_4spp": lambda: _golden_integrator("ao"),
}


def golden_tile(info):
    """The 12x12 tile whose per-sample radiance a fixture holds: the middle of the film (objects, their shadows and reflections)."""
    sb = list(info.sample_bounds)
    cx, cy = (sb[0] + sb[2]) // 2, (sb[1] + sb[3]) // 2
    return (cx - 6, cy - 6, cx + 6, cy + 6)


def scene_instance_swarm(split="sah", maxnodeprims=4, n_inst=36, res=48, spp=8, depth=6):
    """What the fixed instancing scene does not reach: world leaves that hold SEVERAL instances, with triangles before, between and behind them (a ray that
    enters an instance leaves the rest of its leaf on its stack and comes back to it), overlapping instances of a tree deep enough to matter
    (3 000 triangles, some of them glass), one-triangle objects among them (wrapped without an accelerator: tested on the spot), any-hit rays
    that end inside an object."""
    b = base(res=res, spp=spp, depth=depth)
    b.accelerator_bvh(splitmethod=split, maxnodeprims=maxnodeprims)
    T = scenes
    rng = np.random.default_rng(41)
    n = 3000
    c = rng.uniform(-0.5, 0.5, (n, 1, 3)).astype(np.float32)
    verts = (c + rng.uniform(-0.06, 0.06, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    b.object_begin("cloud")
    b.material_matte((0.6, 0.45, 0.3))
    b.shape_trianglemesh(verts[:6000], np.arange(6000))
    b.material_glass()
    b.shape_trianglemesh(verts[6000:], np.arange(3 * n - 6000))
    b.object_end()
    b.object_begin("shard")
    b.material_mirror()
    b.shape_trianglemesh([(-0.3, 0.0, 0.0), (0.3, 0.0, 0.1), (0.0, 0.5, 0.05)], [0, 1, 2])
    b.object_end()
    room(b)
    b.material_matte((0.4, 0.4, 0.7))
    for k in range(n_inst):
        p = rng.uniform(-1.2, 1.2, 3)
        sc = float(rng.uniform(0.4, 1.1))
        m = T.transform_mul(T.transform_translate(float(p[0]), float(p[1]), float(p[2])), T.transform_mul(T.transform_rotate_x(float(rng.uniform(-60, 60))), T.transform_scale(sc, sc * (-1.0 if k % 7 == 3 else 1.0), sc)))
        b.object_instance("shard" if k % 5 == 4 else "cloud", m)
        if k % 3 == 0:          # a world triangle between the instances (primitive order = creation order)
            q = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
            b.shape_trianglemesh([tuple(q), tuple(q + np.float32([0.3, 0.0, 0.05])), tuple(q + np.float32([0.0, 0.3, 0.1]))], [0, 1, 2])
    return b.build()
