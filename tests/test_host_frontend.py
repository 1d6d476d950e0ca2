"""The .pbrt front end (include/pbrtgpu_host.h): parser known-answer tests restated from the
reference's own parser tests, and scene-context parity (a .pbrt Cornell box must flatten to a scene
that renders bit-identically to the programmatic one).  No GPU needed."""
import os

import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, pkg, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
capi = pkg.capi


def test_host_symbols_exported():
    lib = capi.load_library()
    for s in capi.HOST_SYMBOLS:
        assert hasattr(lib, s), s


def test_parse_ops_001():
    """src/core/parser/parser.rs:757-770."""
    log = capi.parse_to_log('Integrator "path" \n\n        WorldBegin')
    assert log[0].startswith('Integrator "path"') and log[1] == "WorldBegin"


def test_parse_ops_002():
    """parser.rs:771-823: argument counts of Translate / Rotate / LookAt / Transform / Texture."""
    log = capi.parse_to_log('''
            Translate 0 0 -140
            Rotate 0 1 2 3
            LookAt 0 1 2 3 4 5 6 7 8

            Transform [0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15]
            Texture "a" "b" "c"
        ''')
    assert log[0] == "Translate 0 0 -140"
    assert log[1] == "Rotate 0 1 2 3"
    assert log[2] == "LookAt 0 1 2 3 4 5 6 7 8"
    assert log[3] == "Transform [ 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 ]"
    assert log[4] == 'Texture "a" "b" "c"'


def test_parse_ops_003():
    """parser.rs:825-852 (the killeroo-simple Film line)."""
    log = capi.parse_to_log('''
        Film "image" "integer xresolution" [700] "integer yresolution" [700]
            "string filename" "killeroo-simple.exr"
        ''')
    assert log == ['Film "image" "integer xresolution" [ 700 ] "integer yresolution" [ 700 ] "string filename" [ "killeroo-simple.exr" ]']


def test_parse_ops_004():
    """parser.rs:854-910: comments, colour / point / float / integer arrays."""
    log = capi.parse_to_log('''
        AttributeBegin # A
            Material "matte" "color Kd" [.5 .5 .8]
            Translate 0 0 -140
            Shape "trianglemesh" "point P" [ -1000 -1000 0 1000 -1000 0 1000 1000 0 -1000 1000 0 ]
                "float uv" [ 0 0 5 0 5 5 0 5 ]
                "integer indices" [ 0 1 2 2 3 0]
            Shape "trianglemesh" "point P" [ -400 -1000 -1000   -400 1000 -1000   -400 1000 1000 -400 -1000 1000 ]
                "float uv" [ 0 0 5 0 5 5 0 5 ]
                "integer indices" [ 0 1 2 2 3 0]
        AttributeEnd
        ''')
    assert [l.split()[0] for l in log] == ["AttributeBegin", "Material", "Translate", "Shape", "Shape", "AttributeEnd"]
    assert log[1] == 'Material "matte" "rgb Kd" [ 0.5 0.5 0.800000012 ]'
    assert '"point P" [ -1000 -1000 0 1000 -1000 0 1000 1000 0 -1000 1000 0 ]' in log[3]
    assert '"float uv" [ 0 0 5 0 5 5 0 5 ]' in log[3] and '"integer indices" [ 0 1 2 2 3 0 ]' in log[3]


def test_syntax_errors_are_reported():
    for bad in ('Translate 1 2', 'Shape "trianglemesh" "point P" [ 1 2 3', 'Frobnicate 1', 'Film "image" "quux x" [1]'):
        with pytest.raises(capi.PtError) as e:
            capi.parse_to_log(bad)
        assert e.value.status == 1


@pytest.mark.parametrize("text,needle", [
    ('Sampler "sobol"\nWorldBegin\nShape "cylinder" "float radius" 1\nWorldEnd', "cylinder"),
    ('Sampler "stratified"\nWorldBegin\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd', "stratified"),
    ('Sampler "sobol"\nWorldBegin\nLightSource "point"\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd', "LightSource"),
    ('Sampler "sobol"\nWorldBegin\nMaterial "disney"\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd', "disney"),
])
def test_out_of_scope_features_fail_loudly(text, needle):
    """Nothing outside the accelerated path is silently approximated."""
    with pytest.raises(capi.PtError) as e:
        capi.ParsedScene(text=text)
    assert e.value.status == 4 and needle in str(e.value)


def test_cornell_pbrt_equals_programmatic_scene(oracle):
    """tests/scenes/cornell.pbrt through the C++ parser + scene context renders (oracle) to exactly the
    film of scenes.cornell_box(): transforms, camera, named materials, Include, area lights all agree."""
    ps = capi.ParsedScene(filename=os.path.join(ROOT, "tests", "scenes", "cornell.pbrt"))
    assert ps.output_filename == "cornell.exr"
    d = ps.desc
    ref = scenes.cornell_box(res=64, spp=16)
    assert (d.n_triangles, d.xres, d.yres, d.spp, d.max_depth) == (36, 64, 64, 16, 5)
    assert np.array_equal(bits(list(d.camera_to_world)), bits(list(ref.desc.camera_to_world)))
    assert np.array_equal(bits(list(d.screen_window)), bits(list(ref.desc.screen_window)))
    assert np.array_equal(bits(np.ctypeslib.as_array(d.P, (d.n_vertices * 3,))), bits(ref.buffers["P"].reshape(-1)))
    a = oracle.scene(ps)
    b = oracle.scene(ref)
    xa, ca, _ = a.render(threads=4)
    xb, cb, _ = b.render(threads=4)
    assert np.array_equal(bits(xa), bits(xb))
    assert ca["regular_rays"] == cb["regular_rays"] and ca["shadow_rays"] == cb["shadow_rays"]
    a.close(); b.close()


def test_transform_stack_and_overrides(oracle):
    """CTM products (Translate/Scale/Rotate/ConcatTransform), TransformBegin/End, CoordSysTransform,
    ReverseOrientation, per-shape material overrides and --pixelsamples."""
    text = '''
    LookAt 0 0 -5  0 0 0  0 1 0
    Camera "perspective" "float fov" 45
    Film "image" "integer xresolution" 24 "integer yresolution" 16 "float cropwindow" [0.1 0.9 0 1]
    Sampler "sobol" "integer pixelsamples" 4
    PixelFilter "gaussian" "float xwidth" 1.5 "float ywidth" 1.5
    WorldBegin
      CoordinateSystem "base"
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [4 4 4] "bool twosided" "true"
        Translate 0 1.9 0
        Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1  1 0 -1  1 0 1  -1 0 1]
      AttributeEnd
      AttributeBegin
        Material "matte" "rgb Kd" [0.2 0.2 0.2]
        Scale 2 1 2
        Rotate 30 0 1 0
        TransformBegin
          ConcatTransform [1 0 0 0  0 1 0 0  0 0 1 0  0 -2 0 1]
          Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-1 0 -1  -1 0 1  1 0 1  1 0 -1] "rgb Kd" [0.8 0.3 0.1] "float sigma" 20
        TransformEnd
        ReverseOrientation
        Scale -1 1 1
        Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 -1 0.5  1 0 0.5  -1 0 0.5] "bool twosided" "false"
        CoordSysTransform "base"
        Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 -1.5 2  1.5 0.5 2  -1.5 0.5 2]
      AttributeEnd
    WorldEnd
    '''
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    assert d.n_triangles == 6 and d.n_meshes == 4
    m = [d.meshes[i] for i in range(4)]
    assert m[0].area_light == 0 and m[1].area_light == -1
    # texture_params.rs:36-83: the material's Kd wins over the shape's; sigma is only on the shape, so it is used
    assert d.materials[m[1].material].sigma == 20.0 and abs(d.materials[m[1].material].kd[0] - 0.2) < 1e-6
    assert abs(d.materials[m[2].material].kd[0] - 0.2) < 1e-6
    f2 = m[2].flags
    assert (f2 & capi.PT_MESH_REVERSE_ORIENTATION) and (f2 & capi.PT_MESH_SWAPS_HANDEDNESS) and not (f2 & capi.PT_MESH_TWO_SIDED)
    assert not (m[3].flags & capi.PT_MESH_SWAPS_HANDEDNESS)                 # CoordSysTransform restored the base CTM
    P = np.ctypeslib.as_array(d.P, (d.n_vertices * 3,)).reshape(-1, 3)
    assert np.allclose(P[:4, 1], 1.9)                                          # Translate
    assert np.allclose(P[4:8, 1], -2.0) and np.abs(P[4:8, 0]).max() > 1.5     # ConcatTransform after Scale * Rotate
    assert np.allclose(P[-3:], [[0, -1.5, 2], [1.5, 0.5, 2], [-1.5, 0.5, 2]])
    sc = oracle.scene(ps)
    x, _, _ = sc.render(threads=2)
    assert x.shape[:2] == (16, 20) and np.isfinite(x).all() and x[..., :3].max() > 0
    ps.set_pixelsamples(8)
    sc2 = oracle.scene(ps)
    assert sc2.info.spp == 8
    sc.close(); sc2.close()


def test_sphere_shape_from_pbrt_text(oracle):
    """Shape "sphere" (shapes/sphere.rs:401-420): CTM and its stored inverse, parameters, ReverseOrientation, material and
    area light per sphere, and its place in the primitive order -- against the programmatic scene, bit for bit."""
    text = '''
    LookAt 0 0 -6.5  0 0 0  0 1 0
    Camera "perspective" "float fov" 40
    Film "image" "integer xresolution" 24 "integer yresolution" 24
    Sampler "sobol" "integer pixelsamples" 4
    Integrator "path" "integer maxdepth" 4
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [30 26 20]
        Translate -1.2 1.3 0.4
        Rotate 30 1 0 0
        Scale 0.25 0.15 0.2
        Shape "sphere"
      AttributeEnd
      Material "matte" "rgb Kd" [0.7 0.7 0.7]
      Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [2 -2 -2  -2 -2 -2  -2 -2 2  2 -2 2]
      AttributeBegin
        Material "mirror"
        Translate 0.2 0.1 1.1
        Shape "sphere" "float radius" 0.55 "float zmin" -0.3 "float zmax" 0.45 "float phimax" 250
      AttributeEnd
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [2 -2 2  -2 -2 2  0 2 2]
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [6 8 12] "bool twosided" "true"
        ReverseOrientation
        Translate 1.3 0.9 0.8
        Shape "sphere" "float radius" 0.2
      AttributeEnd
    WorldEnd
    '''
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    assert (d.n_triangles, d.n_spheres) == (3, 3)
    sp = [d.spheres[i] for i in range(3)]
    assert [s.before_triangle for s in sp] == [0, 2, 3]
    assert [s.flags for s in sp] == [0, 0, capi.PT_SPHERE_REVERSE_ORIENTATION]
    assert (sp[0].radius, sp[0].zmin, sp[0].zmax, sp[0].phimax) == (1.0, -1.0, 1.0, 360.0)
    assert abs(sp[1].zmin + 0.3) < 1e-7 and sp[1].phimax == 250.0
    assert sp[0].area_light == 0 and sp[1].area_light == -1 and sp[2].area_light == 1
    assert d.materials[sp[1].material].type == capi.PT_MATERIAL_MIRROR
    T = scenes
    t0 = T.transform_mul(T.transform_mul(T.transform_translate(-1.2, 1.3, 0.4), T.transform_rotate_x(30.0)), T.transform_scale(0.25, 0.15, 0.2))
    assert np.array_equal(bits(list(sp[0].object_to_world)), bits(t0[0]))
    assert np.array_equal(bits(list(sp[0].world_to_object)), bits(t0[1]))
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -6.5), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0); b.film(xresolution=24, yresolution=24)
    b.pixel_filter_box(); b.sampler_sobol(4); b.integrator_path(maxdepth=4)
    b.area_light_source_diffuse(L=(30, 26, 20))
    b.shape_sphere(object_to_world=t0[0], world_to_object=t0[1])
    b.no_area_light()
    b.material_matte((0.7, 0.7, 0.7))
    b.shape_trianglemesh([(2, -2, -2), (-2, -2, -2), (-2, -2, 2), (2, -2, 2)], [0, 1, 2, 0, 2, 3])
    b.material_mirror()
    t1 = T.transform_translate(0.2, 0.1, 1.1)
    b.shape_sphere(radius=0.55, zmin=-0.3, zmax=0.45, phimax=250.0, object_to_world=t1[0], world_to_object=t1[1])
    b.material_matte((0.7, 0.7, 0.7))
    b.shape_trianglemesh([(2, -2, 2), (-2, -2, 2), (0, 2, 2)], [0, 1, 2])
    b.area_light_source_diffuse(L=(6, 8, 12), twosided=True)
    b.reverse_orientation = True
    t2 = T.transform_translate(1.3, 0.9, 0.8)
    b.shape_sphere(radius=0.2, object_to_world=t2[0], world_to_object=t2[1])
    ref = b.build()
    a, r = oracle.scene(ps), oracle.scene(ref)
    xa, ca, _ = a.render(threads=2)
    xr, cr, _ = r.render(threads=2)
    assert np.array_equal(bits(xa), bits(xr)) and ca == cr and xa[..., :3].max() > 0
    a.close(); r.close()


def test_noise_texture_directives(oracle):
    """Texture "dots" / "fbm" / "wrinkled" / "windy" / "marble" (create_texture.rs:44-110): parameters, defaults and the CTM they
    record, against the SceneBuilder's nodes byte for byte; float "marble" does not exist in the reference."""
    text = '''
    Sampler "sobol" "integer pixelsamples" 1
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [1 1 1]
        Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 2 0 1 2 0 0 2 1]
      AttributeEnd
      Texture "d" "spectrum" "dots" "rgb tex1" [0.8 0.8 0.7] "rgb tex2" [0.7 0.1 0.1] "float uscale" 5 "float vscale" 5
      AttributeBegin
        Scale 2 2 2
        Texture "w" "float" "wrinkled" "integer octaves" 6 "float roughness" 0.6
        Texture "f" "float" "fbm"
        Texture "wi" "spectrum" "windy"
        Texture "m" "spectrum" "marble" "float scale" 4 "float variation" 0.4
        Texture "fm" "float" "marble"
      AttributeEnd
      Material "matte" "texture Kd" "m" "texture sigma" "w"
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]
      Material "matte" "texture Kd" "d" "texture sigma" "f"
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 1 1 0 1 0 1 1]
      Material "matte" "texture Kd" "wi" "texture sigma" "fm"
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 2 1 0 2 0 1 2]
    WorldEnd
    '''
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    assert d.n_textures == 5 and 'Float texture "marble" unknown' in ps.warnings
    b = scenes.SceneBuilder()
    t = scenes.transform_scale(2.0, 2.0, 2.0)
    b.texture_dots((0.8, 0.8, 0.7), (0.7, 0.1, 0.1), uscale=5.0, vscale=5.0)
    b.texture_wrinkled(octaves=6, roughness=0.6, to_world=t)
    b.texture_fbm(to_world=t)
    b.texture_windy(to_world=t)
    b.texture_marble(scale=4.0, variation=0.4, to_world=t)
    for i in range(5):
        assert bytes(d.textures[i]) == bytes(b.textures[i]), i
    m = [d.materials[d.meshes[i].material] for i in (1, 2, 3)]
    assert (m[0].tex_kd, m[0].tex_sigma) == (5, 2) and (m[1].tex_kd, m[1].tex_sigma) == (1, 3) and (m[2].tex_kd, m[2].tex_sigma) == (4, 0)
    sc = oracle.scene(ps)
    x, _, _ = sc.render(threads=2)
    assert np.isfinite(x).all()
    sc.close()


def test_bumpmap_parameter(oracle):
    """"bumpmap" (get_float_texture_or_null, texture_params.rs:107-116): a named float texture, or a number (a constant texture)."""
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]'
    text = '''
    Sampler "sobol" "integer pixelsamples" 1
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [1 1 1]
        %(tri)s
      AttributeEnd
      Texture "w" "float" "wrinkled"
      Texture "k" "float" "constant" "float value" 0.25
      Material "matte" "texture bumpmap" "w"
      %(tri)s
      Material "plastic" "float bumpmap" 0.1
      %(tri)s
      Material "mirror"
      %(tri)s "texture bumpmap" "k"
    WorldEnd
    ''' % {"tri": tri}
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    m = [d.materials[d.meshes[i].material] for i in (1, 2, 3)]
    assert d.n_textures == 3 and d.textures[0].type == capi.PT_TEX_WRINKLED
    assert m[0].tex_bump == 1
    assert m[1].tex_bump == 2 and d.textures[1].type == capi.PT_TEX_CONSTANT and abs(d.textures[1].value[0][0] - 0.1) < 1e-7
    assert m[2].tex_bump == 3 and d.textures[2].value[0][0] == 0.25
    sc = oracle.scene(ps)
    x, _, _ = sc.render(threads=2)
    assert np.isfinite(x).all()
    sc.close()


def test_object_instancing_from_pbrt_text(oracle):
    """ObjectBegin / ObjectEnd / ObjectInstance (scene_context.rs:1327-1391): object membership of meshes and spheres, the
    instance CTMs with their stored inverses, positions in the world primitive order, the dropped area light inside an object,
    ObjectInstance ignored inside a definition; rendered against the SceneBuilder scene bit for bit."""
    text = '''
    LookAt 0 0 -6.5  0 0 0  0 1 0
    Camera "perspective" "float fov" 40
    Film "image" "integer xresolution" 32 "integer yresolution" 32
    Sampler "sobol" "integer pixelsamples" 4
    Integrator "path" "integer maxdepth" 4
    WorldBegin
      ObjectBegin "thing"
        Material "matte" "rgb Kd" [0.3 0.5 0.8]
        Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [-0.5 -0.4 -0.5  0.5 -0.4 -0.5  0.5 -0.4 0.5  -0.5 -0.4 0.5]
        Material "glass"
        AttributeBegin
          Translate 0 0.2 0
          AreaLightSource "diffuse" "rgb L" [50 50 50]
          Shape "sphere" "float radius" 0.35
        AttributeEnd
        ObjectInstance "thing"
      ObjectEnd
      ObjectBegin "shard"
        Material "mirror"
        Shape "trianglemesh" "integer indices" [0 1 2] "point P" [-0.5 0 0  0.5 0 0.1  0 0.9 0.05]
      ObjectEnd
      AttributeBegin
        Translate -1.1 -1.5 0.5
        ObjectInstance "thing"
      AttributeEnd
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [10 9 8]
        Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [0.5 1.99 -0.5  0.5 1.99 0.5  -0.5 1.99 0.5  -0.5 1.99 -0.5]
      AttributeEnd
      Material "matte" "rgb Kd" [0.7 0.7 0.7]
      Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [2 -2 -2  -2 -2 -2  -2 -2 2  2 -2 2]
      AttributeBegin
        Translate 1.0 -1.3 0.2
        Rotate 35 1 0 0
        Scale 1.2 0.7 1.0
        ObjectInstance "thing"
        ObjectInstance "nothing"
      AttributeEnd
      Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [2 -2 2  -2 -2 2  -2 2 2  2 2 2]
      AttributeBegin
        Translate 0.2 -1.9 -0.9
        ObjectInstance "shard"
      AttributeEnd
    WorldEnd
    '''
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    assert d.n_instances == 3 and d.n_spheres == 1 and d.n_triangles == 9
    assert [d.meshes[i].object for i in range(d.n_meshes)] == [1, 2, 0, 0, 0]
    assert d.spheres[0].object == 1 and d.spheres[0].area_light == -1
    assert [(d.instances[i].object, d.instances[i].before_triangle) for i in range(3)] == [(0, 3), (0, 7), (1, 9)]
    T = scenes
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -6.5), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0); b.film(xresolution=32, yresolution=32)
    b.pixel_filter_box(); b.sampler_sobol(4); b.integrator_path(maxdepth=4)
    b.object_begin("thing")
    b.material_matte((0.3, 0.5, 0.8))
    b.shape_trianglemesh([(-0.5, -0.4, -0.5), (0.5, -0.4, -0.5), (0.5, -0.4, 0.5), (-0.5, -0.4, 0.5)], [0, 1, 2, 0, 2, 3])
    b.material_glass()
    ts = T.transform_translate(0.0, 0.2, 0.0)
    b.shape_sphere(radius=0.35, object_to_world=ts[0], world_to_object=ts[1])
    b.object_end()
    b.object_begin("shard")
    b.material_mirror()
    b.shape_trianglemesh([(-0.5, 0.0, 0.0), (0.5, 0.0, 0.1), (0.0, 0.9, 0.05)], [0, 1, 2])
    b.object_end()
    b.object_instance("thing", T.transform_translate(-1.1, -1.5, 0.5))
    b.area_light_source_diffuse(L=(10, 9, 8))
    b.shape_trianglemesh([(0.5, 1.99, -0.5), (0.5, 1.99, 0.5), (-0.5, 1.99, 0.5), (-0.5, 1.99, -0.5)], [0, 1, 2, 0, 2, 3])
    b.no_area_light()
    b.material_matte((0.7, 0.7, 0.7))
    b.shape_trianglemesh([(2, -2, -2), (-2, -2, -2), (-2, -2, 2), (2, -2, 2)], [0, 1, 2, 0, 2, 3])
    b.object_instance("thing", T.transform_mul(T.transform_mul(T.transform_translate(1.0, -1.3, 0.2), T.transform_rotate_x(35.0)), T.transform_scale(1.2, 0.7, 1.0)))
    b.shape_trianglemesh([(2, -2, 2), (-2, -2, 2), (-2, 2, 2), (2, 2, 2)], [0, 1, 2, 0, 2, 3])
    b.object_instance("shard", T.transform_translate(0.2, -1.9, -0.9))
    ref = b.build()
    for i in range(3):
        assert np.array_equal(bits(list(d.instances[i].instance_to_world)), bits(list(ref.desc.instances[i].instance_to_world))), i
        assert np.array_equal(bits(list(d.instances[i].world_to_instance)), bits(list(ref.desc.instances[i].world_to_instance))), i
    a, r = oracle.scene(ps), oracle.scene(ref)
    assert a.info.n_lights == 2
    xa, ca, _ = a.render(threads=2)
    xr, cr, _ = r.render(threads=2)
    assert np.array_equal(bits(xa), bits(xr)) and ca == cr and xa[..., :3].max() > 0
    a.close(); r.close()


def test_materials_from_pbrt_text():
    """Material / MakeNamedMaterial for every supported type: parameters, defaults (create_*_material) and
    TextureParams' precedence (core/param_set/texture_params.rs:36-83: constant values come from the material
    first, from the shape only when the material has none -- the reverse of pbrt-v3)."""
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]'
    text = '''
    Sampler "sobol" "integer pixelsamples" 1
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [1 1 1]
        %(tri)s
      AttributeEnd
      Material "plastic"
      %(tri)s
      Material "plastic" "rgb Kd" [0.1 0.2 0.3] "rgb Ks" [0.4 0.5 0.6] "float roughness" 0.3 "bool remaproughness" "false"
      %(tri)s
      Material "mirror"
      %(tri)s "rgb Kr" [0.5 0.6 0.7]
      Material "glass" "float index" 1.33 "float uroughness" 0.05
      %(tri)s
      Material "glass" "float eta" 1.7 "float index" 1.2
      %(tri)s
      Material "metal" "rgb eta" [0.2 0.9 1.1] "rgb k" [3.9 2.4 2.1] "float vroughness" 0.2
      %(tri)s
      Material "plastic" "rgb Kd" [0.1 0.2 0.3] "float roughness" 0.3
      %(tri)s "rgb Kd" [0.9 0.9 0.9] "float roughness" 0.7 "rgb Ks" [0.6 0.6 0.6]
      MakeNamedMaterial "u" "string type" "uber" "rgb Kt" [0.1 0.1 0.1] "rgb opacity" [0.5 0.5 0.5] "float roughness" 0.25
      NamedMaterial "u"
      %(tri)s
      Material "substrate" "float uroughness" 0.3
      %(tri)s
    WorldEnd
    ''' % {"tri": tri}
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    mats = [d.materials[d.meshes[i].material] for i in range(d.n_meshes)]
    T = capi
    m = mats[1]
    assert m.type == T.PT_MATERIAL_PLASTIC and list(m.kd) == [0.25] * 3 and list(m.ks) == [0.25] * 3
    assert abs(m.roughness - 0.1) < 1e-7 and m.remap_roughness == 1
    m = mats[2]
    assert np.allclose(list(m.kd), [0.1, 0.2, 0.3]) and np.allclose(list(m.ks), [0.4, 0.5, 0.6]) and abs(m.roughness - 0.3) < 1e-7 and m.remap_roughness == 0
    assert mats[3].type == T.PT_MATERIAL_MIRROR and np.allclose(list(mats[3].kr), [0.5, 0.6, 0.7])       # the material gives no Kr: the shape's is used
    m = mats[4]
    assert m.type == T.PT_MATERIAL_GLASS and abs(m.eta - 1.33) < 1e-6 and abs(m.uroughness - 0.05) < 1e-7 and m.vroughness == 0.0
    assert list(m.kr) == [1.0] * 3 and list(m.kt) == [1.0] * 3
    assert abs(mats[5].eta - 1.7) < 1e-6                                                                   # "eta" before "index"
    m = mats[6]
    assert m.type == T.PT_MATERIAL_METAL and np.allclose(list(m.metal_k), [3.9, 2.4, 2.1]) and abs(m.roughness - 0.01) < 1e-8
    assert m.uroughness == T.PT_ROUGHNESS_UNSET and abs(m.vroughness - 0.2) < 1e-7
    m = mats[7]        # material values first, shape values only for what the material leaves out
    assert np.allclose(list(m.kd), [0.1, 0.2, 0.3]) and abs(m.roughness - 0.3) < 1e-7 and np.allclose(list(m.ks), [0.6] * 3)
    m = mats[8]
    assert m.type == T.PT_MATERIAL_UBER and np.allclose(list(m.opacity), [0.5] * 3) and np.allclose(list(m.kt), [0.1] * 3)
    assert list(m.kr) == [0.0] * 3 and abs(m.roughness - 0.25) < 1e-7 and abs(m.eta - 1.5) < 1e-7
    m = mats[9]
    assert m.type == T.PT_MATERIAL_SUBSTRATE and abs(m.uroughness - 0.3) < 1e-7 and abs(m.vroughness - 0.1) < 1e-7 and list(m.kd) == [0.5] * 3


@pytest.mark.parametrize("name,params,build", [
    ("box", "", lambda b: b.pixel_filter_box()),
    ("gaussian", '"float xwidth" 2.5 "float alpha" 1.5', lambda b: b.pixel_filter_gaussian(2.5, 2.0, 1.5)),
    ("mitchell", '"float ywidth" 3 "float B" 0.2 "float C" 0.4', lambda b: b.pixel_filter_mitchell(2.0, 3.0, 0.2, 0.4)),
    ("triangle", '"float xwidth" 1.5', lambda b: b.pixel_filter_triangle(1.5, 2.0)),
    ("sinc", '"float xwidth" 3 "float tau" 2', lambda b: b.pixel_filter_sinc(3.0, 4.0, 2.0)),
])
def test_filter_tables_match_scene_builder(name, params, build):
    """Film::new's 16x16 filter table (film.rs:102-120) from the C++ front end == the Python restatement, bit for bit,
    for every reconstruction filter of src/filters/ (defaults included)."""
    text = ('Sampler "sobol"\nPixelFilter "%s" %s\nWorldBegin\nAreaLightSource "diffuse"\n'
            'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\nWorldEnd' % (name, params))
    ps = capi.ParsedScene(text=text)        # keep it alive: desc points into it
    d = ps.desc
    b = scenes.SceneBuilder()
    build(b)
    assert np.array_equal(bits(list(d.filter_table)), bits(b.filter_table))
    assert np.array_equal(bits(list(d.filter_radius)), bits(np.array(b.filter_radius, np.float32)))


# ---- spectra: SPD -> RGB in the front end (pth_spectrum.cpp) against an independent f64 numpy evaluation
def _cie():
    raw = open(os.path.join(ROOT, "pbrt-r3_amd", "data", "spectrum_tables.bin"), "rb").read()
    assert raw[:8] == b"PTSPECT1"
    n = int(np.frombuffer(raw, np.uint32, 1, 8)[0])
    a = np.frombuffer(raw, np.float32, 4 * n + 1, 12).astype(np.float64)
    off = 12 + 4 * (4 * n + 1)
    m = int(np.frombuffer(raw, np.uint32, 1, off)[0])
    cu = np.frombuffer(raw, np.float32, 3 * m, off + 4).astype(np.float64)
    return a[:n], a[n:2 * n], a[2 * n:3 * n], a[3 * n:4 * n], a[4 * n], cu[:m], cu[m:2 * m], cu[2 * m:]


_XYZ2RGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])


def _rgb_60bins(lam, val):
    """SampledSpectrum pipeline in f64: piecewise-linear SPD averaged over 60 bins of [400,700], times the
    bin-averaged matching functions, scaled by 300 / (Yint * 60)."""
    X, Y, Z, L, yint, *_ = _cie()
    order = np.argsort(lam, kind="stable")
    lam, val = np.asarray(lam, np.float64)[order], np.asarray(val, np.float64)[order]

    def bins(l, v):
        out = np.empty(60)
        for i in range(60):
            w = np.linspace(400 + 5 * i, 405 + 5 * i, 2001)
            out[i] = np.trapezoid(np.interp(w, l, v), w) / 5.0
        return out
    c = bins(lam, val)
    xyz = np.array([(bins(L, m) * c).sum() for m in (X, Y, Z)]) * (300.0 / (yint * 60))
    return _XYZ2RGB @ xyz


def _scene_with(material_line, light='AreaLightSource "diffuse" "rgb L" [1 1 1]'):
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]'
    return 'Sampler "sobol"\nWorldBegin\nAttributeBegin\n%s\n%s\nAttributeEnd\n%s\n%s\nWorldEnd' % (light, tri, material_line, tri)


def test_metal_defaults_are_the_copper_spectrum():
    """create_metal_material (metal.rs:127-132): eta / k default to the measured copper SPD through
    RGBSpectrum::from_sampled (1 nm CIE integration, rgb.rs:124-149)."""
    ps = capi.ParsedScene(text=_scene_with('Material "metal"'))
    m = ps.desc.materials[ps.desc.meshes[1].material]
    X, Y, Z, L, yint, cw, cn, ck = _cie()
    for got, vals in ((list(m.metal_eta), cn), (list(m.metal_k), ck)):
        v = np.interp(L, cw, vals)
        xyz = np.array([(v * X).sum(), (v * Y).sum(), (v * Z).sum()]) * ((L[-1] - L[0]) / (yint * len(L)))
        assert np.allclose(got, _XYZ2RGB @ xyz, rtol=2e-5)
    assert np.allclose(list(m.metal_eta), [0.2004, 0.9240, 1.1022], atol=5e-3)     # the familiar copper RGB constants (60-bin route)
    assert np.allclose(list(m.metal_k), [3.9129, 2.4528, 2.1421], atol=1e-2)
    assert abs(m.roughness - 0.01) < 1e-9


def test_spectrum_files_and_blackbody(tmp_path):
    """"spectrum" parameters name .spd files (parser/common.rs:108-111, scene_context.rs:733-760): '#' lines are
    dropped, samples may be unsorted; "blackbody L" [T scale] goes through blackbody_normalized."""
    spd = tmp_path / "kd.spd"
    lam = np.array([700, 380, 450, 500, 560, 620, 780], np.float64)
    val = np.array([0.8, 0.05, 0.1, 0.3, 0.7, 0.9, 0.6], np.float64)
    spd.write_text("# wavelength value\n" + "\n".join("%g %g" % (l, v) for l, v in zip(lam, val)) + "\n550 0.123 # dropped line\n")
    text = _scene_with('Material "matte" "spectrum Kd" "kd.spd"', light='AreaLightSource "diffuse" "blackbody L" [6500 2.5]')
    ps = capi.ParsedScene(text=text, work_dir=str(tmp_path))
    d = ps.desc
    kd = list(d.materials[d.meshes[1].material].kd)
    assert np.allclose(kd, _rgb_60bins(lam, val), rtol=3e-4, atol=1e-5)
    # blackbody: Planck's law normalised at Wien's peak, times the scale
    X, Y, Z, L, yint, *_ = _cie()
    h, c, kb, T = 6.62606957e-34, 299792458.0, 1.3806488e-23, 6500.0
    planck = lambda nm: (2 * h * c * c) / ((nm * 1e-9) ** 5 * (np.exp((h * c) / (nm * 1e-9 * kb * T)) - 1))
    le = planck(L) / planck(2.8977721e-3 / T * 1e9) * 2.5
    want = _rgb_60bins(L, le)
    got = list(d.area_lights[0].L)
    assert np.allclose(got, want, rtol=3e-4)
    assert got[0] > got[2] * 0.8 and all(g > 0.5 for g in got)           # 6500 K is near white
    # an inline numeric spectrum is refused, not silently defaulted
    with pytest.raises(capi.PtError) as e:
        capi.ParsedScene(text=_scene_with('Material "matte" "spectrum Kd" [400 0.5 700 0.5]'))
    assert e.value.status == 4


def test_reference_kat_spectrum_blackbody():
    """/root/reference/tests/spectrum.rs:10-40 (`spectrum_blackbody`) restated 1:1 against the front end's Planck function
    (host/pth_spectrum.cpp, the one "blackbody L" goes through): four radiance values from spectralcalc.com to 1e-3 relative, and
    Wien's displacement law -- the returned radiance peaks at lambda_max = 2.8977721e-3 / T for five temperatures."""
    import ctypes as C
    lib = capi.load_library()
    lib.pth_blackbody.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_double)]
    lib.pth_blackbody.restype = None

    def blackbody(lams, t):
        a = (C.c_double * len(lams))(*[float(x) for x in lams])
        o = (C.c_double * len(lams))()
        lib.pth_blackbody(a, len(lams), float(t), o)
        return list(o)
    for lam, t, expected in ((483, 6000, 3.1849e13), (600, 6000, 2.86772e13), (500, 3700, 1.59845e12), (600, 4500, 7.46497e12)):
        le = blackbody([lam], t)[0]
        assert abs(le - expected) / expected < 0.001
    for t in (2700, 3000, 4500, 5600, 6000):
        lambda_max = 2.8977721e-3 / t * 1e9
        le = blackbody([0.999 * lambda_max, lambda_max, 1.001 * lambda_max], t)
        assert le[0] < le[1] and le[1] > le[2]
    assert blackbody([500.0], 0.0) == [0.0] and blackbody([500.0], -10.0) == [0.0]      # blackbody.rs:6-8


def _write_ply(path, fmt, P, N, UV, faces, gz=False):
    import gzip, struct
    hdr = ["ply", "format %s 1.0" % fmt, "comment made by the test", "element vertex %d" % len(P),
           "property float x", "property float y", "property float z"]
    if N is not None:
        hdr += ["property float nx", "property float ny", "property float nz"]
    if UV is not None:
        hdr += ["property float u", "property float v"]
    hdr += ["property uchar red", "element face %d" % len(faces), "property list uchar int vertex_indices", "property float quality", "end_header"]
    data = ("\n".join(hdr) + "\n").encode()
    e = "<" if fmt == "binary_little_endian" else ">"
    rows = []
    for i in range(len(P)):
        row = list(P[i]) + (list(N[i]) if N is not None else []) + (list(UV[i]) if UV is not None else [])
        if fmt == "ascii":
            rows.append((" ".join(repr(float(np.float32(x))) for x in row) + " 200\n").encode())
        else:
            rows.append(struct.pack(e + "%df" % len(row), *row) + struct.pack("B", 200))
    for f in faces:
        if fmt == "ascii":
            rows.append(("%d %s 0.5\n" % (len(f), " ".join(str(i) for i in f))).encode())
        else:
            rows.append(struct.pack("B", len(f)) + struct.pack(e + "%di" % len(f), *f) + struct.pack(e + "f", 0.5))
    data += b"".join(rows)
    (gzip.open if gz else open)(path, "wb").write(data)


@pytest.mark.parametrize("fmt,gz", [("ascii", False), ("binary_little_endian", False), ("binary_big_endian", False), ("binary_little_endian", True)])
def test_plymesh_equals_trianglemesh(tmp_path, fmt, gz):
    """Shape "plymesh" (shapes/plymesh.rs:251-380): every encoding flattens to the arrays of the equivalent
    trianglemesh; quads split as (i0,i1,i2),(i3,i0,i2); unknown vertex / face properties are skipped."""
    rng = np.random.default_rng(3)
    P = rng.random((9, 3)).astype(np.float32) * 2 - 1
    N = rng.standard_normal((9, 3)).astype(np.float32)
    UV = rng.random((9, 2)).astype(np.float32)
    faces = [(0, 1, 2), (3, 4, 5, 6), (6, 7, 8), (1, 3, 5, 7)]
    name = "m.ply.gz" if gz else "m.ply"
    _write_ply(str(tmp_path / name), fmt, P, N, UV, faces, gz)
    idx = []
    for f in faces:
        idx += list(f) if len(f) == 3 else [f[0], f[1], f[2], f[3], f[0], f[2]]
    arr = lambda a: " ".join(repr(float(x)) for x in np.asarray(a).reshape(-1))
    head = 'Sampler "sobol"\nWorldBegin\nAttributeBegin\nAreaLightSource "diffuse"\nShape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 9 1 0 9 0 1 9]\nAttributeEnd\nTranslate 0.5 0 0\nRotate 30 0 1 0\n'
    a = capi.ParsedScene(text=head + 'Shape "plymesh" "string filename" "%s"\nWorldEnd' % name, work_dir=str(tmp_path))
    b = capi.ParsedScene(text=head + 'Shape "trianglemesh" "integer indices" [%s] "point P" [%s] "normal N" [%s] "float uv" [%s]\nWorldEnd'
                         % (" ".join(map(str, idx)), arr(P), arr(N), arr(UV)))
    da, db = a.desc, b.desc
    assert (da.n_vertices, da.n_triangles) == (db.n_vertices, db.n_triangles) == (12, 7)
    for field, w in (("P", 3), ("N", 3), ("UV", 2)):
        ga = np.ctypeslib.as_array(getattr(da, field), (da.n_vertices * w,))
        gb = np.ctypeslib.as_array(getattr(db, field), (db.n_vertices * w,))
        assert np.array_equal(bits(ga), bits(gb)), field
    assert np.array_equal(np.ctypeslib.as_array(da.indices, (21,)), np.ctypeslib.as_array(db.indices, (21,)))
    assert da.meshes[1].flags == db.meshes[1].flags


def test_plymesh_errors(tmp_path):
    (tmp_path / "bad.ply").write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty double x\nproperty double y\nproperty double z\n"
                                      "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    (tmp_path / "pent.ply").write_text("ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
                                       "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n0 2 0\n5 0 1 2 3 4\n")
    for f, needle in (("bad.ply", "float"), ("pent.ply", "5 vertices"), ("missing.ply", "open")):
        with pytest.raises(capi.PtError) as e:
            capi.ParsedScene(text='Sampler "sobol"\nWorldBegin\nShape "plymesh" "string filename" "%s"\nWorldEnd' % f, work_dir=str(tmp_path))
        assert needle in str(e.value)


def test_constant_folding_textures():
    """Texture "constant" / "scale" / "mix" (textures/{constant,scale,mix}.rs) fold to values; a texture binding wins
    over constant values (texture_params.rs:107-139); texture tables are global, not attribute-scoped (pbrt_texture
    writes through the shared map, scene_context.rs:1093-1110); position-dependent textures fail only when used."""
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]'
    text = '''
    Sampler "sobol" "integer pixelsamples" 1
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [1 1 1]
        %(tri)s
        Texture "red" "spectrum" "constant" "rgb value" [0.8 0.1 0.1]
        Texture "half" "float" "constant" "float value" 0.5
      AttributeEnd
      Texture "dim" "spectrum" "scale" "texture tex1" "red" "rgb tex2" [0.5 0.5 0.5]
      Texture "blend" "color" "mix" "texture tex1" "red" "rgb tex2" [0 0 1] "float amount" 0.25
      Texture "rough" "float" "mix" "float tex1" 0.1 "float tex2" 0.3 "texture amount" "half"
      Texture "chk" "spectrum" "checkerboard"
      Material "plastic" "texture Kd" "dim" "rgb Ks" [0.2 0.2 0.2] "texture roughness" "rough"
      %(tri)s
      Material "matte" "rgb Kd" [0.3 0.3 0.3]
      %(tri)s "texture Kd" "blend"
    WorldEnd
    ''' % {"tri": tri}
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    f = np.float32
    m1 = d.materials[d.meshes[1].material]
    assert np.array_equal(bits(list(m1.kd)), bits([f(0.8) * f(0.5), f(0.1) * f(0.5), f(0.1) * f(0.5)]))
    assert bits([m1.roughness])[0] == bits([f(f(0.1) * f(f(1.0) - f(0.5))) + f(f(0.3) * f(0.5))])[0]
    m2 = d.materials[d.meshes[2].material]          # the shape's texture binding beats the material's constant Kd
    want = [f(f(f(c1) * f(f(1.0) - f(0.25))) + f(f(c2) * f(0.25))) for c1, c2 in zip((0.8, 0.1, 0.1), (0.0, 0.0, 1.0))]
    assert np.array_equal(bits(list(m2.kd)), bits(want))
    # a checkerboard varies over the surface: it becomes a device texture node bound to the parameter
    ps2 = capi.ParsedScene(text=text.replace('"texture Kd" "blend"', '"texture Kd" "chk"'))
    d2 = ps2.desc
    m3 = d2.materials[d2.meshes[2].material]
    assert d2.n_textures == 1 and m3.tex_kd == 1 and d2.textures[0].type == capi.PT_TEX_CHECKERBOARD_2D
    # ... and so does a roughness (ABI 6: roughness_to_alpha runs on the device at every hit); an image map whose file is missing is reported
    ps3 = capi.ParsedScene(text=text.replace('Texture "chk" "spectrum" "checkerboard"', 'Texture "fchk" "float" "checkerboard"')
                           .replace('"texture roughness" "rough"', '"texture roughness" "fchk"'))
    d3 = ps3.desc
    m4 = d3.materials[d3.meshes[1].material]
    assert m4.tex_roughness == 1 and m4.tex_uroughness == 0 and d3.textures[0].type == capi.PT_TEX_CHECKERBOARD_2D and m4.remap_roughness == 1
    with pytest.raises(capi.PtError) as e:
        capi.ParsedScene(text=text.replace('"spectrum" "checkerboard"', '"spectrum" "imagemap" "string filename" "x.png"')
                         .replace('"texture Kd" "blend"', '"texture Kd" "chk"'))
    assert e.value.status == 4 and "File not found" in str(e.value)


def test_texture_directives_equal_programmatic_scene(oracle):
    """Texture "checkerboard" (2-D with uv / planar / spherical mappings, 3-D under a CTM), "uv", "bilerp", and "scale" /
    "mix" over them, bound to material parameters through the shape-first / material-first rules: the flattened scene must
    render exactly like the SceneBuilder one (textures/*.rs, mapping2d.rs:178-214, texture_params.rs)."""
    text = fs.TEXTURED_PBRT
    ps = capi.ParsedScene(text=text)
    d = ps.desc
    assert d.n_textures == 10
    T = scenes
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -6.5), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0); b.film(xresolution=32, yresolution=32)
    b.pixel_filter_box(); b.sampler_sobol(4); b.integrator_path(maxdepth=4)
    b.area_light_source_diffuse(L=(10, 9, 8))
    b.shape_trianglemesh([(0.5, 1.99, -0.5), (0.5, 1.99, 0.5), (-0.5, 1.99, 0.5), (-0.5, 1.99, -0.5)], [0, 1, 2, 0, 2, 3])
    b.no_area_light()
    uvt = b.texture_uv(uscale=3.0, vscale=2.0)
    amt = b.texture_bilerp(0.1, 0.9, 0.6, 0.3)
    mixt = b.texture_mix((0.8, 0.2, 0.1), (0.1, 0.3, 0.8), amount=amt)
    floor = b.texture_checkerboard(mixt, uvt, uscale=6.0, vscale=6.0, udelta=0.25, vdelta=0.1)
    b.material_matte(floor)
    b.shape_trianglemesh([(2, -2, -2), (-2, -2, -2), (-2, -2, 2), (2, -2, 2)], [0, 1, 2, 0, 2, 3])
    planar = b.texture_checkerboard((0.9, 0.9, 0.2), (0.2, 0.2, 0.2), mapping="planar", v1=(1.5, 0.0, 0.0), v2=(0.0, 1.5, 0.3), udelta=0.2, vdelta=0.4, aamode="none")
    fine = b.texture_checkerboard(1.0, 0.2, uscale=10.0, vscale=10.0)
    ks = b.texture_scale((0.5, 0.5, 0.5), fine)
    b.material_plastic(Kd=planar, Ks=ks, roughness=0.05)
    b.shape_trianglemesh([(2, -2, 2), (-2, -2, 2), (-2, 2, 2), (2, 2, 2)], [0, 1, 2, 0, 2, 3])
    t3 = T.transform_mul(T.transform_rotate_x(25.0), T.transform_scale(3.0, 3.0, 3.0))
    c3 = b.texture_checkerboard((0.9, 0.5, 0.1), (0.1, 0.1, 0.4), dimension=3, to_world=t3)
    ts = T.transform_translate(0.9, -1.3, -0.2)
    sph = b.texture_checkerboard((0.8, 0.8, 0.8), (0.15, 0.3, 0.15), mapping="spherical", to_world=ts)
    sig = b.texture_bilerp(0.0, 40.0, 10.0, 60.0, to_world=ts)       # the CTM is recorded whatever the mapping
    b.material_matte(c3, sigma=sig)
    b.shape_sphere(radius=0.65, object_to_world=ts[0], world_to_object=ts[1])
    b.material_uber(Kd=sph, Ks=(0.2, 0.2, 0.2))
    b.shape_trianglemesh([(-2, -2, 2), (-2, -2, -2), (-2, 2, 0)], [0, 1, 2])
    ref = b.build()
    assert ref.desc.n_textures == 10
    for i in range(10):
        ta, tb = d.textures[i], ref.desc.textures[i]
        assert bytes(ta) == bytes(tb), i
    a, r = oracle.scene(ps), oracle.scene(ref)
    xa, ca, _ = a.render(threads=2)
    xr, cr, _ = r.render(threads=2)
    assert np.array_equal(bits(xa), bits(xr)) and ca == cr and xa[..., :3].max() > 0
    a.close(); r.close()


def test_parse_options_quick_and_pixelsamples():
    """--quick (bin/pbrt.rs:360-366): pixelsamples 1 and resolution / 4 (create_film, film.rs:548-554; the camera's
    frame aspect follows the reduced resolution); --quick_full_resolution keeps the resolution; --pixelsamples
    overrides afterwards (bin/pbrt.rs:234-238)."""
    import ctypes as C

    class Opts(C.Structure):
        _fields_ = [("quick", C.c_int32), ("quick_full_resolution", C.c_int32), ("pixelsamples", C.c_int32), ("reserved", C.c_int32)]
    lib = capi.load_library()
    lib.pth_parse_file_opts.argtypes = [C.c_char_p, C.POINTER(Opts), C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    lib.pth_scene_get_desc.restype = C.POINTER(capi.pt_scene_desc)
    lib.pth_scene_get_desc.argtypes = [C.c_void_p]
    lib.pth_scene_free.argtypes = [C.c_void_p]
    scene = os.path.join(ROOT, "tests", "scenes", "cornell.pbrt").encode()

    def parse(**kw):
        h, err = C.c_void_p(), C.create_string_buffer(512)
        assert lib.pth_parse_file_opts(scene, C.byref(Opts(**kw)), C.byref(h), err, 512) == 0, err.value
        d = lib.pth_scene_get_desc(h).contents
        out = (d.xres, d.yres, d.spp)
        lib.pth_scene_free(h)
        return out
    assert parse() == (64, 64, 16)
    assert parse(quick=1) == (16, 16, 1)
    assert parse(quick_full_resolution=1) == (64, 64, 1)
    assert parse(quick=1, pixelsamples=5) == (16, 16, 5)


def test_image_writers(tmp_path):
    """pth_write_image: EXR (uncompressed float scanlines, data window inside the display window), PNG with the
    reference's to_byte (write_image.rs:16-18) and PFM, decoded here independently."""
    import ctypes as C, struct, zlib
    lib = capi.load_library()
    lib.pth_write_image.argtypes = [C.c_char_p, C.c_void_p] + [C.c_int] * 6
    rng = np.random.default_rng(4)
    w, h = 7, 5
    img = (rng.random((h, w, 3), dtype=np.float32) * 1.4 - 0.1).astype(np.float32)
    img[0, 0] = (0.0, 0.002, 1.0)
    ptr = img.ctypes.data_as(C.c_void_p)
    # ---- PNG
    p = str(tmp_path / "a.png")
    assert lib.pth_write_image(p.encode(), ptr, w, h, 0, 0, w, h) == 0
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR":
            assert struct.unpack(">IIBBBBB", body) == (w, h, 8, 2, 0, 0, 0)
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    f = np.float32
    with np.errstate(invalid="ignore"):
        g = np.where(img <= f(0.0031308), f(12.92) * img, f(1.055) * np.power(img, f(1.0 / 2.4), dtype=np.float32) - f(0.055)).astype(np.float32)
    want = np.clip(np.nan_to_num(f(255.0) * g, nan=0.0), 0, 255).astype(np.uint8)
    got = rows[:, 1:].reshape(h, w, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.97      # numpy's powf may differ in the last ulp
    # ---- EXR: 4 x 3 crop at (2, 1) of a 10 x 8 frame
    p = str(tmp_path / "a.exr")
    assert lib.pth_write_image(p.encode(), ptr, w, h, 2, 1, 10, 8) == 0
    raw = open(p, "rb").read()
    assert struct.unpack("<II", raw[:8]) == (20000630, 2)
    pos, attrs = 8, {}
    while raw[pos] != 0:
        e = raw.index(b"\0", pos); name = raw[pos:e].decode(); pos = e + 1
        e = raw.index(b"\0", pos); typ = raw[pos:e].decode(); pos = e + 1
        n = struct.unpack("<I", raw[pos:pos + 4])[0]
        attrs[name] = (typ, raw[pos + 4:pos + 4 + n]); pos += 4 + n
    pos += 1
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (2, 1, 2 + w - 1, 1 + h - 1)
    assert struct.unpack("<4i", attrs["displayWindow"][1]) == (0, 0, 9, 7) and attrs["compression"][1] == b"\0"
    offs = struct.unpack("<%dQ" % h, raw[pos:pos + 8 * h])
    back = np.empty((h, w, 3), np.float32)
    for y in range(h):
        yy, n = struct.unpack("<ii", raw[offs[y]:offs[y] + 8])
        assert yy == 1 + y and n == 12 * w
        line = np.frombuffer(raw, np.float32, 3 * w, offs[y] + 8).reshape(3, w)       # B, G, R planes
        back[y, :, 2], back[y, :, 1], back[y, :, 0] = line[0], line[1], line[2]
    assert np.array_equal(bits(back), bits(img))
    # ---- PFM and an unknown extension
    p = str(tmp_path / "a.pfm")
    assert lib.pth_write_image(p.encode(), ptr, w, h, 0, 0, w, h) == 0
    body = open(p, "rb").read().split(b"\n", 3)[3]
    assert np.array_equal(bits(np.frombuffer(body, "<f4").reshape(h, w, 3)[::-1]), bits(img))
    assert lib.pth_write_image(str(tmp_path / "a.tga").encode(), ptr, w, h, 0, 0, w, h) == 4
