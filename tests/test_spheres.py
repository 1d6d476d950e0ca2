"""Analytic spheres (src/shapes/sphere.rs) on the accelerated path: CPU-side checks.  The reference's own
sphere tests (tests/shapes.rs:280-329, :416-457) run in oracle/orc_kat (test_oracle_kat.py); here the oracle's
BVH with spheres is checked against its exhaustive loop, the product's host builder against the oracle's
primitive order, and the C ABI's validation of pt_sphere against what it documents."""
import ctypes as C

import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, pkg, random_rays, scenes


@pytest.mark.parametrize("split", ["sah", "hlbvh", "middle", "equal"])
def test_host_bvh_with_spheres_matches_oracle_order(oracle, split):
    sd = fs.scene_spheres(split=split, res=16, spp=1)
    n = sd.desc.n_triangles + sd.desc.n_spheres
    assert sd.desc.n_spheres == 6
    for leaf in (1, 4):
        sd.desc.max_node_prims = leaf
        order, n_nodes, n_leaves, _ = pkg.capi.bvh_leaf_order(sd)
        sc = oracle.scene(sd)
        assert np.array_equal(order, sc.ordered_prims())
        assert (n_nodes, n_leaves) == (sc.info.n_nodes, sc.info.n_leaves)
        assert sorted(order) == list(range(n))
        sc.close()


def test_oracle_bvh_equals_exhaustive_with_spheres(oracle):
    """accelerators/exhaustive as the cross-check: same closest primitive and t through the QBVH."""
    sc = oracle.scene(fs.scene_spheres(res=16, spp=1))
    o, d, tmax = random_rays(sc.info, 20000, 5)
    a, _ = sc.trace_closest(o, d, tmax)
    b, _ = sc.trace_closest(o, d, tmax, exhaustive=True)
    assert np.array_equal(a["prim"], b["prim"])
    hit = a["prim"] >= 0
    assert np.array_equal(bits(a["t"][hit]), bits(b["t"][hit]))
    # the first sphere is primitive 0 (it precedes every triangle), the last one is the last primitive
    n = sc.sd.desc.n_triangles + sc.sd.desc.n_spheres
    assert (a["prim"] == 0).any() and (a["prim"] == n - 1).any()
    assert sc.info.n_lights == 1 + 2 + 1          # sphere, the quad's two triangles, sphere
    sc.close()


def test_sphere_light_only_scene_is_lit(oracle):
    """A scene whose only light is a sphere: Sphere::sample_from's cone sampling feeds next-event estimation."""
    b = fs.base(res=24, spp=64, depth=3)
    b.material_matte((0.7, 0.7, 0.7))
    scenes._quad(b, (3, -1, -3), (-3, -1, -3), (-3, -1, 3), (3, -1, 3))
    b.area_light_source_diffuse(L=(20, 20, 20))
    t = scenes.transform_translate(0.0, 1.0, 0.0)
    b.shape_sphere(radius=0.3, object_to_world=t[0], world_to_object=t[1])
    sc = oracle.scene(b.build())
    x, cnt, _ = sc.render(threads=4)
    rgb = sc.resolve_rgb(x)
    assert sc.info.n_lights == 1
    assert cnt["shadow_rays"] > 0
    floor = rgb[16:, :, :]            # lower half of the image sees the floor
    assert floor.mean() > 0.05
    # analytic check at the floor point under the sphere: E = L * pi * (r/d)^2, Lo = Kd/pi * E.  The brightest floor row
    # of the central columns averages a pixel footprint around that point (observed 0.89 of the point value).
    peak = rgb[14:, 10:14, 1].mean(axis=1).max()
    assert 0.75 < peak / (0.7 * 20 * (0.3 / 2.0) ** 2) < 1.05
    sc.close()


def test_abi_rejects_bad_spheres():
    lib = pkg.capi.load_library()
    sd = fs.scene_spheres(res=16, spp=1)
    order = np.empty(sd.desc.n_triangles + sd.desc.n_spheres, np.uint32)
    sph = sd.buffers["spheres"]
    keep = sph[2].before_triangle
    sph[2].before_triangle = sd.desc.n_triangles + 1          # past the end of the primitive list
    st = lib.pt_bvh_leaf_order(C.byref(sd.desc), order.ctypes.data_as(C.c_void_p), None, None, None)
    assert st != 0
    sph[2].before_triangle = 0                                # not ordered (spheres[1] sits after the room)
    st = lib.pt_bvh_leaf_order(C.byref(sd.desc), order.ctypes.data_as(C.c_void_p), None, None, None)
    assert st != 0
    sph[2].before_triangle = keep
    st = lib.pt_bvh_leaf_order(C.byref(sd.desc), order.ctypes.data_as(C.c_void_p), None, None, None)
    assert st == 0
