"""The lower half of the HLBVH build on the GPU (pt_hlbvh.hip: Morton codes, radix sort, treelets, emit_lbvh) against the host
builder -- which tests/test_host.py pins to the oracle's restatement of hlbvh.rs -- through the C ABI: the uploaded 4-wide node array
and leaf record array must be byte-identical wherever the build ran."""
import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, pkg, random_rays, scenes

pytestmark = pytest.mark.gpu
HOST, DEVICE, AUTO = pkg.capi.BVH_BUILD_HOST, pkg.capi.BVH_BUILD_DEVICE, pkg.capi.BVH_BUILD_AUTO


def _digests(sd, expect_device=True):
    ctx = pkg.Context(0)
    try:
        ctx.set_bvh_build(HOST)
        info = ctx.upload(sd)
        assert info.bvh_on_device == 0
        host = (ctx.bvh_digest(), info.n_nodes, info.n_leaves, tuple(info.world_bound))
        o, d, tmax = random_rays(info, 20000, 5)
        hits_host = ctx.trace_closest(o, d, tmax)
        ctx.set_bvh_build(DEVICE)
        info = ctx.upload(sd)
        assert info.bvh_on_device == (1 if expect_device else 0)
        dev = (ctx.bvh_digest(), info.n_nodes, info.n_leaves, tuple(info.world_bound))
        hits_dev = ctx.trace_closest(o, d, tmax)
        assert np.array_equal(hits_host["prim"], hits_dev["prim"]) and np.array_equal(bits(hits_host["t"]), bits(hits_dev["t"]))
        return host, dev
    finally:
        ctx.close()


@pytest.mark.parametrize("n,leaf", [(12, 4), (13, 1), (1024, 4), (1025, 5), (4000, 4), (70001, 4), (300000, 8), (300000, 255)])
def test_device_build_matches_host_build(n, leaf):
    """Sizes around the sort tile (1024 keys) and the wave (64); leaf sizes 1 .. 255.  The enclosure's ceiling and light quads are
    four triangles whose boxes share one centre, so below four primitives per leaf emit_lbvh runs out of code bits and goes on by
    centroid medians (split_node) -- on the device as well since round 3."""
    sd = scenes.rt1m(n, res=16, spp=1, max_depth=1)
    sd.desc.split_method = 1
    sd.desc.max_node_prims = leaf
    host, dev = _digests(sd, expect_device=True)
    assert host == dev


def test_device_build_clustered_geometry():
    """Geometry concentrated in a few Morton cells: a handful of large treelets, deep bit levels, uneven level widths."""
    rng = np.random.default_rng(11)
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -5), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0)
    b.film(xresolution=16, yresolution=16); b.pixel_filter_box(); b.sampler_sobol(1); b.integrator_path(maxdepth=1)
    b.accelerator_bvh("hlbvh", 8)
    b.material_matte((0.5, 0.5, 0.5))
    centers = np.concatenate([rng.normal((0.5, 0.2, -0.3), 0.1, (60000, 3)), rng.normal((-0.7, -0.6, 0.6), 0.05, (30000, 3)),
                              rng.uniform(-2, 2, (500, 3))]).astype(np.float32)
    off = rng.uniform(-0.002, 0.002, (len(centers), 3, 3)).astype(np.float32)
    verts = (centers[:, None, :] + off).reshape(-1, 3)
    b.shape_trianglemesh_fast(verts, np.arange(len(verts)), twosided=True)
    host, dev = _digests(b.build())
    assert host == dev


def test_device_build_median_fallback_on_the_device():
    """More primitives than a leaf may hold inside one Morton cell (a cluster of 60 within 1e-5, 40 exact duplicates): split_node's
    centroid medians (hlbvh.rs:102-157) run on the device -- stable order along the cycling axis, cut in the middle -- and give the
    host builder's arrays."""
    host, dev = _digests(fs.scene_hlbvh_cluster(), expect_device=True)
    assert host == dev
    sd = fs.scene_hlbvh_cluster()
    sd.desc.max_node_prims = 1
    host, dev = _digests(sd, expect_device=True)
    assert host == dev


def test_auto_picks_device_from_64k_primitives():
    ctx = pkg.Context(0)
    try:
        for n, want in ((65535 - 12, 0), (70000, 1)):
            sd = scenes.rt1m(n, res=16, spp=1, max_depth=1)
            sd.desc.split_method = 1
            assert ctx.upload(sd).bvh_on_device == want
        sd.desc.split_method = 2           # "middle" (and "equal") always build on the host
        ctx.set_bvh_build(DEVICE)
        assert ctx.upload(sd).bvh_on_device == 0
        with pytest.raises(pkg.PtError):
            ctx.set_bvh_build(7)
    finally:
        ctx.close()


def test_instanced_scene_with_device_built_object_trees(oracle):
    """Every primitive list (the world and each object) goes through the device builder; per-sample radiance stays the oracle's."""
    sd = fs.scene_instances(split="hlbvh")
    ctx = pkg.Context(0)
    try:
        ctx.set_bvh_build(DEVICE)
        info = ctx.upload(sd)
        assert info.bvh_on_device == 1
        osc = oracle.scene(sd)
        sb = list(info.sample_bounds)
        cx, cy = (sb[0] + sb[2]) // 2, (sb[1] + sb[3]) // 2
        tile = (cx - 8, cy - 8, cx + 8, cy + 8)
        assert np.array_equal(bits(ctx.radiance_samples(tile)), bits(osc.radiance_samples(tile)))
        osc.close()
    finally:
        ctx.close()


# ---- the SAH build on the GPU (pt_sah.hip)
@pytest.mark.parametrize("n,leaf", [(12, 4), (13, 2), (65, 4), (1024, 4), (1025, 3), (4000, 4), (70001, 4), (300000, 8), (300000, 2), (1000000, 4)])
def test_device_sah_matches_host_sah(n, leaf, oracle):
    """Level-by-level SAH on the device against the host's recursive builder: byte-identical node and record arrays, and the leaf
    order the oracle's restatement of sah.rs produces."""
    sd = scenes.rt1m(n, res=16, spp=1, max_depth=1)
    sd.desc.split_method = 0
    sd.desc.max_node_prims = leaf
    host, dev = _digests(sd)
    assert host == dev
    if n <= 300000:
        ctx = pkg.Context(0)
        try:
            ctx.set_bvh_build(DEVICE)
            info = ctx.upload(sd)
            assert info.bvh_on_device == 1
            osc = oracle.scene(sd)
            assert (osc.info.n_nodes, osc.info.n_leaves) == (info.n_nodes, info.n_leaves)
            # the leaf order: the device-built record array is byte-identical to the host builder's (digests above), and the host
            # builder's order is the oracle's restatement of the reference's ordered_prims (build/node.rs:138-151)
            order, n_nodes, n_leaves, _ = pkg.capi.bvh_leaf_order(sd)
            assert np.array_equal(order, osc.ordered_prims()) and (n_nodes, n_leaves) == (info.n_nodes, info.n_leaves)
            o, d, tmax = random_rays(info, 20000, 9)
            g = ctx.trace_closest(o, d, tmax)
            r, _ = osc.trace_closest(o, d, tmax)
            assert np.array_equal(g["prim"], r["prim"]) and np.array_equal(bits(g["t"]), bits(r["t"]))
            osc.close()
        finally:
            ctx.close()


def test_device_sah_clustered_and_feature_scenes(oracle):
    """Clustered geometry (deep, lopsided levels), a scene with spheres and one with instances (every primitive list goes through the
    device builder)."""
    rng = np.random.default_rng(12)
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -5), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0)
    b.film(xresolution=16, yresolution=16); b.pixel_filter_box(); b.sampler_sobol(1); b.integrator_path(maxdepth=1)
    b.accelerator_bvh("sah", 4)
    b.material_matte((0.5, 0.5, 0.5))
    centers = np.concatenate([rng.normal((0.5, 0.2, -0.3), 0.1, (60000, 3)), rng.normal((-0.7, -0.6, 0.6), 0.05, (30000, 3)),
                              rng.uniform(-2, 2, (500, 3))]).astype(np.float32)
    off = rng.uniform(-0.002, 0.002, (len(centers), 3, 3)).astype(np.float32)
    verts = (centers[:, None, :] + off).reshape(-1, 3)
    b.shape_trianglemesh_fast(verts, np.arange(len(verts)), twosided=True)
    host, dev = _digests(b.build())
    assert host == dev
    for sd in (fs.scene_spheres(), fs.scene_instances(split="sah")):
        ctx = pkg.Context(0)
        try:
            ctx.set_bvh_build(DEVICE)
            info = ctx.upload(sd)
            osc = oracle.scene(sd)
            sb = list(info.sample_bounds)
            cx, cy = (sb[0] + sb[2]) // 2, (sb[1] + sb[3]) // 2
            tile = (cx - 8, cy - 8, cx + 8, cy + 8)
            assert np.array_equal(bits(ctx.radiance_samples(tile)), bits(osc.radiance_samples(tile)))
            osc.close()
        finally:
            ctx.close()


# ---- the whole upload on the device (pt_sah.hip device_sah_scene): bounds, SAH, records, shading records, collapse, finish
def _eligible(sd):
    d = sd.desc
    return d.n_spheres == 0 and d.n_instances == 0 and d.split_method == 0 and all(d.meshes[i].object == 0 for i in range(d.n_meshes))


def test_device_scene_path_shading_records_and_lights(oracle):
    """A triangle-only world list under SAH is built, collapsed and finished on the device, which also writes the shading records
    (vertex indices, mesh flags, material, light) and numbers the lights' records.  The digests of the tests above pin the node and
    leaf-record arrays; here the same path is held to the oracle where those other arrays matter: textured and bump-mapped materials
    (uv / normals through PtTriInfo), several emissive meshes with `nsamples`, one-sided lights, materials of every kind -- per-sample
    radiance, film and every counter, on scenes small enough that only the forced device path runs them there."""
    from test_gpu_features import _compare
    makes = [lambda: scenes.cornell_box(res=40, spp=8), lambda: fs.scene_materials_lights("spatial"), lambda: fs.scene_materials_lights("power"),
             lambda: fs.scene_bump(), lambda: fs.scene_imagemaps(), lambda: fs.scene_textures(), lambda: fs.scene_noise_textures(), lambda: fs.scene_attributes(),
             lambda: fs.scene_materials_render(["glass", "mirror", "plastic"], spp=4), lambda: fs.scene_accel("sah", 2)]
    ran = 0
    for make in makes:
        sd = make()
        if not _eligible(sd):
            continue
        ctx = pkg.Context(0)
        try:
            ctx.set_bvh_build(DEVICE)
            info = ctx.upload(sd)
            assert info.bvh_on_device == 1
            osc = oracle.scene(sd)
            assert (osc.info.n_nodes, osc.info.n_leaves, osc.info.n_lights) == (info.n_nodes, info.n_leaves, info.n_lights)
            _compare(ctx, osc, exact_film=True)
            osc.close()
            ran += 1
        finally:
            ctx.close()
    assert ran >= 5


def test_device_scene_path_equals_host_finish(monkeypatch):
    """The same device-built binary tree finished on the host (PBRTGPU_HOST_FINISH=1: tree and order read back, records, collapse and
    finishing pass on the host's threads) and on the device: byte-identical node and record arrays, same counts and bounds."""
    sd = scenes.rt1m(200000, res=16, spp=1, max_depth=1)
    got = []
    for host_finish in ("1", None):
        if host_finish:
            monkeypatch.setenv("PBRTGPU_HOST_FINISH", host_finish)
        else:
            monkeypatch.delenv("PBRTGPU_HOST_FINISH", raising=False)
        ctx = pkg.Context(0)
        try:
            ctx.set_bvh_build(DEVICE)
            info = ctx.upload(sd)
            assert info.bvh_on_device == 1
            got.append((ctx.bvh_digest(), info.n_nodes, info.n_leaves, tuple(info.world_bound)))
        finally:
            ctx.close()
    assert got[0] == got[1]


def test_device_scene_path_with_world_spheres(oracle):
    """Round 4: analytic spheres of the WORLD list ride along the device upload (killeroo-simple's lights are spheres: config 4 used to take the
    host builder for that alone).  A sphere is one primitive of the merged list -- spliced in before triangle `before_triangle` --, its world
    bound comes from the host, its leaf record, shading record and light number are written by the same kernels as the triangles'.  Node and
    record arrays byte-identical to the host path's (SAH and HLBVH, sphere objects and sphere lights, a sphere as the very first and the very
    last primitive), then per-sample radiance, film and counters against the oracle on the device-built scene."""
    from test_gpu_features import _compare

    def rt_sphere(n, split):
        sd = scenes.rt1m(n, res=32, spp=4, max_depth=4, light="sphere")
        sd.desc.split_method = split
        return sd
    makes = [lambda: fs.scene_spheres("spatial"), lambda: fs.scene_spheres("power", split="hlbvh"), lambda: fs.scene_spheres("spatial", lights_only=True),
             lambda: rt_sphere(70001, 0), lambda: rt_sphere(70001, 1)]
    ran = 0
    for make in makes:
        sd = make()
        d = sd.desc
        if d.n_spheres == 0 or d.n_instances or d.split_method not in (0, 1) or any(d.meshes[i].object for i in range(d.n_meshes)) or any(d.spheres[i].object for i in range(d.n_spheres)):
            continue
        host, dev = _digests(sd)
        assert host == dev
        ctx = pkg.Context(0)
        try:
            ctx.set_bvh_build(DEVICE)
            info = ctx.upload(sd)
            assert info.bvh_on_device == 1
            osc = oracle.scene(sd)
            assert (osc.info.n_nodes, osc.info.n_leaves, osc.info.n_lights) == (info.n_nodes, info.n_leaves, info.n_lights)
            _compare(ctx, osc, exact_film=True)
            osc.close()
            ran += 1
        finally:
            ctx.close()
    assert ran >= 3
