"""Ray-level parity of the HOT traversal kernel.  pt_trace_closest / pt_trace_any run a plain one-ray-per-lane kernel
(k_trace_batch); pt_render spends its time in k_trace and its variants (persistent lanes, three-stage ray prefetch, per-XCD
queue segments with stealing, pooled leaf rounds with bpermuted ray constants and the owner-side t_max replay).  These tests
push caller rays -- camera rays plus hostile ones: axis-aligned directions (infinite reciprocals, the NaN-exact slab form),
finite t_max, unnormalised shadow-like segments -- through THAT kernel via pt_trace_wavefront, as the mix of continuation /
shadow / probe work items a bounce launches, and compare every hit, t, barycentric, occlusion flag and the node / triangle
counters with the oracle (qbvh_x86.rs:230-343).  Bit-exact."""
import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, random_rays, scenes

pytestmark = pytest.mark.gpu


def _rays(ctx, osc, n, seed, kinds=(1, 2, 3)):
    """n rays: a third camera rays, a third hostile random rays, a third shadow-like segments; kinds dealt at random."""
    rng = np.random.default_rng(seed)
    sb = list(ctx.info.sample_bounds)
    n_cam = n // 3
    px = np.stack([rng.integers(sb[0], sb[2], n_cam), rng.integers(sb[1], sb[3], n_cam)], 1).astype(np.int32)
    si = rng.integers(0, max(1, ctx.info.spp), n_cam).astype(np.uint32)
    co, cd, _ = osc.generate_camera_rays(px, si)
    n_rand = (n - n_cam) // 2
    ro, rd, rt = random_rays(ctx.info, n_rand, seed + 1)
    so, sd_, st = random_rays(ctx.info, n - n_cam - n_rand, seed + 2, shadow_like=True)
    o = np.concatenate([co, ro, so]); d = np.concatenate([cd, rd, sd_])
    t = np.concatenate([np.full(n_cam, np.inf, np.float32), rt, st])
    perm = rng.permutation(n)
    kind = np.asarray(kinds, np.uint8)[rng.integers(0, len(kinds), n)]
    return o[perm], d[perm], t[perm], kind


def _check(ctx, osc, o, d, t, kind, probe_prims=True, barycentrics=True):
    ctx.reset_counters()
    g, gocc = ctx.trace_wavefront(o, d, t, kind)
    gc = ctx.counters()
    m1, m2, m3 = kind == 1, kind == 2, kind == 3
    closest = m1 | m3
    r, oc = osc.trace_closest(o[closest], d[closest], t[closest])
    ro, oa = osc.trace_any(o[m2], d[m2], t[m2])
    rk = kind[closest]
    r1, r3 = r[rk == 1], r[rk == 3]
    # continuation rays: primitive, t, barycentrics
    assert np.array_equal(g["prim"][m1], r1["prim"])
    hit = r1["prim"] >= 0
    for f in ("t", "b0", "b1") if barycentrics else ("t",):
        assert np.array_equal(bits(g[f][m1][hit]), bits(r1[f][hit])), f
    # shadow rays: the occlusion flag
    assert np.array_equal(gocc[m2], ro)
    # probe rays: the primitive the closest hit lands on
    if probe_prims:
        assert np.array_equal(g["prim"][m3], r3["prim"])
    # the same traversal step for step: ray, node and primitive-test counters
    assert gc["regular_rays"] == int(closest.sum()) and gc["shadow_rays"] == int(m2.sum())
    assert gc["nodes_visited"] == oc["nodes_visited"] + oa["nodes_visited"]
    assert gc["tris_tested"] == oc["tris_tested"] + oa["tris_tested"]
    return int(hit.sum()), int(ro.sum())


SCENES = {
    # k_trace (pooled leaf rounds); the Cornell light is one-sided, so the leaf rounds carry the ray direction
    "cornell": lambda: scenes.cornell_box(res=64, spp=16),
    "rt20k": lambda: scenes.rt1m(20000, res=64, spp=8),
    "rt4k_hlbvh_leaf2": lambda: fs.scene_accel("hlbvh", 2),
    # k_trace_seq: leaves of more than 8 triangles are walked by the owning lane
    "rt4k_leaf12": lambda: fs.scene_accel("sah", 12),
    "rt4k_equal_leaf40": lambda: fs.scene_accel("equal", 40),
    # k_trace_sph_dist / k_trace_sph: analytic spheres next to triangles
    "spheres": lambda: fs.scene_spheres(),
    # k_trace_inst: object instances (nested traversal on the lane's stack)
    "instances": lambda: fs.scene_instances(),
}


@pytest.fixture(scope="module", params=list(SCENES))
def pair(request, gpu_ctx, oracle):
    sd = SCENES[request.param]()
    osc = oracle.scene(sd)
    gpu_ctx.upload(sd)
    yield request.param, sd, gpu_ctx, osc
    osc.close()


def test_wavefront_mixed_kinds_exact(pair):
    name, sd, ctx, osc = pair
    # probe results of an instanced scene keep the inner record (the renderer only compares them with light records,
    # and lights inside objects are dropped): primitives are compared for continuation rays there; the hooks report no
    # barycentrics for a hit inside an instance (pt_trace_closest does not either)
    inst = name == "instances"
    o, d, t, kind = _rays(ctx, osc, 150001, 21)          # not a multiple of 64
    n_hit, n_occ = _check(ctx, osc, o, d, t, kind, probe_prims=not inst, barycentrics=not inst)
    assert n_hit > 1000 and n_occ > 1000


@pytest.mark.parametrize("n", [1, 63, 100, 513, 4097])
def test_wavefront_small_batches_leave_segments_empty(pair, n):
    """Totals far below 8 x 64 items: most of the eight per-XCD queue segments are empty and every wave has to walk past
    dry segments (the seg_dry path) before it finds work or retires."""
    name, sd, ctx, osc = pair
    o, d, t, kind = _rays(ctx, osc, max(n, 3), 31 + n)
    o, d, t, kind = o[:n], d[:n], t[:n], kind[:n]
    _check(ctx, osc, o, d, t, kind, probe_prims=name != "instances", barycentrics=name != "instances")


@pytest.mark.parametrize("kinds", [(1,), (2,), (3,), (1, 2)])
def test_wavefront_single_kind(pair, kinds):
    """One kind only: the other two queues are empty (a first bounce has no next-event work; a last one has only that)."""
    name, sd, ctx, osc = pair
    o, d, t, kind = _rays(ctx, osc, 20011, 41, kinds=kinds)
    _check(ctx, osc, o, d, t, kind, probe_prims=name != "instances", barycentrics=name != "instances")


def test_wavefront_rt1m_2m_rays(gpu_ctx, oracle):
    """BASELINE config 2's scene at full size, 2.1 M mixed work items in one launch (every segment many tickets long,
    lanes refilled thousands of times, leaf rounds packed from up to 64 parked lanes)."""
    sd = scenes.rt1m(1000000, res=1024, spp=4)
    osc = oracle.scene(sd)
    gpu_ctx.upload(sd)
    o, d, t, kind = _rays(gpu_ctx, osc, 2100037, 51)
    n_hit, n_occ = _check(gpu_ctx, osc, o, d, t, kind)
    assert n_hit > 100000 and n_occ > 100000
    osc.close()


def test_wavefront_rt16m_beyond_the_infinity_cache(gpu_ctx, oracle):
    """16 M triangles: nodes + leaf records are ~1.1 GB, four times the 256 MiB Infinity Cache, so node and record fetches really
    come from HBM (the regime BASELINE config 5 is written for; the tree is ~4 levels deeper than RT1M's).  300 k mixed work items,
    every hit / t / barycentric / occlusion flag and the node / triangle counters equal to the oracle's."""
    sd = scenes.rt1m(16000000, res=1024, spp=4)
    osc = oracle.scene(sd)
    info = gpu_ctx.upload(sd)
    assert 128 * info.n_nodes + 48 * 16000000 > 4 * 256 * 2 ** 20
    o, d, t, kind = _rays(gpu_ctx, osc, 300007, 61)
    n_hit, n_occ = _check(gpu_ctx, osc, o, d, t, kind)
    assert n_hit > 10000 and n_occ > 10000
    osc.close()


def test_top_of_tree_is_served_from_lds(gpu_ctx, oracle):
    """The upload numbers the top of the world tree breadth-first and k_trace keeps its first nodes in LDS: the visits counted as
    served from there are a real share of all node visits (every ray starts at the root), never more than all of them, and the
    owner-walks-the-leaf kernel (leaves of more than 8 triangles), which has no such copy, reports none.  What the rays hit is
    covered by the parity tests above -- the renumbering moves nodes, it does not change a reference's meaning."""
    for make, expect in ((SCENES["rt20k"], True), (SCENES["rt4k_leaf12"], False)):
        sd = make()
        osc = oracle.scene(sd)
        gpu_ctx.upload(sd)
        o, d, t, kind = _rays(gpu_ctx, osc, 30011, 77)
        _check(gpu_ctx, osc, o, d, t, kind)
        c = gpu_ctx.counters()
        if expect:
            assert 0.05 * c["nodes_visited"] < c["nodes_from_lds"] <= c["nodes_visited"]
        else:
            assert c["nodes_from_lds"] == 0
        osc.close()
