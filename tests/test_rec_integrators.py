"""Integrators "directlighting" and "whitted" (integrators/directlighting.rs, integrators/whitted.rs, with specular_reflect /
specular_transmit of core/integrator/sampler.rs): the oracle's restatement against closed forms and against each other, the front
end, and -- on the GPU -- the depth-first wavefront walk against the oracle per camera sample."""
import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, pkg, rel_l2, scenes

capi = pkg.capi


def _with(sd, integrator, strategy="all", maxdepth=5):
    sd.desc.integrator = {"directlighting": capi.PT_INTEGRATOR_DIRECTLIGHTING, "whitted": capi.PT_INTEGRATOR_WHITTED}[integrator]
    sd.desc.direct_strategy = capi.PT_DIRECT_ONE if strategy == "one" else capi.PT_DIRECT_ALL
    sd.desc.max_depth = maxdepth
    return sd


def _nsamples(sd, n):
    """AreaLightSource "nsamples" n on every emissive shape: the size of the sample arrays DirectLighting's "all" strategy gives each of its
    lights (one light per emissive triangle / sphere)."""
    for i in range(sd.desc.n_area_lights):
        sd.desc.area_lights[i].n_samples = n
    return sd


SCENES = {
    # diffuse only: no tree, one node per camera sample
    "cornell_dl_all": lambda: _with(scenes.cornell_box(res=40, spp=8), "directlighting"),
    "cornell_dl_one_halton": lambda: _with(scenes.cornell_box(res=40, spp=6, sampler="halton"), "directlighting", "one"),
    "cornell_whitted": lambda: _with(scenes.cornell_box(res=40, spp=8), "whitted"),
    # "nsamples" > 1: arrays of n estimates per light, drawn at sample numbers s n + k (not a power of two with Halton)
    "cornell_dl_all_ns4": lambda: _nsamples(_with(scenes.cornell_box(res=40, spp=4), "directlighting"), 4),
    "specular_dl_all_ns3_halton": lambda: _nsamples(_with(fs.scene_materials_render(["glass", "mirror", "plastic"], spp=3, sampler="halton"), "directlighting", maxdepth=4), 3),
    "spheres_dl_all_ns2": lambda: _nsamples(_with(fs.scene_spheres(spp=4), "directlighting", maxdepth=3), 2),
    "attributes_dl_all_ns5": lambda: _nsamples(_with(fs.scene_attributes(), "directlighting", maxdepth=3), 5),     # surfaces without a material use the arrays up
    # glass + mirror + plastic: reflect and transmit subtrees, the sampler consumed depth first
    "specular_dl_all": lambda: _with(fs.scene_materials_render(["glass", "mirror", "plastic"], spp=8), "directlighting", maxdepth=5),
    "specular_dl_one": lambda: _with(fs.scene_materials_render(["glass", "mirror", "uber_translucent"], spp=8), "directlighting", "one", maxdepth=4),
    "specular_whitted": lambda: _with(fs.scene_materials_render(["glass", "mirror", "metal"], spp=8), "whitted", maxdepth=6),
    "specular_dl_depth1": lambda: _with(fs.scene_materials_render(["glass", "mirror", "plastic"], spp=4), "directlighting", maxdepth=1),
    # differentials through the specular bounces feed the texture filters; spheres; thin lens
    "textures_dl": lambda: _with(fs.scene_textures(spp=4), "directlighting", maxdepth=4),
    "textures_whitted_lens": lambda: _with(fs.scene_textures(spp=4, lens=True), "whitted", maxdepth=4),
    "spheres_dl": lambda: _with(fs.scene_spheres(spp=4), "directlighting", maxdepth=5),
    "imagemaps_dl": lambda: _with(fs.scene_imagemaps(spp=4), "directlighting", maxdepth=3),
    "bump_whitted": lambda: _with(fs.scene_bump(spp=4), "whitted", maxdepth=3),
    # instances (hits rebuilt in instance space) and surfaces without a material (directlighting passes through, whitted stops)
    "instances_dl": lambda: _with(fs.scene_instances(spp=4), "directlighting", maxdepth=4),
    "attributes_dl": lambda: _with(fs.scene_attributes(), "directlighting", "one", maxdepth=3),
    "attributes_whitted": lambda: _with(fs.scene_attributes(), "whitted", maxdepth=3),
}


def test_oracle_direct_strategies_agree_in_the_mean(oracle):
    """"all" and "one" estimate the same integral; Whitted's unweighted light sampling too (it has no emitted term, so compare away
    from the lamp)."""
    imgs = {}
    for name, args in (("all", ("directlighting", "all")), ("one", ("directlighting", "one")), ("whitted", ("whitted",))):
        osc = oracle.scene(_with(scenes.cornell_box(res=24, spp=64), *args))
        x, _, _ = osc.render(threads=8)
        imgs[name] = osc.resolve_rgb(x)
        osc.close()
    lower = slice(8, 24)          # rows below the lamp
    a, o, w = (imgs[k][lower].mean() for k in ("all", "one", "whitted"))
    assert abs(a - o) < 0.02 * a and abs(a - w) < 0.03 * a
    assert imgs["whitted"][:3].max() < imgs["all"][:3].max()          # the lamp itself is black for Whitted


def test_oracle_light_sample_count_keeps_the_mean(oracle):
    """uniform_sample_all_lights with arrays of four estimates per light (sample_lights.rs:43-57: their mean) against one estimate per
    light: the same image up to noise -- and less noise."""
    imgs = []
    for n in (1, 4):
        osc = oracle.scene(_nsamples(_with(scenes.cornell_box(res=24, spp=16), "directlighting", maxdepth=1), n))
        x, c, _ = osc.render(threads=8)
        imgs.append(osc.resolve_rgb(x))
        assert c["shadow_rays"] > 0
        osc.close()
    a, b = imgs
    assert abs(a.mean() - b.mean()) <= 0.02 * a.mean()
    assert not np.array_equal(a, b)


def test_oracle_depth_one_is_local_lighting(oracle):
    """maxdepth 1: depth + 1 < maxdepth never holds, no specular recursion and no sampler use beyond the node's own light samples."""
    sd = _with(fs.scene_materials_render(["glass", "mirror", "plastic"], spp=4), "directlighting", maxdepth=1)
    osc = oracle.scene(sd)
    x, c, _ = osc.render(threads=8)
    assert c["regular_rays"] >= c["camera_rays"] and c["path_vertices"] <= c["camera_rays"]
    osc.close()


def test_front_end_directlighting_and_whitted(tmp_path):
    text = '''
    Integrator "directlighting" "integer maxdepth" 7 "string strategy" "one"
    Sampler "sobol" "integer pixelsamples" 2
    WorldBegin
      AttributeBegin
        AreaLightSource "diffuse" "rgb L" [1 1 1]
        Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 2 0 1 2 0 0 2 1]
      AttributeEnd
      Material "matte"
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]
    WorldEnd
    '''
    d = capi.ParsedScene(text=text, work_dir=str(tmp_path)).desc
    assert (d.integrator, d.direct_strategy, d.max_depth) == (capi.PT_INTEGRATOR_DIRECTLIGHTING, capi.PT_DIRECT_ONE, 7)
    d = capi.ParsedScene(text=text.replace('"integer maxdepth" 7 "string strategy" "one"', ""), work_dir=str(tmp_path)).desc
    assert (d.integrator, d.direct_strategy, d.max_depth) == (capi.PT_INTEGRATOR_DIRECTLIGHTING, capi.PT_DIRECT_ALL, 5)
    d = capi.ParsedScene(text=text.replace('"directlighting" "integer maxdepth" 7 "string strategy" "one"', '"whitted"'), work_dir=str(tmp_path)).desc
    assert (d.integrator, d.max_depth) == (capi.PT_INTEGRATOR_WHITTED, 5)
    # the "all" strategy sizes its sample arrays by the lights' "nsamples" (ABI 8 carries it per area light source)
    d = capi.ParsedScene(text=text.replace('"string strategy" "one"', "").replace('"rgb L" [1 1 1]', '"rgb L" [1 1 1] "integer nsamples" 4'), work_dir=str(tmp_path)).desc
    assert d.n_area_lights == 1 and d.area_lights[0].n_samples == 4
    assert capi.ParsedScene(text=text, work_dir=str(tmp_path)).desc.area_lights[0].n_samples == 1
    with pytest.raises(capi.PtError) as e:        # the reference divides the summed estimates by it
        capi.ParsedScene(text=text.replace('"rgb L" [1 1 1]', '"rgb L" [1 1 1] "integer nsamples" 0'), work_dir=str(tmp_path))
    assert "nsamples" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_gpu_rec_integrators_match_oracle(oracle, name):
    sd = SCENES[name]()
    ctx = pkg.Context(0)
    osc = oracle.scene(sd)
    try:
        info = ctx.upload(sd)
        sb = list(info.sample_bounds)
        cx, cy = (sb[0] + sb[2]) // 2, (sb[1] + sb[3]) // 2
        tile = (cx - 8, cy - 8, cx + 8, cy + 8)
        gs, rs = ctx.radiance_samples(tile), osc.radiance_samples(tile)
        assert rs.sum() > 0
        same = np.all(bits(gs) == bits(rs), axis=-1)
        assert same.all(), "%d of %d camera samples differ" % ((~same).sum(), same.size)      # per-sample radiance: bit-identical
        ctx.film_clear(); ctx.reset_counters(); ctx.render()
        gx, grgb, gc = ctx.film_xyzw(), ctx.film_rgb(), ctx.counters()
        ox, oc, _ = osc.render(threads=8)
        assert rel_l2(grgb, osc.resolve_rgb(ox)) <= 1e-3                 # north_star tolerance
        for k in ("camera_rays", "regular_rays", "shadow_rays", "path_vertices", "nodes_visited", "tris_tested"):
            assert gc[k] == oc[k], (k, gc[k], oc[k])
    finally:
        osc.close(); ctx.close()
