"""Integrator "ao" (integrators/ao.rs): the oracle's restatement against what the estimator must return analytically, the front
end, and -- on the GPU -- the wavefront AO pipeline against the oracle per sample."""
import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, pkg, rel_l2, scenes

capi = pkg.capi


def _open_floor(cossample, nsamples, sampler):
    b = scenes.SceneBuilder()
    b.look_at((0, 2, -3), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=30.0)
    b.film(xresolution=8, yresolution=8); b.pixel_filter_box()
    b.sampler_sobol(4) if sampler == "sobol" else b.sampler_halton(4)
    b.integrator_ao(nsamples=nsamples, cossample=cossample)
    b.material_matte((0.5, 0.5, 0.5))
    b.shape_trianglemesh([(50, 0, -50), (-50, 0, -50), (-50, 0, 50), (50, 0, 50)], [0, 1, 2, 0, 2, 3])
    return b.build()


@pytest.mark.parametrize("sampler", ["sobol", "halton"])
def test_oracle_unoccluded_surface_returns_pi(oracle, sampler):
    """Nothing above an infinite-looking floor: every term is cos / (cos / pi) / n, so L = pi up to rounding for cosine sampling
    (ao.rs:80-99 has no 1 / pi albedo factor); with uniform sampling the terms are 2 pi cos / n and only their mean is pi."""
    osc = oracle.scene(_open_floor(True, 16, sampler))
    r = osc.radiance_samples((2, 2, 6, 6))
    assert np.allclose(r, np.pi, rtol=2e-6, atol=0)
    osc.close()
    osc = oracle.scene(_open_floor(False, 256, sampler))
    r = osc.radiance_samples((2, 2, 6, 6))
    assert abs(float(r.mean()) - np.pi) < 0.02 and r.std() > 0
    osc.close()


def test_oracle_array_samples_are_the_documented_slices(oracle):
    """get_2d_array(n) for pixel sample s is sample numbers [s n, s n + n) at dimensions 5 and 6 (sobol.rs:60-75,
    base_sampler.rs:59-70): with n = 1 the occlusion direction of camera sample s comes from the same Sobol' point as its
    film / lens sample, so two renders that differ only in spp share their first samples exactly."""
    a = oracle.scene(fs.scene_ao(nsamples=1, spp=2, res=24))
    b = oracle.scene(fs.scene_ao(nsamples=1, spp=4, res=24))
    ra, rb = a.radiance_samples((8, 4, 12, 8)), b.radiance_samples((8, 4, 12, 8))
    assert np.array_equal(bits(ra[:, :2]), bits(rb[:, :2]))
    a.close(); b.close()


def test_front_end_ao_integrator(tmp_path):
    text = '''
    Integrator "ao" "integer nsamples" 24 "bool cossample" "false"
    Sampler "sobol" "integer pixelsamples" 2
    WorldBegin
      Material "matte"
      Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]
    WorldEnd
    '''
    ps = capi.ParsedScene(text=text, work_dir=str(tmp_path))
    d = ps.desc
    assert (d.integrator, d.ao_samples, d.ao_cos_sample) == (capi.PT_INTEGRATOR_AO, 24, 0)
    ps2 = capi.ParsedScene(text=text.replace('"integer nsamples" 24 "bool cossample" "false"', ""), work_dir=str(tmp_path))
    assert (ps2.desc.integrator, ps2.desc.ao_samples, ps2.desc.ao_cos_sample) == (capi.PT_INTEGRATOR_AO, 64, 1)
    with pytest.raises(capi.PtError) as e:
        capi.ParsedScene(text=text.replace('"ao"', '"bdpt"'), work_dir=str(tmp_path))
    assert "only path, ao, directlighting and whitted" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", [
    ("sobol_cos", lambda: fs.scene_ao()),
    ("halton_uniform", lambda: fs.scene_ao(sampler="halton", cossample=False, nsamples=7, spp=3)),
    ("spheres", lambda: fs.scene_ao(kind="spheres", nsamples=8)),
    ("instances", lambda: fs.scene_ao(kind="instances", nsamples=8)),
    ("default_64", lambda: fs.scene_ao(nsamples=64, spp=2, res=32)),
])
def test_gpu_ao_matches_oracle(oracle, name, make):
    sd = make()
    ctx = pkg.Context(0)
    osc = oracle.scene(sd)
    try:
        info = ctx.upload(sd)
        sb = list(info.sample_bounds)
        cx, cy = (sb[0] + sb[2]) // 2, (sb[1] + sb[3]) // 2
        tile = (cx - 8, cy - 8, cx + 8, cy + 8)
        gs, rs = ctx.radiance_samples(tile), osc.radiance_samples(tile)
        assert np.array_equal(bits(gs), bits(rs))                      # per-sample radiance: bit-identical
        ctx.film_clear(); ctx.reset_counters(); ctx.render()
        gx, grgb, gc = ctx.film_xyzw(), ctx.film_rgb(), ctx.counters()
        ox, oc, _ = osc.render(threads=8)
        assert rel_l2(grgb, osc.resolve_rgb(ox)) <= 1e-3                 # north_star tolerance
        assert np.array_equal(bits(gx[..., 3]), bits(ox[..., 3]))       # box filter: the weights are bit-identical
        assert np.allclose(gx[..., :3], ox[..., :3], rtol=1e-5, atol=1e-6)
        for k in ("camera_rays", "regular_rays", "shadow_rays", "path_vertices", "nodes_visited", "tris_tested"):
            assert gc[k] == oc[k], (k, gc[k], oc[k])
    finally:
        osc.close(); ctx.close()


@pytest.mark.gpu
def test_gpu_ao_refuses_what_the_reference_cannot_render():
    b = scenes.SceneBuilder()
    b.look_at((0, 0, -3), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=30.0)
    b.film(xresolution=8, yresolution=8); b.pixel_filter_box(); b.sampler_sobol(1); b.integrator_ao(nsamples=4)
    b.material_none()
    b.shape_trianglemesh([(1, -1, 0), (-1, -1, 0), (0, 1, 0)], [0, 1, 2])
    ctx = pkg.Context(0)
    try:
        with pytest.raises(pkg.PtError) as e:
            ctx.upload(b.build())
        assert "without a material" in str(e.value)
    finally:
        ctx.close()
