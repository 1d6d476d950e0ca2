"""Materials beyond matte (plastic, mirror, glass, metal, uber, substrate): CPU checks of the oracle's
BxDFs -- the reference's chi^2 sampling test (tests/bsdfs.rs:369-419) restated, closed-form Fresnel values,
energy bounds -- and, on the GPU, bit-exact BSDF evaluation / sampling and rendered parity."""
import ctypes as C

import numpy as np
import pytest

import feature_scenes as fs
from helpers import bits, rel_l2

ALL, NOSPEC = 31, 31 & ~16
REFL, TRANS, DIFFUSE, GLOSSY, SPECULAR = 1, 2, 4, 8, 16


@pytest.fixture(scope="module")
def palette(oracle):
    sd = fs.scene_material_palette()
    sc = oracle.scene(sd)
    yield sd, sc
    sc.close()


def sphere_dirs(rng, n):
    v = rng.standard_normal((n, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(np.float32)
    return v.astype(np.float32)


def hemi_dirs(rng, n):
    v = sphere_dirs(rng, n)
    v[:, 2] = np.abs(v[:, 2])
    return v


# ------------------------------------------------------------------ closed forms
def test_roughness_to_alpha_and_fresnel_closed_forms(palette, oracle):
    """trowbridge_reitz.rs:104-113 polynomial in ln(roughness); mirror = FresnelNoOp (f * cos = Kr);
    smooth glass at normal incidence reflects ((eta-1)/(eta+1))^2."""
    sd, sc = palette
    for r in (1e-4, 0.01, 0.1, 0.5, 1.0):
        x = np.log(np.float64(max(r, 1e-3)))
        want = 1.62142 + 0.819955 * x + 0.1734 * x * x + 0.0171201 * x ** 3 + 0.000640711 * x ** 4
        assert abs(oracle.lib.orc_roughness_to_alpha(r) - want) < 2e-6
    wo = hemi_dirs(np.random.default_rng(1), 64)
    u = np.random.default_rng(2).random((64, 2), dtype=np.float32)
    f, wi, pdf, t = sc.bsdf_sample(sd.material_index["mirror"], wo, u)
    assert (t == (REFL | SPECULAR)).all() and (pdf == 1.0).all()
    assert np.allclose(wi, wo * np.array([-1, -1, 1], np.float32))
    assert np.allclose(f * np.abs(wi[:, 2:3]), np.array([0.9, 0.85, 0.8], np.float32), rtol=1e-6)
    # glass, normal incidence: reflection chosen with probability R0 = 0.04
    wo0 = np.tile(np.array([[0, 0, 1]], np.float32), (2, 1))
    f, wi, pdf, t = sc.bsdf_sample(sd.material_index["glass"], wo0, np.array([[0.01, 0.5], [0.9, 0.5]], np.float32))
    assert t[0] == (REFL | SPECULAR) and abs(pdf[0] - 0.04) < 1e-6
    assert t[1] == (TRANS | SPECULAR) and abs(pdf[1] - 0.96) < 1e-6 and np.allclose(wi[1], [0, 0, -1])
    assert np.allclose(f[1], 0.96 / 2.25, rtol=1e-5)           # (1 - F) * (eta_i / eta_t)^2 / |cos|


def test_no_bsdf_and_lobe_counts(palette):
    """glass with Kr = Kt = 0 has no BSDF (glass.rs:65-67); specular-only materials have no non-specular lobes."""
    sd, sc = palette
    wo = hemi_dirs(np.random.default_rng(3), 16)
    u = np.random.default_rng(4).random((16, 2), dtype=np.float32)
    for kind in ("glass_black",):
        f, wi, pdf, t = sc.bsdf_sample(sd.material_index[kind], wo, u)
        assert (t == 0).all()
    for kind in ("mirror", "glass"):
        f, wi, pdf, t = sc.bsdf_sample(sd.material_index[kind], wo, u, NOSPEC)
        assert (t == 0).all()
        fv, pv = sc.bsdf_eval(sd.material_index[kind], wo, sphere_dirs(np.random.default_rng(5), 16), ALL)
        assert (fv == 0).all() and (pv == 0).all()


# ------------------------------------------------------------------ reference chi^2 test, restated
def chi2_bsdf(sc, mat, rng, runs=3, theta_res=10, phi_res=20, samples=200000, sub=12):
    """tests/bsdfs.rs:197-419: histogram of sampled wi over (theta, phi) cells against the integral of
    BSDF::pdf over the cells; cells pooled until the expected count reaches 5; significance 0.01 with the
    Sidak correction over the runs."""
    from scipy import stats
    worst = 1.0
    for _ in range(runs):
        s2 = rng.random(2, dtype=np.float32)
        r, ph = np.sqrt(s2[0]), 2 * np.pi * s2[1]
        wo = np.array([[r * np.cos(ph), r * np.sin(ph), np.sqrt(max(0.0, 1 - r * r))]], np.float32)
        u = rng.random((samples, 2), dtype=np.float32)
        f, wi, pdf, t = sc.bsdf_sample(mat, np.repeat(wo, samples, 0), u)
        ok = (t != 0) & ((t & SPECULAR) == 0) & (pdf > 0) & (f != 0).any(axis=1)
        w = wi[ok]
        th = np.arccos(np.clip(w[:, 2], -1, 1)) * (theta_res / np.pi)
        p = np.arctan2(w[:, 1], w[:, 0])
        p[p < 0] += 2 * np.pi
        tb = np.clip(np.floor(th).astype(int), 0, theta_res - 1)
        pb = np.clip(np.floor(p * (phi_res / (2 * np.pi))).astype(int), 0, phi_res - 1)
        freq = np.bincount(tb * phi_res + pb, minlength=theta_res * phi_res).astype(np.float64)
        # expected: midpoint rule on a sub x sub grid per cell
        tt = (np.arange(theta_res * sub) + 0.5) * (np.pi / (theta_res * sub))
        pp = (np.arange(phi_res * sub) + 0.5) * (2 * np.pi / (phi_res * sub))
        T, P = np.meshgrid(tt, pp, indexing="ij")
        d = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1).reshape(-1, 3).astype(np.float32)
        _, pv = sc.bsdf_eval(mat, np.repeat(wo, len(d), 0), d, ALL)
        dens = (pv.astype(np.float64).reshape(T.shape) * np.sin(T)) * (np.pi / (theta_res * sub)) * (2 * np.pi / (phi_res * sub))
        exp = dens.reshape(theta_res, sub, phi_res, sub).sum(axis=(1, 3)).reshape(-1) * samples
        # the sampler may fail (None) for part of the domain: compare conditional distributions
        order = np.argsort(exp)
        e, o = exp[order], freq[order]
        pooled_e, pooled_o, dof, chi = 0.0, 0.0, 0, 0.0
        for ei, oi in zip(e, o):
            if ei == 0:
                assert oi <= samples * 1e-5, "samples in a zero-probability cell"
                continue
            if ei < 5:
                pooled_e += ei; pooled_o += oi
                continue
            chi += (oi - ei) ** 2 / ei
            dof += 1
        if pooled_e > 0:
            chi += (pooled_o - pooled_e) ** 2 / pooled_e
            dof += 1
        pval = 1.0 - stats.chi2.cdf(chi, dof - 1)
        worst = min(worst, pval)
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / runs)
    return worst, alpha


@pytest.mark.parametrize("kind", ["matte", "oren_nayar", "plastic", "plastic_noremap", "metal", "metal_aniso", "substrate", "substrate_aniso",
                                  "glass_reflect_only"])
def test_bsdf_sampling_matches_pdf_chi2(palette, kind):
    """The sampled directions of every non-specular BSDF follow its pdf (the reference's own acceptance test
    for BxDF sampling, bsdfs.rs:482-606, here through the material -> BSDF path)."""
    sd, sc = palette
    rng = np.random.default_rng(sum(ord(c) for c in kind) + 7)      # fixed per kind: the test is deterministic
    pval, alpha = chi2_bsdf(sc, sd.material_index[kind], rng)
    assert pval > alpha, (kind, pval, alpha)


def test_rough_glass_sample_and_eval_agree(palette):
    """MicrofacetTransmission (microfacet.rs:102-263).  Its pdf() keeps no `wo . wh < 0` guard, so it also counts
    half vectors the sampler never produces and integrates to slightly more than one at grazing angles (1.08 at
    54 degrees for these roughnesses) -- the reference's chi^2 suite leaves it out for that reason, and so do we.
    What must hold: the pdf and f reported by sample_f are the ones pdf()/f() return for the sampled direction."""
    sd, sc = palette
    rng = np.random.default_rng(11)
    mat = sd.material_index["glass_rough"]
    wo = sphere_dirs(rng, 50000)
    f, wi, pdf, t = sc.bsdf_sample(mat, wo, rng.random((50000, 2), dtype=np.float32))
    ok = t != 0
    assert 0.5 < ok.mean() <= 1.0
    assert set(np.unique(t[ok])) == {REFL | GLOSSY, TRANS | GLOSSY}
    fe, pe = sc.bsdf_eval(mat, wo[ok], wi[ok], ALL)
    # transmission lobe: sample_f calls pdf()/f() with the very same operands -> identical up to the two-lobe average
    assert np.allclose(pe, pdf[ok], rtol=2e-5, atol=1e-7)
    assert np.allclose(fe, f[ok], rtol=2e-4, atol=1e-6)
    tr = (t[ok] & TRANS) != 0
    assert ((wi[ok][tr, 2] * wo[ok][tr, 2]) < 0).all() and ((wi[ok][~tr, 2] * wo[ok][~tr, 2]) > 0).all()


def test_white_furnace_bounds(palette):
    """E[f cos / pdf] over sample_f stays <= 1 (+ noise) for energy-conserving parameter sets."""
    sd, sc = palette
    rng = np.random.default_rng(21)
    for kind in ("plastic", "substrate", "uber", "glass", "glass_rough", "mirror", "matte"):
        wo = np.repeat(hemi_dirs(rng, 1), 100000, 0)
        f, wi, pdf, t = sc.bsdf_sample(sd.material_index[kind], wo, rng.random((100000, 2), dtype=np.float32))
        ok = t != 0
        est = np.zeros((100000, 3))
        est[ok] = f[ok] * np.abs(wi[ok, 2:3]) / pdf[ok, None]
        albedo = est.mean(0)
        assert (albedo <= 1.02).all() and (albedo >= 0).all(), (kind, albedo)


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
def test_gpu_bsdf_eval_and_sample_bit_exact(gpu_ctx, palette):
    """Every lobe kind, evaluated and sampled on the device, equals the oracle bit for bit:
    f, pdf, wi, sampled type (incl. sin/cos of the normal-incidence branch and powf(x, 5) of FresnelBlend)."""
    sd, sc = palette
    gpu_ctx.upload(sd)
    rng = np.random.default_rng(31)
    n = 40000
    for kind, mat in sd.material_index.items():
        wo, wi = sphere_dirs(rng, n), sphere_dirs(rng, n)
        wo[:200, 2] = 1.0; wo[:200, :2] *= 1e-3           # near-normal incidence (TR sample_11 special case)
        wo[:200] /= np.linalg.norm(wo[:200], axis=1, keepdims=True)
        wo[200:300, 2] *= 1e-4                               # grazing
        wo[300] = (1, 0, 0)                                  # wo.z == 0
        wi[:100] = wo[:100] * np.array([-1, -1, 1], np.float32)   # mirror direction
        wi[100:150] = -wo[100:150]                           # wh = 0
        u = rng.random((n, 2), dtype=np.float32)
        u[:50, 0] = 0.0; u[50:100, 1] = 0.0; u[100:150, 0] = np.float32(0.99999994)
        for flags in (ALL, NOSPEC, REFL | GLOSSY | DIFFUSE | SPECULAR):
            gf, gp = gpu_ctx.bsdf_eval(mat, wo, wi, flags)
            of, op = sc.bsdf_eval(mat, wo, wi, flags)
            assert np.array_equal(bits(gf), bits(of)), (kind, flags, int((bits(gf) != bits(of)).any(axis=1).sum()))
            assert np.array_equal(bits(gp), bits(op)), (kind, flags, int((bits(gp) != bits(op)).sum()))
            g = gpu_ctx.bsdf_sample(mat, wo, u, flags)
            o = sc.bsdf_sample(mat, wo, u, flags)
            assert np.array_equal(g[3], o[3]), (kind, flags, "sampled type")
            for a, b, what in zip(g[:3], o[:3], ("f", "wi", "pdf")):
                bad = bits(a) != bits(b)
                assert not bad.any(), (kind, flags, what, int(bad.sum()))


RENDER_SETS = [
    ("plastic_mirror_glass", ["plastic", "mirror", "glass"], "sobol"),
    ("metal_uber_substrate", ["metal", "uber", "substrate"], "sobol"),
    ("roughglass_translucent_aniso", ["glass_rough", "uber_translucent", "metal_aniso"], "halton"),
    ("glass_black_passthrough", ["glass_black", "substrate_aniso", "plastic_noremap"], "sobol"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kinds,sampler", RENDER_SETS)
def test_gpu_material_scene_parity(gpu_ctx, oracle, name, kinds, sampler):
    """PathIntegrator::li with specular bounces, eta_scale, MIS over multi-lobe BSDFs: per-sample radiance
    identical to the oracle, image within the 1e-3 rel-L2 bar, same ray counts."""
    sd = fs.scene_materials_render(kinds, sampler=sampler)
    osc = oracle.scene(sd)
    gpu_ctx.upload(sd)
    sb = list(gpu_ctx.info.sample_bounds)
    tile = (sb[0] + 8, sb[1] + 8, sb[0] + 32, sb[1] + 32)
    g = gpu_ctx.radiance_samples(tile)
    r = osc.radiance_samples(tile)
    same = np.all(bits(g) == bits(r), axis=-1)
    print("\n[%s] per-sample radiance: %d samples, %.4f%% not bit-identical, rel-L2 %.3e" % (name, same.size, 100 * (1 - same.mean()), rel_l2(g, r)))
    assert r.sum() > 0
    assert same.all()
    gpu_ctx.film_clear(); gpu_ctx.reset_counters(); gpu_ctx.render()
    grgb, gc = gpu_ctx.film_rgb(), gpu_ctx.counters()
    ox, oc, _ = osc.render(threads=8)
    assert rel_l2(grgb, osc.resolve_rgb(ox)) <= 1e-3
    for k in ("regular_rays", "shadow_rays", "path_vertices", "nodes_visited", "tris_tested"):
        assert gc[k] == oc[k], (k, gc[k], oc[k])
    osc.close()


@pytest.mark.gpu
def test_gpu_many_materials_share_sort_bins(gpu_ctx, oracle):
    """More materials than shade-queue bins (128 per class): the overflow shares the class's last bin;
    200 plastics + 150 mattes render exactly as on the CPU."""
    b = fs.base(res=32, spp=4, depth=4)
    fs.room(b)
    rng = np.random.default_rng(5)
    for k in range(350):
        c = rng.random(3) * 3.2 - 1.6
        kd = tuple(0.2 + 0.6 * rng.random(3))
        if k < 200:
            b.material_plastic(Kd=kd, Ks=(0.3, 0.3, 0.3), roughness=0.05 + 0.3 * rng.random())
        else:
            b.material_matte(kd, sigma=float(rng.integers(0, 2)) * 20.0)
        b.shape_trianglemesh([tuple(c), tuple(c + [0.45, 0, 0.1]), tuple(c + [0, 0.45, 0.05])], [0, 1, 2])
    sd = b.build()
    assert sd.desc.n_materials > 256
    osc = oracle.scene(sd)
    gpu_ctx.upload(sd)
    sb = list(gpu_ctx.info.sample_bounds)
    tile = (sb[0], sb[1], sb[2], sb[3])
    g = gpu_ctx.radiance_samples(tile)
    r = osc.radiance_samples(tile)
    assert r.sum() > 0 and np.array_equal(bits(g), bits(r))
    osc.close()
