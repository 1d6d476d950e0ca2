#!/usr/bin/env python3
"""Golden vectors produced BY THE ORACLE (the reference cannot run here: no Rust toolchain,
and it holds no image fixtures for this path -- SURVEY.md section 8c).  They pin the oracle
against accidental change and give the GPU tests a committed target:
  tests/golden/cornell_32x32_8spp_xyzw.npy   film {X,Y,Z,weight}
  tests/golden/cornell_32x32_8spp_rays.npz   camera rays + closest hits for sample 0
  tests/golden/rt2k_32x32_4spp_xyzw.npy
  tests/golden/materials_halton_40x40_6spp_xyzw.npy   plastic sphere + mirror + glass slabs, Halton, HLBVH
  tests/golden/materials_sobol_40x40_8spp_samples.npy per-sample radiance of a 12x12 tile: metal / uber / substrate
  tests/golden/{directlighting_all_ns3,whitted_depth4,ao_16cos_cornell}_*.npz   the other integrators: film, per-sample radiance of the
                                                      middle tile, ray counters (tests/feature_scenes.py::GOLDEN_INTEGRATORS)
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
pkg = importlib.import_module("pbrt-r3_amd")
G = os.path.join(ROOT, "tests", "golden")
orc = oracle_lib.load()

sd = pkg.scenes.cornell_box(res=32, spp=8)
sc = orc.scene(sd)
xyzw, cnt, _ = sc.render(threads=1)
np.save(os.path.join(G, "cornell_32x32_8spp_xyzw.npy"), xyzw)
sb = list(sc.info.sample_bounds)
ys, xs = np.mgrid[sb[1]:sb[3], sb[0]:sb[2]]
px = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
o, d, pf = sc.generate_camera_rays(px, np.zeros(len(px), np.uint32))
hits, _ = sc.trace_closest(o, d, np.full(len(px), np.inf, np.float32))
np.savez_compressed(os.path.join(G, "cornell_32x32_8spp_rays.npz"), pixel=px, o=o, d=d, p_film=pf, t=hits["t"], prim=hits["prim"],
                    b0=hits["b0"], b1=hits["b1"], counters=np.array([cnt[k] for k in ("camera_rays", "regular_rays", "shadow_rays", "path_vertices")], np.int64))
sd2 = pkg.scenes.rt1m(2000, res=32, spp=4)
sc2 = orc.scene(sd2)
xyzw2, _, _ = sc2.render(threads=1)
np.save(os.path.join(G, "rt2k_32x32_4spp_xyzw.npy"), xyzw2)
import feature_scenes as fs
sd3 = fs.scene_materials_render(["plastic", "mirror", "glass"], spp=6, sampler="halton")
sd3.desc.split_method = 1
sc3 = orc.scene(sd3)
xyzw3, _, _ = sc3.render(threads=1)
np.save(os.path.join(G, "materials_halton_40x40_6spp_xyzw.npy"), xyzw3)
sd4 = fs.scene_materials_render(["metal", "uber", "substrate"], spp=8)
sc4 = orc.scene(sd4)
sb4 = list(sc4.info.sample_bounds)
np.save(os.path.join(G, "materials_sobol_40x40_8spp_samples.npy"), sc4.radiance_samples((sb4[0] + 14, sb4[1] + 14, sb4[0] + 26, sb4[1] + 26)))
# the other SamplerIntegrators: film + per-sample radiance of the middle tile + ray counters, so an edit to the oracle's
# directlighting / whitted / ao paths (orc_render.hpp) cannot pass unnoticed
for name, make in fs.GOLDEN_INTEGRATORS.items():
    sdi = make()
    sci = orc.scene(sdi)
    xyzw_i, cnt_i, _ = sci.render(threads=1)
    rad_i = sci.radiance_samples(fs.golden_tile(sci.info))
    np.savez_compressed(os.path.join(G, name + ".npz"), xyzw=xyzw_i, radiance=rad_i,
                        counters=np.array([cnt_i[k] for k in ("camera_rays", "regular_rays", "shadow_rays", "path_vertices")], np.int64))
    sci.close()
print("golden written:", os.listdir(G))
