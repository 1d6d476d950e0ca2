#!/usr/bin/env python3
"""Upload and render time of an instanced scene: one object of N random triangles, K instances of it in a lit room.
usage: python tools/r04_inst_upload.py [N=500000] [K=50] [spp=16]      (PBRTGPU_BUILD_TRACE=1 prints the upload's stages)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pbrt-r3_amd")
scenes = pkg.scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
b = scenes.SceneBuilder()
b.look_at((0, 0, -3.4), (0, 0, 0), (0, 1, 0)); b.camera_perspective(fov=40.0); b.film(xresolution=512, yresolution=512); b.pixel_filter_box()
b.sampler_sobol(spp); b.integrator_path(maxdepth=5, rrthreshold=1.0, lightsamplestrategy="spatial"); b.accelerator_bvh("sah", 4)
b.material_matte((0.5, 0.5, 0.5))
q = scenes._quad
q(b, (1, -1, -1), (-1, -1, -1), (-1, -1, 1), (1, -1, 1)); q(b, (1, 1, -1), (1, 1, 1), (-1, 1, 1), (-1, 1, -1)); q(b, (1, -1, 1), (-1, -1, 1), (-1, 1, 1), (1, 1, 1))
q(b, (-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)); q(b, (1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1))
b.area_light_source_diffuse(L=(17, 12, 4)); q(b, (0.3, 0.999, -0.3), (0.3, 0.999, 0.3), (-0.3, 0.999, 0.3), (-0.3, 0.999, -0.3)); b.no_area_light()
rng = np.random.default_rng(7)
c = rng.uniform(-0.5, 0.5, (N, 1, 3)).astype(np.float32)
verts = (c + rng.uniform(-0.02, 0.02, (N, 3, 3)).astype(np.float32)).reshape(-1, 3)
b.object_begin("blob"); b.material_matte((0.6, 0.4, 0.3)); b.shape_trianglemesh_fast(verts, np.arange(3 * N), twosided=True); b.object_end()
T = scenes
for k in range(K):
    p = rng.uniform(-0.8, 0.8, 3)
    b.object_instance("blob", T.transform_mul(T.transform_translate(float(p[0]), float(p[1]), float(p[2])), T.transform_scale(0.25, 0.25, 0.25)))
sd = b.build()
ctx = pkg.Context(0)
for rep in range(2):
    t0 = time.time(); info = ctx.upload(sd); t1 = time.time()
    print("upload %d: %.1f ms (bvh %.1f + rest)  nodes %d" % (rep, 1e3 * (t1 - t0), info.bvh_build_ms, info.n_nodes), flush=True)
ctx.film_clear(); ctx.reset_counters()
t0 = time.time(); ctx.render(); t1 = time.time()
cn = ctx.counters()
rays = cn["regular_rays"] + cn["shadow_rays"]
print("render %.1f ms, %.1f Mrays/s (%d rays)" % (1e3 * (t1 - t0), rays / (t1 - t0) / 1e6, rays))
ctx.close()
