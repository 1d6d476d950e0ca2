#!/usr/bin/env python3
"""Throughput of the instanced-scene kernels (k_trace_inst, k_shade_general_inst) on the feature scene at a useful size.
usage: [PBRTGPU_LIB=...] python3 tools/instances_probe.py [res] [spp]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("pbrt-r3_amd")
import feature_scenes as fs
res = int(sys.argv[1]) if len(sys.argv) > 1 else 768
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sd = fs.scene_instances(res=res, spp=spp)
ctx = pkg.Context(0)
ctx.upload(sd)
ctx.film_clear(); ctx.render()
for _ in range(2):
    ctx.film_clear(); ctx.reset_counters()
    t0 = time.time(); ctx.render(); ctx.film_device_xyzw(); dt = time.time() - t0
    c = ctx.counters()
    rays = c["regular_rays"] + c["shadow_rays"]
    print("instances %dx%d %d spp: %.1f Mrays/s  trace %.1f ms shade %.1f ms of %.1f ms" % (res, res, spp, rays / dt / 1e6, c["trace_ms"], c["shade_ms"], c["render_ms"]), flush=True)
