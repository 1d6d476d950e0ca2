#!/usr/bin/env python3
"""One-off robustness check at crown scale (BASELINE config 5 stand-in): 3.5 M random triangles, both SAH and
HLBVH builds; a 16x16 tile of per-sample radiance against the oracle; build / upload times."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("pbrt-r3_amd")
import oracle_lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500000
ctx = pkg.Context(0)
orc = oracle_lib.load()
out = {}
for split, code in (("sah", 0), ("hlbvh", 1)):
    sd = pkg.scenes.rt1m(n, res=256, spp=8, max_depth=8)
    sd.desc.split_method = code
    t0 = time.time(); ctx.upload(sd); t_up = time.time() - t0
    info = ctx.info
    t0 = time.time(); ctx.film_clear(); ctx.reset_counters(); ctx.render(); t_r = time.time() - t0
    c = ctx.counters()
    rays = c["regular_rays"] + c["shadow_rays"]
    rec = {"bvh_build_ms": round(info.bvh_build_ms, 1), "upload_s": round(t_up, 2), "nodes": info.n_nodes, "leaves": info.n_leaves,
           "render_s": round(t_r, 3), "mrays_s": round(rays / t_r / 1e6, 1), "nodes_per_ray": round(c["nodes_visited"] / rays, 1),
           "tris_per_ray": round(c["tris_tested"] / rays, 1)}
    print(split, rec, flush=True)
    sb = list(info.sample_bounds)
    tile = (sb[0] + 120, sb[1] + 120, sb[0] + 136, sb[1] + 136)
    g = ctx.radiance_samples(tile)
    t0 = time.time(); osc = orc.scene(sd); rec["oracle_build_s"] = round(time.time() - t0, 1)
    r = osc.radiance_samples(tile)
    same = (g.view(np.uint32) == r.view(np.uint32)).all(axis=-1)
    rec["tile_samples"] = int(same.size); rec["tile_bit_identical"] = bool(same.all())
    assert (osc.info.n_nodes, osc.info.n_leaves) == (info.n_nodes, info.n_leaves)
    osc.close()
    out[split] = rec
    print(split, "parity", rec["tile_bit_identical"], flush=True)
print(json.dumps({"triangles": n, **out}))
