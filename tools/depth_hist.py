#!/usr/bin/env python3
"""Where in the 4-wide tree do the node visits of the bench workload fall?  Renders a sample of RT1M tiles with the CPU oracle and prints
node visits by tree depth (diagnostic behind the decision to keep the top of the tree in LDS).
usage: python tools/depth_hist.py [triangles] [spp] [n_tiles]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
pkg = importlib.import_module("pbrt-r3_amd")
import oracle_lib
ntri = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 64
sd = pkg.scenes.rt1m(ntri, res=1024, spp=spp, max_depth=8)
osc = oracle_lib.load().scene(sd)
lib = osc.lib
lib.orc_depth_hist.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.orc_depth_hist.restype = None
tiles = pkg.partition_tiles(tuple(osc.info.sample_bounds), 16) if hasattr(pkg, "partition_tiles") else None
if tiles is None:
    sb = list(osc.info.sample_bounds)
    tiles = [(x, y, min(x + 16, sb[2]), min(y + 16, sb[3])) for y in range(sb[1], sb[3], 16) for x in range(sb[0], sb[2], 16)]
tiles = tiles[:: max(1, len(tiles) // nt)]
lib.orc_depth_hist(osc.h, 1, None)
_, cnt, secs = osc.render(tiles, threads=8, want_image=False)
out = np.zeros(32, np.uint64)
lib.orc_depth_hist(osc.h, 0, out.ctypes.data_as(C.c_void_p))
tot = float(out.sum())
rays = cnt["regular_rays"] + cnt["shadow_rays"]
print("tiles %d, rays %d, node visits %d (hist %d), %.1f s" % (len(tiles), rays, cnt["nodes_visited"], int(tot), secs))
cum = 0.0
for d in range(32):
    if out[d] == 0: continue
    cum += out[d]
    print("depth %2d  nodes<=4^d %8d  visits/ray %6.2f  share %5.1f %%  cumulative %5.1f %%" % (d, 4 ** d, out[d] / rays, 100 * out[d] / tot, 100 * cum / tot))
