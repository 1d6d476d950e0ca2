#!/usr/bin/env python3
"""One-GPU probe of the multi-GPU partition: time what rank 0 of N would render (tiles[0::N]) against the whole
frame.  Says how much of strong scaling the per-rank work itself allows (the reduce is a 16 MiB all-reduce)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("pbrt-r3_amd")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd = pkg.scenes.rt1m(1000000, res=1024, spp=spp, max_depth=8)
ctx = pkg.Context(0)
info = ctx.upload(sd)
tiles = pkg.scenes.all_tiles(info)
base = None
for n in (1, 2, 4, 8):
    mine = tiles[0::n]
    ctx.film_clear(); ctx.render(mine); ctx.film_xyzw()          # warm-up (pool allocation on the first pass)
    t0 = time.time()
    for _ in range(2):
        ctx.film_clear(); ctx.render(mine); ctx.film_device_xyzw()
    dt = (time.time() - t0) / 2
    base = base or dt
    print("N=%d: %5d tiles, %.3f s per frame share, ideal %.3f s, efficiency %.3f" % (n, len(mine), dt, base / n, base / n / dt), flush=True)
