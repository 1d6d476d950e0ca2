#!/usr/bin/env python3
"""First upload of a fresh context (cold host and device buffers) against the second: PBRTGPU_BUILD_TRACE=1 python3 tools/cold_upload_trace.py [n_triangles]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("pbrt-r3_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sd = pkg.scenes.rt1m(n, res=64, spp=1, max_depth=1)
ctx = pkg.Context(0)
for k in range(3):
    print("---- upload %d" % k, file=sys.stderr, flush=True)
    info = ctx.upload(sd)
    print("upload %d: bvh_build_ms %.1f upload_ms %.1f" % (k, info.bvh_build_ms, info.upload_ms), file=sys.stderr, flush=True)
