import importlib, sys, os
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("pbrt-r3_amd")
ctx = pkg.Context(0)
sd = pkg.scenes.rt1m(1000000, res=64, spp=1, max_depth=1)
ctx.set_bvh_build(pkg.capi.BVH_BUILD_DEVICE)
for _ in range(2):
    print(ctx.upload(sd).bvh_build_ms)
ctx.close()
