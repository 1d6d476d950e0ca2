#!/bin/bash
# Round 3, run N: kernel trace of the device SAH scene build (1M triangles, three uploads).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03n
cat > gpurun_out/r03n/build3.py <<'PY'
import importlib, sys, os
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("pbrt-r3_amd")
sd = pkg.scenes.rt1m(1000000, res=64, spp=1)
ctx = pkg.Context(0)
for k in range(3):
    info = ctx.upload(sd)
    print("upload %d: bvh_build_ms %.1f upload_ms %.1f" % (k, info.bvh_build_ms, info.upload_ms), flush=True)
ctx.close()
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r03n/prof -o build -- python3 gpurun_out/r03n/build3.py > gpurun_out/r03n/log.txt 2>&1
echo "rc=$?"; tail -5 gpurun_out/r03n/log.txt
python3 tools/rocpd_levels.py gpurun_out/r03n/prof/build_results.db > gpurun_out/r03n/levels.txt; head -30 gpurun_out/r03n/levels.txt
