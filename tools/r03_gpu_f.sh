#!/bin/bash
# Round 3, run F: the device scene path -- digests / parity, then build times.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03f
timeout -k 10 900 python3 -m pytest tests/test_gpu_hlbvh.py -x -q > gpurun_out/r03f/pytest_hlbvh.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03f/pytest_hlbvh.txt | cut -c1-300
PBRTGPU_BUILD_TRACE=1 python3 - > gpurun_out/r03f/build_trace.txt 2>&1 <<'PY'
import importlib, sys, os
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("pbrt-r3_amd")
for split in (0, 1):
  for n in (1000000, 3500000, 16000000):
    sd = pkg.scenes.rt1m(n, res=64, spp=1)
    sd.desc.split_method = split
    ctx = pkg.Context(0)
    for k in range(3):
        info = ctx.upload(sd)
        print("split=%s n=%d upload %d: bvh_build_ms %.1f upload_ms %.1f on_device %d nodes %d" % ("sah" if split == 0 else "hlbvh", n, k, info.bvh_build_ms, info.upload_ms, info.bvh_on_device, info.n_nodes), flush=True)
    ctx.close()
PY
grep -E "n=|device scene \(" gpurun_out/r03f/build_trace.txt | cut -c1-260
