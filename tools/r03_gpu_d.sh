#!/bin/bash
# Round 3, run D: kernel split of the mixed / textured benches, the quad-visit micro-benchmark, the 16 M build trace.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03d
( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o visit_quad visit_quad.hip 2>/dev/null; for mb in 2 21 69; do ./visit_quad $mb; done ) > gpurun_out/r03d/ubench_visit_quad.txt 2>&1
tail -5 gpurun_out/r03d/ubench_visit_quad.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for w in mixed textured; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03d/prof_$w -- python3 bench.py --materials $w --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-spp1024 > gpurun_out/r03d/bench_$w.json 2> gpurun_out/r03d/bench_$w.err
  f=$(find gpurun_out/r03d/prof_$w -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03d/kernel_stats_$w.csv; head -12 "$f" | cut -c1-150
done
PBRTGPU_BUILD_TRACE=1 python3 - > gpurun_out/r03d/build_trace_16m.txt 2>&1 <<'PY'
import importlib, sys, os
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("pbrt-r3_amd")
for n in (1000000, 3500000, 16000000):
    sd = pkg.scenes.rt1m(n, res=64, spp=1)
    ctx = pkg.Context(0)
    for k in range(2):
        info = ctx.upload(sd)
        print("n=%d upload %d: bvh_build_ms %.1f upload_ms %.1f on_device %d nodes %d" % (n, k, info.bvh_build_ms, info.upload_ms, info.bvh_on_device, info.n_nodes), flush=True)
    ctx.close()
PY
tail -40 gpurun_out/r03d/build_trace_16m.txt | cut -c1-200
