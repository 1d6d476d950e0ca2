#!/bin/bash
# Round 3, run P: where the textured shading kernel's clocks go (-DPT_PROFILE_SHADE build on the box), SQ counters of the textured workload.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03p
make -s -j16 -C pbrt-r3_amd/csrc OUT=/tmp/libpbrtgpu_shp.so EXTRA=-DPT_PROFILE_SHADE /tmp/libpbrtgpu_shp.so > gpurun_out/r03p/make.txt 2>&1 || { tail -5 gpurun_out/r03p/make.txt; exit 1; }
for m in textured mixed matte; do
  PBRTGPU_LIB=/tmp/libpbrtgpu_shp.so timeout -k 10 300 python3 bench.py --materials $m --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 > gpurun_out/r03p/bench_$m.json 2> gpurun_out/r03p/bench_$m.err
  echo "== $m: $(cut -c1-60 gpurun_out/r03p/bench_$m.json)"; grep "shade phases" gpurun_out/r03p/bench_$m.err | tail -1 | cut -c1-700
done
