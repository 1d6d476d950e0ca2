#!/bin/bash
# Round 3: the deferred-store divergence of the textured / instanced shading kernels (pt_kernels.hip, PT_DEFER_WIDE).
# Builds the variants ON THE GPU BOX (gpurun_out/ does not travel) and runs tools/defer_probe.py with each.
# usage: tools/r03_defer_probe.sh "NAME:EXTRA flags[:CODEGEN flags]" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/defer
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; flags="${rest%%:*}"; cg="${rest#*:}"
  so="gpurun_out/defer/lib_$name.so"
  if [ "$cg" != "$rest" ]; then
    make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$flags" CODEGEN="$cg" "../../$so" > gpurun_out/defer/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; tail -3 gpurun_out/defer/build_$name.log; continue; }
  else
    make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$flags" "../../$so" > gpurun_out/defer/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; tail -3 gpurun_out/defer/build_$name.log; continue; }
  fi
  echo "== $name   (EXTRA=$flags${cg:+ CODEGEN=$cg})"
  PBRTGPU_LIB="$PWD/$so" timeout -k 10 400 python3 tools/defer_probe.py $PROBE_SCENES 2>&1 | tee gpurun_out/defer/probe_$name.txt | tail -12
done
