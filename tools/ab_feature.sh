#!/bin/bash
# usage: tools/ab_feature.sh "NAME:flags" ...  -> builds each variant and runs the feature-scene tests with it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  so="gpurun_out/tune/lib_$name.so"
  make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$flags" "../../$so" > gpurun_out/tune/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; continue; }
  echo "== $name"
  PBRTGPU_LIB="$PWD/$so" timeout -k 10 300 python -m pytest tests/test_gpu_features.py -q -m gpu -k "test_feature_scene" 2>&1 | grep -E "^FAILED|passed|failed"
done
