#!/bin/bash
# usage: tools/tune_mixed.sh "NAME:make-args" ...   -> mixed-material bench, three runs per variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
for spec in "$@"; do
  name="${spec%%:*}"; margs="${spec#*:}"
  so="gpurun_out/tune/lib_$name.so"
  make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" $margs "../../$so" > gpurun_out/tune/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; continue; }
  for i in 1 2 3; do
    PBRTGPU_LIB="$PWD/$so" timeout -k 10 200 python bench.py --materials mixed --spp 64 --no-cpu-baseline --no-spp1024 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s %8.1f Mrays/s' % ('$name', d['value']))"
  done
done
