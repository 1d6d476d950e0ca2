#!/bin/bash
# usage: tools/tune_any.sh "bench args" "NAME:make-args" ...   -> the given bench line, two runs per build variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
bargs="$1"; shift
for spec in "$@"; do
  name="${spec%%:*}"; margs="${spec#*:}"
  so="gpurun_out/tune/lib_$name.so"
  make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" $margs "../../$so" > gpurun_out/tune/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; continue; }
  for i in 1 2; do
    PBRTGPU_LIB="$PWD/$so" timeout -k 10 300 python bench.py $bargs --no-cpu-baseline --no-spp1024 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s %-44s %8.1f Mrays/s' % ('$name', '$bargs', d['value']))"
  done
done
