#!/bin/bash
# usage: tools/tune_shade.sh "NAME:flags:shade_bpc" ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; flags="${rest%%:*}"; bpc="${rest##*:}"
  so="gpurun_out/tune/lib_$name.so"
  make -s -j8 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$flags" "../../$so" > gpurun_out/tune/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; continue; }
  PBRTGPU_LIB="$PWD/$so" PBRTGPU_SHADE_BLOCKS_PER_CU="$bpc" timeout -k 10 120 python bench.py --spp ${SPP:-32} --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2> gpurun_out/tune/err_$name.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s %8.1f Mrays/s  ms/step %.1f trace-share %.3f' % ('$name', d['value'], d['ms_per_step'], r['trace_share_of_render']))"
  grep -h "phases\]" gpurun_out/tune/err_$name.log | tail -2
done
