#!/bin/bash
# usage: tools/tune_tex.sh "NAME:flags" ...  -> textured-material bench per variant, with the kernel split from rocprofv3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  so="gpurun_out/tune/lib_$name.so"
  make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$flags" "../../$so" > gpurun_out/tune/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; continue; }
  PBRTGPU_LIB="$PWD/$so" timeout -k 10 200 python bench.py --materials textured --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2>gpurun_out/tune/err_$name.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-20s %8.1f Mrays/s  ms/step %.1f  trace %.3f shade %.3f' % ('$name', d['value'], d['ms_per_step'], r['trace_share_of_render'], r['shade_share_of_render']))"
done
