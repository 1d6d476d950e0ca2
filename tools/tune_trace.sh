#!/bin/bash
# Build k_trace variants on the GPU box and time each with a short bench run.
# usage: tools/tune_trace.sh "NAME:-DPT_REFILL_MIN=16:4" ...   (name : extra flags : blocks per CU)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
for spec in "$@"; do
  name="${spec%%:*}"; rest="${spec#*:}"; flags="${rest%%:*}"; bpc="${rest##*:}"
  so="gpurun_out/tune/lib_$name.so"
  make -s -j8 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$flags" "../../$so" > gpurun_out/tune/build_$name.log 2>&1 || { echo "$name BUILD FAILED"; tail -3 gpurun_out/tune/build_$name.log; continue; }
  PBRTGPU_LIB="$PWD/$so" PBRTGPU_TRACE_BLOCKS_PER_CU="$bpc" timeout -k 10 120 python bench.py --spp ${SPP:-32} --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2>gpurun_out/tune/err_$name.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s %8.1f Mrays/s  trace %.3f ms/launch  frac %.3f  share %.3f' % ('$name', d['value'], r['avg_launch_ms'], r['frac'], r['trace_share_of_render']))"
done
