#!/bin/bash
# same-box A/B of two prebuilt libraries under tmp_ab/ (usage: NAMES="base pf" tools/r03_gpu_k.sh)
cd "$(dirname "$0")/.."
for round in 1 2; do
for name in ${NAMES:-base pf}; do
  for w in "--spp 64" "--materials mixed --spp 64" "--materials textured --spp 64"; do
    PBRTGPU_LIB="$PWD/tmp_ab/lib_$name.so" python3 bench.py $w --steps 2 --warmup 1 --no-cpu-baseline --no-spp1024 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s %-30s %8.1f Mrays/s  k_trace %.3f ms  trace %.3f shade %.3f' % ('$name', '$w', d['value'], r['avg_launch_ms'], r['trace_share_of_render'], r['shade_share_of_render']))"
  done
done
done
