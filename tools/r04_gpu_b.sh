#!/bin/bash
# Round 4, run B: the vertex in two kernels (PBRTGPU_NEE_SPLIT) -- parity of the split kernels, then A/B bench lines at 64 spp.
# usage: tools/r04_gpu_b.sh [tests|bench|both]
cd "$(dirname "$0")/.."
what="${1:-both}"
out="gpurun_out/r04b"
mkdir -p "$out"
declare -A W
W[head]=""
W[mixed]="--materials mixed"
W[killeroo]="--materials mixed --light sphere --sampler halton"
W[crown]="--triangles 3500000 --materials textured"
W[sphere]="--light sphere"
if [ "$what" != bench ]; then
  PBRTGPU_NEE_SPLIT=15 timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_materials.py tests/test_spheres.py tests/test_gpu_fuzz.py -m gpu -x -q > "$out/pytest_split15.txt" 2>&1
  echo "pytest split=15 rc=$?"; tail -5 "$out/pytest_split15.txt"
fi
if [ "$what" != tests ]; then
  for spec in ${SPECS:-head:0 head:1 mixed:0 mixed:1 mixed:3 killeroo:0 killeroo:4 killeroo:7 crown:0 crown:3 crown:11}; do
    w="${spec%%:*}"; sp="${spec##*:}"
    PBRTGPU_NEE_SPLIT=$sp timeout -k 10 400 python3 bench.py ${W[$w]} --spp ${SPP:-64} --steps 2 --warmup 1 --no-cpu-baseline --no-spp1024 > "$out/bench_${w}_$sp.json" 2> "$out/bench_${w}_$sp.err" || { echo "$spec failed"; tail -5 "$out/bench_${w}_$sp.err"; exit 1; }
    python3 -c "
import json; d=json.load(open('$out/bench_${w}_$sp.json')); r=d['roofline']
print('%-9s split %2s  %8.1f Mrays/s  %8.2f ms/frame  trace share %.3f  shade share %.3f' % ('$w', '$sp', d['value'], d['ms_per_step'], r['trace_share_of_render'], r['shade_share_of_render']))"
  done
fi
