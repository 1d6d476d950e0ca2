#!/bin/bash
# Round 4, run C: A/B lines of prebuilt variant libraries (variants/lib_NAME.so, built in the container: `make OUT=../../variants/lib_NAME.so EXTRA=...`).
# usage: tools/r04_gpu_c.sh "NAME:WORKLOAD[:ENV=VAL,...]" ...       NAME "default" = the shipped library
cd "$(dirname "$0")/.."
out="gpurun_out/r04c"
mkdir -p "$out"
declare -A W
W[head]=""
W[mixed]="--materials mixed"
W[killeroo]="--materials mixed --light sphere --sampler halton"
W[crown]="--triangles 3500000 --materials textured"
W[textured]="--materials textured"
W[sphere]="--light sphere"
W[sparse16]="--triangles 16000000 --tri-size 0.00125"
W[t4m]="--triangles 4000000"
W[t8m]="--triangles 8000000"
W[dense16]="--triangles 16000000"
W[sparse8]="--triangles 8000000 --tri-size 0.00177"
W[sparse4]="--triangles 4000000 --tri-size 0.0025"
W[halton]="--sampler halton"
W[direct]="--integrator directlighting"
W[whitted]="--integrator whitted"
for spec in "$@"; do
  IFS=: read -r name w envs <<< "$spec"
  lib="$PWD/variants/lib_$name.so"; [ "$name" = default ] && lib="$PWD/pbrt-r3_amd/csrc/libpbrtgpu.so"
  [ -f "$lib" ] || { echo "$spec: no $lib"; continue; }
  tag="${name}_${w}_$(echo "$envs" | tr -c 'A-Za-z0-9\n' '_')"
  ( export PBRTGPU_LIB="$lib"; for kv in $(echo "$envs" | tr ',' ' '); do export "$kv"; done
    timeout -k 10 500 python3 bench.py ${W[$w]} --spp ${SPP:-64} --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-spp1024 > "$out/bench_$tag.json" 2> "$out/bench_$tag.err" ) || { echo "$spec failed"; tail -5 "$out/bench_$tag.err"; exit 1; }
  python3 -c "
import json; d=json.load(open('$out/bench_$tag.json')); r=d['roofline']
print('%-10s %-9s %-34s %8.1f Mrays/s %9.2f ms/frame  trace %.3f ms/launch (share %.3f)  shade share %.3f' % ('$name', '$w', '$envs', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_share_of_render'], r['shade_share_of_render']))"
done
