#!/bin/bash
# Round 3, run Q: what the parts of k_shade_general_tex cost -- timing-only builds (PT_TEX_EXP: 1 no texture program runs, 2 image maps answer
# with a constant, 3 image maps and noise do, 4 lobes built from the unevaluated parameter block: the second half of a split kernel) and
# candidate builds (PT_TEX_SLOTS=1), each with the phase profile on the textured bench.  VARIANTS="name:flags ..."
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03q
for spec in ${VARIANTS:-base: exp1:-DPT_TEX_EXP=1 exp2:-DPT_TEX_EXP=2 exp3:-DPT_TEX_EXP=3}; do
  v="${spec%%:*}"; flags="${spec#*:}"; flags="${flags//,/ }"
  make -s -j16 -C pbrt-r3_amd/csrc OUT=/tmp/libpbrtgpu_q$v.so EXTRA="${PROF--DPT_PROFILE_SHADE} $flags" /tmp/libpbrtgpu_q$v.so > gpurun_out/r03q/make_$v.txt 2>&1 || { tail -5 gpurun_out/r03q/make_$v.txt; exit 1; }
  PBRTGPU_LIB=/tmp/libpbrtgpu_q$v.so timeout -k 10 300 python3 bench.py --materials ${MATERIALS:-textured} --spp 32 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 > gpurun_out/r03q/bench_$v.json 2> gpurun_out/r03q/bench_$v.err
  echo "== $v ($flags): $(cut -c1-60 gpurun_out/r03q/bench_$v.json)"; grep "shade phases" gpurun_out/r03q/bench_$v.err | tail -1 | cut -c1-420
done
