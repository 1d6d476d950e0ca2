#!/bin/bash
# Round 3, run C: the opaque-zero fix -- full GPU suite, the held-back variant on the probe scenes, and the benches either way.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03c
python3 -m pytest tests -m gpu -x -q > gpurun_out/r03c/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03c/pytest_gpu.txt
PROBE_DEPTHS=4,5 bash tools/r03_defer_probe.sh "fixed_defer:-DPT_DEFER_WIDE=1" > gpurun_out/r03c/probe.txt 2>&1; tail -14 gpurun_out/r03c/probe.txt
for v in "" "-DPT_DEFER_WIDE=1"; do
  name=base; [ -n "$v" ] && name=defer
  so="gpurun_out/defer/lib_bench_$name.so"
  make -s -j16 -C pbrt-r3_amd/csrc OUT="../../$so" EXTRA="$v" "../../$so" > /dev/null 2>&1
  for w in "--materials textured --spp 64" "--materials mixed --spp 64" "--spp 64"; do
    PBRTGPU_LIB="$PWD/$so" python3 bench.py $w --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-6s %-32s %8.1f Mrays/s  trace %.3f shade %.3f' % ('$name', '$w', d['value'], r['trace_share_of_render'], r['shade_share_of_render']))"
  done
done
