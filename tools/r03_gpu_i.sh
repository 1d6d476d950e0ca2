#!/bin/bash
# Round 3, run I: does the continuation-key plumbing in the shading kernels cost anything where the sort is off?  Same box, two builds, two rounds each.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/tune
for spec in "with:" "without:-DPT_NO_CKEY"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  make -s -j16 -C pbrt-r3_amd/csrc OUT="../../gpurun_out/tune/lib_$name.so" EXTRA="$flags" "../../gpurun_out/tune/lib_$name.so" > /dev/null 2>&1 || echo "build $name failed"
done
for round in 1 2; do
for name in with without; do
  for w in "--materials mixed --spp 64" "--materials textured --spp 64" "--spp 64"; do
    PBRTGPU_LIB="$PWD/gpurun_out/tune/lib_$name.so" PBRTGPU_SORT_CONT=0 python3 bench.py $w --steps 2 --warmup 1 --no-cpu-baseline --no-spp1024 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-8s %-30s %8.1f Mrays/s  shade %.3f' % ('$name', '$w', d['value'], r['shade_share_of_render']))"
  done
done
done
bash tools/pmc_traffic_workload.sh r03_16m_sparse2 --triangles 16000000 --tri-size 0.00125 > /dev/null && bash tools/pmc_traffic_workload.sh r03_16m2 --triangles 16000000 > /dev/null
python3 -c "
import json
for t in ('16m_sparse2','16m2'):
    e=json.load(open('gpurun_out/pmc_r03_%s/traffic_entry.json'%t)); print(t, e['fabric_bytes_per_launch']/e['avg_launch_ms_under_pmc']/1e9/8, e['avg_launch_ms_under_pmc'])
"
