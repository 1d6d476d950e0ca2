#!/bin/bash
# Round 3, run H: material-sorted shade queues against one unsorted lobe-list kernel.
cd "$(dirname "$0")/.."
run() { PBRTGPU_SHADE_UNSORTED=$2 python3 bench.py $3 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-30s unsorted=%s  %8.1f Mrays/s  trace %.3f shade %.3f' % ('$1', '$2', d['value'], r['trace_share_of_render'], r['shade_share_of_render']))"; }
for m in 0 1; do run "mixed 64spp" $m "--materials mixed --spp 64"; done
for m in 0 1; do run "textured 64spp" $m "--materials textured --spp 64"; done
for m in 0 1; do run "crown-class 3.5M textured" $m "--triangles 3500000 --materials textured --spp 64"; done
PBRTGPU_SHADE_UNSORTED=1 python3 -m pytest tests/test_materials.py tests/test_gpu_features.py -x -q -m gpu -k "not deferred and not stack_spill" 2>&1 | tail -3
