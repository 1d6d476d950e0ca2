#!/bin/bash
# Round 3, run E: continuation-ray sort -- parity with it forced on, then what it buys by scene size and mode.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03e
python3 -m pytest tests/test_gpu_features.py -x -q -k "sorting_changes_nothing or deferred_store" > gpurun_out/r03e/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03e/pytest.txt
run() { # name, env mode, bench args
  PBRTGPU_SORT_CONT=$2 python3 bench.py $3 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s cont-sort %s  %8.1f Mrays/s  k_trace %.2f ms/launch  trace %.3f shade %.3f' % ('$1', '$2', d['value'], r['avg_launch_ms'], r['trace_share_of_render'], r['shade_share_of_render']))"
}
for m in 0 1 2; do run "16M sparse" $m "--triangles 16000000 --tri-size 0.00125 --spp 64"; done
for m in 0 1 2; do run "3.5M textured (crown-class)" $m "--triangles 3500000 --materials textured --spp 64"; done
for m in 0 1 2; do run "16M dense" $m "--triangles 16000000 --spp 64"; done
for m in 0 1 2; do run "RT1M" $m "--spp 64"; done
for m in 0 1; do run "4M matte" $m "--triangles 4000000 --tri-size 0.0025 --spp 64"; done
