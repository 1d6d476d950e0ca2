#!/bin/bash
# k_shade's cost per path against the size of the path pool (TLB reach experiment).  usage: tools/shade_pool_probe.sh
cd "$(dirname "$0")/.."
for pool in 67108864 16777216 4194304; do
  PBRTGPU_POOL_PATHS=$pool timeout -k 10 120 python bench.py --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pool %9d  %8.1f Mrays/s  trace %.3f shade %.3f of render, ms/step %.1f  launches %d' % ($pool, d['value'], r['trace_share_of_render'], r['shade_share_of_render'], d['ms_per_step'], r['launches']))"
done
