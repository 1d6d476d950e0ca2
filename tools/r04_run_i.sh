PBRTGPU_SHADE_LOCAL=1 WORKLOADS="mixed" bash tools/r04_gpu_a.sh local
bash tools/r04_gpu_c.sh default:head default:mixed default:mixed:PBRTGPU_SHADE_LOCAL=1 lc1k:mixed:PBRTGPU_SHADE_LOCAL=1 lc16k:mixed:PBRTGPU_SHADE_LOCAL=1 lc16k:killeroo:PBRTGPU_SHADE_LOCAL=1
