mkdir -p gpurun_out/r04h
PBRTGPU_SHADE_LOCAL=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_features.py tests/test_materials.py tests/test_spheres.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04h/pytest_local.txt 2>&1; echo "pytest local rc=$?"; tail -3 gpurun_out/r04h/pytest_local.txt
bash tools/r04_gpu_c.sh default:mixed default:mixed:PBRTGPU_SHADE_LOCAL=1 default:killeroo default:killeroo:PBRTGPU_SHADE_LOCAL=1 default:sphere default:sphere:PBRTGPU_SHADE_LOCAL=1
