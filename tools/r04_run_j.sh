mkdir -p gpurun_out/r04j
PBRTGPU_SHADE_LOCAL=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_features.py tests/test_materials.py tests/test_image_textures.py -m gpu -x -q > gpurun_out/r04j/pytest_local_tex.txt 2>&1; echo "pytest local rc=$?"; tail -3 gpurun_out/r04j/pytest_local_tex.txt
bash tools/r04_gpu_c.sh default:crown default:crown:PBRTGPU_SHADE_LOCAL=1 default:textured default:textured:PBRTGPU_SHADE_LOCAL=1
