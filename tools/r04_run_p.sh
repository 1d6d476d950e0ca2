mkdir -p gpurun_out/r04p
export PBRTGPU_DATA_DIR=$PWD/pbrt-r3_amd/data
timeout -k 10 900 python3 -m pytest tests/test_gpu_features.py tests/test_materials.py tests/test_image_textures.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04p/pytest_tex.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04p/pytest_tex.txt
bash tools/r04_gpu_c.sh base:crown default:crown base:textured default:textured base:crown default:crown
