mkdir -p gpurun_out/r04n
export PBRTGPU_DATA_DIR=$PWD/pbrt-r3_amd/data
timeout -k 10 900 python3 -m pytest tests/test_spheres.py tests/test_gpu_features.py tests/test_gpu_fuzz.py tests/test_gpu_wavefront.py -m gpu -x -q > gpurun_out/r04n/pytest_sph.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04n/pytest_sph.txt
bash tools/r04_gpu_c.sh default:killeroo sl4:killeroo sl2:killeroo default:sphere sl4:sphere sl2:sphere
