mkdir -p gpurun_out/r04o
export PBRTGPU_DATA_DIR=$PWD/pbrt-r3_amd/data
timeout -k 10 900 python3 -m pytest tests/test_halton.py tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_fuzz.py tests/test_rec_integrators.py tests/test_ao.py -m gpu -x -q > gpurun_out/r04o/pytest_halton.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04o/pytest_halton.txt
bash tools/r04_gpu_c.sh nohb:halton default:halton nohb:killeroo default:killeroo nohb:halton default:halton
