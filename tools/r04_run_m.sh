mkdir -p gpurun_out/r04m
PBRTGPU_LIB=$PWD/variants/lib_lds.so timeout -k 10 900 python3 -m pytest tests/test_gpu_features.py tests/test_materials.py tests/test_spheres.py -m gpu -x -q > gpurun_out/r04m/pytest_lds.txt 2>&1; echo "pytest lds rc=$?"; tail -3 gpurun_out/r04m/pytest_lds.txt
bash tools/r04_gpu_c.sh default:mixed lds:mixed default:killeroo lds:killeroo default:crown lds:crown default:mixed lds:mixed
