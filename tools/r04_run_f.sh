mkdir -p gpurun_out/r04f
PBRTGPU_LIB=$PWD/variants/lib_wide.so timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_wavefront.py tests/test_spheres.py -m gpu -x -q > gpurun_out/r04f/pytest_wide.txt 2>&1; echo "pytest wide rc=$?"; tail -3 gpurun_out/r04f/pytest_wide.txt
bash tools/r04_gpu_c.sh default:head wide:head default:mixed wide:mixed default:killeroo wide:killeroo default:crown wide:crown touch2:crown touch3:crown default:sparse16 wide:sparse16 touch:sparse16 touch2:sparse16 touch3:sparse16
