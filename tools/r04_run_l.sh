mkdir -p gpurun_out/r04l
timeout -k 10 900 python3 -m pytest tests/test_rec_integrators.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04l/pytest_rec.txt 2>&1; echo "pytest rec rc=$?"; tail -3 gpurun_out/r04l/pytest_rec.txt
bash tools/r04_gpu_c.sh default:direct default:whitted
