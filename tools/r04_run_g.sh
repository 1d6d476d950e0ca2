mkdir -p gpurun_out/r04g
timeout -k 10 900 python3 -m pytest tests/test_gpu_features.py -m gpu -x -q -k "far_traversal" > gpurun_out/r04g/pytest_far.txt 2>&1; echo "pytest far rc=$?"; tail -3 gpurun_out/r04g/pytest_far.txt
PBRTGPU_BUILD_TRACE=1 bash tools/r04_gpu_c.sh default:sparse16 default:dense16 default:head default:crown default:t4m 2>&1
grep -h "\[trace\] trial" gpurun_out/r04c/bench_default_sparse16_.err gpurun_out/r04c/bench_default_dense16_.err gpurun_out/r04c/bench_default_t4m_.err
