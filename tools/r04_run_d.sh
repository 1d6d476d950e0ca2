mkdir -p gpurun_out/r04d
timeout -k 10 900 python3 -m pytest tests/test_rec_integrators.py -m gpu -x -q > gpurun_out/r04d/pytest_rec.txt 2>&1; echo "pytest rec rc=$?"; tail -3 gpurun_out/r04d/pytest_rec.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_dist.py "tests/test_gpu_features.py::test_crown_class_3p5m_textured_tiles" -m gpu -x -q > gpurun_out/r04d/pytest_dist.txt 2>&1; echo "pytest dist rc=$?"; tail -3 gpurun_out/r04d/pytest_dist.txt
bash tools/r04_gpu_c.sh default:direct default:whitted default:direct:PBRTGPU_SORT_SHADOW_MIN=0 default:whitted:PBRTGPU_SORT_SHADOW_MIN=0
for v in prof profdeep; do
  PBRTGPU_LIB=$PWD/variants/lib_$v.so timeout -k 10 300 python3 bench.py --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 2> gpurun_out/r04d/err_$v.log > gpurun_out/r04d/bench_$v.json; echo "== $v"; grep -h "phases\]" gpurun_out/r04d/err_$v.log | tail -2
done
