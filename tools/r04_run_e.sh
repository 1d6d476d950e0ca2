mkdir -p gpurun_out/r04e
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r04e/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04e/pytest_gpu.txt
bash tools/r04_gpu_c.sh default:killeroo default:sphere
python3 -c "
import json
for w in ('killeroo','sphere'):
    d=json.load(open('gpurun_out/r04c/bench_default_%s_.json' % w)); c=d['config']; print(w, 'bvh_build_ms', c['bvh_build_ms'], 'upload_ms', c['upload_ms'], 'first', c['bvh_build_first_upload_ms'], c['first_upload_ms'])"
