#!/bin/bash
# Round 3, evidence run B: fabric traffic (FETCH_SIZE / WRITE_SIZE) of the crown-class and beyond-the-cache workloads, the sparse 16 M scene,
# and the multi-rank rehearsals on one device.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03b
python3 -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r03b/pytest_dist.txt 2>&1; echo "pytest dist rc=$?"; tail -3 gpurun_out/r03b/pytest_dist.txt
BENCH_REHEARSE=gloo python3 bench.py --gpus 2 --spp 32 --steps 2 --warmup 1 > gpurun_out/r03b/bench_rehearse_2ranks.json 2> gpurun_out/r03b/bench_rehearse_2ranks.err; echo "rehearse rc=$?"
python3 bench.py --triangles 16000000 --tri-size 0.00125 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > gpurun_out/r03b/bench_16m_sparse.json 2> gpurun_out/r03b/bench_16m_sparse.err; echo "16m sparse rc=$?"
bash tools/pmc_traffic_workload.sh r03_crown --triangles 3500000 --materials textured --spp 1024 && \
bash tools/pmc_traffic_workload.sh r03_16m --triangles 16000000 && \
bash tools/pmc_traffic_workload.sh r03_16m_sparse --triangles 16000000 --tri-size 0.00125
echo "pmc rc=$?"
cat gpurun_out/r03b/*.json
