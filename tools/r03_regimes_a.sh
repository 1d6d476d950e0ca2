#!/bin/bash
# Round 3, evidence run A: the regimes never measured -- crown-class (3.5 M triangles, textured, 1024 spp) and beyond the Infinity Cache (16 M triangles).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03a
python3 bench.py --triangles 3500000 --materials textured --spp 1024 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > gpurun_out/r03a/bench_crown_class.json 2> gpurun_out/r03a/bench_crown_class.err && \
python3 bench.py --triangles 16000000 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > gpurun_out/r03a/bench_16m.json 2> gpurun_out/r03a/bench_16m.err && \
python3 -m pytest tests/test_gpu_wavefront.py -x -q -k "rt16m or rt1m" > gpurun_out/r03a/pytest_16m.txt 2>&1
echo "rc=$?"; tail -3 gpurun_out/r03a/*.err gpurun_out/r03a/pytest_16m.txt; cat gpurun_out/r03a/*.json
