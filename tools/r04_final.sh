#!/bin/bash
# Round-4 end-of-round evidence, run on the GPU box: tools/r04_final.sh [a|b|c|d]
#   a  the default bench line (the driver's command), the same under rocprofv3 --kernel-trace --stats (kernel_stats.csv, per-bounce launches),
#      FETCH_SIZE / WRITE_SIZE of the default workload (counter-only passes) -> a stamped traffic entry
#   b  the regime lines at full sample counts: crown-class (3.5 M triangles, textured, 1024 spp), 16 M dense, 16 M sparse (+ its traffic entry, k_trace_far
#      forced), killeroo-class (mixed materials, sphere light, Halton, 512 spp), and the secondary lines at 64 spp
#   c  the whole -m gpu suite
#   d  the N > 1 rehearsals a one-GPU box allows: bench.py with 4 ranks (gloo, BENCH_REHEARSE) at full size, pbrt_gpu --devices 0 x 8
cd "$(dirname "$0")/.."
part="${1:-a}"
out="gpurun_out/r04_final_$part"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$part" = a ]; then
  # counters first: the bench lines behind them then quote the traffic measured on this very source (bench.py checks the stamp)
  bash tools/pmc_traffic_workload.sh r04_default && cp gpurun_out/pmc_r04_default/traffic_entry.json "$out/traffic_entry_default.json" && cp gpurun_out/pmc_r04_default/summary.txt "$out/pmc_fetch_write_default.txt" && python3 tools/r04_stamp.py "$out/traffic_entry_default.json"
  timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
  cut -c1-300 "$out/bench.json"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$out/prof" --output-format csv -- python3 bench.py --no-cpu-baseline --no-spp1024 > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err" || { echo "rocprof run failed"; tail -5 "$out/bench_under_rocprof.err"; exit 1; }
  find "$out/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
  find "$out/prof" -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/per_bounce.py {} > "$out/per_bounce.txt"; rm -rf "$out/prof"
  head -8 "$out/kernel_stats.csv" | cut -c1-140
elif [ "$part" = b ]; then
  PBRTGPU_TRACE_FAR=1 bash tools/pmc_traffic_workload.sh r04_16m_sparse --triangles 16000000 --tri-size 0.00125 && cp gpurun_out/pmc_r04_16m_sparse/traffic_entry.json "$out/traffic_entry_16m_sparse.json" && cp gpurun_out/pmc_r04_16m_sparse/summary.txt "$out/pmc_fetch_write_16m_sparse.txt" && python3 tools/r04_stamp.py "$out/traffic_entry_16m_sparse.json" "PBRTGPU_TRACE_FAR=1 for the whole frame (what the library's trial picks for this scene)"
  timeout -k 10 600 python3 bench.py --triangles 3500000 --materials textured --spp 1024 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_crown_class.json" 2> "$out/bench_crown_class.err"
  timeout -k 10 600 python3 bench.py --triangles 16000000 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_16m.json" 2> "$out/bench_16m.err"
  timeout -k 10 600 python3 bench.py --triangles 16000000 --tri-size 0.00125 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_16m_sparse.json" 2> "$out/bench_16m_sparse.err"
  timeout -k 10 600 python3 bench.py --materials mixed --light sphere --sampler halton --spp 512 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_killeroo_class.json" 2> "$out/bench_killeroo_class.err"
  timeout -k 10 600 python3 bench.py --materials mixed --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_mixed.json" 2> "$out/bench_mixed.err"
  for f in crown_class 16m 16m_sparse killeroo_class mixed; do python3 -c "import json,sys; d=json.load(open('$out/bench_$f.json')); r=d['roofline']; print('%-14s %8.1f Mrays/s  alg frac %.3f | l1 %.3f hbm_measured %s | trace share %.3f shade %.3f | parity %.1e | upload %.1f+%.1f ms' % ('$f', d['value'], r['frac'], r['l1_req']['frac'], r['hbm_measured'] and r['hbm_measured']['frac'], r['trace_share_of_render'], r['shade_share_of_render'], d['parity']['rel_l2'], d['config']['bvh_build_ms'], d['config']['upload_ms']))"; done
  for v in "--integrator ao" "--sampler halton --spp 64" "--integrator directlighting --spp 64" "--integrator whitted --spp 64" "--light sphere --spp 64" "--materials textured --spp 64" "--integrator directlighting" "--integrator whitted"; do
    name=$(echo "$v" | tr -d '-' | tr ' ' '_')
    timeout -k 10 300 python3 bench.py $v --no-cpu-baseline --no-spp1024 > "$out/bench_$name.json" 2> "$out/bench_$name.err" || { echo "bench $v failed"; tail -3 "$out/bench_$name.err"; exit 1; }
    echo "$v: $(cut -c1-60 $out/bench_$name.json)"
  done
elif [ "$part" = c ]; then
  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$out/pytest_gpu.txt"
else
  BENCH_REHEARSE=gloo timeout -k 10 900 python3 bench.py --gpus 4 --steps 1 --warmup 1 > "$out/bench_rehearse_4ranks_one_gpu_gloo.json" 2> "$out/bench_rehearse_4ranks.err" || { echo "rehearsal failed"; tail -5 "$out/bench_rehearse_4ranks.err"; exit 1; }
  python3 -c "import json; d=json.load(open('$out/bench_rehearse_4ranks_one_gpu_gloo.json')); print('4 ranks on one GPU: %.1f Mrays/s, reduced_film_ok %s, parity %.1e, tiles %s' % (d['value'], d['reduced_film_ok'], d['parity']['rel_l2'], [e['tiles'] for e in d['per_rank']]))"
  exe=pbrt-r3_amd/csrc/pbrt_gpu
  sed 's/"integer xresolution" \[64\] "integer yresolution" \[64\]/"integer xresolution" [1024] "integer yresolution" [1024]/' tests/scenes/cornell.pbrt > "$out/cornell_1024.pbrt"; cp tests/scenes/cornell_blocks.pbrt "$out/"
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 $exe "$out/cornell_1024.pbrt" -o "$out/cornell_8ranks.pfm" --pixelsamples 64 --devices 0,0,0,0,0,0,0,0 --stats 2> "$out/pbrt_gpu_8ranks_one_gpu.txt"; echo "pbrt_gpu 8 ranks rc=$?"; grep -E "^ *rank|tiles" "$out/pbrt_gpu_8ranks_one_gpu.txt" | head -12
  rm -f "$out/cornell_8ranks.pfm"
fi
