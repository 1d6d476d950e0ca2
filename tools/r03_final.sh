#!/bin/bash
# Round-3 end-of-round evidence, run on the GPU box: tools/r03_final.sh [a|b|c]
#   a  the default bench line (driver's command), the same under rocprofv3 --kernel-trace --stats, FETCH_SIZE / WRITE_SIZE of the default
#      workload (one frame each, separate counter-only passes), SQ and TCP counter passes on the final build
#   b  the regime lines: crown-class (3.5 M triangles, textured, 1024 spp), 16 M dense, 16 M sparse -- with the traffic file in place --, the
#      killeroo-class stand-in for config 4 (mixed materials, sphere light, Halton, 512 spp) and the secondary lines of round 2's table
#   c  the whole -m gpu suite
cd "$(dirname "$0")/.."
part="${1:-a}"
out="gpurun_out/r03_final_$part"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$part" = a ]; then
  timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
  cut -c1-400 "$out/bench.json"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$out/prof" --output-format csv -- python3 bench.py --no-cpu-baseline --no-spp1024 > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err" || { echo "rocprof run failed"; tail -5 "$out/bench_under_rocprof.err"; exit 1; }
  find "$out/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
  # the last frame's launches one by one (bounce by bounce): what the late, small bounces cost
  find "$out/prof" -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/per_bounce.py {} > "$out/per_bounce.txt"; rm -rf "$out/prof"
  head -10 "$out/kernel_stats.csv" | cut -c1-140
  bash tools/pmc_traffic_workload.sh r03_default && cp gpurun_out/pmc_r03_default/traffic_entry.json "$out/traffic_entry_default.json" && cp gpurun_out/pmc_r03_default/summary.txt "$out/pmc_fetch_write_default.txt"
  SPP=52 bash tools/pmc_sq.sh r03 > "$out/sq_summary.txt" 2>&1; tail -32 "$out/sq_summary.txt"
  SPP=32 bash tools/pmc_sets.sh r03tcp -- "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum" > "$out/tcp_summary.txt" 2>&1; tail -40 "$out/tcp_summary.txt"
elif [ "$part" = b ]; then
  timeout -k 10 600 python3 bench.py --triangles 3500000 --materials textured --spp 1024 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_crown_class.json" 2> "$out/bench_crown_class.err"
  timeout -k 10 600 python3 bench.py --triangles 16000000 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_16m.json" 2> "$out/bench_16m.err"
  timeout -k 10 600 python3 bench.py --triangles 16000000 --tri-size 0.00125 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_16m_sparse.json" 2> "$out/bench_16m_sparse.err"
  timeout -k 10 600 python3 bench.py --materials mixed --light sphere --sampler halton --spp 512 --steps 1 --warmup 1 --no-spp1024 --cpu-tiles 16 > "$out/bench_killeroo_class.json" 2> "$out/bench_killeroo_class.err"
  for f in crown_class 16m 16m_sparse killeroo_class; do python3 -c "import json,sys; d=json.load(open('$out/bench_$f.json')); r=d['roofline']; print('%-12s %8.1f Mrays/s  bound %s frac %.3f | l1 %.3f hbm %s | parity %.1e' % ('$f', d['value'], r['bound'], r['frac'], r['l1_req']['frac'], r['hbm'] and r['hbm']['frac'], d['parity']['rel_l2']))"; done
  for v in "--integrator ao" "--materials mixed --spp 64" "--sampler halton --spp 64" "--integrator directlighting --spp 64" "--integrator whitted --spp 64" "--light sphere --spp 64" "--materials textured --spp 64"; do
    name=$(echo "$v" | tr -d '-' | tr ' ' '_')
    timeout -k 10 300 python3 bench.py $v --no-cpu-baseline --no-spp1024 > "$out/bench_$name.json" 2> "$out/bench_$name.err" || { echo "bench $v failed"; tail -3 "$out/bench_$name.err"; exit 1; }
    echo "$v: $(cut -c1-60 $out/bench_$name.json)"
  done
else
  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > "$out/pytest_gpu.txt" 2>&1; echo "pytest rc=$?"; tail -4 "$out/pytest_gpu.txt"
fi
