#!/bin/bash
# Round-end evidence, run on the GPU box: the default bench line, the same command under rocprofv3 --kernel-trace
# --stats, and the FETCH_SIZE / WRITE_SIZE counter passes.  usage: tools/final_profile.sh TAG
cd "$(dirname "$0")/.."
tag="$1"
out="gpurun_out/final_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== bench (default flags)"
timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
cut -c1-400 "$out/bench.json"
echo "== rocprofv3 --kernel-trace --stats (no cpu baseline leg)"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$out/prof" --output-format csv -- python3 bench.py --no-cpu-baseline --no-spp1024 > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err" || { echo "rocprof run failed"; tail -5 "$out/bench_under_rocprof.err"; exit 1; }
find "$out/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
head -8 "$out/kernel_stats.csv"
echo "== PMC traffic"
SPP=52 tools/pmc_traffic.sh "$tag" 2>&1 | tail -3
cp "gpurun_out/pmc_$tag/traffic.json" "$out/traffic.json"
cp "gpurun_out/pmc_$tag/summary.txt" "$out/pmc_summary.txt"
echo "== mixed-material workload"
timeout -k 10 300 python3 bench.py --spp 64 --materials mixed --no-cpu-baseline --no-spp1024 > "$out/bench_mixed.json" 2> /dev/null
cut -c1-200 "$out/bench_mixed.json"
