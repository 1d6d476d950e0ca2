#!/bin/bash
# Round 3, run R: kernel stats of the textured bench with the textured segment in one kernel (PBRTGPU_TEX_SPLIT=0) and in two (1).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03r
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for s in ${SPLITS:-0 1}; do
  export PBRTGPU_TEX_SPLIT=$s
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r03r/prof$s --output-format csv -- python3 bench.py --materials ${MATERIALS:-textured} --spp 64 --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 ${BENCH_EXTRA} > gpurun_out/r03r/bench_$s.json 2> gpurun_out/r03r/bench_$s.err
  find gpurun_out/r03r/prof$s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03r/kernel_stats_split$s.csv; rm -rf gpurun_out/r03r/prof$s
  echo "== split $s: $(cut -c1-60 gpurun_out/r03r/bench_$s.json)"
  python3 - gpurun_out/r03r/kernel_stats_split$s.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("   %-34s calls %4s total %9.2f ms avg %8.3f ms %6s%%" % (r["Name"][:34], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
done
