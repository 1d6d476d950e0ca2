#!/bin/bash
# Round-2 evidence, part B (run on the GPU box): L1 access-pattern micro-benchmark, FETCH_SIZE / WRITE_SIZE of the default bench
# workload, the secondary bench lines, BVH build times and the one-GPU strong-scaling probe.  usage: tools/r02_evidence_b.sh TAG
cd "$(dirname "$0")/.."
tag="$1"
out="gpurun_out/ev_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== L1 access patterns"
timeout -k 10 120 tools/ubench/l1_patterns > "$out/l1_patterns.txt" 2>&1 || { echo "ubench failed"; tail -3 "$out/l1_patterns.txt"; exit 1; }
cat "$out/l1_patterns.txt"
echo "== PMC traffic (default workload, one step)"
SPP=256 tools/pmc_traffic.sh "$tag" 2>&1 | tail -3
cp "gpurun_out/pmc_$tag/traffic.json" "$out/traffic_raw.json"
cp "gpurun_out/pmc_$tag/summary.txt" "$out/pmc_fetch_write_summary.txt"
cp "gpurun_out/pmc_$tag/fetch/bench.json" "$out/pmc_fetch_bench.json"
echo "== secondary bench lines"
for v in "--integrator ao" "--materials mixed --spp 64" "--sampler halton --spp 64" "--integrator directlighting --spp 64" "--integrator whitted --spp 64" "--light sphere --spp 64" "--materials textured --spp 64"; do
  name=$(echo "$v" | tr -d '-' | tr ' ' '_')
  timeout -k 10 300 python3 bench.py $v --no-cpu-baseline --no-spp1024 > "$out/bench_$name.json" 2> "$out/bench_$name.err" || { echo "bench $v failed"; tail -3 "$out/bench_$name.err"; exit 1; }
  echo "$v: $(cut -c1-60 $out/bench_$name.json)"
done
echo "== BVH build"
PBRTGPU_BUILD_TRACE=1 timeout -k 10 300 python3 tools/bvh_build_time.py 1000000 3500000 > "$out/bvh_build.txt" 2>&1 || { echo "bvh build time failed"; tail -3 "$out/bvh_build.txt"; exit 1; }
tail -12 "$out/bvh_build.txt"
echo "== one-GPU strong-scaling probe"
timeout -k 10 400 python3 tools/strong_scaling_probe.py 256 > "$out/strong_scaling_probe.txt" 2>&1 || { echo "probe failed"; tail -3 "$out/strong_scaling_probe.txt"; exit 1; }
cat "$out/strong_scaling_probe.txt"
