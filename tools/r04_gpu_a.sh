#!/bin/bash
# Round 4, run A: where the time goes, kernel by kernel, on the workloads the last review named -- the killeroo-class and crown-class
# stand-ins, mixed materials, the two recursive integrators -- at 64 spp under rocprofv3 --kernel-trace --stats.
# usage: tools/r04_gpu_a.sh [tag]     (PBRTGPU_LIB selects a variant build; WORKLOADS="killeroo crown mixed direct whitted" picks a subset)
cd "$(dirname "$0")/.."
tag="${1:-base}"
out="gpurun_out/r04a_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
declare -A W
W[killeroo]="--materials mixed --light sphere --sampler halton"
W[crown]="--triangles 3500000 --materials textured"
W[mixed]="--materials mixed"
W[textured]="--materials textured"
W[sphere]="--light sphere"
W[direct]="--integrator directlighting"
W[whitted]="--integrator whitted"
W[head]=""
for w in ${WORKLOADS:-killeroo crown mixed direct whitted}; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out/prof_$w" --output-format csv -- python3 bench.py ${W[$w]} --spp ${SPP:-64} --steps 1 --warmup 1 --no-cpu-baseline --no-spp1024 > "$out/bench_$w.json" 2> "$out/bench_$w.err" || { echo "$w failed"; tail -5 "$out/bench_$w.err"; exit 1; }
  find "$out/prof_$w" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats_$w.csv"; rm -rf "$out/prof_$w"
  python3 - "$out/bench_$w.json" "$out/kernel_stats_$w.csv" "$w" <<'PY'
import csv, json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("== %-9s %8.1f Mrays/s  trace share %.3f shade share %.3f  bvh %.1f + upload %.1f ms" % (sys.argv[3], d["value"], r["trace_share_of_render"], r["shade_share_of_render"],
      d["config"]["bvh_build_ms"], d["config"]["upload_ms"]))
rows = list(csv.DictReader(open(sys.argv[2])))
tot = sum(float(x["TotalDurationNs"]) for x in rows)
for x in rows[:10]:
    print("   %-34s calls %5s total %9.2f ms avg %8.3f ms %6.2f%%" % (x["Name"][:34], x["Calls"], float(x["TotalDurationNs"]) / 1e6, float(x["AverageNs"]) / 1e6, 100.0 * float(x["TotalDurationNs"]) / tot))
PY
done
