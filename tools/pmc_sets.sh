#!/bin/bash
# Hardware counters of k_trace in counter-only passes (one rocprofv3 --pmc run per set; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc_sets.sh TAG [LIB.so] -- "SET1 COUNTERS" "SET2 COUNTERS" ...   -> gpurun_out/pmc_TAG/summary.txt
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag="$1"; shift
lib=""
if [ "$1" != "--" ]; then lib="$1"; shift; fi
shift
[ -n "$lib" ] && export PBRTGPU_LIB="$PWD/$lib"
i=0
for set in "$@"; do
  out="gpurun_out/pmc_$tag/pass$i"
  mkdir -p "$out"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out" -- python3 bench.py --spp ${SPP:-32} --steps 1 --warmup 0 --no-cpu-baseline --no-spp1024 > "$out/bench.json" 2> "$out/bench.err" || { echo "pass $i failed"; tail -5 "$out/bench.err"; }
  echo "pass $i done: $(cut -c1-60 $out/bench.json)"
  i=$((i+1))
done
python3 - "$tag" <<'PY' | tee "gpurun_out/pmc_$1/summary.txt"
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_%s/pass*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k in sorted(tot):
    if not (k.startswith("k_trace") or k.startswith("k_shade")): continue
    print("==", k)
    for c in sorted(tot[k]): print("  %-36s %18.0f  over %d launches  (%.4g per launch)" % (c, tot[k][c], calls[k][c], tot[k][c] / max(1, calls[k][c])))
PY
