#!/bin/bash
# Where k_trace's wave cycles go: SQ counters in counter-only passes (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
# usage: tools/pmc_sq.sh TAG  -> gpurun_out/sq_TAG/passN/..., gpurun_out/sq_TAG/summary.txt
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag="$1"
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
      "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAVES")
i=0
for set in "${sets[@]}"; do
  out="gpurun_out/sq_$tag/pass$i"
  mkdir -p "$out"
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d "$out" -- python3 bench.py --spp ${SPP:-52} --steps 1 --warmup 0 --no-cpu-baseline --no-spp1024 > "$out/bench.json" 2> "$out/bench.err" || { echo "pass $i failed"; tail -5 "$out/bench.err"; }
  echo "pass $i done"
  i=$((i+1))
done
python3 - "$tag" <<'PY' | tee "gpurun_out/sq_$1/summary.txt"
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/sq_%s/pass*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k in ("k_trace", "k_shade"):
    print("==", k)
    for c in sorted(tot[k]): print("  %-24s %16.0f  over %d launches" % (c, tot[k][c], calls[k][c]))
    t = tot[k]
    if t.get("SQ_WAVE_CYCLES"):
        w = t["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in t: print("  %-24s / WAVE_CYCLES = %.3f" % (c, t[c] / w))
    if t.get("SQ_THREAD_CYCLES_VALU") and t.get("SQ_ACTIVE_INST_VALU"):
        print("  lanes active per VALU cycle = %.1f of 64" % (t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"]))
PY
