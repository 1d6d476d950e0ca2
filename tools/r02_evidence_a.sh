#!/bin/bash
# Round-2 evidence, part A (run on the GPU box): FETCH_SIZE calibration, the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, the FETCH_SIZE / WRITE_SIZE passes and the SQ counter passes.  usage: tools/r02_evidence_a.sh TAG
cd "$(dirname "$0")/.."
tag="$1"
out="gpurun_out/ev_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== FETCH_SIZE calibration"
timeout -k 10 200 tools/ubench/fetch_calib > "$out/fetch_calib_plain.txt" 2>&1 || { echo "calib failed"; tail -3 "$out/fetch_calib_plain.txt"; exit 1; }
cat "$out/fetch_calib_plain.txt"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/calib" -- tools/ubench/fetch_calib > "$out/fetch_calib_pmc.txt" 2>&1 || { echo "calib pmc failed"; tail -3 "$out/fetch_calib_pmc.txt"; exit 1; }
python3 - "$out" <<'PY' | tee "$out/fetch_calib_counters.txt"
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-14s %s = %s" % (r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Counter_Value"]))
PY
echo "== bench (default flags)"
timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
cut -c1-600 "$out/bench.json"
echo "== rocprofv3 --kernel-trace --stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$out/prof" --output-format csv -- python3 bench.py --no-cpu-baseline --no-spp1024 > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err" || { echo "rocprof run failed"; tail -5 "$out/bench_under_rocprof.err"; exit 1; }
find "$out/prof" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
head -12 "$out/kernel_stats.csv"
rm -rf "$out/prof"
echo "== PMC traffic"
SPP=52 tools/pmc_traffic.sh "$tag" 2>&1 | tail -3
cp "gpurun_out/pmc_$tag/traffic.json" "$out/traffic_raw.json"
cp "gpurun_out/pmc_$tag/summary.txt" "$out/pmc_fetch_write_summary.txt"
echo "== SQ counters"
SPP=52 tools/pmc_sq.sh "$tag" > "$out/sq_summary.txt" 2>&1
tail -40 "$out/sq_summary.txt"
