#!/bin/bash
# Fabric (L2-miss) traffic of k_trace for ANY bench workload: FETCH_SIZE and WRITE_SIZE in separate counter-only passes
# (MI355X_MICROARCH.md, HBM section), one frame each, on exactly the workload the bench line is quoted on.
# usage: tools/pmc_traffic_workload.sh TAG [bench.py workload flags...]
#   -> gpurun_out/pmc_TAG/{fetch,write}/..., gpurun_out/pmc_TAG/traffic_entry.json (an entry for profiles/traffic_rNN.json)
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag="$1"; shift
for pass in fetch write; do
  if [ $pass = fetch ]; then set="FETCH_SIZE"; else set="WRITE_SIZE"; fi
  out="gpurun_out/pmc_$tag/$pass"
  mkdir -p "$out"
  timeout -k 10 ${PMC_TIMEOUT:-500} rocprofv3 --pmc $set --output-format csv -d "$out" -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-spp1024 > "$out/bench.json" 2> "$out/bench.err" || { echo "pass $pass failed"; tail -5 "$out/bench.err"; exit 1; }
  echo "pass $pass done"
done
kname=$(python3 -c "import json; print(json.loads(open('gpurun_out/pmc_$tag/fetch/bench.json').read().strip().splitlines()[-1])['roofline']['kernel'].split()[0])")
python3 tools/pmc_summarize.py "gpurun_out/pmc_$tag" "$kname" > "gpurun_out/pmc_$tag/summary.txt"
python3 - "$tag" "$*" <<'PY'
import hashlib, json, sys
tag, flags = sys.argv[1], sys.argv[2]
raw = json.load(open("gpurun_out/pmc_%s/k_trace_traffic_raw.json" % tag))
line = json.loads(open("gpurun_out/pmc_%s/fetch/bench.json" % tag).read().strip().splitlines()[-1])
f, w = raw["fetch_kib_per_launch_raw"], raw["write_kib_per_launch_raw"]
ksha = hashlib.sha256(open("pbrt-r3_amd/csrc/pt_kernels.hip", "rb").read()).hexdigest()[:16]      # bench.py quotes an entry only for the kernel source it was measured on
out = {"kernel": line["roofline"]["kernel"], "kernels_sha16": ksha, "workload": line["config"]["workload_key"], "workload_text": line["config"]["workload"],
       "measured_on": "bench.py %s --steps 1 --warmup 0 --no-cpu-baseline --no-spp1024 under rocprofv3 --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE; mean over the frame's k_trace launches" % flags,
       "launches_in_pass": line["roofline"]["launches"], "avg_launch_ms_under_pmc": line["roofline"]["avg_launch_ms"],
       "fetch_kib_per_launch_raw": f, "write_kib_per_launch_raw": w, "fetch_factor": 2.0,
       "calibration": "profiles/r02_fetch_size_calibration.txt (FETCH_SIZE counts L2 line fills at half their size for 16-byte lane gathers -> x2; WRITE_SIZE exact)",
       "fabric_bytes_per_launch": (2.0 * f + w) * 1024.0,
       "note": "L2 misses of k_trace (served by the Infinity Cache or HBM), FETCH_SIZE x 2 + WRITE_SIZE, mean per launch of one frame of this workload"}
json.dump(out, open("gpurun_out/pmc_%s/traffic_entry.json" % tag, "w"), indent=1)
print(json.dumps(out))
PY
