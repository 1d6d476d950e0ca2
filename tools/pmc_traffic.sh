#!/bin/bash
# HBM traffic of k_trace: FETCH_SIZE and WRITE_SIZE in separate counter-only passes (MI355X_MICROARCH.md, HBM).
# usage: tools/pmc_traffic.sh TAG  -> gpurun_out/pmc_TAG/{fetch,write}/..., gpurun_out/pmc_TAG/traffic.json
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag="$1"
for pass in fetch write; do
  if [ $pass = fetch ]; then set="FETCH_SIZE"; else set="WRITE_SIZE"; fi
  out="gpurun_out/pmc_$tag/$pass"
  mkdir -p "$out"
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d "$out" -- python3 bench.py --spp ${SPP:-256} --steps 1 --warmup 0 --no-cpu-baseline --no-spp1024 > "$out/bench.json" 2> "$out/bench.err" || { echo "pass $pass failed"; tail -5 "$out/bench.err"; exit 1; }
  echo "pass $pass done"
done
python3 tools/pmc_summarize.py "gpurun_out/pmc_$tag" > "gpurun_out/pmc_$tag/summary.txt"
SPP="${SPP:-256}" python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
raw = json.load(open("gpurun_out/pmc_%s/k_trace_traffic_raw.json" % tag))
f, w = raw["fetch_kib_per_launch_raw"], raw["write_kib_per_launch_raw"]
import os
out = {"kernel": "k_trace", "config": "RT1M 1024x1024, one pass of %s spp" % os.environ.get("SPP", "256"),
       "fetch_kib_per_launch_raw": f, "write_kib_per_launch_raw": w,
       "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (guide, HBM section); WRITE_SIZE exact",
       "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
json.dump(out, open("gpurun_out/pmc_%s/traffic.json" % tag, "w"), indent=1)
print(out)
PY
