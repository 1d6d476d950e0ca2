#!/bin/bash
# PMC passes over a short bench run (counters only: no trace domains mixed in).
# usage: tools/pmc_run.sh TAG   -> gpurun_out/pmc_TAG/<pass>/...csv
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag="$1"; shift
i=0
for set in ${PMC_SETS_OVERRIDE:+} "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  out="gpurun_out/pmc_$tag/p$i"
  mkdir -p "$out"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out" -- python3 bench.py --spp ${SPP:-16} --steps 1 --warmup 0 --no-cpu-baseline --no-spp1024 > "$out/bench.json" 2> "$out/bench.err"
  echo "pass $i ($set): exit $?"
done
python3 tools/pmc_summarize.py "gpurun_out/pmc_$tag" | tee "gpurun_out/pmc_$tag/summary.txt"
