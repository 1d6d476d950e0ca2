#!/bin/bash
# Round-end evidence, run on the GPU box: tools/final_profile.sh TAG
#   part A  FETCH_SIZE calibration, the default bench line, the same command under rocprofv3 --kernel-trace --stats, the
#           FETCH_SIZE / WRITE_SIZE passes (separate counter-only runs) and the SQ counter passes   -> gpurun_out/ev_TAGa
#   part B  L1 access-pattern micro-benchmark, traffic of the default workload, the secondary bench lines, BVH build times and the
#           one-GPU strong-scaling probe                                                            -> gpurun_out/ev_TAGb
# Copy what is to be judged into profiles/ afterwards (profiles/README.md lists the files).
cd "$(dirname "$0")/.."
tools/r02_evidence_a.sh "${1}a" && tools/r02_evidence_b.sh "${1}b"
