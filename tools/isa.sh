#!/bin/bash
# Device ISA of one kernel of pt_kernels.hip: tools/isa.sh k_trace [extra flags] -> /tmp/isa/<kernel>.s + register counts
k="${1:-k_trace}"; shift
mkdir -p /tmp/isa
cd "$(dirname "$0")/../pbrt-r3_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -mllvm -amdgpu-spill-vgpr-to-agpr=0 -S --cuda-device-only "$@" -o /tmp/isa/pt_kernels.s pt_kernels.hip 2>&1 | grep -E "error" -A5
awk "/^$k:/,/\.end_amdhsa_kernel/" /tmp/isa/pt_kernels.s > /tmp/isa/$k.s
awk "/\.name: *$k\$/,/\.wavefront_size/" /tmp/isa/pt_kernels.s | grep -E "vgpr_count|sgpr_count|spill_count|private_segment_fixed|group_segment_fixed"
echo "/tmp/isa/$k.s: $(wc -l < /tmp/isa/$k.s) lines"
