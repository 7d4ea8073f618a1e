#!/bin/bash
# Print VGPR / scratch / occupancy for every kernel instantiation whose (demangled) name matches a regex.
# Compile-only, no GPU needed.   usage: scripts/kernel_resources.sh [name-regex] [extra hipcc flags...]
filt=${1:-gemm_mfma_kernel}; shift
mkdir -p /tmp/cdna4_tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c -Rpass-analysis=kernel-resource-usage \
  "$(dirname "$(readlink -f "$0")")/../ik_llama.cpp_amd/csrc/cdna4_api.hip" -o /tmp/cdna4_tmp/api_res.o 2>&1 |
sed 's/ \[-Rpass.*//' |
awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {print name, "vgpr=" v, "scratch=" s, "occ=" $NF}' |
c++filt | grep -E "$filt"
