#!/bin/bash
# VGPR / spill summary of every step kernel (cross-compile only; no GPU needed)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -std=c++17 -c planeverb_amd/csrc/pv_kernels.hip -o /tmp/pv_k.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - | grep -E "${1:-step|small}" | sed -e 's/Function Name: _ZN3pva[0-9]*//'
