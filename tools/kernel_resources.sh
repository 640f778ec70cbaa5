#!/bin/bash
# Print VGPR / spill / scratch per kernel of one HIP source (compiler view).
src=${1:-/root/repo/cerberusnet_amd/csrc/corr_d4.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -Rpass-analysis=kernel-resource-usage \
  -c "$src" -o /tmp/_kr.o 2>&1 | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" \
  | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' -e 's/Function Name: _ZN4cerb12_GLOBAL__N_1[0-9]*/\n/' | tr '\n' ' ' | sed 's/corr_/\ncorr_/g; s/warp_/\nwarp_/g'; echo
