#!/bin/bash
# round 4: phase stamps of the warp-backward tile workgroups (a -DCERB_STAMP build made on the box)
cd "$(dirname "$0")/.."
CERB_EXTRA_HIPCC_FLAGS=-DCERB_STAMP python -m cerberusnet_amd.build --force > /dev/null 2>&1
for lvl in ${LEVELS:-3 2 1}; do
  echo "=== level $lvl forward role ==="; python tools/stamp_warp_fwd.py $lvl 2>&1 | grep -v amdgpu.ids
  echo "=== level $lvl lists ==="; python tools/stamp_warp.py $lvl smooth 2>&1 | grep -v amdgpu.ids
  [ -n "$NOSCAN" ] || { echo "=== level $lvl scan ==="; CERB_OPT=warp_no_lists python tools/stamp_warp.py $lvl smooth 2>&1 | grep -v amdgpu.ids; }
done
