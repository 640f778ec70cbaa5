#!/bin/bash
# same-box A/B: the working tree's warp.hip against the committed one (tools/_warp_head.hip)
cd "$(dirname "$0")/.."
for rep in 1 2; do
echo "--- new (rep $rep) ---"; python tools/quick_warp.py smooth 2>&1 | tail -3
cp cerberusnet_amd/csrc/warp.hip /tmp/warp_new.hip; cp tools/_warp_head.hip cerberusnet_amd/csrc/warp.hip
python -m cerberusnet_amd.build > /dev/null 2>&1
echo "--- committed (rep $rep) ---"; python tools/quick_warp.py smooth 2>&1 | tail -3
cp /tmp/warp_new.hip cerberusnet_amd/csrc/warp.hip; python -m cerberusnet_amd.build > /dev/null 2>&1
done
