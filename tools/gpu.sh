#!/bin/bash
# usage: tools/gpu.sh <timeout_s> '<command>'  -- gpurun with retries while the pod's GPU slots are busy (nothing is charged then)
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /tmp/_gpu_last.log 2>&1
  if grep -q "status=transient" /tmp/_gpu_last.log; then sleep 45; else break; fi
done
grep -v amdgpu.ids /tmp/_gpu_last.log
