#!/bin/bash
# usage: tools/isa.sh <file.hip> [extra flags]  -> /tmp/<file>.s (device ISA of every kernel in the file, product flags)
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -S --cuda-device-only /root/repo/cerberusnet_amd/csrc/$F -o /tmp/${F%.hip}.s "$@" 2>&1 | grep -v "warning: argument unused"
echo /tmp/${F%.hip}.s
