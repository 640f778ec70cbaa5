#!/usr/bin/env python3
"""GPU, -DCERB_ABLATE build (CERB_EXTRA_HIPCC_FLAGS=-DCERB_ABLATE python -m cerberusnet_amd.build):
which part of the matrix-core kernels owns the time (results are WRONG while the mask is set).
forward : 1 no global loads, 2 no LDS staging writes, 4 no T tile / read-back / stores, 8 T writes
          but no read-back / stores.
backward (argv[4] == "bwd"): 1 no gradOutput loads, 2 no window loads, 4 no window LDS writes,
          8 no band-row writes, 16 no MFMA loop, 32 no stores."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
from tools.tune_corr import timeit, P
ops = torch.ops.cerberus
C, H, W = (32, 256, 512) if len(sys.argv) < 4 else tuple(int(a) for a in sys.argv[1:4])
bwd = len(sys.argv) > 4 and sys.argv[4] == "bwd"
B = 4
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).half().cuda()
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).half().cuda()
go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).half().cuda()
for m in ((0, 1, 2, 3, 4, 7, 8, 16, 32, 63, 47, 31) if bwd else (0, 1, 2, 3, 4, 8, 7, 15)):
    _lib.set_option("corr_debug_ablate", m)
    if bwd:
        med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
    else:
        med, mn = timeit(lambda: ops.correlation(x1, x2, *P), 20, 5)
    print("mask %2d: %.1f us (min %.1f) %s" % (m, med, mn, _lib.last_kernel(1 if bwd else 0)), flush=True)
_lib.set_option("corr_debug_ablate", 0)
