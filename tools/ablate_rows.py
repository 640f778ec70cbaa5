#!/usr/bin/env python3
"""GPU, -DCERB_ABLATE build: which part of corr_bwd_d4_rows owns the time (results are WRONG
while the mask is set).  1: no gradOutput DMA, 2: no row DMA, 4: no FMAs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit, P
ops = torch.ops.cerberus
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
C, H, W = pyramid_shapes()[lvl]
B = 4
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
_lib.set_option("corr_bwd_variant", 6)
for m in (0, 1, 2, 3, 4, 5, 6, 7):
    _lib.set_option("corr_debug_ablate", m)
    med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
    print("mask %d: %.1f us (min %.1f)" % (m, med, mn), flush=True)
_lib.set_option("corr_debug_ablate", 0)
