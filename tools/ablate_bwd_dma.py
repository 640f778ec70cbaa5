#!/usr/bin/env python3
"""Phase ablation of corr_bwd_d4_dma_kernel on one pyramid level (needs a -DCERB_ABLATE build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
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
_lib.set_option("corr_bwd_variant", 4)
names = {0: "full", 64: "launch only", 32: "prologue + gather", 2: "no DMA in loop", 1: "no stores",
         3: "no DMA, no stores", 4: "no LDS reads/FMA", 2 + 4 + 1: "gather + empty loop (barriers)"}
for m in names:
    _lib.set_option("corr_debug_ablate", m)
    med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
    print("%-32s %6.1f us (min %.1f)" % (names[m], med, mn), flush=True)
_lib.set_option("corr_debug_ablate", 0)
