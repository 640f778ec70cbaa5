#!/usr/bin/env python3
"""Phase ablation of the fused warp backward tile kernel (needs a -DCERB_ABLATE build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
C, H, W = pyramid_shapes()[lvl]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
_, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
names = {0: "full", 1: "no grad_flow section", 2: "no atomics", 4: "no scan at all", 5: "zero + write back only",
         3: "scan loads + max only"}
for m in names:
    _lib.set_option("corr_debug_ablate", m)
    t, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True), 20, 5)
    print("L%d %-28s %6.1f us" % (lvl, names[m], t), flush=True)
_lib.set_option("corr_debug_ablate", 0)
