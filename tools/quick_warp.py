#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cerberusnet_amd
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tune_corr import timeit
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload
ops = torch.ops.cerberus
for mode in (0, 1, 2):
    _lib.set_option("warp_pair_taps", mode)
    for lvl, (C, H, W) in enumerate(pyramid_shapes()):
        if lvl == 0:
            continue
        B = 4
        img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
        go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
        fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
        f, _ = timeit(lambda: ops.flow_warp(img, fl, 1, 0), 10, 5)
        k1, _ = timeit(lambda: ops.flow_warp_backward(img, fl, go, 1, 0, False, True), 10, 5)
        full, _ = timeit(lambda: ops.flow_warp_backward(img, fl, go, 1, 0, True, True), 10, 5)
        print("mode %d L%d  fwd %.1f  grad_flow only %.1f  full bwd %.1f us" % (mode, lvl, f, k1, full))
