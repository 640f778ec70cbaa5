#!/usr/bin/env python3
"""warp backward time vs flow amplitude (where does the tiled path hand over to the scatter?)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cerberusnet_amd
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tune_corr import timeit
ops = torch.ops.cerberus
C, H, W = pyramid_shapes()[3]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
for amp in (2, 6, 10, 14, 18, 24, 40):
    coarse = torch.from_numpy(hash_uniform((B, 2, H // 8, W // 8), 3, -float(amp), float(amp)))
    fl = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True).contiguous().cuda()
    t, _ = timeit(lambda: ops.flow_warp_backward(img, fl, go, 1, 0, True, True), 10, 5)
    f, _ = timeit(lambda: ops.flow_warp(img, fl, 1, 0), 10, 5)
    print("amp %2d px: warp fwd %.1f us, bwd %.1f us" % (amp, f, t))
