#!/usr/bin/env python3
"""Warp backward time vs flow magnitude at level 3 (4 pairs): smooth fields of growing
amplitude, and a +-2 px field with ONE 40x60 region moving by `amp` (per-tile scan regions:
only the tiles around it should pay)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
ops = torch.ops.cerberus
C, H, W = pyramid_shapes()[3]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()


def smooth(amp, seed=3):
    coarse = torch.from_numpy(hash_uniform((B, 2, H // 8, W // 8), seed, -float(amp), float(amp)))
    return torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear",
                                           align_corners=True).contiguous()


for amp in (2, 6, 10, 16, 24, 40, 64, 100):
    for kind in ("whole field", "one object"):
        if kind == "whole field":
            fl = smooth(amp)
        else:
            fl = smooth(2)
            fl[:, :, 40:80, 100:160] += float(amp)
        fl = fl.cuda()
        out, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
        t, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True), 10, 5)
        f, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 10, 5)
        print("amp %3d px, %-11s: warp fwd %6.1f us, bwd %7.1f us" % (amp, kind, f, t), flush=True)
