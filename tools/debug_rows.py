#!/usr/bin/env python3
"""GPU: where does corr_bwd variant 6 differ from the generic kernel?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
P = (4, 1, 4, 1, 1, 1)
ops = torch.ops.cerberus
for shape in [(1, 4, 8, 64), (1, 32, 16, 128)]:
    B, C, H, W = shape
    x1 = torch.from_numpy(hash_uniform(shape, 1)).cuda()
    x2 = torch.from_numpy(hash_uniform(shape, 2)).cuda()
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
    _lib.set_option("corr_force_generic", 1)
    r1, r2 = ops.correlation_backward(x1, x2, go, *P)
    _lib.set_option("corr_force_generic", 0)
    _lib.set_option("corr_bwd_variant", 6)
    g1, g2 = ops.correlation_backward(x1, x2, go, *P)
    _lib.set_option("corr_bwd_variant", 0)
    for name, g, r in (("g1", g1, r1), ("g2", g2, r2)):
        e = (g - r).abs() / r.abs().max()
        bad = e > 1e-5
        print(shape, name, "max err %.3g  bad fraction %.3f" % (float(e.max()), float(bad.float().mean())))
        print("  bad per channel:", bad.float().mean(dim=(0, 2, 3)).cpu().numpy().round(2))
        print("  bad per row    :", bad.float().mean(dim=(0, 1, 3)).cpu().numpy().round(2))
        print("  bad per col/4  :", bad.float().mean(dim=(0, 1, 2)).view(-1, 4).mean(1).cpu().numpy().round(2))
    # which displacement rows are wrong: feed gO with a single non-zero displacement row
    for s in range(9):
        go1 = torch.zeros_like(go)
        go1[:, s * 9:(s + 1) * 9] = go[:, s * 9:(s + 1) * 9]
        _lib.set_option("corr_force_generic", 1)
        r1, r2 = ops.correlation_backward(x1, x2, go1, *P)
        _lib.set_option("corr_force_generic", 0)
        _lib.set_option("corr_bwd_variant", 6)
        g1, g2 = ops.correlation_backward(x1, x2, go1, *P)
        _lib.set_option("corr_bwd_variant", 0)
        print("  dy row %d: err g1 %.3g g2 %.3g" % (s, float((g1 - r1).abs().max() / r1.abs().max()),
                                                     float((g2 - r2).abs().max() / r2.abs().max())))
