#!/usr/bin/env python3
"""GPU: check backward variants against the generic kernel and time them.
usage: quick_bwd.py [variants=1,4] [cslices=0] [pairs=4]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit, P

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,4").split(",")]
cslices = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ops = torch.ops.cerberus
shapes = list(pyramid_shapes()) + [(32, 124, 252), (30, 64, 128), (5, 9, 68), (7, 40, 72)]
for C, H, W in shapes:
    B = pairs
    x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
    _lib.set_option("corr_force_generic", 1)
    r1, r2 = ops.correlation_backward(x1, x2, go, *P)
    _lib.set_option("corr_force_generic", 0)
    bb = (4 * C + 81) * B * H * W * 4
    for v in variants:
        _lib.set_option("corr_bwd_variant", v)
        for cs in cslices:
            if cs > C:
                continue
            _lib.set_option("corr_bwd_cslice", cs)
            g1, g2 = ops.correlation_backward(x1, x2, go, *P)
            name = _lib.last_kernel(1)
            e1 = float((g1 - r1).abs().max() / r1.abs().max())
            e2 = float((g2 - r2).abs().max() / r2.abs().max())
            med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
            print(json.dumps(dict(shape=[B, C, H, W], variant=v, cslice=cs, kernel=name,
                                  err=[e1, e2], us=round(med, 2), us_min=round(mn, 2),
                                  TBps=round(bb / med / 1e6, 2))), flush=True)
    _lib.set_option("corr_bwd_variant", 0)
    _lib.set_option("corr_bwd_cslice", 0)
