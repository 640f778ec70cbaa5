#!/usr/bin/env python3
"""GPU: check forward variants against the default kernel (bit pattern of the sum order
differs -> 1e-6) and time them on the pyramid levels.  usage: quick_fwd.py v1,v2,..."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit, P

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,7,9").split(",")]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ops = torch.ops.cerberus
shapes = list(pyramid_shapes()) + [(32, 127, 252), (32, 64, 128), (48, 9, 68), (16, 33, 40)]
for lvl, (C, H, W) in enumerate(shapes):
    B = pairs
    x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    _lib.set_option("corr_force_generic", 1)
    ref = ops.correlation(x1, x2, *P)
    _lib.set_option("corr_force_generic", 0)
    fb = (2 * C + 81) * B * H * W * 4
    for v in variants:
        _lib.set_option("corr_fwd_variant", v)
        out = ops.correlation(x1, x2, *P)
        name = _lib.last_kernel(0)
        err = float((out - ref).abs().max() / ref.abs().max())
        med, mn = timeit(lambda: ops.correlation(x1, x2, *P), 20, 5)
        print(json.dumps(dict(shape=[B, C, H, W], variant=v, kernel=name, err=err,
                              us=round(med, 2), us_min=round(mn, 2),
                              TBps=round(fb / med / 1e6, 2))), flush=True)
    _lib.set_option("corr_fwd_variant", 0)
