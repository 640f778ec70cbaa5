#!/usr/bin/env python3
"""fp16 / bf16 storage vs fp32 on the pyramid levels (forward, backward), us per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit, P
ops = torch.ops.cerberus
for lvl, (C, H, W) in enumerate(pyramid_shapes()):
    B = 4
    row = []
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda().to(dt)
        f, _ = timeit(lambda: ops.correlation(x1, x2, *P), 20, 8)
        kf = _lib.last_kernel(0)
        b, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 8)
        kb = _lib.last_kernel(1)
        row.append("%s fwd %.1f bwd %.1f [%s | %s]" % (str(dt).split(".")[-1], f, b, kf, kb))
    print("L%d  " % lvl + "   ".join(row), flush=True)
