#!/usr/bin/env python3
"""GPU: channels per workgroup of the 16-bit warp forward (option warp_staged = N), us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
for name, (w, h) in (("bf16", (1024, 512)), ("f16", (2048, 1024))):
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}[name]
    for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
        if lvl == 0:
            continue
        B = 4
        img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
        fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
        row = []
        for cr in (0, 4, 8, 16, 32, 64, 128):
            if cr > C:
                continue
            _lib.set_option("warp_staged", cr)
            t, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
            row.append("%d: %.1f" % (cr, t))
        _lib.set_option("warp_staged", 0)
        print("%s %dx%d L%d (%d,%d,%d): %s" % (name, w, h, lvl, C, H, W, " | ".join(row)), flush=True)
