#!/usr/bin/env python3
"""GPU: warp forward / backward timings for a dtype on the config-5 (2048x1024) or config-3 levels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
w, h = (2048, 1024) if (len(sys.argv) > 2 and sys.argv[2] == "5") else (1024, 512)
for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
    if lvl == 0:
        continue
    B = 4
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
    go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
    fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
    f, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
    _, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
    b, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True), 20, 5)
    t, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, False), 20, 5)
    print("%s %dx%d L%d (%d,%d,%d): fwd+ctx %.1f us, bwd %.1f us (tiles only %.1f)" % (sys.argv[1] if len(sys.argv) > 1 else "f16", w, h, lvl, C, H, W, f, b, t), flush=True)
