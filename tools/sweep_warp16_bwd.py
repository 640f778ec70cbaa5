#!/usr/bin/env python3
"""GPU: decomposition sweep of the 16-bit warp backward (tile height, channel ranges per tile, phase shift).
    python tools/sweep_warp16_bwd.py [f16|bf16] [5|3] [levels]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
name = sys.argv[1] if len(sys.argv) > 1 else "f16"
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[name]
w, h = (2048, 1024) if (len(sys.argv) > 2 and sys.argv[2] == "5") else (1024, 512)
levels = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "3").split(",")]
for lvl in levels:
    C, H, W = pyramid_shapes(w, h, 32)[lvl]
    B = 4
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
    go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
    fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
    _, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
    for th in (0, 8, 16):
        for rg in (0, 1, 2, 4):
            for stg in (0, -1):
                _lib.set_option("warp_tile_h", th); _lib.set_option("warp_tile_ranges", rg); _lib.set_option("warp_stagger", stg)
                b, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True), 20, 5)
                t, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, False), 20, 5)
                print("%s %dx%d L%d (%d,%d,%d) tile_h %2d ranges %d stagger %2d: bwd %.1f us, tiles only %.1f" %
                      (name, w, h, lvl, C, H, W, th, rg, stg, b, t), flush=True)
    for k in ("warp_tile_h", "warp_tile_ranges", "warp_stagger"):
        _lib.set_option(k, 0)
