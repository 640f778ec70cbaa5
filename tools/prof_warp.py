#!/usr/bin/env python3
"""A few plain launches of the warp ops per level (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from bench import Workload
ops = torch.ops.cerberus
levels = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3").split(",")]
for lvl in levels:
    C, H, W = pyramid_shapes()[lvl]
    B = 4
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
    for _ in range(20):
        out, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
        ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True)
    torch.cuda.synchronize()
