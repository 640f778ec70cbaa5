#!/usr/bin/env python3
"""Driver for rocprofv3 passes over the loss side of the training step (bench.py extra.loss_side): the loss pyramid, the
full-resolution RGB warps (forward, grad_flow) and the gradOutput pass of the concat-buffer backward, a few launches each
on rotating copies (inputs from HBM); no timing here, the profiler collects."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401,E402
from cerberusnet_amd.synth import hash_uniform  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
ops = torch.ops.cerberus
B, H, W = 4, 512, 1024
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t = lambda shape, seed, lo=-1.0, hi=1.0: torch.from_numpy(hash_uniform(shape, seed, lo, hi)).to(dev)
sizes = [H, W, H // 2, W // 2, H // 4, W // 4, H // 8, W // 8]
for i in range(reps):
    img = t((B, 3, H, W), 10 + i, -2.0, 2.0)
    flo = bench.Workload._flow(B, H, W, 20 + i, "smooth", dev)
    go = t((B, 3, H, W), 30 + i)
    ops.area_pyramid(img, sizes)
    ops.flow_warp(img, flo, 1, 0)
    ops.flow_warp_backward(img, flo, go, 1, 0, False, True)
    x1, x2 = t((B, 32, H // 4, W // 4), 40 + i), t((B, 32, H // 4, W // 4), 50 + i)
    g, f = t((B, 115, H // 4, W // 4), 60 + i), t((B, 115, H // 4, W // 4), 70 + i)
    ops.correlation_backward_leaky(x1, x2, g, f, 0, 4, 1, 4, 1, 1, 1, 0.1)
torch.cuda.synchronize()
