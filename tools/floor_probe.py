#!/usr/bin/env python3
"""GPU: the fixed floor of the correlation kernels on the coarse-level shapes -- time per launch (hipGraph of 20) against the
channel count, and a one-element torch kernel for the graph-node gap.  DESIGN.md 3.1b."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd
from tune_corr import timeit
P = (4, 1, 4, 1, 1, 1)
ops = torch.ops.cerberus
dev = "cuda:0"
B = 4
def run(C, H, W):
    x1 = torch.randn(B, C, H, W, device=dev); x2 = torch.randn(B, C, H, W, device=dev)
    go = torch.randn(B, 81, H, W, device=dev)
    f = timeit(lambda: ops.correlation(x1, x2, *P), 20, 7)
    b = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 7)
    return f, b
t = torch.zeros(1024, device=dev)
print("tiny torch add_: %.2f us" % timeit(lambda: t.add_(1.0), 20, 7)[0])
for (H, W) in ((16, 32), (32, 64), (64, 128)):
    for C in (16, 32, 64, 128, 256):
        f, b = run(C, H, W)
        print("H%3d W%3d C%3d  fwd %6.2f (min %6.2f)  bwd %6.2f (min %6.2f)" % (H, W, C, f[0], f[1], b[0], b[1]), flush=True)
