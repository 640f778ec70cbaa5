#!/usr/bin/env python3
"""Driver for rocprofv3 counter passes: launches the L-level correlation kernels a
few times each (no timing here; the profiler collects)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes

P = (4, 1, 4, 1, 1, 1)
ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="3")
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--fwd-variant", type=int, default=0)
ap.add_argument("--bwd-cslice", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--warp", action="store_true")
ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"])
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--height", type=int, default=512)
args = ap.parse_args()
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
dev = "cuda:0"
ops = torch.ops.cerberus
_lib.set_option("corr_fwd_variant", args.fwd_variant)
_lib.set_option("corr_bwd_cslice", args.bwd_cslice)
for lvl in [int(x) for x in args.levels.split(",")]:
    C, H, W = pyramid_shapes(args.width, args.height)[lvl]
    B = args.pairs
    x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).to(DT).to(dev)
    x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).to(DT).to(dev)
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).to(DT).to(dev)
    from bench import Workload
    fl = Workload._flow(B, H, W, 4, "smooth", dev)   # the bench's flow field
    for _ in range(args.reps):
        ops.correlation(x1, x2, *P)
        ops.correlation_backward(x1, x2, go, *P)
        if args.warp:
            w, ctx = ops.flow_warp_ctx(x2, fl, 1, 0)
            ops.flow_warp_backward_ctx(x2, fl, ctx, w, 1, 0, True, True)
    torch.cuda.synchronize()
