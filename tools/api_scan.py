#!/usr/bin/env python3
"""GPU: every public op on the benched level shapes with its less-travelled arguments (no context, one gradient, other
padding, other batch sizes, mixed flow dtype), us per launch -- a scan for paths that are slower than a path that
computes MORE (round 6 found grad_flow-alone slower than both gradients this way)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
P = (4, 1, 4, 1, 1, 1)
def T(fn):
    return timeit(fn, 20, 5)[0]
for (w, h) in ((1024, 512), (2048, 1024)):
    for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
        if lvl in (0, 2): continue
        for dt in (torch.float32, torch.float16):
            for B in (1, 4, 8):
                if B != 4 and (w, h) != (1024, 512): continue
                x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
                x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
                go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda().to(dt)
                gi = torch.from_numpy(hash_uniform((B, C, H, W), 4)).cuda().to(dt)
                fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
                fl32 = fl.float()
                out = ops.correlation_leaky(x1, x2, *P, 0.1)
                _, ctx = ops.flow_warp_ctx(x2, fl, 1, 0)
                r = {}
                r["corr"] = T(lambda: ops.correlation(x1, x2, *P))
                r["corr_leaky"] = T(lambda: ops.correlation_leaky(x1, x2, *P, 0.1))
                r["corr_bwd"] = T(lambda: ops.correlation_backward(x1, x2, go, *P))
                r["warp"] = T(lambda: ops.flow_warp(x2, fl, 1, 0))
                r["warp zeros"] = T(lambda: ops.flow_warp(x2, fl, 0, 0))
                r["warp_ctx"] = T(lambda: ops.flow_warp_ctx(x2, fl, 1, 0))
                r["wbwd_ctx"] = T(lambda: ops.flow_warp_backward_ctx(x2, fl, ctx, gi, 1, 0, True, True))
                r["wbwd_ctx zeros"] = T(lambda: ops.flow_warp_backward_ctx(x2, fl, ops.flow_warp_ctx(x2, fl, 0, 0)[1], gi, 0, 0, True, True)) - T(lambda: ops.flow_warp_ctx(x2, fl, 0, 0))
                r["wbwd noctx"] = T(lambda: ops.flow_warp_backward(x2, fl, gi, 1, 0, True, True))
                r["wbwd noctx gflow"] = T(lambda: ops.flow_warp_backward(x2, fl, gi, 1, 0, False, True))
                r["wbwd noctx gimg"] = T(lambda: ops.flow_warp_backward(x2, fl, gi, 1, 0, True, False))
                if dt != torch.float32:
                    r["warp f32flow"] = T(lambda: ops.flow_warp_ctx(x2, fl32, 1, 0))
                    _, ctx32 = ops.flow_warp_ctx(x2, fl32, 1, 0)
                    r["wbwd f32flow"] = T(lambda: ops.flow_warp_backward_ctx(x2, fl32, ctx32, gi, 1, 0, True, True))
                print("%dx%d L%d (%d,%d,%d,%d) %s: %s" % (w, h, lvl, B, C, H, W, str(dt)[6:], " | ".join("%s %.1f" % kv for kv in r.items())), flush=True)
