#!/usr/bin/env python3
"""GPU: the correlation / warp kernels on the W32 pyramids of frames off the tuned widths (896 x 448, 1216 x 352), us per
launch at 4 pairs: default dispatch, and the strip backward forced (corr_bwd_variant 12)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload, corr_bytes, warp_bytes
ops = torch.ops.cerberus
P = (4, 1, 4, 1, 1, 1)
frames = [(896, 448), (1216, 352), (1024, 512)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for (w, h) in frames:
    for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
        B = 4
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
        cf, cb = corr_bytes(C, B, H, W)
        f, _ = timeit(lambda: ops.correlation(x1, x2, *P), 20, 5)
        kf = _lib.last_kernel(0)
        b0, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
        k0 = _lib.last_kernel(1)
        _lib.set_option("corr_bwd_variant", 12)
        b1, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
        k1 = _lib.last_kernel(1)
        _lib.set_option("corr_bwd_variant", 15)
        b2, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
        k2 = _lib.last_kernel(1)
        _lib.set_option("corr_bwd_variant", 0)
        line = "%dx%d L%d (%d,%d,%d): corr fwd %.1f us (%.2f) %s | bwd %.1f us (%.2f) %s | strip forced %.1f us %s | no coarse %.1f us %s" % (
            w, h, lvl, C, H, W, f, cf / f / 8e6, kf, b0, cb / b0 / 8e6, k0, b1, k1, b2, k2)
        if lvl > 0:
            fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
            wf, wb = warp_bytes(C, B, H, W)
            tf, _ = timeit(lambda: ops.flow_warp_ctx(x2, fl, 1, 0), 20, 5)
            _, ctx = ops.flow_warp_ctx(x2, fl, 1, 0)
            tb, _ = timeit(lambda: ops.flow_warp_backward_ctx(x2, fl, ctx, x1, 1, 0, True, True), 20, 5)
            line += " | warp fwd %.1f us (%.2f) bwd %.1f us (%.2f)" % (tf, wf / tf / 8e6, tb, wb / tb / 8e6)
        print(line, flush=True)
