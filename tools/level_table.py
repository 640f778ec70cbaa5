#!/usr/bin/env python3
"""GPU: every launch of the op step per level, dtype and frame, hot, us per launch at 4 pairs + the kernel the dispatch took:
    python tools/level_table.py [W H] [dtypes]
(the table behind DESIGN.md's 16-bit-vs-fp32 comparison of round 6)."""
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
frames = [(1024, 512), (2048, 1024)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
dts = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
names = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "f16", "bf16"]
B = 4
for (w, h) in frames:
    for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
        for nm in names:
            dt = dts[nm]
            x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
            x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
            go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda().to(dt)
            f, _ = timeit(lambda: ops.correlation(x1, x2, *P), 20, 5)
            kf = _lib.last_kernel(0)
            b, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
            kb = _lib.last_kernel(1)
            line = "%dx%d L%d (%d,%d,%d) %-4s corr fwd %6.1f  bwd %6.1f" % (w, h, lvl, C, H, W, nm, f, b)
            if lvl > 0:
                fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
                tf, _ = timeit(lambda: ops.flow_warp_ctx(x2, fl, 1, 0), 20, 5)
                _, ctx = ops.flow_warp_ctx(x2, fl, 1, 0)
                tb, _ = timeit(lambda: ops.flow_warp_backward_ctx(x2, fl, ctx, x1, 1, 0, True, True), 20, 5)
                line += "  warp fwd %6.1f  bwd %6.1f" % (tf, tb)
            print(line + "   [%s | %s]" % (kf, kb), flush=True)
