#!/usr/bin/env python3
"""GPU: the 16-bit warp kernels of warp16.hip (two elements per lane) against the general kernels they replace
(option warp_pair16 = -1): bit-equality of outputs / context / gradients on the benched levels, then us per launch.
    python tools/ab_warp16.py [f16|bf16] [5|3] [smooth|noise]"""
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
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[name]
w, h = (2048, 1024) if (len(sys.argv) > 2 and sys.argv[2] == "5") else (1024, 512)
kind = sys.argv[3] if len(sys.argv) > 3 else "smooth"
for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
    if lvl == 0:
        continue
    B = 4
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
    go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
    for fdt in (dt, torch.float32):
        fl = Workload._flow(B, H, W, 3, kind, "cuda").to(fdt)
        res = {}
        for opt in (-1, 0):
            _lib.set_option("warp_pair16", opt)
            out, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
            plain = ops.flow_warp(img, fl, 1, 0)
            gi, gf = ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True)
            f, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
            b, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True), 20, 5)
            t, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, False), 20, 5)
            res[opt] = (out, ctx, plain, gi, gf, f, b, t)
        _lib.set_option("warp_pair16", 0)
        a, n = res[-1], res[0]
        same = [bool(torch.equal(a[i].view(torch.int16) if a[i].dtype != torch.float32 and a[i].dtype != torch.int64 else a[i],
                                 n[i].view(torch.int16) if n[i].dtype != torch.float32 and n[i].dtype != torch.int64 else n[i])) for i in range(5)]
        print("%s flow %s %dx%d L%d (%d,%d,%d) %s: fwd+ctx %.1f -> %.1f us, bwd %.1f -> %.1f us (tiles only %.1f -> %.1f) "
              "| same bits out/ctx/plain/gi/gf: %s" % (name, "f32" if fdt == torch.float32 else name, w, h, lvl, C, H, W, kind,
                                                        a[5], n[5], a[6], n[6], a[7], n[7], same), flush=True)
