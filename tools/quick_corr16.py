#!/usr/bin/env python3
"""GPU: 16-bit correlation forward / backward timings on the 2048x1024 (config 5) and 1024x512
pyramid levels, per backward variant (0 = auto = matrix cores, 1 = VALU all-81-per-lane)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
from tools.tune_corr import timeit
ops = torch.ops.cerberus
p = (4, 1, 4, 1, 1, 1)
dts = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
names = sys.argv[1:] or ["f16"]
for (C, H, W) in [(256, 16, 32), (128, 32, 64), (64, 64, 128), (32, 128, 256),
                  (256, 32, 64), (128, 64, 128), (64, 128, 256), (32, 256, 512)]:
    B = 4
    for name in names:
        dt = dts[name]
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).to(dt).cuda()
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).to(dt).cuda()
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).to(dt).cuda()
        f, _ = timeit(lambda: ops.correlation(x1, x2, *p), 20, 5)
        res = []
        for v in ((0, 1) if dt != torch.float32 else (0,)):
            _lib.set_option("corr_bwd_variant", v)
            t, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *p), 20, 5)
            res.append("%d:%.1f (%s)" % (v, t, _lib.last_kernel(1)))
        _lib.set_option("corr_bwd_variant", 0)
        e = x1.element_size()
        bb = (4 * C + 81) * B * H * W * e
        print("%-14s %s fwd %.1f us | bwd %s | bwd bytes %.0f MB" % ((C, H, W), name, f, "  ".join(res), bb / 1e6), flush=True)
