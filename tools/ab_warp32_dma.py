#!/usr/bin/env python3
"""GPU: the fp32 warp forward through the LDS-DMA window kernel (warp16.hip; option warp_pair16 = 1) against the
general staged kernel: bit-equality of outputs and context, us per launch, channels per workgroup."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
for (w, h) in ((1024, 512), (2048, 1024), (896, 448)):
    for kind in ("smooth", "noise"):
        for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
            if lvl == 0:
                continue
            B = 4
            img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
            fl = Workload._flow(B, H, W, 3, kind, "cuda")
            _lib.set_option("warp_pair16", 0)
            o0, c0 = ops.flow_warp_ctx(img, fl, 1, 0)
            t0, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
            row = []
            same = True
            for cr in (0, 8, 16, 32):
                _lib.set_option("warp_pair16", 1); _lib.set_option("warp_staged", cr)
                o1, c1 = ops.flow_warp_ctx(img, fl, 1, 0)
                same = same and bool(torch.equal(o0.view(torch.int32), o1.view(torch.int32)) and torch.equal(c0, c1))
                t1, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
                row.append("%d: %.1f" % (cr, t1))
            _lib.set_option("warp_pair16", 0); _lib.set_option("warp_staged", 0)
            print("f32 %dx%d L%d (%d,%d,%d) %s: staged %.1f us | dma window by channels per workgroup %s | same bits %s"
                  % (w, h, lvl, C, H, W, kind, t0, " | ".join(row), same), flush=True)
