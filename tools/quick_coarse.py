#!/usr/bin/env python3
"""GPU: the coarse-level kernels (corr_coarse.hip) forced (forward variant 15, backward variant 14) against the oracle,
and timed against the dispatch without them (16 / 15).  DESIGN.md 3.1b."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
import oracle
from tune_corr import timeit
P = (4, 1, 4, 1, 1, 1)
ops = torch.ops.cerberus
dev = "cuda:0"
def rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
for shape in ((2, 128, 7, 32), (1, 64, 5, 64), (2, 128, 8, 16), (4, 256, 16, 32), (4, 128, 32, 64)):
    B, C, H, W = shape
    x1, x2, go = hash_uniform(shape, 43), hash_uniform(shape, 44), hash_uniform((B, 81, H, W), 45)
    ref = oracle.corr_forward_ref(x1, x2, 4, 1, 4, 1, 1)
    r1, r2 = oracle.corr_backward_ref(x1, x2, go, 4, 1, 4, 1, 1)
    _lib.set_option("corr_fwd_variant", 15); _lib.set_option("corr_bwd_variant", 14)
    t1, t2, tg = (torch.from_numpy(a).to(dev) for a in (x1, x2, go))
    out = ops.correlation(t1, t2, *P); fn = _lib.last_kernel(0)
    g1, g2 = ops.correlation_backward(t1, t2, tg, *P); bn = _lib.last_kernel(1)
    _lib.set_option("corr_fwd_variant", 0); _lib.set_option("corr_bwd_variant", 0)
    print(shape, fn, bn, "rel err fwd %.2e g1 %.2e g2 %.2e" % (rel(out.cpu().numpy(), ref), rel(g1.cpu().numpy(), r1), rel(g2.cpu().numpy(), r2)), flush=True)
for shape in ((4, 256, 16, 32), (4, 128, 32, 64), (8, 256, 16, 32), (8, 128, 32, 64), (1, 256, 16, 32), (1, 128, 32, 64)):
    B, C, H, W = shape
    x1 = torch.from_numpy(hash_uniform(shape, 1)).to(dev); x2 = torch.from_numpy(hash_uniform(shape, 2)).to(dev)
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).to(dev)
    res = []
    for fv, bv in ((16, 15), (15, 14), (16, 15), (15, 14)):
        _lib.set_option("corr_fwd_variant", fv); _lib.set_option("corr_bwd_variant", bv)
        ops.correlation(x1, x2, *P); fn = _lib.last_kernel(0).replace("corr_fwd_d4_", "")
        ops.correlation_backward(x1, x2, go, *P); bn = _lib.last_kernel(1).replace("corr_bwd_d4_", "")
        tf = timeit(lambda: ops.correlation(x1, x2, *P), 20, 7)[0]
        tb = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 7)[0]
        res.append("fwd %s %.2f  bwd %s %.2f" % (fn, tf, bn, tb))
    _lib.set_option("corr_fwd_variant", 0); _lib.set_option("corr_bwd_variant", 0)
    print(shape, " | ".join(res), flush=True)
