#!/usr/bin/env python3
"""GPU: time the strip backward with experiment flags (corr_bwd_cslice carries StripCfg::FLAGS)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
from tools.tune_corr import timeit, P
ops = torch.ops.cerberus
flags = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4,8,16,24,28").split(",")]
for B, C, H, W in [(4, 32, 128, 256), (8, 32, 128, 256)]:
    x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
    _lib.set_option("corr_force_generic", 1)
    r1, r2 = ops.correlation_backward(x1, x2, go, *P)
    _lib.set_option("corr_force_generic", 0)
    med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
    print(json.dumps(dict(shape=[B, C, H, W], kernel=_lib.last_kernel(1), us=round(med, 2), us_min=round(mn, 2))), flush=True)
    _lib.set_option("corr_bwd_variant", 12)
    for f in flags:
        _lib.set_option("corr_bwd_cslice", f)
        g1, g2 = ops.correlation_backward(x1, x2, go, *P)
        e1 = float((g1 - r1).abs().max() / r1.abs().max()); e2 = float((g2 - r2).abs().max() / r2.abs().max())
        med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 20, 5)
        print(json.dumps(dict(shape=[B, C, H, W], flags=f, kernel=_lib.last_kernel(1), err=[e1, e2], us=round(med, 2), us_min=round(mn, 2))), flush=True)
    _lib.set_option("corr_bwd_variant", 0); _lib.set_option("corr_bwd_cslice", 0)
