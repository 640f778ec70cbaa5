#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tune_corr import timeit, P
ops = torch.ops.cerberus
for lvl, (C, H, W) in enumerate(pyramid_shapes()):
    B = 4
    x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
    mf, _ = timeit(lambda: ops.correlation(x1, x2, *P), 10, 5)
    mb, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 10, 5)
    for v in (2, 3):
        _lib.set_option("corr_bwd_variant", v)
        for cs in (0, 8, 16, 32):
            if cs > C: continue
            _lib.set_option("corr_bwd_cslice", cs)
            mb2, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 10, 5)
            print("   variant %d cslice %d bwd %.1f us [%s]" % (v, cs, mb2, _lib.last_kernel(1)))
    _lib.set_option("corr_bwd_cslice", 0)
    _lib.set_option("corr_bwd_variant", 0)
    print("L%d fwd %.1f us (%.0f GB/s)  bwd %.1f us (%.0f GB/s)  [%s | %s]" % (
        lvl, mf, (2*C+81)*B*H*W*4/mf/1e3, mb, (4*C+81)*B*H*W*4/mb/1e3, _lib.last_kernel(0), _lib.last_kernel(1)))
