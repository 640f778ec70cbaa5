#!/usr/bin/env python3
"""How does L3 forward/backward time scale with the batch (rounds of workgroups per CU)?"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tune_corr import timeit, P
ops = torch.ops.cerberus
C, H, W = pyramid_shapes()[3]
for B in (1, 2, 4, 8, 16):
    x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
    for v in (2, 7):
        _lib.set_option("corr_fwd_variant", v)
        med, mn = timeit(lambda: ops.correlation(x1, x2, *P), 10, 5)
        print("fwd B=%2d variant=%d  %.1f us  (%.2f us/pair)  %.0f GB/s" % (B, v, med, med / B, (2*C+81)*B*H*W*4/med/1e3))
    for cs in (16, 32):
        _lib.set_option("corr_bwd_cslice", cs)
        med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 10, 5)
        print("bwd B=%2d cslice=%d  %.1f us  (%.2f us/pair)  %.0f GB/s" % (B, cs, med, med / B, (4*C+81)*B*H*W*4/med/1e3))
