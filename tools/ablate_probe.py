#!/usr/bin/env python3
"""Timing ablation of the L3 kernels: which phase owns the time?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tune_corr import timeit, P
ops = torch.ops.cerberus
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
C, H, W = pyramid_shapes()[lvl]
B = 4
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
names = {0: "full", 1: "no stores", 2: "no loads", 4: "no fma", 3: "no stores+loads", 5: "no stores+fma",
         6: "no loads+fma", 7: "nothing"}
names.update({64: "launch only", 7 + 32: "nothing, no epilogue", 7 + 32 + 8: "... no commit",
              7 + 32 + 8 + 16: "... no barrier", 7 + 8: "nothing, no commit", 7 + 16: "nothing, no barrier"})
for v in (0,):
    _lib.set_option("corr_fwd_variant", v)
    for m in (0, 7, 64, 7 + 32, 7 + 32 + 8, 7 + 32 + 8 + 16, 7 + 8, 7 + 16, 1, 4, 5):
        _lib.set_option("corr_debug_ablate", m)
        med, _ = timeit(lambda: ops.correlation(x1, x2, *P), 10, 5)
        print("fwd variant %d  %-16s %.1f us" % (v, names[m], med))
_lib.set_option("corr_bwd_cslice", 0)
names.update({32: "gather only"})
for m in (0, 64, 32, 7, 4, 2, 1):
    _lib.set_option("corr_debug_ablate", m)
    med, _ = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), 10, 5)
    print("bwd cslice 32  %-16s %.1f us" % (names[m], med))
_lib.set_option("corr_debug_ablate", 0)
