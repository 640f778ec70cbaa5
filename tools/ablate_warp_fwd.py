#!/usr/bin/env python3
"""Phase ablation of the warp forward at level 3 (needs a -DCERB_ABLATE build)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
C, H, W = pyramid_shapes()[lvl]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
names = {0: "full", 1: "no stores", 2: "taps at the pixel itself (coords still computed, then dropped)",
         8: "taps at the pixel itself, coordinate chain skipped", 3: "no stores + identity taps"}
for m in names:
    _lib.set_option("corr_debug_ablate", m)
    t, _ = timeit(lambda: ops.flow_warp(img, fl, 1, 0), 20, 8)
    print("L%d %-64s %6.1f us" % (lvl, names[m], t), flush=True)
_lib.set_option("corr_debug_ablate", 0)
