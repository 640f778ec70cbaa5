#!/usr/bin/env python3
"""GPU, -DCERB_ABLATE build (python -m cerberusnet_amd.build --variant ablate -DCERB_ABLATE; CERBERUS_HIP_LIB=...):
parts of the 16-bit warp forward switched off (wrong results), us per launch.
    python tools/ablate_warp16.py [f16|bf16] [5|3]"""
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
names = {0: "full", 1: "no stores", 2: "no staging loads", 4: "no taps / blends", 5: "no taps, no stores", 6: "no staging, no taps",
         7: "nothing but the set-up", 32: "exit at once", 64: "exit after flow + coordinates", 128: "exit after the context", 256: "exit after the box", 512: "exit after the cell map", 7 + 1024: "set-up, channel loop without barriers"}
for lvl, (C, H, W) in enumerate(pyramid_shapes(w, h, 32)):
    if lvl == 0:
        continue
    B = 4
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
    fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
    row = []
    for m in names:
        _lib.set_option("corr_debug_ablate", m)
        t, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
        row.append("%s %.1f" % (names[m], t))
    _lib.set_option("corr_debug_ablate", 0)
    print("%s %dx%d L%d (%d,%d,%d): %s" % (name, w, h, lvl, C, H, W, " | ".join(row)), flush=True)
