#!/usr/bin/env python3
"""A few plain launches of the correlation forward + backward on one level (for rocprofv3 --kernel-trace / --pmc).
    python3 tools/prof_corr16.py [f16|bf16|f32] [5|3] [level] [n]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
ops = torch.ops.cerberus
name = sys.argv[1] if len(sys.argv) > 1 else "f16"
dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[name]
w, h = (2048, 1024) if (len(sys.argv) > 2 and sys.argv[2] == "5") else (1024, 512)
lvl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
C, H, W = pyramid_shapes(w, h, 32)[lvl]
B = 4
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda().to(dt)
P = (4, 1, 4, 1, 1, 1)
for _ in range(n):
    ops.correlation(x1, x2, *P)
    ops.correlation_backward(x1, x2, go, *P)
torch.cuda.synchronize()
