#!/usr/bin/env python3
"""GPU soak test of the hand-synchronised 16-bit correlation forward (column walk: counted vmcnt waits, raw barriers, LDS-DMA
ring): random shapes and data for `seconds`, every result compared bit for bit with the register-staged form (variant 20),
half of the launches with another stream keeping the memory system busy, some with forced tiles-per-walk.
    python tools/soak_corr16.py [seconds=120] [seed=1]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
ops = torch.ops.cerberus
P = (4, 1, 4, 1, 1, 1)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device="cuda")
noise_b = torch.empty_like(noise_a)
t_end = time.time() + seconds
n = bad = 0
while time.time() < t_end:
    B = int(rng.integers(1, 5)); C = int(rng.integers(17, 65)); H = int(rng.integers(1, 97)); W = 8 * int(rng.integers(1, 41))
    dt = torch.float16 if rng.integers(0, 2) else torch.bfloat16
    x1 = torch.randn(B, C, H, W, device="cuda").to(dt)
    x2 = torch.randn(B, C, H, W, device="cuda").to(dt)
    slope = float(rng.choice([0.1, 1.0, 0.0, 2.0]))
    _lib.set_option("corr_fwd_variant", 20)
    ref = ops.correlation_leaky(x1, x2, *P, slope)
    _lib.set_option("corr_fwd_variant", 0)
    nw = int(rng.choice([0, 0, 1, 2, 3, 7]))
    _lib.set_option("corr_bwd_cslice", nw)
    busy = bool(rng.integers(0, 2))
    if busy:
        with torch.cuda.stream(side):
            for _ in range(3):
                noise_b.copy_(noise_a)
    outs = [ops.correlation_leaky(x1, x2, *P, slope) for _ in range(3)]
    name = _lib.last_kernel(0)
    _lib.set_option("corr_bwd_cslice", 0)
    torch.cuda.synchronize()
    for o in outs:
        if not torch.equal(o.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            print("MISMATCH", (B, C, H, W), dt, slope, nw, busy, name, int((o.view(torch.int16) != ref.view(torch.int16)).sum()), flush=True)
            break
    n += 1
print("soak: %d shapes x 3 launches in %.0f s, %d mismatches (last kernel %s)" % (n, seconds, bad, name))
sys.exit(1 if bad else 0)
