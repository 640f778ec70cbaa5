#!/usr/bin/env python3
"""GPU: SURVEY 8(f)-2, "sample x2 through the flow while staging the correlation's LDS window"
(eval path, no saved tensor), priced with this package's own kernels.

A fused kernel must produce every pixel of a tile's (TH+8) x (TW+8) window by bilinear
sampling, i.e. (TH+8)(TW+8)/(TH*TW) times the samples of the stand-alone warp, and saves the
warped tensor's write + read.  Lower bound of the fused staging cost = the stand-alone warp's
measured cost per pixel x that factor (the gather is the same four taps per pixel-channel; in
the fused kernel it would additionally sit on the critical path of the tile).  Compared with
what fusion saves: warp_fwd + corr_fwd today vs corr_fwd + bound."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit, P
from bench import Workload
ops = torch.ops.cerberus
C, H, W = pyramid_shapes()[3]
res = {}
for B in (4, 7, 9, 14):
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
    t, _ = timeit(lambda: ops.flow_warp(img, fl, 1, 0), 20, 5)
    res[B] = t
    print("warp_fwd (eval, no context) 32x128x256, B=%2d (%.2fx the pixels of B=4): %.1f us" % (B, B / 4, t))
x1 = torch.from_numpy(hash_uniform((4, C, H, W), 1)).cuda()
x2 = torch.from_numpy(hash_uniform((4, C, H, W), 2)).cuda()
tc, _ = timeit(lambda: ops.correlation(x1, x2, *P), 20, 5)
print("corr_fwd B=4: %.1f us; warp_fwd + corr_fwd today: %.1f us" % (tc, res[4] + tc))
per_px = (res[14] - res[4]) / (14 - 4)          # marginal us per batch item of 32768 px
base = res[4] - 4 * per_px                       # fixed cost
for name, f in (("4x64 tile (window 3.375x)", 3.375), ("8x64 tile (2.25x)", 2.25), ("16x64 tile (1.69x)", 1.6875)):
    extra = (f - 1.0) * 4 * per_px
    saved = 2 * 4 * C * H * W * 4 / 5.5e6       # warped tensor write + read at the ~5.5 TB/s these kernels sustain
    print("%-28s extra sampling >= %.1f us, saved round trip <= %.1f us -> fused >= %.1f us vs %.1f us today"
          % (name, extra, saved, res[4] + tc + extra - saved, res[4] + tc))
