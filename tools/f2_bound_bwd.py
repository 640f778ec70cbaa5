#!/usr/bin/env python3
"""GPU: the BACKWARD half of SURVEY 8(f)-2 priced: correlation_backward -> flow_warp_backward fused, so
that g2 -- the gradient w.r.t. the warped feature map, (B, C, H, W), written by corr_bwd and re-read by warp_bwd -- never
makes the round trip through memory ("backward then produces grad_image / grad_flow directly", pwcnet_sfd.py:176-182
seen from autograd).

What fusion can save per level and direction: the write + read of g2 (2 x B*C*H*W*4 bytes at the rate these launches
move bytes) and ONE graph node.  What it must do instead: a warp-backward workgroup (a TH x 64 tile of grad_image for 8
channels at a time, fed by the source pixels whose taps land in it: the tile shifted by the flow) needs g2 AT ITS SOURCE
PIXELS, i.e. it has to run the second-gradient half of the correlation backward -- 81 FMAs per (pixel, channel) over a
9 x 9 window of gradOutput and x1 -- for its scan region:
  (1) as a TILE kernel: the strip backward's whole-rows-per-wavefront layout (no halo, neighbours by DPP) does not
      exist for a 2-D region; the tile formulations of the same arithmetic are this package's round-2 kernels
      (corr_bwd_variant 4), slower at every level -- measured here;
  (2) for region / tile x the pixels: the scan regions of neighbouring tiles overlap (measured here from the bench's
      flow field: bounding box of the sources of a 16 x 64 tile);
  (3) with the 81-plane gradOutput window staged per region instead of streamed once per image row.
The bound printed below charges the fused kernel only (1) and (2) on the second-gradient half and gives it the whole
saving; it is generous to fusion."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401,E402
from cerberusnet_amd import _lib  # noqa: E402
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes  # noqa: E402
from tools.tune_corr import P  # noqa: E402
from bench import Workload, _time_graph  # noqa: E402

ops = torch.ops.cerberus
NODE = 2.15          # an empty graph node, us (tools/ubench/launch_floor.hip)


def region_over_tile(flow, th=16, tw=64):
    """mean (bounding box of the pixels whose sample position falls into a th x tw tile) / (tile), from the flow"""
    B, _, H, W = flow.shape
    f = flow.cpu().numpy()
    ys, xs = np.mgrid[0:H, 0:W]
    ratios = []
    for b in range(B):
        px = np.clip((xs + f[b, 0]) * W / (W - 1) - 0.5, 0, W - 1)
        py = np.clip((ys + f[b, 1]) * H / (H - 1) - 0.5, 0, H - 1)
        tx, ty = (px // tw).astype(int), (py // th).astype(int)
        for j in range(H // th):
            for i in range(W // tw):
                m = (tx == i) & (ty == j)
                if m.any():
                    yy, xx = np.nonzero(m)
                    ratios.append((yy.max() - yy.min() + 1) * (xx.max() - xx.min() + 1) / float(th * tw))
    return float(np.mean(ratios))


for lvl in (1, 2, 3):
    C, H, W = pyramid_shapes()[lvl]
    B = 4
    mk = lambda shape, s: torch.from_numpy(hash_uniform(shape, s)).cuda()
    f1, f2, go = mk((B, C, H, W), 1), mk((B, C, H, W), 2), mk((B, 81, H, W), 3)
    fl = Workload._flow(B, H, W, 4, "smooth", "cuda")
    warped, ctx = ops.flow_warp_ctx(f2, fl, 1, 0)
    t_cb = _time_graph([lambda: ops.correlation_backward(f1, warped, go, *P)], 20) * 1e6
    k_cb = _lib.last_kernel(1)
    g2 = ops.correlation_backward(f1, warped, go, *P)[1]
    t_wb = _time_graph([lambda: ops.flow_warp_backward_ctx(f2, fl, ctx, g2, 1, 0, True, True)], 20) * 1e6

    def pair():
        a, b = ops.correlation_backward(f1, warped, go, *P)
        return a, ops.flow_warp_backward_ctx(f2, fl, ctx, b, 1, 0, True, True)
    t_pair = _time_graph([pair], 20) * 1e6
    # the tile formulation of the same correlation backward (what a 2-D region needs)
    _lib.set_option("corr_bwd_variant", 4)
    try:
        t_tile = _time_graph([lambda: ops.correlation_backward(f1, warped, go, *P)], 20) * 1e6
        k_tile = _lib.last_kernel(1)
    finally:
        _lib.set_option("corr_bwd_variant", 0)
    ratio = region_over_tile(fl, 16 if B * H * W > 64 * 128 * 4 else 8)
    g2_bytes = 4 * B * C * H * W
    step_bytes = (4 * C + 81) * B * H * W * 4 + (3 * C + 4) * B * H * W * 4
    rate = step_bytes / (t_cb + t_wb) / 1e6        # TB/s these two launches move their algorithmic bytes at
    saved = 2 * g2_bytes / rate / 1e6 + NODE
    extra = 0.5 * (t_tile - t_cb) + 0.5 * t_tile * (ratio - 1.0)
    print("level %d (%dx%dx%d, %d pairs): corr_bwd %.1f us (%s) + warp_bwd %.1f us = %.1f; back to back in one graph %.1f us"
          % (lvl, C, H, W, B, t_cb, k_cb, t_wb, t_cb + t_wb, t_pair))
    print("   g2 round trip: 2 x %.1f MB at the %.2f TB/s of these launches = %.1f us, + one node %.2f us -> fusion can save <= %.1f us"
          % (g2_bytes / 1e6, rate, 2 * g2_bytes / rate / 1e6, NODE, saved))
    print("   second-gradient half as a TILE kernel: %s %.1f us whole launch vs %.1f -> +%.1f us for its half; scan region / tile = %.2f "
          "-> +%.1f us of re-computed overlap" % (k_tile, t_tile, t_cb, 0.5 * (t_tile - t_cb), ratio, 0.5 * t_tile * (ratio - 1.0)))
    print("   fused >= today %.1f - saved %.1f + extra %.1f = %.1f us  (%+.1f us per direction)"
          % (t_pair, saved, extra, t_pair - saved + extra, extra - saved))
