#!/usr/bin/env python3
"""GPU: SURVEY 8(f)-2 re-priced at the COARSE levels, where launches, not bytes, set the cost.

Training needs the warped tensor anyway (correlation_backward reads it), so a fused warp->correlation forward
still writes it: what fusion can save is ONE graph node (launch + ramp + tail) and the correlation's re-read of
`warped` from L2.  What it costs depends on who samples:
  (A) a row-band workgroup (R output rows x all channels) samples rows [y0-4, y0+R+4) once: (R+8)/R of the samples,
      but only B*H/R workgroups exist (the channel sum cannot be split over workgroups without a second pass);
  (B) the coarse recipe's (image, row, displacement row) workgroups sample their own x2 row: 9x the samples.
Measured here with this package's own kernels: the two launches today (separately and back to back in one graph),
the marginal cost of sampling k x the pixels (the warp at k x the batch), the correlation at 1 / R of the rows
(what ONE band of (A) costs when it has the chip to itself is bounded below by the kernel on that band alone)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit, P
from bench import Workload, _time_graph
ops = torch.ops.cerberus
for lvl in (1, 2):
    C, H, W = pyramid_shapes()[lvl]
    B = 4
    mk = lambda b, s: torch.from_numpy(hash_uniform((b, C, H, W), s)).cuda()
    f1, f2 = mk(B, 1), mk(B, 2)
    fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
    tw = _time_graph([lambda: ops.flow_warp_ctx(f2, fl, 1, 0)], 20) * 1e6
    warped, _ = ops.flow_warp_ctx(f2, fl, 1, 0)
    tc = _time_graph([lambda: ops.correlation(f1, warped, *P)], 20) * 1e6
    kern = _lib.last_kernel(0)
    def pair():
        w, c = ops.flow_warp_ctx(f2, fl, 1, 0)
        return ops.correlation(f1, w, *P), c
    tp = _time_graph([pair], 20) * 1e6 * 1.0
    print("level %d (%dx%dx%d, %d pairs): warp_fwd+ctx %.1f us, corr_fwd %.1f us (%s), the two back to back in one graph %.1f us"
          % (lvl, C, H, W, B, tw, tc, kern, tp))
    # (B): marginal cost of sampling 9x the pixels with the stand-alone warp (eval form: no context, no extra store)
    res = {}
    for k in (1, 3, 9):
        img, flo = mk(B * k, 4), Workload._flow(B * k, H, W, 5, "smooth", "cuda")
        res[k] = _time_graph([lambda: ops.flow_warp(img, flo, 1, 0)], 20) * 1e6
    per = (res[9] - res[1]) / 8.0
    print("   warp (no context) at 1x / 3x / 9x the pixels: %.1f / %.1f / %.1f us -> %.2f us per extra 1x" % (res[1], res[3], res[9], per))
    node = 2.15                                   # an empty graph node (tools/ubench/launch_floor.hip)
    reread = 4 * B * C * H * W / 10e6             # `warped` re-read from L2 at ~10 TB/s
    print("   (B) fused >= corr_fwd %.1f + warp %.1f + 8 x %.2f extra sampling - node %.2f - re-read %.2f = %.1f us vs %.1f us today"
          % (tc, res[1], per, node, reread, tc + res[1] + 8 * per - node - reread, tp))
    # (A): bands of R rows; sampling (R+8)/R; B*H/R workgroups.  A band cannot be faster than the correlation kernel run on
    # that band ALONE (one image, R rows: the whole chip serving 1/(B*H/R) of the work)
    for R in (8, 16):
        if H % R:
            continue
        x1b, x2b = mk(1, 6)[:, :, :R].contiguous(), mk(1, 7)[:, :, :R + 8].contiguous()
        flb = Workload._flow(1, R + 8, W, 8, "smooth", "cuda")
        tband_c = _time_graph([lambda: ops.correlation(x1b, x2b[:, :, :R].contiguous(), *P)], 20) * 1e6
        tband_w = _time_graph([lambda: ops.flow_warp(x2b, flb, 1, 0)], 20) * 1e6
        print("   (A) R = %2d: %d bands (workgroups) for 256 CUs, sampling x %.2f; ONE band alone on the chip: warp %.1f us + correlation %.1f us"
              " -> a band-per-workgroup kernel >= %.1f us (no other band can help it) vs %.1f us today"
              % (R, B * H // R, (R + 8) / R, tband_w, tband_c, max(tband_c, tband_w), tp))
