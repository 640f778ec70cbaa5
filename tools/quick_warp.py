#!/usr/bin/env python3
"""GPU: time warp forward / grad_flow-only / full backward on the pyramid levels with the
bench's smooth flow, and check the full backward against the scatter path (no workspace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import timeit
from bench import Workload
ops = torch.ops.cerberus
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
if len(sys.argv) > 2:
    _lib.set_option("warp_tile_h", int(sys.argv[2]))
if len(sys.argv) > 3:
    _lib.set_option("warp_tile_ranges", int(sys.argv[3]))
for name in os.environ.get("CERB_OPT", "").split(","):      # e.g. CERB_OPT=warp_stagger=-1
    if name:
        k, _, v = name.partition("=")
        _lib.set_option(k, int(v or 1))
for lvl, (C, H, W) in enumerate(pyramid_shapes()):
    if lvl == 0:
        continue
    B = 4
    img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
    go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
    fl = Workload._flow(B, H, W, 3, kind, "cuda")
    f, _ = timeit(lambda: ops.flow_warp(img, fl, 1, 0), 20, 5)
    fc, _ = timeit(lambda: ops.flow_warp_ctx(img, fl, 1, 0), 20, 5)
    _, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
    k1, _ = timeit(lambda: ops.flow_warp_backward(img, fl, go, 1, 0, False, True), 20, 5)
    noctx, _ = timeit(lambda: ops.flow_warp_backward(img, fl, go, 1, 0, True, True), 20, 5)
    full, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True), 20, 5)
    tiles, _ = timeit(lambda: ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, False), 20, 5)
    gi, gf = ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True)
    gi2, gf2 = ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, True)
    wb = (3 * C + 4) * B * H * W * 4
    print("L%d %s fwd %.1f us (+ctx %.1f)  grad_flow only %.1f us  bwd no-ctx %.1f us  bwd ctx %.1f us "
          "(%.2f TB/s; tiles only %.1f us)  reproducible=%s"
          % (lvl, kind, f, fc, k1, noctx, full, wb / full / 1e6, tiles,
             bool(torch.equal(gi, gi2) and torch.equal(gf, gf2))), flush=True)
