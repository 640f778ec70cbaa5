#!/usr/bin/env python3
"""GPU: the warp backward per level, hot / cold / gradOutput-hot-and-the-rest-cold inputs, with and without the grad_flow role
(DESIGN.md 3.4, round 3)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
dev = "cuda:0"
ops = torch.ops.cerberus
def smooth_flow(B, H, W, seed):
    g = torch.from_numpy(hash_uniform((B, 2, H // 8 + 1, W // 8 + 1), seed, -3.0, 3.0))
    f = torch.nn.functional.interpolate(g, size=(H, W), mode="bilinear", align_corners=True)
    return (f + torch.from_numpy(hash_uniform((B, 2, H, W), seed + 1, -0.25, 0.25))).to(dev)
def time_graph(fns, rounds=7):
    for f in fns: f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for f in fns: f()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [f() for f in fns]
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / len(fns))
    return float(np.median(ts))
B = 4
for (C, H, W) in ((128, 32, 64), (64, 64, 128), (32, 128, 256)):
    ncopy = max(8, int(300e6 / (2 * B * C * H * W * 4)) + 1)
    sets = []
    for k in range(ncopy):
        img = torch.from_numpy(hash_uniform((B, C, H, W), 10 + k)).to(dev)
        flo = smooth_flow(B, H, W, 100 + 2 * k)
        out, ctx = ops.flow_warp_ctx(img, flo, 0, 0)
        go = torch.from_numpy(hash_uniform((B, C, H, W), 50 + k)).to(dev)
        sets.append((img, flo, ctx, go))
    for need_flow in (True, False):
        hot = time_graph([lambda s=sets[0]: ops.flow_warp_backward_ctx(s[0], s[1], s[2], s[3], 0, 0, True, need_flow) for _ in range(20)])
        cold = time_graph([lambda s=s: ops.flow_warp_backward_ctx(s[0], s[1], s[2], s[3], 0, 0, True, need_flow) for s in sets])
        # gradOutput hot (one tensor), image / flow / context cold: what the step sees
        go0 = sets[0][3]
        mixed = time_graph([lambda s=s: ops.flow_warp_backward_ctx(s[0], s[1], s[2], go0, 0, 0, True, need_flow) for s in sets])
        print("C%d %dx%d need_flow=%s: hot %.2f  cold %.2f  gO-hot/rest-cold %.2f us (%d copies)" % (C, H, W, need_flow, hot, cold, mixed, ncopy), flush=True)
    fh = time_graph([lambda s=sets[0]: ops.flow_warp_ctx(s[0], s[1], 0, 0) for _ in range(20)])
    fc = time_graph([lambda s=s: ops.flow_warp_ctx(s[0], s[1], 0, 0) for s in sets])
    print("   forward: hot %.2f cold %.2f" % (fh, fc), flush=True)
    del sets
