#!/usr/bin/env python3
"""Distribution of one kernel's time over many graph replays (20 launches per replay)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from tools.tune_corr import P
ops = torch.ops.cerberus
C, H, W = pyramid_shapes()[3]
B = 4
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
fn = lambda: ops.correlation_backward(x1, x2, go, *P)
fn(); torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    keep = [fn() for _ in range(20)]
for gap_ms in (0, 0, 5, 50):
    ts = []
    for _ in range(40):
        if gap_ms:
            torch.cuda.synchronize(); 
            import time; time.sleep(gap_ms / 1e3)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / 20)
    ts = np.array(ts)
    print("idle gap %2d ms: min %.1f p25 %.1f median %.1f p75 %.1f max %.1f   first5 %s"
          % (gap_ms, ts.min(), np.percentile(ts, 25), np.median(ts), np.percentile(ts, 75), ts.max(),
             np.round(ts[:5], 1)), flush=True)
# back-to-back replays without sync in between
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    g.replay()
b.record(); torch.cuda.synchronize()
print("50 replays back to back: %.1f us per launch" % (a.elapsed_time(b) * 1e3 / 1000))
