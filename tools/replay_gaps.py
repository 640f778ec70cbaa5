#!/usr/bin/env python3
"""GPU: where do the ~0.5 ms go that a 20-step timed region loses against a 200-step one?  Events around every
replay of the captured step after a synchronize: GPU-side duration of each replay and the gaps between them."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload
dev = torch.device("cuda", 0)
wl = Workload(4, 1024, 512, dev, "smooth", False, 1, torch.float32)
streams = [torch.cuda.Stream()]
for _ in range(3):
    wl.step(streams)
torch.cuda.synchronize()
cap = torch.cuda.Stream(); cap.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cap):
    wl.step(streams)
torch.cuda.current_stream().wait_stream(cap)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    held = wl.step(streams)
for _ in range(5):
    g.replay()
for trial in range(3):
    torch.cuda.synchronize()
    N = 24
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(N):
        h0 = time.perf_counter()
        g.replay()
        host.append(time.perf_counter() - h0)
        ev[i + 1].record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
    print("trial %d: wall %.3f ms for %d replays (%.4f ms each), host issue of all %.3f ms (per replay call: first %.0f us, median %.0f us)"
          % (trial, wall * 1e3, N, wall * 1e3 / N, t_issue * 1e3, host[0] * 1e6, sorted(host)[N // 2] * 1e6))
    print("   event-to-event ms:", " ".join("%.3f" % x for x in d))
