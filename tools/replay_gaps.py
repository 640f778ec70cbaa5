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

# ---- is the ramp a property of the GRAPH EXEC (its first ~25 launches) or of the chip's state? ----
# the chip has just run ~80 replays of graph `g`; capture a SECOND graph of the same step now and time its first replays
def capture():
    c = torch.cuda.Stream(); c.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(c):
        wl.step(streams)
    torch.cuda.current_stream().wait_stream(c)
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        h = wl.step(streams)
    return gg, h
for _ in range(30):
    g.replay()
torch.cuda.synchronize()
g2, held2 = capture()
for _ in range(30):          # the chip busy with the OLD graph right up to the new graph's first replay
    g.replay()
N = 24
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    g2.replay(); ev[i + 1].record()
torch.cuda.synchronize()
print("a FRESH graph exec right after 30 replays of the old one (no idle gap): event-to-event ms:",
      " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(N)))
# and the old graph after an idle gap of 20 ms
time.sleep(0.02)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    g.replay(); ev[i + 1].record()
torch.cuda.synchronize()
print("the OLD graph after 20 ms of idle: event-to-event ms:", " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(N)))
time.sleep(1.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    g.replay(); ev[i + 1].record()
torch.cuda.synchronize()
print("the OLD graph after 1 s of idle: event-to-event ms:", " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(N)))
