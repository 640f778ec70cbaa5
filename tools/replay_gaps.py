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

# ---- how long an idle gap does it take to trigger the transient? ----
for gap_ms in (0.0, 0.2, 1.0, 3.0, 10.0):
    for _ in range(40):
        g.replay()
    torch.cuda.synchronize()
    if gap_ms:
        t_end = time.perf_counter() + gap_ms * 1e-3
        while time.perf_counter() < t_end:
            pass
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record(); torch.cuda.synchronize()
    print("40 replays, synchronize, %.1f ms of idle, then 20 replays: %.4f ms per replay" % (gap_ms, a.elapsed_time(b) / 20))
# ---- and what kind of load keeps the chip in its working state: single-kernel graphs with a synchronize after each (the probe passes) ----
ops = torch.ops.cerberus
t3 = wl.dirs[0][3]
def one():
    return ops.correlation_backward(t3["f1"], t3["warped"], t3["gout"], 4, 1, 4, 1, 1, 1)
one(); torch.cuda.synchronize()
sg = torch.cuda.CUDAGraph()
cs = torch.cuda.Stream(); cs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cs):
    one()
torch.cuda.current_stream().wait_stream(cs)
with torch.cuda.graph(sg):
    keep = [one() for _ in range(20)]
time.sleep(0.5)
for _ in range(18):
    sg.replay(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    g.replay()
b.record(); torch.cuda.synchronize()
print("0.5 s idle, 18 x (replay of a 20-launch single-kernel graph + synchronize) = ~12 ms of bursty load, then 20 step replays: %.4f ms per replay" % (a.elapsed_time(b) / 20))
