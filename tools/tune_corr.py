#!/usr/bin/env python3
"""GPU-side tuning sweep: time every tuned forward variant / backward channel
slice on the config-3 pyramid levels (HIP events, interleaved rounds in one
process).  Output: one line per (level, variant)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes

P = (4, 1, 4, 1, 1, 1)


def timeit(fn, reps, rounds):
    """`reps` back-to-back launches captured in one hipGraph (no host launch gaps),
    replayed `rounds` times between HIP events; returns (median, min) us per launch."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep = [fn() for _ in range(reps)]
    graph.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) * 1e3 / reps)
    del keep, graph
    return float(np.median(best)), float(np.min(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    dev = "cuda:0"
    ops = torch.ops.cerberus
    rows = []
    for lvl, (C, H, W) in enumerate(pyramid_shapes()):
        B = args.pairs
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).to(dev)
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).to(dev)
        go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).to(dev)
        fb = (2 * C + 81) * B * H * W * 4
        bb = (4 * C + 81) * B * H * W * 4
        for variant in range(0, 9):
            _lib.set_option("corr_fwd_variant", variant)
            ops.correlation(x1, x2, *P)
            name = _lib.last_kernel(0)
            med, mn = timeit(lambda: ops.correlation(x1, x2, *P), args.reps, args.rounds)
            rows.append(dict(level=lvl, op="fwd", variant=variant, kernel=name,
                             us=round(med, 2), us_min=round(mn, 2), GBps=round(fb / med / 1e3, 1)))
            print(json.dumps(rows[-1]), flush=True)
        _lib.set_option("corr_fwd_variant", 0)
        for cs in (0, 4, 8, 16, 32, 64):
            if cs > C:
                continue
            _lib.set_option("corr_bwd_cslice", cs)
            ops.correlation_backward(x1, x2, go, *P)
            name = _lib.last_kernel(1)
            med, mn = timeit(lambda: ops.correlation_backward(x1, x2, go, *P), args.reps,
                             args.rounds)
            rows.append(dict(level=lvl, op="bwd", cslice=cs, kernel=name, us=round(med, 2),
                             us_min=round(mn, 2), GBps=round(bb / med / 1e3, 1)))
            print(json.dumps(rows[-1]), flush=True)
        _lib.set_option("corr_bwd_cslice", 0)


if __name__ == "__main__":
    main()
