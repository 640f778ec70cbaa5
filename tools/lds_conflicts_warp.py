#!/usr/bin/env python3
"""CPU model of the LDS-atomic traffic of warp_bwd_tile_kernel's add phase ("more than one conflict
cycle per LDS instruction -- try a padded or XOR-swizzled slot index").

For the bench's flow fields at a level, rebuild what a tile workgroup does: the sources whose north-west tap lies in
the padded tile, in region raster order (the compaction keeps it), dealt to the 256 threads as list entries
e = 256 j + t, and the four ds_add_u64 each source issues per channel pair (slots o, o + 1, o + PW, o + PW + 1 of a
(TH + 2) x (TW + 2) plane of 64-bit accumulators).  A 64-lane 64-bit DS access is served in four groups of 16
consecutive lanes (MI355X_MICROARCH.md, LDS table: ds_write_b64), 32 dword banks: a group takes as many LDS cycles as
its busiest bank has DISTINCT-OR-EQUAL accesses -- an atomic cannot broadcast, so two lanes adding to the SAME slot
serialise exactly like two lanes on different slots of one bank.  Reported per wave-instruction (ideal: 4 cycles):
  same   extra cycles that come from lanes adding to the same slot (the scatter itself: no layout removes them)
  bank   extra cycles from different slots on one bank (what padding / swizzling the slot index could remove)
for the plain layout (PW = 66) and for candidate layouts.
    python tools/lds_conflicts_warp.py [level] [smooth|noise] [width height]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cerberusnet_amd.synth import pyramid_shapes
from bench import Workload

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
wh = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 512)
C, H, W = pyramid_shapes(wh[0], wh[1], 32)[lvl]
TH, TW = 16, 64
flow = Workload._flow(1, H, W, 3 + 4 * lvl, kind, "cpu")[0].numpy().astype(np.float64)
xs, ys = np.meshgrid(np.arange(W), np.arange(H))
ix = ((2.0 * (xs + flow[0]) / (W - 1) - 1.0 + 1.0) * W - 1.0) / 2.0
iy = ((2.0 * (ys + flow[1]) / (H - 1) - 1.0 + 1.0) * H - 1.0) / 2.0
ix, iy = np.clip(ix, 0, W - 1), np.clip(iy, 0, H - 1)          # border padding
x0, y0 = np.floor(ix).astype(int), np.floor(iy).astype(int)


def tile_sources(tx0, ty0):
    lx, ly = x0 - tx0 + 1, y0 - ty0 + 1
    m = (lx >= 0) & (lx <= TW) & (ly >= 0) & (ly <= TH)
    rows, cols = np.nonzero(m)                                  # raster order of the region
    return lx[rows, cols], ly[rows, cols]


def cost(slots):
    """slots: (n,) slot index per lane of ONE ds_add_u64 (n <= 64, -1 = idle).  Returns (cycles, same, bank) extra split."""
    cyc = same = bank = 0
    for g in range(0, 64, 16):
        grp = [s for s in slots[g:g + 16] if s >= 0]
        if not grp:
            continue
        per_bank = {}
        for s in grp:
            per_bank.setdefault((2 * s) % 32, []).append(s)
        worst = max(per_bank.values(), key=len)
        c = len(worst)
        cyc += c
        # of the c - 1 extra cycles of the busiest bank: those that remain if different slots never shared a bank
        dup = max(np.unique(worst, return_counts=True)[1])
        same += dup - 1
        bank += (c - 1) - (dup - 1)
    return cyc, same, bank


layouts = {
    "plain PW=66": lambda lx, ly: ly * 66 + lx,
    "PW=67": lambda lx, ly: ly * 67 + lx,
    "PW=68": lambda lx, ly: ly * 68 + lx,
    "PW=72": lambda lx, ly: ly * 72 + lx,
    "PW=66, x ^= (y & 7)": lambda lx, ly: ly * 66 + (lx ^ (ly & 7)),
    "PW=80, x + 4 (y & 3)": lambda lx, ly: ly * 80 + lx + 4 * (ly & 3),
}
tot = {k: np.zeros(4) for k in layouts}
ntiles = 0
for ty0 in range(0, H, TH):
    for tx0 in range(0, W, TW):
        lx, ly = tile_sources(tx0, ty0)
        n = len(lx)
        if n == 0:
            continue
        ntiles += 1
        for name, f in layouts.items():
            for j in range(0, n, 256):
                for w0 in range(j, min(n, j + 256), 64):
                    lxx, lyy = lx[w0:w0 + 64], ly[w0:w0 + 64]
                    for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
                        # taps address the padded plane: (lx + dx, ly + dy); the swizzles apply to the tap's own coordinates
                        s = np.full(64, -1)
                        s[:len(lxx)] = f(lxx + dx, lyy + dy)
                        c, sm, bk = cost(s)
                        tot[name] += (1, c, sm, bk)
print("level %d (%d x %d x %d), %s flow, %d tiles of %d x %d, ds_add_u64 wave-instructions per channel pair and tile: %.1f"
      % (lvl, C, H, W, kind, ntiles, TH, TW, tot["plain PW=66"][0] / ntiles))
for name, t in tot.items():
    print("  %-24s cycles / instruction %.2f (ideal 4): +%.2f same-slot, +%.2f other-slot-same-bank" % (name, t[1] / t[0], t[2] / t[0], t[3] / t[0]))
