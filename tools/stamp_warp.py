#!/usr/bin/env python3
"""GPU, diagnostic build (CERB_EXTRA_HIPCC_FLAGS=-DCERB_STAMP python -m cerberusnet_amd.build --force):
where a warp-backward TILE workgroup spends its time.  Prints the phase durations (in shader
cycles and us at 2.4 GHz-ish; read the SHARES) of the first tile workgroups of one launch."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from bench import Workload
ops = torch.ops.cerberus
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kind = sys.argv[2] if len(sys.argv) > 2 else "smooth"
dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[4] if len(sys.argv) > 4 else "f32"]
wh = (2048, 1024) if (len(sys.argv) > 5 and sys.argv[5] == "5") else (1024, 512)      # config 5 / config 3 frames
C, H, W = pyramid_shapes(wh[0], wh[1], 32)[lvl]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
go = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda().to(dt)
fl = Workload._flow(B, H, W, 3, kind, "cuda").to(dt)
print("stamps: level %d (%d, %d, %d) %s %s" % (lvl, C, H, W, kind, dt))
_, ctx = ops.flow_warp_ctx(img, fl, 1, 0)
want_flow = (sys.argv[3] if len(sys.argv) > 3 else "flow") == "flow"     # "tiles": grad_image alone
for _ in range(5):
    ops.flow_warp_backward_ctx(img, fl, ctx, go, 1, 0, True, want_flow)
torch.cuda.synchronize()
lib = _lib.get()
buf = np.zeros((64, 16), dtype=np.uint64)
rc = lib.cerberus_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
names = ["strip scan", "region scan+compact", "deal+zero+load g", "density", "max+scale", "adds",
         "barrier", "write-out", "rest (other groups)"]
d = np.diff(buf[:, :10].astype(np.int64), axis=1)
t0 = buf[:, 0].astype(np.int64)
print("start skew of the 64 workgroups (cycles): min %d max %d" % (0, int(t0.max() - t0.min())))
tot = (buf[:, 9].astype(np.int64) - buf[:, 0].astype(np.int64))
print("total per workgroup: median %d cycles, min %d, max %d" % (np.median(tot), tot.min(), tot.max()))
for k, nme in enumerate(names):
    print("  %-22s median %7d  (%4.1f %%)" % (nme, np.median(d[:, k]), 100.0 * np.median(d[:, k]) / np.median(tot)))
