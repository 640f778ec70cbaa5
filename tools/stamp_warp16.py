#!/usr/bin/env python3
"""GPU, -DCERB_STAMP build (python -m cerberusnet_amd.build --variant stamp -DCERB_STAMP; CERBERUS_HIP_LIB=...): where a
workgroup of the 16-bit warp forward spends its time (cycle-counter stamps of thread 0; read the SHARES).
    python tools/stamp_warp16.py [f16|bf16] [5|3] [level]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform, pyramid_shapes
from bench import Workload
ops = torch.ops.cerberus
name = sys.argv[1] if len(sys.argv) > 1 else "f16"
dt = {"f16": torch.float16, "bf16": torch.bfloat16}[name]
wh = (2048, 1024) if (len(sys.argv) > 2 and sys.argv[2] == "5") else (1024, 512)
lvl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
C, H, W = pyramid_shapes(wh[0], wh[1], 32)[lvl]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda().to(dt)
fl = Workload._flow(B, H, W, 3, "smooth", "cuda").to(dt)
for _ in range(5):
    ops.flow_warp_ctx(img, fl, 1, 0)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((64, 16), dtype=np.uint64)
rc = lib.cerberus_debug_stamps16(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(buf.nbytes))
assert rc == 0, rc
names = ["flow + coordinates", "context", "box + barrier", "cell map", "staging issue (pass 1)", "barrier = staging wait",
         "channel loop (pass 1)", "second pass (all of it)"]
d = np.diff(buf[:, :9].astype(np.int64), axis=1)
tot = (buf[:, 8].astype(np.int64) - buf[:, 0].astype(np.int64))
t0 = buf[:, 0].astype(np.int64)
print("%s %dx%d L%d (%d,%d,%d): total per workgroup median %d cycles (min %d, max %d); start skew of the 64: %d" %
      (name, wh[0], wh[1], lvl, C, H, W, np.median(tot), tot.min(), tot.max(), int(t0.max() - t0.min())))
for k, nme in enumerate(names):
    print("  %-28s median %7d  (%4.1f %%)" % (nme, np.median(d[:, k]), 100.0 * np.median(d[:, k]) / np.median(tot)))
