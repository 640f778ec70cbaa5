#!/usr/bin/env python3
"""GPU, -DCERB_STAMP build: phases of the lists_role workgroups of the warp forward (the first blocks of the launch)."""
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
C, H, W = pyramid_shapes()[lvl]
B = 4
img = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
fl = Workload._flow(B, H, W, 3, "smooth", "cuda")
for _ in range(5):
    ops.flow_warp_ctx(img, fl, 1, 0)
torch.cuda.synchronize()
buf = np.zeros((64, 16), dtype=np.uint64)
assert _lib.get().cerberus_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
d = np.diff(buf[:, :6].astype(np.int64), axis=1)
print("role workgroup total: median %d cycles (min %d max %d)" % (np.median(d.sum(1)), d.sum(1).min(), d.sum(1).max()))
for k, n in enumerate(["positions + box", "masks + ballots", "density adds + prefix", "scan + header", "records"]):
    print("  %-24s median %6d" % (n, np.median(d[:, k])))
