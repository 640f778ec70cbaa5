#!/usr/bin/env python3
"""GPU, diagnostic build (CERB_EXTRA_HIPCC_FLAGS=-DCERB_STAMP python -m cerberusnet_amd.build --force):
where the waves of the coarse-level forward (corr_coarse.hip) spend their time.
usage: stamp_coarse.py [B C H W]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
P = (4, 1, 4, 1, 1, 1)
B, C, H, W = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (4, 256, 16, 32)
ops = torch.ops.cerberus
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
_lib.set_option("corr_fwd_variant", 15)
_lib.set_option("corr_bwd_cslice", int(os.environ.get("ALT", "0")))
for _ in range(5):
    ops.correlation(x1, x2, *P)
torch.cuda.synchronize()
print(_lib.last_kernel(0))
lib = _lib.get()
nwg = min(B * 8 * ((H * 9 + 7) // 8), 2048)
NW = 4 if "w4" in _lib.last_kernel(0) else 8
buf = np.zeros((2048, 8, 8), dtype=np.uint64)
assert lib.cerberus_debug_coarse_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
t = buf[:nwg, :NW].astype(np.int64)
t = t[t[:, 0, 0] > 0]
live = t[:, 0, 4] > 0          # workgroups whose displacement row is inside the image
t0 = t[:, :, 0].min()
us = (t - t0) / 100.0
print("workgroups %d (%d with loads); all times us after the first wave start" % (nwg, live.sum()))
names = ["wave start", "loads issued", "first load back", "all loads back", "FMAs done", "after barrier", "stores issued", "stores acked"]
for k, n in enumerate(names):
    v = us[live][:, :, k]
    print("  %-16s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (n, v.min(), np.median(v), np.percentile(v, 90), v.max()))
d = np.diff(us[live], axis=2)
print("phase lengths (median / p90 over waves):")
for k in range(7):
    print("  %-16s -> %-16s %6.2f / %6.2f" % (names[k], names[k + 1], np.median(d[:, :, k]), np.percentile(d[:, :, k], 90)))
print("kernel span %.2f us; wave lifetime median %.2f" % (us[:, :, 7].max(), np.median(us[live][:, :, 7] - us[live][:, :, 0])))
print("WG start histogram (us bins 0,.25,.5,1,1.5,2,3,4,6):", np.histogram(us[:, 0, 0], bins=[0, .25, .5, 1, 1.5, 2, 3, 4, 6, 99])[0].tolist())
