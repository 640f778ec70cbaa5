#!/usr/bin/env python3
"""GPU: the three forms of the 16-bit matrix-core correlation forward, A/B in one process (us per launch, 4 pairs, hot; median and
min of 7 replays of a 20-launch graph after a warm-up pass): corr_fwd_variant 20 = register-staged (rounds 4-5), 26 = LDS-DMA +
transposing reads, standing still (C <= 32), 0 = the column walk.  With an -DCERB_ABLATE build (CERBERUS_HIP_LIB=...ablate.so):
the parts of the default form switched off (1 no copies, 4 no T tile / stores, 8 T writes only, 5 = 1 + 4)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
from tools.tune_corr import timeit, P
ops = torch.ops.cerberus
B = 4
abl = "ablate" in _lib.LIB_PATH
for dt in (torch.float16, torch.bfloat16):
    for (C, H, W) in ((32, 256, 512), (64, 128, 256), (32, 128, 256), (64, 64, 128)):
        x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).to(dt).cuda()
        x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).to(dt).cuda()
        nbytes = (2 * C + 81) * B * H * W * 2
        if abl:
            if dt != torch.float16:
                continue
            for m in (0, 1, 4, 8, 5):
                _lib.set_option("corr_debug_ablate", m)
                timeit(lambda: ops.correlation_leaky(x1, x2, *P, 0.1), 20, 3)
                med, mn = timeit(lambda: ops.correlation_leaky(x1, x2, *P, 0.1), 20, 7)
                print("(%d,%d,%d) ablate mask %d: %.1f us (min %.1f) %s" % (C, H, W, m, med, mn, _lib.last_kernel(0)), flush=True)
            _lib.set_option("corr_debug_ablate", 0)
            continue
        for v in ((20, 26, 0) if C <= 32 else (20, 0)):
            _lib.set_option("corr_fwd_variant", v)
            timeit(lambda: ops.correlation_leaky(x1, x2, *P, 0.1), 20, 3)
            med, mn = timeit(lambda: ops.correlation_leaky(x1, x2, *P, 0.1), 20, 7)
            print("%s (%d,%d,%d) variant %2d %-32s %.1f us (min %.1f)  = %.3f of 8 TB/s on %d MB" % (
                str(dt)[6:], C, H, W, v, _lib.last_kernel(0), med, mn, nbytes / med / 8e6, nbytes // 1000000), flush=True)
        _lib.set_option("corr_fwd_variant", 0)
