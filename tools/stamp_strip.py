#!/usr/bin/env python3
"""GPU, diagnostic build (CERB_EXTRA_HIPCC_FLAGS=-DCERB_STAMP python -m cerberusnet_amd.build --force):
where the waves of the strip backward (corr_strip.hip) spend their time, per step.
usage: stamp_strip.py [flags]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cerberusnet_amd  # noqa: F401
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
P = (4, 1, 4, 1, 1, 1)
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, C, H, W = 4, 32, 128, 256
ops = torch.ops.cerberus
x1 = torch.from_numpy(hash_uniform((B, C, H, W), 1)).cuda()
x2 = torch.from_numpy(hash_uniform((B, C, H, W), 2)).cuda()
go = torch.from_numpy(hash_uniform((B, 81, H, W), 3)).cuda()
_lib.set_option("corr_bwd_variant", 12)
_lib.set_option("corr_bwd_cslice", flags)
for _ in range(5):
    ops.correlation_backward(x1, x2, go, *P)
torch.cuda.synchronize()
print(_lib.last_kernel(1))
lib = _lib.get()
print('occupancy (workgroups per CU) by the API:', lib.cerberus_debug_strip_occupancy())
buf = np.zeros((64, 8, 64), dtype=np.uint64)
rc = lib.cerberus_debug_strip_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
t = buf.astype(np.int64)
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
print('per-WG start (cycles after the first):', sorted(((t[:, 0, 0] - t0) // 100).tolist()))
print('per-WG end:', sorted(((t[:, 0, 43] - t0) // 100).tolist()))
print("first stamp skew over 64 WGs x 8 waves: %d cycles" % (t[:, :, 0].max() - t0))
print("prologue: load wait %d, patch+barrier %d (median cycles)" % (np.median(t[:, :, 1] - t[:, :, 0]), np.median(t[:, :, 2] - t[:, :, 1])))
print("step | x wait+issue+windows | row blocks | vmcnt wait+patch | barrier wait | step total   (median over waves, cycles)")
tot = np.zeros(4)
for n in range(10):
    prev = t[:, :, 2] if n == 0 else t[:, :, 6 + 4 * (n - 1)]
    a = t[:, :, 3 + 4 * n] - prev
    b = t[:, :, 4 + 4 * n] - t[:, :, 3 + 4 * n]
    c = t[:, :, 5 + 4 * n] - t[:, :, 4 + 4 * n]
    d = t[:, :, 6 + 4 * n] - t[:, :, 5 + 4 * n]
    m = [np.median(a), np.median(b), np.median(c), np.median(d)]
    tot += m
    print("%4d | %7d | %7d | %7d | %7d | %7d    (max over waves: %d %d %d %d)" % (n, *m, sum(m), a.max(), b.max(), c.max(), d.max()))
print(" sum | %7d | %7d | %7d | %7d | %7d" % (*tot, tot.sum()))
real = (t[:, :, 45] - t[:, :, 44]) / 100.0
print('wave lifetime by s_memrealtime (100 MHz): median %.2f us, min %.2f, max %.2f; realtime start spread over all waves %.2f us, end spread %.2f us' % (np.median(real), real.min(), real.max(), (t[:, :, 44].max() - t[:, :, 44].min()) / 100.0, (t[:, :, 45].max() - t[:, :, 45].min()) / 100.0))
print('kernel span by s_memrealtime: %.2f us' % ((t[:, :, 45].max() - t[:, :, 44].min()) / 100.0))
life = t[:, :, 43] - t[:, :, 0]
print("wave lifetime start -> before stores: median %d, min %d, max %d cycles" % (np.median(life), life.min(), life.max()))
print("all 64 WGs: first start %d, last end %d -> span %d cycles" % (0, t[:, :, 43].max() - t0, t[:, :, 43].max() - t0))

life = np.zeros((512, 8, 4), dtype=np.uint64)
assert lib.cerberus_debug_strip_life(life.ctypes.data_as(ctypes.c_void_p), life.nbytes) == 0
L = life.astype(np.int64)
z = L[:, :, 0].min()
st, en, ack = (L[:, :, 0] - z) / 100.0, (L[:, :, 1] - z) / 100.0, (L[:, :, 2] - z) / 100.0
print("ALL 512 WGs x 8 waves, s_memrealtime relative to the first wave start (us):")
print("  wave start : min %.2f  median %.2f  p90 %.2f  max %.2f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("  loop end   : min %.2f  median %.2f  p90 %.2f  max %.2f" % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
print("  stores ack : min %.2f  median %.2f  p90 %.2f  max %.2f" % (ack.min(), np.median(ack), np.percentile(ack, 90), ack.max()))
print("  lifetime   : median %.2f  max %.2f ; store wait median %.2f max %.2f" % (np.median(en - st), (en - st).max(), np.median(ack - en), (ack - en).max()))
wg_start = st.min(axis=1)
print("  WG start histogram (us bins):", np.histogram(wg_start, bins=[0, 0.5, 1, 2, 3, 5, 8, 12, 20, 40])[0].tolist())
# group lifetimes by side / XCD / rotation offset / image border
bid = np.arange(512)
mapped = (bid % 8) * 64 + bid // 8
side = mapped & 1; yb = (mapped >> 1) % 64; img = (mapped >> 1) // 64
lt = (en - st).max(axis=1)      # slowest wave of the WG
e_wg = en.max(axis=1)
for name, key in (("side", side), ("xcd", bid % 8), ("rot", (10 - (2 * yb) % 10) % 10), ("img", img)):
    print("  by %-5s" % name, " ".join("%s: life %.1f end %.1f |" % (k, np.median(lt[key == k]), np.median(e_wg[key == k])) for k in sorted(set(key.tolist()))))
print("  border rows (yb<2 or yb>61): life %.1f ; interior %.1f" % (np.median(lt[(yb < 2) | (yb > 61)]), np.median(lt[(yb >= 2) & (yb <= 61)])))
order = np.argsort(e_wg)
print("  10 latest WGs: ", [(int(mapped[i]), int(side[i]), int(yb[i]), round(float(st.min(axis=1)[i]), 1), round(float(e_wg[i]), 1)) for i in order[-10:]])
print("  10 earliest WGs:", [(int(mapped[i]), int(side[i]), int(yb[i]), round(float(st.min(axis=1)[i]), 1), round(float(e_wg[i]), 1)) for i in order[:10]])
# per-step totals of the detailed stamps, by workgroup (the first 64 blockIdx: yb 0..3, both sides)
print("per-step total cycles by (side, yb): steps 0..9, then sum")
for b in range(64):
    m = (b % 8) * 64 + b // 8
    sd, y = m & 1, (m >> 1) % 64
    if b % 8 != 0: continue
    row = []
    for n in range(10):
        prev = t[b, :, 2] if n == 0 else t[b, :, 6 + 4 * (n - 1)]
        row.append(int(np.median(t[b, :, 6 + 4 * n] - prev)))
    parts = [int(np.median(sum(t[b, :, k + 4 * n] - (t[b, :, k - 1 + 4 * n] if k > 3 else (t[b, :, 2] if n == 0 else t[b, :, 6 + 4 * (n - 1)])) for n in range(10)))) for k in (3, 4, 5, 6)]
    print("  side %d yb %d rot %d:" % (sd, y, (10 - 2 * y % 10) % 10), row, sum(row), " phases(issue/blocks/wait/barrier):", parts)

iss = [int(np.median(t[:, :, 48 + n] - (t[:, :, 2] if n == 0 else t[:, :, 6 + 4 * (n - 1)]))) for n in range(10)]
wt = [int(np.median(t[:, :, 3 + 4 * n] - t[:, :, 48 + n])) for n in range(10)]
print("issue of loads + DMAs per step (cycles):", iss, " then x wait + windows:", wt)
