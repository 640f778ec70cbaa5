"""GPU parity tests for the fused flow_warp HIP kernels (through the C ABI)."""
import numpy as np
import pytest
import torch

import cerberusnet_amd as ca
from cerberusnet_amd.synth import hash_uniform, W32_PYRAMID_1024x512
from conftest import rel_err
import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def hip_warp_with_grads(img, flo, go, pad):
    i = dev(img).requires_grad_(True)
    f = dev(flo).requires_grad_(True)
    out = ca.flow_warp(i, f, pad=pad)
    gi, gf = torch.autograd.grad(out, (i, f), dev(go))
    return out.detach().cpu().numpy(), gi.cpu().numpy(), gf.cpu().numpy()


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("pad", ["border", "zeros"])
def test_golden_vectors(golden, tag, pad):
    g = golden("warp_" + tag)
    out, gi, gf = hip_warp_with_grads(g["image"], g["flow"], g["gout"], pad)
    assert rel_err(out, g["out_" + pad]) < TOL
    assert rel_err(gi, g["gimage_" + pad]) < TOL
    assert rel_err(gf, g["gflow_" + pad]) < TOL
    near = ca.flow_warp(dev(g["image"]), dev(g["flow"]), pad=pad, mode="nearest")
    assert np.array_equal(near.cpu().numpy(), g["nearest_" + pad])


def test_q2_zero_flow_is_not_identity(golden):
    g = golden("warp_q2")
    out = ca.flow_warp(dev(g["image"]), torch.zeros(1, 2, 4, 6, device=DEV))
    assert rel_err(out.cpu().numpy(), g["out"]) < 1e-6
    assert abs(float(out[0, 0, 0, 1]) - 0.7) < 1e-5


SHAPES = [(1, 1, 2, 2), (2, 3, 7, 9), (1, 16, 12, 20), (2, 5, 33, 65), (1, 130, 9, 17),
          (3, 32, 64, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("pad", ["border", "zeros"])
def test_against_torch_cpu_oracle(shape, pad):
    B, C, H, W = shape
    img = hash_uniform(shape, 1)
    flo = hash_uniform((B, 2, H, W), 2, -6.0, 6.0)
    go = hash_uniform(shape, 3)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                               torch.from_numpy(go), pad)
    out, gi, gf = hip_warp_with_grads(img, flo, go, pad)
    assert rel_err(out, ref.numpy()) < TOL
    assert rel_err(gi, rgi.numpy()) < TOL
    assert rel_err(gf, rgf.numpy()) < TOL


@pytest.mark.parametrize("lvl", [1, 2, 3])
def test_fullsize_feature_warps(lvl):
    """Config-3 feature-warp shapes (levels 1-3) at the benched batch of 4: EVERY item's output,
    grad_image (the fixed-point tile path) and grad_flow against the torch-CPU oracle
    (UnFlowLoss.py:83-94 + autograd), plus batch-independence bit for bit; flows in [-6, 6) px so
    borders are hit."""
    C, H, W = W32_PYRAMID_1024x512[lvl]
    img = hash_uniform((4, C, H, W), 4)
    flo = hash_uniform((4, 2, H, W), 5, -6.0, 6.0)
    go = hash_uniform((4, C, H, W), 6)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                               torch.from_numpy(go), "border")
    out, gi, gf = hip_warp_with_grads(img, flo, go, "border")
    for b in range(4):
        assert rel_err(out[b], ref[b].numpy()) < TOL, b
        assert rel_err(gi[b], rgi[b].numpy()) < TOL, b
        assert rel_err(gf[b], rgf[b].numpy()) < TOL, b
    o1, i1, f1 = hip_warp_with_grads(img[3:], flo[3:], go[3:], "border")
    assert np.array_equal(o1[0], out[3]) and np.array_equal(f1[0], gf[3])
    # grad_image: one image alone takes 8-row tiles at the finest level, four take 16-row tiles; the
    # fixed-point scale is chosen per (tile, 8-channel group), so the two runs round the same integer
    # sums at scales up to 2^-29 of the group maximum apart: equal to well below one fp32 ulp of the
    # maximum, not bit for bit
    assert rel_err(i1[0], gi[3]) < 1e-7


@pytest.mark.parametrize("lvl", [1, 2, 3])
def test_fullsize_feature_warps_smooth_flow_as_benched(lvl):
    """The same shapes with the flow field bench.py feeds (a coarse field upsampled x8 + a
    +-0.25 px residual: what PWCNetHead produces): the tile workgroups of the backward then see
    shifted, not scattered, source regions -- the branch the benchmark times.  All 4 items."""
    import torch.nn.functional as F
    C, H, W = W32_PYRAMID_1024x512[lvl]
    img = hash_uniform((4, C, H, W), 14)
    coarse = torch.from_numpy(hash_uniform((4, 2, max(2, H // 8), max(2, W // 8)), 15, -6.0, 6.0))
    flo = F.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True)
    flo = (flo + torch.from_numpy(hash_uniform((4, 2, H, W), 115, -0.25, 0.25))).contiguous().numpy()
    go = hash_uniform((4, C, H, W), 16)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                               torch.from_numpy(go), "border")
    out, gi, gf = hip_warp_with_grads(img, flo, go, "border")
    for b in range(4):
        assert rel_err(out[b], ref[b].numpy()) < TOL, b
        assert rel_err(gi[b], rgi[b].numpy()) < TOL, b
        assert rel_err(gf[b], rgf[b].numpy()) < TOL, b


def test_rgb_loss_warps_and_reflection_nearest_modes():
    img = hash_uniform((2, 3, 64, 128), 7)
    flo = hash_uniform((2, 2, 64, 128), 8, -9.0, 9.0)
    for pad in ("border", "zeros", "reflection"):
        for mode in ("bilinear", "nearest"):
            ref = oracle.flow_warp_ref(torch.from_numpy(img), torch.from_numpy(flo), pad, mode)
            out = ca.flow_warp(dev(img), dev(flo), pad=pad, mode=mode)
            assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL, (pad, mode)


def test_fp64_and_half_dtypes():
    img = hash_uniform((1, 8, 20, 30), 9).astype(np.float64)
    flo = hash_uniform((1, 2, 20, 30), 10, -5.0, 5.0).astype(np.float64)
    go = hash_uniform((1, 8, 20, 30), 11).astype(np.float64)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                               torch.from_numpy(go), "border")
    out, gi, gf = hip_warp_with_grads(img, flo, go, "border")
    assert rel_err(out, ref.numpy()) < 1e-12
    assert rel_err(gi, rgi.numpy()) < 1e-12
    assert rel_err(gf, rgf.numpy()) < 1e-11
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        i16 = torch.from_numpy(img).to(dt)
        f16 = torch.from_numpy(flo).to(dt)
        r = oracle.flow_warp_ref(i16.double(), f16.double(), "border")
        o = ca.flow_warp(i16.to(DEV), f16.to(DEV))
        assert o.dtype == dt
        assert rel_err(o.double().cpu().numpy(), r.numpy()) < tol
        ii = i16.to(DEV).requires_grad_(True)
        ff = f16.to(DEV).requires_grad_(True)
        ca.flow_warp(ii, ff).float().sum().backward()
        assert ii.grad is not None and ff.grad is not None and ii.grad.dtype == dt


def test_errors():
    img = dev(hash_uniform((1, 3, 8, 8), 1))
    with pytest.raises(RuntimeError, match="flow shape"):
        ca.flow_warp(img, torch.zeros(1, 2, 8, 9, device=DEV))
    with pytest.raises(ValueError):
        ca.flow_warp(img, torch.zeros(1, 2, 8, 8, device=DEV), pad="wrap")
    with pytest.raises(ValueError):
        ca.flow_warp(img, torch.zeros(1, 2, 8, 8, device=DEV), mode="bicubic")


@pytest.mark.parametrize("amp", [3.0, 15.9, 31.0, 60.0])
@pytest.mark.parametrize("pad", ["border", "zeros"])
def test_grad_image_tiles_at_any_flow_magnitude(amp, pad):
    """The owner-computes grad_image kernel at small and large tap extents (a tile's scan
    region follows the extents of the strips that reach it; there is no limit and no scatter
    fallback on this path), against the torch-CPU oracle."""
    shape = (2, 12, 80, 136)
    img = hash_uniform(shape, 21)
    flo = hash_uniform((2, 2, 80, 136), 22, -amp, amp)
    go = hash_uniform(shape, 23)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                               torch.from_numpy(go), pad)
    out, gi, gf = hip_warp_with_grads(img, flo, go, pad)
    assert rel_err(out, ref.numpy()) < TOL
    assert rel_err(gi, rgi.numpy()) < TOL
    assert rel_err(gf, rgf.numpy()) < TOL


def test_backward_without_workspace_uses_scatter_and_matches():
    """C ABI called directly with workspace = NULL (ATen-style global atomics)."""
    import ctypes
    from cerberusnet_amd import _lib
    shape = (1, 6, 20, 36)
    img, go = dev(hash_uniform(shape, 31)), dev(hash_uniform(shape, 33))
    flo = dev(hash_uniform((1, 2, 20, 36), 32, -5.0, 5.0))
    gi = torch.full_like(img, 7.0)
    gf = torch.full_like(flo, 7.0)
    rc = _lib.get().cerberus_flow_warp_backward(
        img.data_ptr(), flo.data_ptr(), go.data_ptr(), gi.data_ptr(), gf.data_ptr(), None, 0,
        None, 0, 1, 6, 20, 36, 1, 0, 0, 0,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    _, rgi, rgf = oracle.flow_warp_grads_ref(img.cpu(), flo.cpu(), go.cpu(), "border")
    assert rel_err(gi.cpu().numpy(), rgi.numpy()) < TOL
    assert rel_err(gf.cpu().numpy(), rgf.numpy()) < TOL
    # and only one of the two gradients
    # a workspace that is too small is not an error either: same scatter path
    gi2 = torch.full_like(img, 7.0)
    ws = torch.empty(2, dtype=torch.int64, device=DEV)
    rc = _lib.get().cerberus_flow_warp_backward(
        img.data_ptr(), flo.data_ptr(), go.data_ptr(), gi2.data_ptr(), None, None, 0,
        ws.data_ptr(), 16, 1, 6, 20, 36, 1, 0, 0, 0,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(gi2.cpu().numpy(), rgi.numpy()) < TOL
    # full workspace, no context: tiled path with the context derived from the flow
    need = _lib.get().cerberus_flow_warp_backward_workspace_bytes(1, 6, 20, 36)
    ws = torch.empty((need + 7) // 8, dtype=torch.int64, device=DEV)
    gi3, gf3 = torch.full_like(img, 7.0), torch.full_like(flo, 7.0)
    rc = _lib.get().cerberus_flow_warp_backward(
        img.data_ptr(), flo.data_ptr(), go.data_ptr(), gi3.data_ptr(), gf3.data_ptr(), None, 0,
        ws.data_ptr(), need, 1, 6, 20, 36, 1, 0, 0, 0,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(gi3.cpu().numpy(), rgi.numpy()) < TOL
    assert rel_err(gf3.cpu().numpy(), rgf.numpy()) < TOL
    # a context of the wrong size is rejected
    rc = _lib.get().cerberus_flow_warp_backward(
        img.data_ptr(), flo.data_ptr(), go.data_ptr(), gi3.data_ptr(), gf3.data_ptr(),
        ws.data_ptr(), 64, ws.data_ptr(), need, 1, 6, 20, 36, 1, 0, 0, 0,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == -1


@pytest.mark.parametrize("amp", [0.4, 6.0, 40.0])
@pytest.mark.parametrize("shape", [(2, 12, 80, 136), (1, 5, 17, 70), (3, 32, 32, 64)])
def test_forward_context_path_equals_the_contextless_path(amp, shape):
    """Training saves the sample positions in the forward (flow_warp_ctx) and the backward
    reuses them; the raw backward derives them from the flow.  Same kernels, same numbers:
    outputs and both gradients must agree bit for bit, at small and large flows."""
    B, C, H, W = shape
    img, go = dev(hash_uniform(shape, 51)), dev(hash_uniform(shape, 53))
    flo = dev(hash_uniform((B, 2, H, W), 52, -amp, amp))
    out0 = torch.ops.cerberus.flow_warp(img, flo, 1, 0)
    out1, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, 1, 0)
    assert torch.equal(out0, out1)
    gi0, gf0 = torch.ops.cerberus.flow_warp_backward(img, flo, go, 1, 0, True, True)
    gi1, gf1 = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, 1, 0, True, True)
    assert torch.equal(gf0, gf1)
    assert torch.equal(gi0, gi1)
    _, rgi, rgf = oracle.flow_warp_grads_ref(img.cpu(), flo.cpu(), go.cpu(), "border")
    assert rel_err(gi1.cpu().numpy(), rgi.numpy()) < TOL
    assert rel_err(gf1.cpu().numpy(), rgf.numpy()) < TOL
    with pytest.raises(RuntimeError, match="context"):
        torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx[:8], go, 1, 0, True, True)


@pytest.mark.parametrize("mag", [1e-20, 1.0, 3e18, 0.0])
def test_tiled_grad_image_fixed_point_is_scale_free_and_deterministic(mag):
    """The tiled kernel accumulates in 64-bit fixed point scaled by max|grad_out| (measured
    on the device): results must not depend on the magnitude of the gradient, must be exactly
    linear in a power-of-two rescale, and must be bit-reproducible run to run."""
    shape = (2, 9, 33, 70)
    img = hash_uniform(shape, 41)
    flo = hash_uniform((2, 2, 33, 70), 42, -7.0, 7.0)
    go = hash_uniform(shape, 43)
    _, rgi, _ = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                           torch.from_numpy(go), "border")
    i, f = dev(img), dev(flo)
    g = dev(go) * mag
    a, _ = torch.ops.cerberus.flow_warp_backward(i, f, g, 1, 0, True, False)
    b, _ = torch.ops.cerberus.flow_warp_backward(i, f, g, 1, 0, True, False)
    assert torch.equal(a, b)
    if mag == 0.0:
        assert float(a.abs().max()) == 0.0
    else:
        assert rel_err(a.double().cpu().numpy() / mag, rgi.numpy()) < TOL
    c, _ = torch.ops.cerberus.flow_warp_backward(i, f, g * 8.0, 1, 0, True, False)
    assert torch.equal(c, a * 8.0)


def test_backward_with_a_fast_object_on_a_large_map():
    """One fast object on an otherwise smooth flow: only the tiles it feeds scan wide."""
    shape = (1, 4, 300, 520)     # 2438 strips; W not a multiple of 64: strips straddle rows
    img, go = dev(hash_uniform(shape, 71)), dev(hash_uniform(shape, 73))
    flo = hash_uniform((1, 2, 300, 520), 72, -3.0, 3.0)
    flo[0, :, 100:140, 200:260] += 45.0          # one fast object: only nearby tiles scan wide
    flo = dev(flo)
    out, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, 1, 0)
    gi, gf = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, 1, 0, True, True)
    gi0, gf0 = torch.ops.cerberus.flow_warp_backward(img, flo, go, 1, 0, True, True)
    assert torch.equal(gi, gi0) and torch.equal(gf, gf0)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(img.cpu(), flo.cpu(), go.cpu(), "border")
    assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL
    assert rel_err(gi.cpu().numpy(), rgi.numpy()) < TOL
    assert rel_err(gf.cpu().numpy(), rgf.numpy()) < TOL


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
def test_lds_staged_gather_equals_the_direct_gather_bit_for_bit(dt):
    """The forward takes its taps from an LDS copy of the tile's source window (one coalesced
    pass over the window instead of a line-granular gather); same arithmetic in the same order,
    so every output bit and the whole context must equal the direct-gather kernel's -- for
    smooth flows (window fits), windows that only fit for a few channels at a time, diverged
    flows (direct-gather fallback inside the staged kernel), taps off every border, ragged
    tiles, non-finite flow values, and channel counts that are not a multiple of the
    workgroup's channel range."""
    from cerberusnet_amd import _lib
    cases = [((2, 19, 37, 132), 0.7), ((1, 8, 64, 128), 6.0), ((2, 5, 24, 64), 30.0),
             ((1, 33, 16, 32), 300.0), ((1, 3, 2, 4), 1.0), ((1, 9, 130, 260), 14.0)]
    for k, (shape, amp) in enumerate(cases):
        B, C, H, W = shape
        img = dev(hash_uniform(shape, 600 + k)).to(dt)
        flo = hash_uniform((B, 2, H, W), 700 + k, -amp, amp)
        flo[0, 0, H // 2, W // 2] = np.nan
        flo[0, 1, 0, W - 1] = np.inf
        flo[-1, 0, H - 1, 0] = -np.inf
        flo = dev(flo)
        for pad in (0, 1):
            for crange in (0, 4, 16, 32):
                _lib.set_option("warp_staged", crange)
                try:
                    out, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, pad, 0)
                    plain = torch.ops.cerberus.flow_warp(img, flo, pad, 0)
                    _lib.set_option("warp_staged", 2)
                    out0, ctx0 = torch.ops.cerberus.flow_warp_ctx(img, flo, pad, 0)
                finally:
                    _lib.set_option("warp_staged", 0)
                assert torch.equal(out.view(torch.int16 if dt != torch.float32 else torch.int32),
                                   out0.view(torch.int16 if dt != torch.float32 else torch.int32)), (shape, amp, pad, crange)
                assert torch.equal(plain.view(torch.int16 if dt != torch.float32 else torch.int32),
                                   out0.view(torch.int16 if dt != torch.float32 else torch.int32))
                assert torch.equal(ctx, ctx0), (shape, amp, pad, crange)
            # backward: the staged grad_flow role against the four-wave strip role
            go = dev(hash_uniform(shape, 800 + k)).to(dt)
            _lib.set_option("warp_staged", 2)
            try:
                gi0, gf0 = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx0, go, pad, 0, True, True)
            finally:
                _lib.set_option("warp_staged", 0)
            bits = torch.int32 if gf0.dtype == torch.float32 else torch.int16
            for force in (0, 8):          # auto (staged on large maps) / staged at any size
                _lib.set_option("warp_staged", force)
                try:
                    gi, gf = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx0, go, pad, 0, True, True)
                finally:
                    _lib.set_option("warp_staged", 0)
                assert torch.equal(gf.view(bits), gf0.view(bits)), (shape, amp, pad, force)
                assert torch.equal(gi.view(torch.int16 if dt != torch.float32 else torch.int32),
                                   gi0.view(torch.int16 if dt != torch.float32 else torch.int32))


def test_seeded_random_shape_sweep_against_the_oracle():
    """24 seeded random shapes / flow magnitudes / paddings: forward, grad_image and grad_flow of
    the context path against the torch-CPU oracle (ragged tiles, single rows, odd channel
    counts, flows from sub-pixel to larger than the image)."""
    import numpy as np
    rng = np.random.RandomState(777)
    for trial in range(24):
        B = int(rng.randint(1, 3))
        C = int(rng.choice([1, 3, 4, 7, 16, 33]))
        H = int(rng.randint(2, 50))
        W = int(rng.randint(2, 150))
        amp = float(rng.choice([0.3, 2.0, 9.0, 40.0, 300.0]))
        pad = ["border", "zeros"][int(rng.randint(0, 2))]
        shape = (B, C, H, W)
        img, go = hash_uniform(shape, 3000 + trial), hash_uniform(shape, 4000 + trial)
        flo = hash_uniform((B, 2, H, W), 5000 + trial, -amp, amp)
        ref, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                                   torch.from_numpy(go), pad)
        out, gi, gf = hip_warp_with_grads(img, flo, go, pad)
        assert rel_err(out, ref.numpy()) < TOL, (shape, amp, pad)
        assert rel_err(gi, rgi.numpy()) < TOL, (shape, amp, pad)
        assert rel_err(gf, rgf.numpy()) < TOL, (shape, amp, pad)


def test_nonfinite_grad_out_reaches_the_same_elements_as_the_oracle():
    """A NaN / Inf in grad_out must reach grad_image exactly where ATen's scatter puts it (the
    four taps of that source, that channel) and nowhere else; the fixed-point tiles cannot
    represent them, so such a (tile, channel group) accumulates in float (the first
    tiled kernel laundered NaN into 0 and Inf into a finite value)."""
    shape = (2, 9, 40, 140)
    img = hash_uniform(shape, 81)
    flo = hash_uniform((2, 2, 40, 140), 82, -5.0, 5.0)
    go = hash_uniform(shape, 83)
    go[0, 1, 3, 3] = np.nan
    go[0, 5, 20, 77] = np.inf
    go[1, 8, 39, 139] = -np.inf
    go[1, 0, 17, 64] = np.nan
    for pad in ("border", "zeros"):
        _, rgi, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo),
                                                 torch.from_numpy(go), pad)
        _, gi, gf = hip_warp_with_grads(img, flo, go, pad)
        rgi, rgf = rgi.numpy(), rgf.numpy()
        assert np.array_equal(np.isnan(gi), np.isnan(rgi))
        assert np.array_equal(np.isposinf(gi), np.isposinf(rgi))
        assert np.array_equal(np.isneginf(gi), np.isneginf(rgi))
        assert np.isnan(rgi).sum() >= 4 and np.isinf(rgi).sum() >= 2
        ok = np.isfinite(rgi)
        assert rel_err(np.where(ok, gi, 0), np.where(ok, rgi, 0)) < TOL
        assert np.array_equal(np.isfinite(gf), np.isfinite(rgf))
        okf = np.isfinite(rgf)
        assert rel_err(np.where(okf, gf, 0), np.where(okf, rgf, 0)) < TOL


@pytest.mark.parametrize("dt,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("shape", [(1, 8, 20, 30), (1, 32, 256, 512), (4, 32, 256, 512),
                                   (4, 64, 128, 256), (4, 128, 64, 128)])
def test_half_backward_is_tiled_deterministic_and_matches_the_oracle(dt, tol, shape):
    """Config 5 (fp16 mixed precision, 2048x1024: the feature warps are 128x64x128, 64x128x256 and
    32x256x512, benched at 4 pairs): the
    16-bit backward takes the owner-computes tiles (fp32 fixed-point accumulation, 16-bit I/O),
    not a 16-bit CAS loop on global memory.  Against the torch-CPU oracle run in fp32 on the same
    16-bit inputs; bit-reproducible run to run; equal (to 16-bit rounding) to the scatter path."""
    from cerberusnet_amd import _lib
    B, C, H, W = shape
    img = torch.from_numpy(hash_uniform(shape, 91)).to(dt)
    flo = torch.from_numpy(hash_uniform((B, 2, H, W), 92, -6.0, 6.0)).to(dt)
    go = torch.from_numpy(hash_uniform(shape, 93)).to(dt)
    _, rgi, rgf = oracle.flow_warp_grads_ref(img.float(), flo.float(), go.float(), "border")
    i, f, g = img.to(DEV), flo.to(DEV), go.to(DEV)
    out, ctx = torch.ops.cerberus.flow_warp_ctx(i, f, 1, 0)
    gi, gf = torch.ops.cerberus.flow_warp_backward_ctx(i, f, ctx, g, 1, 0, True, True)
    gi2, gf2 = torch.ops.cerberus.flow_warp_backward_ctx(i, f, ctx, g, 1, 0, True, True)
    assert gi.dtype == dt and gf.dtype == dt
    assert torch.equal(gi, gi2) and torch.equal(gf, gf2)
    assert rel_err(gi.float().cpu().numpy(), rgi.numpy()) < tol
    assert rel_err(gf.float().cpu().numpy(), rgf.numpy()) < tol
    if H * W <= 4096:
        _lib.set_option("warp_force_scatter", 1)
        try:
            gis, gfs = torch.ops.cerberus.flow_warp_backward_ctx(i, f, ctx, g, 1, 0, True, True)
        finally:
            _lib.set_option("warp_force_scatter", 0)
        assert rel_err(gis.float().cpu().numpy(), rgi.numpy()) < 4 * tol
        assert torch.equal(gfs, gf)


def test_fp32_flow_beside_a_half_image_keeps_its_precision():
    """a half image with an fp32 flow is sampled at full flow precision in the
    reference (grid_sample is on autocast's fp32 list); the flow must not be rounded to the
    image's dtype.  At |flow| ~ 200 px an fp16 flow is 0.06-0.12 px off."""
    B, C, H, W = 1, 8, 64, 512
    img16 = torch.from_numpy(hash_uniform((B, C, H, W), 95)).to(torch.float16)
    flo = torch.from_numpy(hash_uniform((B, 2, H, W), 96, -1.0, 1.0))
    flo[:, 0] += 200.37
    go16 = torch.from_numpy(hash_uniform((B, C, H, W), 97)).to(torch.float16)
    ref, rgi, rgf = oracle.flow_warp_grads_ref(img16.float(), flo, go16.float(), "border")
    i = img16.to(DEV).requires_grad_(True)
    f = flo.to(DEV).requires_grad_(True)
    out = ca.flow_warp(i, f)
    assert out.dtype == torch.float16
    out.backward(go16.to(DEV))
    assert f.grad.dtype == torch.float32 and i.grad.dtype == torch.float16
    assert rel_err(out.detach().float().cpu().numpy(), ref.numpy()) < 1e-3      # fp16 output rounding
    assert rel_err(i.grad.float().cpu().numpy(), rgi.numpy()) < 2e-3
    assert rel_err(f.grad.cpu().numpy(), rgf.numpy()) < 1e-4
    # the same call with the flow rounded to fp16 is visibly worse: the test would see it
    bad = ca.flow_warp(img16.to(DEV), flo.to(DEV).half())
    assert rel_err(bad.float().cpu().numpy(), ref.numpy()) > 5e-3


@pytest.mark.parametrize("th", [8, 16])
@pytest.mark.parametrize("ranges", [1, 2, 8])
def test_tile_shapes_and_channel_ranges_agree_bit_for_bit(th, ranges):
    """Every (tile height, channel ranges per tile) decomposition of the tiled backward adds
    the same integers: identical bits, uniform translation (the shifted-region case), a
    zooming flow (several sources per element) and noise."""
    from cerberusnet_amd import _lib
    B, C, H, W = 2, 20, 50, 200
    img, go = dev(hash_uniform((B, C, H, W), 61)), dev(hash_uniform((B, C, H, W), 63))
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    flows = {
        "shift": np.stack([np.full((H, W), 5.3, np.float32), np.full((H, W), -3.6, np.float32)]),
        "zoom": np.stack([-0.4 * (xs - W / 2), -0.4 * (ys - H / 2)]),
        "noise": hash_uniform((2, H, W), 62, -9.0, 9.0),
    }
    for name, fl in flows.items():
        flo = dev(np.broadcast_to(fl, (B, 2, H, W)).copy())
        _, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, 1, 0)
        base = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, 1, 0, True, True)
        _lib.set_option("warp_tile_h", th)
        _lib.set_option("warp_tile_ranges", ranges)
        try:
            got = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, 1, 0, True, True)
        finally:
            _lib.set_option("warp_tile_h", 0)
            _lib.set_option("warp_tile_ranges", 0)
        _, rgi, rgf = oracle.flow_warp_grads_ref(img.cpu(), flo.cpu(), go.cpu(), "border")
        assert rel_err(got[0].cpu().numpy(), rgi.numpy()) < TOL, name
        assert rel_err(got[1].cpu().numpy(), rgf.numpy()) < TOL, name
        assert torch.equal(got[1], base[1]), name
        # the fixed-point scale is per (tile, 4-channel group): channel ranges cut on group
        # boundaries see the same groups, so only the tile height may change the scale
        if th == 16:
            assert torch.equal(got[0], base[0]) or H * W * B <= 64 * 128 * 4, name


@pytest.mark.parametrize("factor", [2, 4])
@pytest.mark.parametrize("shape", [(2, 2, 8, 16), (1, 2, 1, 1), (3, 2, 17, 5), (1, 2, 64, 128), (2, 3, 1, 9)])
def test_flow_upsample_against_torch_interpolate(shape, factor):
    """flow_upsample(flow, k) == F.interpolate(flow * k, scale_factor=k, 'bilinear',
    align_corners=True) (pwcnet_sfd.py:176, :199-201) and its gradient (a deterministic gather
    here, ATen's atomic scatter there), against torch on the CPU; single rows / columns / pixels
    (scale 0 paths) included."""
    import torch.nn.functional as F
    x = torch.from_numpy(hash_uniform(shape, 181, -8.0, 8.0)).requires_grad_(True)
    ref = F.interpolate(x * factor, scale_factor=factor, mode="bilinear", align_corners=True)
    go = torch.from_numpy(hash_uniform(tuple(ref.shape), 182))
    (rg,) = torch.autograd.grad(ref, x, go)
    xd = x.detach().to(DEV).requires_grad_(True)
    out = torch.ops.cerberus.flow_upsample(xd, factor)
    (g,) = torch.autograd.grad(out, xd, go.to(DEV))
    (g2,) = torch.autograd.grad(torch.ops.cerberus.flow_upsample(xd, factor), xd, go.to(DEV))
    assert out.shape == ref.shape
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-6
    assert rel_err(g.cpu().numpy(), rg.numpy()) < 1e-6
    assert torch.equal(g, g2)


def test_flow_upsample_half_and_errors():
    import torch.nn.functional as F
    x = torch.from_numpy(hash_uniform((2, 2, 12, 20), 183, -4.0, 4.0))
    ref = F.interpolate(x * 2, scale_factor=2, mode="bilinear", align_corners=True)
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        out = torch.ops.cerberus.flow_upsample(x.to(DEV).to(dt), 2)
        assert out.dtype == dt
        assert rel_err(out.float().cpu().numpy(), F.interpolate(x.to(dt).float() * 2, scale_factor=2, mode="bilinear",
                                                              align_corners=True).numpy()) < tol
    with pytest.raises(RuntimeError, match="factor"):
        torch.ops.cerberus.flow_upsample(x.to(DEV), 0)
    with pytest.raises(RuntimeError, match="multiple"):
        torch.ops.cerberus.flow_upsample_backward(x.to(DEV), 8)
    with pytest.raises(RuntimeError, match="no CPU"):
        torch.ops.cerberus.flow_upsample(x, 2)


@pytest.mark.parametrize("size", [(16, 24), (20, 36), (7, 5), (64, 96), (32, 48), (1, 1), (100, 130)])
def test_area_resize_is_bit_identical_to_torch_cpu(size):
    """The photometric loss resizes the target images to every flow scale with
    F.interpolate(mode='area') (UnFlowLoss.py:279-280) = adaptive average pooling.  Same window
    bounds, same fp32 row-major sum, same two divisions as ATen's CPU kernel: identical bits for
    integer and fractional ratios, down- and up-sampling (windows up to 16 x 16 here)."""
    x = hash_uniform((2, 3, 64, 96), 901, -2.0, 2.0)
    ref = torch.nn.functional.interpolate(torch.from_numpy(x), size, mode="area")
    out = ca.area_resize(dev(x), size)
    assert out.shape == ref.shape
    if size == (1, 1):
        # a 6144-element window: ATen's CPU kernel sums it in vector lanes, not serially
        assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL
    else:
        assert torch.equal(out.cpu(), ref)


def test_area_resize_pyramid_half_errors():
    # the loss's own pyramid: (B,3,512,1024) targets to the four flow scales
    x = dev(hash_uniform((1, 3, 512, 1024), 902, -2.0, 2.0))
    for h, w in [(512, 1024), (128, 256), (64, 128), (32, 64), (16, 32)]:
        ref = torch.nn.functional.interpolate(x.cpu(), (h, w), mode="area")
        assert torch.equal(ca.area_resize(x, (h, w)).cpu(), ref)
    for dt, tol in [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)]:
        ref = torch.nn.functional.interpolate(x.cpu().to(dt).float(), (64, 128), mode="area")
        got = ca.area_resize(x.to(dt), (64, 128))
        assert got.dtype == dt
        assert float((got.float().cpu() - ref).abs().max()) < tol * 2.0
    with pytest.raises(RuntimeError, match="positive"):
        ca.area_resize(x, (0, 4))
    with pytest.raises(RuntimeError, match="4-D"):
        ca.area_resize(x[0], (4, 4))
    with pytest.raises(RuntimeError, match="float64"):
        ca.area_resize(x.double(), (4, 4))
    xr = x.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="not differentiable"):
        ca.area_resize(xr, (4, 4)).sum().backward()
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ca.area_resize(x.cpu(), (4, 4))


def test_a_misaligned_context_is_rejected_with_a_clear_message():
    """the warp kernels read the context header as int4 (ABI v4: CERB_EINVAL otherwise);
    a context re-viewed at an 8-byte offset must be refused by the binding, not at launch time."""
    shape = (1, 4, 12, 40)
    img = torch.from_numpy(hash_uniform(shape, 1)).to(DEV)
    flo = torch.from_numpy(hash_uniform((1, 2, 12, 40), 2, -2.0, 2.0)).to(DEV)
    go = torch.from_numpy(hash_uniform(shape, 3)).to(DEV)
    _, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, 1, 0)
    assert ctx.data_ptr() % 16 == 0
    shifted = torch.empty(ctx.numel() + 1, dtype=ctx.dtype, device=DEV)[1:]
    shifted.copy_(ctx)
    assert shifted.data_ptr() % 16 == 8
    with pytest.raises(RuntimeError, match="16-byte aligned"):
        torch.ops.cerberus.flow_warp_backward_ctx(img, flo, shifted, go, 1, 0, True, True)


@pytest.mark.parametrize("lvl", [1, 3])
def test_phase_shift_of_coresident_workgroups_changes_nothing_but_time(lvl):
    """Round 4: when the whole warp-backward launch is resident at once some of its workgroups start a few thousand
    cycles late (option warp_stagger, auto rule in launch_tiles).  Speed only: grad_image and grad_flow must be bit-identical
    with the shift off, on (auto) and with an arbitrary explicit pattern, at the two benched shapes the rule fires on."""
    from cerberusnet_amd import _lib
    C, H, W = W32_PYRAMID_1024x512[lvl]
    img, go = dev(hash_uniform((4, C, H, W), 301)), dev(hash_uniform((4, C, H, W), 303))
    flo = dev(hash_uniform((4, 2, H, W), 302, -6.0, 6.0))
    _, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, 1, 0)
    res = {}
    for tag, val in (("off", -1), ("auto", 0), ("explicit", 3 | (5 << 8) | (9 << 16))):
        _lib.set_option("warp_stagger", val)
        try:
            res[tag] = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, 1, 0, True, True)
        finally:
            _lib.set_option("warp_stagger", 0)
    for tag in ("auto", "explicit"):
        assert torch.equal(res[tag][0], res["off"][0]) and torch.equal(res[tag][1], res["off"][1]), tag


def test_fp32_forward_through_the_dma_window_kernel_is_bit_identical_too():
    """warp16.hip's kernel instantiated for fp32 (4-pixel cells, no widening; option warp_pair16 = 1): measured SLOWER
    than the staged kernel (profiles/r06_warp_fp32_dma_ab.txt: twice the LDS per channel, four channels per pass), so it
    is opt-in only -- but it must stay correct: same bits as the default, outputs and context."""
    from cerberusnet_amd import _lib
    for k, (shape, amp) in enumerate([((2, 19, 37, 132), 0.7), ((1, 8, 64, 128), 6.0), ((2, 5, 24, 64), 30.0), ((1, 32, 40, 256), 3.0)]):
        B, C, H, W = shape
        img = dev(hash_uniform(shape, 900 + k))
        flo = hash_uniform((B, 2, H, W), 910 + k, -amp, amp)
        flo[0, 0, H // 2, W // 2] = np.nan
        flo = dev(flo)
        for pad in (0, 1):
            out0, ctx0 = torch.ops.cerberus.flow_warp_ctx(img, flo, pad, 0)
            _lib.set_option("warp_pair16", 1)
            try:
                out1, ctx1 = torch.ops.cerberus.flow_warp_ctx(img, flo, pad, 0)
            finally:
                _lib.set_option("warp_pair16", 0)
            assert torch.equal(out0.view(torch.int32), out1.view(torch.int32)) and torch.equal(ctx0, ctx1), (shape, pad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_grad_flow_alone_takes_the_flow_role_of_the_tile_launch_bit_for_bit(dtype):
    """Round 6: grad_flow WITHOUT grad_image for more than 4 channels (a frozen feature extractor) used to fall to the
    per-pixel gather kernel -- slower than computing both (32 x 256 x 512 fp16: 68.8 us against 54.8).  It now takes the
    tile launch with no tile workgroups (its flow role through the LDS window: 23.7 us): the bits are those of the launch
    that computes both (the gather kernel sums its channel groups in another order: equal within rounding)."""
    from cerberusnet_amd import _lib
    for k, (shape, amp) in enumerate([((2, 32, 64, 128), 6.0), ((1, 19, 37, 132), 0.7), ((4, 128, 32, 64), 3.0), ((1, 8, 24, 64), 30.0),
                                      ((2, 64, 40, 72), 2.0)]):
        B, C, H, W = shape
        img, go = dev(hash_uniform(shape, 3000 + k)).to(dtype), dev(hash_uniform(shape, 3010 + k)).to(dtype)
        flo = dev(hash_uniform((B, 2, H, W), 3020 + k, -amp, amp)).to(dtype)
        for pad in (0, 1):
            _, ctx = torch.ops.cerberus.flow_warp_ctx(img, flo, pad, 0)
            both = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, pad, 0, True, True)
            alone = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, pad, 0, False, True)
            bits = torch.int32 if dtype == torch.float32 else torch.int16
            assert torch.equal(alone[1].view(bits), both[1].view(bits)), (shape, pad)
            _lib.set_option("warp_force_scatter", 1)
            try:
                gather = torch.ops.cerberus.flow_warp_backward_ctx(img, flo, ctx, go, pad, 0, False, True)
            finally:
                _lib.set_option("warp_force_scatter", 0)
            scale = float(gather[1].float().abs().max())
            tol = {torch.float32: 2e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
            assert float((alone[1].float() - gather[1].float()).abs().max()) <= tol * scale, (shape, pad)
