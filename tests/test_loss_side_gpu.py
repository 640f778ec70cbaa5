"""GPU parity tests for the hot path AS THE TRAINING STEP RUNS IT (round 5): the photometric
loss's RGB warps (UnFlowLoss.py:282-283) on the lane-per-pixel kernels, the loss pyramid in one pass
(UnFlowLoss.py:279-280), and the correlation backward fed by the gradient of the estimator's concatenation
buffer (pwcnet_sfd.py:181-187 seen from autograd): cerberus_correlation_backward_ex."""
import ctypes

import numpy as np
import pytest
import torch

import cerberusnet_amd as ca
from cerberusnet_amd import _lib
from cerberusnet_amd.synth import hash_uniform
from conftest import l2_err, rel_err
import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5
DEV = "cuda:0"
CORR_P = (4, 1, 4, 1, 1, 1)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture
def fewc_off():
    """the channel-group kernels (what every warp took before round 5) for comparison"""
    def _set(off):
        _lib.set_option("warp_fewc", -1 if off else 0)
    yield _set
    _lib.set_option("warp_fewc", 0)


# ---- RGB warps ------------------------------------------------------------------------------------------------
def _flows(B, H, W, kind, seed):
    if kind == "noise":
        return hash_uniform((B, 2, H, W), seed, -6.0, 6.0)
    coarse = torch.from_numpy(hash_uniform((B, 2, max(2, H // 8), max(2, W // 8)), seed, -6.0, 6.0))
    up = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True)
    return (up + torch.from_numpy(hash_uniform((B, 2, H, W), seed + 1, -0.25, 0.25))).numpy()


@pytest.mark.parametrize("shape", [(1, 1, 2, 2), (2, 3, 7, 9), (2, 3, 33, 65), (1, 2, 16, 40), (3, 4, 64, 128),
                                   (2, 3, 128, 256), (2, 3, 512, 512), (1, 2, 601, 899), (1, 4, 1024, 520), (1, 1, 800, 700)])
@pytest.mark.parametrize("pad", ["border", "zeros"])
@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_few_channel_warp_against_the_oracle_and_the_channel_group_kernels(shape, pad, kind, fewc_off):
    """An image without gradient (the loss's targets): forward without context, backward = grad_flow alone.  Oracle:
    torch CPU flow_warp + autograd; and bit for bit what the channel-group kernels give.  (From 512 K pixels per call a
    lane owns four pixels: the last four shapes, ragged tails included.)"""
    B, C, H, W = shape
    img = hash_uniform(shape, 11, -2.0, 2.0)
    flo = _flows(B, H, W, kind, 12)
    go = hash_uniform(shape, 14)
    ref, _, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img), torch.from_numpy(flo), torch.from_numpy(go), pad)

    def run():
        f = dev(flo).requires_grad_(True)
        out = ca.flow_warp(dev(img), f, pad=pad)            # image.requires_grad False -> no context
        gf, = torch.autograd.grad(out, f, dev(go))
        return out.detach().cpu(), gf.cpu()

    fewc_off(False)
    out, gf = run()
    assert rel_err(out.numpy(), ref.numpy()) < TOL
    assert rel_err(gf.numpy(), rgf.numpy()) < TOL
    fewc_off(True)
    out_old, gf_old = run()
    assert torch.equal(out, out_old)
    assert torch.equal(gf, gf_old)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
def test_few_channel_warp_16_bit_storage(dtype, tol, fewc_off):
    shape = (2, 3, 48, 96)
    img = torch.from_numpy(hash_uniform(shape, 21, -2.0, 2.0)).to(dtype)
    flo = _flows(2, 48, 96, "smooth", 22)
    go = torch.from_numpy(hash_uniform(shape, 24)).to(dtype)
    ref, _, rgf = oracle.flow_warp_grads_ref(img.float(), torch.from_numpy(flo), go.float(), "border")
    outs = []
    for off in (False, True):
        fewc_off(off)
        f = dev(flo).requires_grad_(True)                  # an fp32 flow beside a 16-bit image keeps its precision
        out = ca.flow_warp(img.to(DEV), f, pad="border")
        gf, = torch.autograd.grad(out, f, go.to(DEV))
        assert out.dtype == dtype and gf.dtype == torch.float32
        assert rel_err(out.detach().float().cpu().numpy(), ref.numpy()) < tol
        assert rel_err(gf.cpu().numpy(), rgf.numpy()) < tol
        outs.append((out.detach().cpu(), gf.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_few_channel_warp_non_finite_flow_and_gradient(fewc_off):
    """NaN / Inf in the flow and in gradOutput reach exactly the elements they reach in the channel-group kernels."""
    shape = (1, 3, 24, 40)
    img = hash_uniform(shape, 31, -2.0, 2.0)
    flo = _flows(1, 24, 40, "noise", 32)
    go = hash_uniform(shape, 34)
    flo[0, 0, 3, 5] = np.nan
    flo[0, 1, 7, 9] = np.inf
    flo[0, 0, 11, 2] = -1e30
    go[0, 1, 5, 6] = np.inf
    go[0, 2, 9, 9] = np.nan
    res = []
    for off in (False, True):
        fewc_off(off)
        f = dev(flo).requires_grad_(True)
        out = ca.flow_warp(dev(img), f, pad="border")
        gf, = torch.autograd.grad(out, f, dev(go))
        res.append((out.detach().cpu().numpy(), gf.cpu().numpy()))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))
        ok = np.isfinite(a)
        assert np.array_equal(a[ok], b[ok])


def test_full_resolution_loss_warp_every_item():
    """The largest warp of the training step: (4, 3, 512, 1024) target images by the full-resolution flow."""
    shape = (4, 3, 512, 1024)
    img = hash_uniform(shape, 41, -2.0, 2.0)
    flo = _flows(4, 512, 1024, "smooth", 42)
    go = hash_uniform(shape, 44)
    f = dev(flo).requires_grad_(True)
    out = ca.flow_warp(dev(img), f, pad="border")
    gf, = torch.autograd.grad(out, f, dev(go))
    out, gf = out.detach().cpu().numpy(), gf.cpu().numpy()
    for b in range(4):
        ref, _, rgf = oracle.flow_warp_grads_ref(torch.from_numpy(img[b:b + 1]), torch.from_numpy(flo[b:b + 1]),
                                                 torch.from_numpy(go[b:b + 1]), "border")
        assert rel_err(out[b:b + 1], ref.numpy()) < TOL
        assert rel_err(gf[b:b + 1], rgf.numpy()) < TOL


# ---- the loss pyramid -----------------------------------------------------------------------------------------
def test_area_pyramid_is_bit_identical_to_torch_cpu_per_scale():
    x = hash_uniform((2, 3, 512, 1024), 902, -2.0, 2.0)
    sizes = [(512, 1024), (256, 512), (128, 256), (64, 128)]        # unFlowLoss's four scales under the host model's head
    outs = ca.area_pyramid(dev(x), sizes)
    assert outs[0].data_ptr() != 0 and torch.equal(outs[0].cpu(), torch.from_numpy(x))     # identity: the image itself
    for o, size in zip(outs, sizes):
        ref = torch.nn.functional.interpolate(torch.from_numpy(x), size, mode="area")
        assert o.shape == ref.shape and torch.equal(o.cpu(), ref)
    # five scales incl. the coarsest one (two launches), a ragged width, and sizes the one-pass kernel does not take
    x2 = hash_uniform((1, 2, 128, 1280), 903, -2.0, 2.0)
    for sizes in ([(32, 320), (16, 160), (8, 80), (4, 40), (2, 20)], [(64, 640), (32, 320)], [(100, 130), (32, 320)], [(128, 640)],
                  [(7, 5)], []):
        outs = ca.area_pyramid(dev(x2), sizes)
        assert len(outs) == len(sizes)
        for o, size in zip(outs, sizes):
            ref = torch.nn.functional.interpolate(torch.from_numpy(x2), size, mode="area")
            assert torch.equal(o.cpu(), ref), size


def test_area_pyramid_narrow_rows_take_the_one_output_per_thread_path():
    """Output rows that are not a whole number of 4-element groups (16-byte stores impossible) and rows that are, ratios
    2 / 4 / 8, several segments per row: the same bits as torch's CPU kernel either way."""
    for shape, sizes in (((1, 2, 32, 40), [(16, 20), (8, 10)]),        # 10-wide rows: one output per thread for every scale
                         ((1, 2, 32, 40), [(16, 20)]),                 # 20-wide rows: four per thread
                         ((2, 1, 64, 2304), [(32, 1152), (16, 576), (8, 288)]),   # three 1024 / 1024 / 256-column segments per row
                         ((1, 1, 8, 8), [(4, 4), (2, 2), (1, 1)])):
        x = hash_uniform(shape, 905, -2.0, 2.0)
        outs = ca.area_pyramid(dev(x), sizes)
        for o, size in zip(outs, sizes):
            ref = torch.nn.functional.interpolate(torch.from_numpy(x), size, mode="area")
            if size == (1, 1):
                assert rel_err(o.cpu().numpy(), ref.numpy()) < TOL     # ATen sums a 1 x 1 output with its vectorised mean
            else:
                assert torch.equal(o.cpu(), ref), (shape, size)


def test_area_pyramid_16_bit_and_errors():
    x = dev(hash_uniform((1, 3, 64, 128), 904, -2.0, 2.0))
    for dt, tol in [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)]:
        outs = ca.area_pyramid(x.to(dt), [(16, 32), (8, 16)])
        for o, size in zip(outs, [(16, 32), (8, 16)]):
            ref = torch.nn.functional.interpolate(x.cpu().to(dt).float(), size, mode="area")
            assert o.dtype == dt and float((o.float().cpu() - ref).abs().max()) < tol * 2.0
            assert torch.equal(o, ca.area_resize(x.to(dt), size))           # same bits as the per-scale kernel
    with pytest.raises(RuntimeError, match="positive"):
        torch.ops.cerberus.area_pyramid(x, [0, 4])
    with pytest.raises(RuntimeError, match="pairs"):
        torch.ops.cerberus.area_pyramid(x, [4])
    with pytest.raises(RuntimeError, match="float64"):
        torch.ops.cerberus.area_pyramid(x.double(), [4, 4])
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        torch.ops.cerberus.area_pyramid(x.cpu(), [4, 4])
    xr = x.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="not differentiable"):
        torch.ops.cerberus.area_pyramid(xr, [4, 4])[0].sum().backward()


# ---- correlation backward from the concatenation buffer's gradient ---------------------------------------------
def _abi_backward_ex(x1, x2, gbuf, fbuf, off, slope, ws=None):
    lib = _lib.get()
    B, C, H, W = x1.shape
    g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
    need = lib.cerberus_correlation_backward_ex_workspace_bytes(B, H, W, 4, 1, 4, 1, 1, 0)
    assert need == B * 81 * H * W * 4
    ws = torch.empty(need // 4, dtype=torch.float32, device=DEV) if ws is None else ws
    rc = lib.cerberus_correlation_backward_ex(
        x1.data_ptr(), x2.data_ptr(), gbuf[:, off].data_ptr(), gbuf.stride(0),
        fbuf[:, off].data_ptr() if fbuf is not None else None, fbuf.stride(0) if fbuf is not None else 0,
        ctypes.c_float(slope), ws.data_ptr() if ws.numel() else None, ws.numel() * 4, g1.data_ptr(), g2.data_ptr(),
        B, C, H, W, 4, 1, 4, 1, 1, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return rc, g1, g2


@pytest.mark.parametrize("shape,others,off", [((2, 8, 12, 20), 5, 0), ((1, 32, 16, 24), 34, 3), ((2, 16, 9, 11), 2, 0),
                                              ((4, 32, 128, 256), 34, 0)])
def test_backward_ex_equals_the_oracle_on_the_masked_gradient(shape, others, off):
    """C ABI: gradOutput and the stored LeakyReLU(corr) volume are channel slices of wider (batch-strided) buffers.
    Oracle: corr_backward_ref on where(stored > 0, g, g * slope)."""
    B, C, H, W = shape
    slope = 0.1
    x1, x2 = hash_uniform(shape, 51), hash_uniform(shape, 52)
    gbuf = hash_uniform((B, 81 + others, H, W), 53)
    fbuf = hash_uniform((B, 81 + others, H, W), 54)
    fbuf[0, off + 3, 1, 2] = np.nan                     # a NaN stored value takes the slope branch, as torch.where does
    fbuf[B - 1, off + 80, H - 1, W - 1] = 0.0
    g = gbuf[:, off:off + 81]
    masked = np.where(fbuf[:, off:off + 81] > 0, g, g * np.float32(slope)).astype(np.float32)
    rc, g1, g2 = _abi_backward_ex(dev(x1), dev(x2), dev(gbuf), dev(fbuf), off, slope)
    assert rc == 0
    if B * H * W <= 2 * 16 * 24 * 2:
        r1, r2 = oracle.corr_backward_ref(x1, x2, masked, 4, 1, 4, 1, 1)
    else:   # full size: the plain backward on the masked gradient (itself pinned against the oracle at this size)
        r = torch.ops.cerberus.correlation_backward(dev(x1), dev(x2), dev(masked), *CORR_P)
        r1, r2 = r[0].cpu().numpy(), r[1].cpu().numpy()
    assert rel_err(g1.cpu().numpy(), r1) < TOL
    assert rel_err(g2.cpu().numpy(), r2) < TOL
    # strided but unmasked; and dense + unmasked = the plain entry point (no workspace needed)
    rc, g1, g2 = _abi_backward_ex(dev(x1), dev(x2), dev(gbuf), None, off, slope)
    assert rc == 0
    r = torch.ops.cerberus.correlation_backward(dev(x1), dev(x2), dev(np.ascontiguousarray(g)), *CORR_P)
    assert torch.equal(g1, r[0]) and torch.equal(g2, r[1])
    rc, g1, g2 = _abi_backward_ex(dev(x1), dev(x2), dev(np.ascontiguousarray(g)), None, 0, slope,
                                  ws=torch.empty(0, device=DEV))
    assert rc == 0 and torch.equal(g1, r[0]) and torch.equal(g2, r[1])


def test_backward_ex_rejects_bad_arguments():
    x = dev(hash_uniform((1, 4, 8, 8), 1))
    buf = dev(hash_uniform((1, 90, 8, 8), 2))
    rc, _, _ = _abi_backward_ex(x, x, buf, buf, 0, 0.1, ws=torch.empty(0, device=DEV))
    assert rc == -1                                      # masked / strided without a workspace
    lib = _lib.get()
    g = torch.empty_like(x)
    args = (x.data_ptr(), x.data_ptr(), buf.data_ptr(), 80 * 64, None, 0, ctypes.c_float(0.1), None, 0, g.data_ptr(),
            g.data_ptr(), 1, 4, 8, 8, 4, 1, 4, 1, 1, 0, None)
    assert lib.cerberus_correlation_backward_ex(*args) == -1            # batch stride smaller than an item
    bad_s1 = list(args); bad_s1[18] = 2
    assert lib.cerberus_correlation_backward_ex(*bad_s1) == -3


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
def test_concat_backward_equals_the_three_pass_formula(dtype, tol):
    """CostVolumeConcat.backward (now one library call) against the stock-op formula it replaced:
    where(buf > 0, g, g * slope) -> contiguous -> correlation_backward."""
    from cerberusnet_amd.correlation_package.correlation import cost_volume_concat
    shape = (2, 32, 32, 64)
    x1 = dev(hash_uniform(shape, 61)).to(dtype).requires_grad_(True)
    x2 = dev(hash_uniform(shape, 62)).to(dtype).requires_grad_(True)
    o1 = dev(hash_uniform((2, 32, 32, 64), 63)).to(dtype).requires_grad_(True)
    o2 = dev(hash_uniform((2, 2, 32, 64), 64)).to(dtype).requires_grad_(True)
    buf = cost_volume_concat(x1, x2, [o1, o2], CORR_P, 0.1)
    gbuf = dev(hash_uniform(tuple(buf.shape), 65)).to(dtype)
    g1, g2, go1, go2 = torch.autograd.grad(buf, (x1, x2, o1, o2), gbuf)
    g = gbuf[:, :81]
    gm = torch.where(buf.detach()[:, :81] > 0, g, g * 0.1)
    r1, r2 = torch.ops.cerberus.correlation_backward(x1.detach(), x2.detach(), gm, *CORR_P)
    assert rel_err(g1.float().cpu().numpy(), r1.float().cpu().numpy()) <= tol
    assert rel_err(g2.float().cpu().numpy(), r2.float().cpu().numpy()) <= tol
    assert torch.equal(go1, gbuf[:, 81:113]) and torch.equal(go2, gbuf[:, 113:115])
    # the raw leaky op's autograd goes the same way
    y = torch.ops.cerberus.correlation_leaky(x1, x2, *CORR_P, 0.1)
    h1, h2 = torch.autograd.grad(y, (x1, x2), g.contiguous())
    assert rel_err(h1.float().cpu().numpy(), r1.float().cpu().numpy()) <= tol
    assert rel_err(h2.float().cpu().numpy(), r2.float().cpu().numpy()) <= tol


def test_concat_backward_takes_an_expanded_gradient():
    """``correlation_leaky(...).sum(0)`` hands the backward ``grad.expand(...)`` -- dense planes, batch
    stride 0.  The library reads a batch stride of 0 as "dense" and would walk B items out of one item's storage; the
    binding must materialise such a gradient.  Against the same call on the materialised gradient, bit for bit."""
    shape = (3, 16, 12, 20)
    x1 = dev(hash_uniform(shape, 66)).requires_grad_(True)
    x2 = dev(hash_uniform(shape, 67)).requires_grad_(True)
    w = dev(hash_uniform((81, 12, 20), 68))
    y = torch.ops.cerberus.correlation_leaky(x1, x2, *CORR_P, 0.1)
    (y.sum(0) * w).sum().backward()                              # the gradient arrives as w.expand(3, 81, 12, 20)
    g1, g2 = x1.grad.clone(), x2.grad.clone()
    x1.grad = x2.grad = None
    y = torch.ops.cerberus.correlation_leaky(x1, x2, *CORR_P, 0.1)
    y.backward(w.expand(3, 81, 12, 20).contiguous())
    assert torch.equal(g1, x1.grad) and torch.equal(g2, x2.grad)
    # the raw op with an expanded gradient AND an expanded stored volume
    fwd = y.detach()[:1].expand(3, 81, 12, 20)
    r = torch.ops.cerberus.correlation_backward_leaky(x1.detach(), x2.detach(), w.expand(3, 81, 12, 20), fwd, 0, *CORR_P, 0.1)
    q = torch.ops.cerberus.correlation_backward_leaky(x1.detach(), x2.detach(), w.expand(3, 81, 12, 20).contiguous(),
                                                      fwd.contiguous(), 0, *CORR_P, 0.1)
    assert torch.equal(r[0], q[0]) and torch.equal(r[1], q[1])


def test_area_pyramid_identity_scale_is_a_copy_not_an_alias():
    """a custom op must not return its input; F.interpolate returns a copy as well."""
    x = dev(hash_uniform((1, 3, 16, 32), 69))
    outs = torch.ops.cerberus.area_pyramid(x, [16, 32, 8, 16])
    assert torch.equal(outs[0], x) and outs[0].data_ptr() != x.data_ptr()
    outs[0].zero_()
    assert float(x.abs().sum()) > 0


def test_unflow_loss_uses_the_pyramid_and_matches_the_torch_backend():
    """unFlowLoss on the HIP ops (one pyramid pass per image, context-free RGB warps) against its own stock-op backend."""
    from cerberusnet_amd.loss_functions.UnFlowLoss import unFlowLoss
    B, H, W = 2, 128, 256
    img1, img2 = dev(hash_uniform((B, 3, H, W), 71, -2.0, 2.0)), dev(hash_uniform((B, 3, H, W), 72, -2.0, 2.0))
    sizes = [(H, W), (H // 4, W // 4), (H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    mk = lambda s: [dev(_flows(B, h, w, "smooth", s + i)).requires_grad_(True) for i, (h, w) in enumerate(sizes)]
    res = []
    for backend in ("hip", "torch"):
        fw, bw = mk(80), mk(90)
        loss = unFlowLoss(backend=backend)({"flow": fw, "flow_b": bw}, {"l_img": img1, "l_seq": img2})
        grads = torch.autograd.grad(loss, fw[:4] + bw[:4])
        res.append((float(loss.detach()), [g.cpu().numpy() for g in grads]))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    # gradients through |.|, SSIM's clamp and the bilinear taps' floor(): a last-bit difference of a sample position
    # (ATen's GPU grid_sample vs this package's) moves single pixels by percents of the maximum: the l2 norm is the yardstick
    for a, b in zip(res[0][1], res[1][1]):
        assert l2_err(a, b) < 5e-3
