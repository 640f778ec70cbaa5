"""CPU restatement of the reference's ``flow_warp`` (TEST INFRASTRUCTURE ONLY).

Reference: /root/reference/nnet_training/loss_functions/UnFlowLoss.py
  mesh_grid :11-20, norm_grid :22-32, flow_warp :83-94.

The sampling arithmetic itself is third-party: ``torch.nn.functional.grid_sample``
(ATen ``grid_sampler_2d``; the reference pinned "Pytorch 1.7" in README.md:27,
this image has torch 2.10 -- bilinear/nearest ``align_corners=False`` semantics
are unchanged).  Two witnesses are provided:

  * ``flow_warp_ref`` / ``flow_warp_grads_ref``: the reference's op sequence on
    torch CPU (mesh -> +flow -> normalise by (W-1),(H-1) -> grid_sample), with
    torch autograd for the two gradients.  This is the executable oracle.
  * ``flow_warp_numpy`` / ``flow_warp_numpy_grads``: an independent numpy
    restatement of ATen's published grid_sampler_2d algorithm (unnormalise
    ((g+1)*size-1)/2, clip/reflect, floor taps, nw/ne/sw/se weights, bounds
    masks; backward = weighted scatter to the image + per-pixel grid gradient
    with the clip multiplier), in the caller's dtype with the same operation
    order.  Used to cross-check the torch witness and for fp64 checks.

Quirk Q2: the grid is normalised with the
align_corners=True convention but sampled with align_corners=False, so zero
flow is NOT the identity: ix = (x+fx)*W/(W-1) - 0.5.  Both witnesses reproduce it.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# torch witness (reference op sequence)
# ----------------------------------------------------------------------------
def _mesh_grid(batch, height, width):
    xs = torch.arange(0, width).repeat(batch, height, 1)
    ys = torch.arange(0, height).repeat(batch, width, 1).transpose(1, 2)
    return torch.stack([xs, ys], 1)  # (B,2,H,W) integer pixel coordinates


def _norm_grid(v):
    _, _, height, width = v.size()
    out = torch.zeros_like(v)
    out[:, 0, :, :] = 2.0 * v[:, 0, :, :] / (width - 1) - 1.0
    out[:, 1, :, :] = 2.0 * v[:, 1, :, :] / (height - 1) - 1.0
    return out.permute(0, 2, 3, 1)


def flow_warp_ref(image, flow12, pad="border", mode="bilinear"):
    b, _, h, w = image.size()
    base = _mesh_grid(b, h, w).type_as(image)
    grid = _norm_grid(base + flow12)
    return F.grid_sample(image, grid, mode=mode, padding_mode=pad,
                         align_corners=False)


def flow_warp_grads_ref(image, flow12, gout, pad="border", mode="bilinear"):
    """Returns (out, grad_image, grad_flow) via torch autograd on CPU."""
    img = image.detach().clone().requires_grad_(True)
    flo = flow12.detach().clone().requires_grad_(True)
    out = flow_warp_ref(img, flo, pad, mode)
    gi, gf = torch.autograd.grad(out, (img, flo), gout)
    return out.detach(), gi, gf


# ----------------------------------------------------------------------------
# numpy witness (ATen grid_sampler_2d algorithm, align_corners=False)
# ----------------------------------------------------------------------------
def _reflect(x, twice_low, twice_high, T):
    # ATen reflect_coordinates
    if twice_low == twice_high:
        return np.zeros_like(x)
    mn = T(twice_low) / T(2)
    span = T(twice_high - twice_low) / T(2)
    x = np.abs(x - mn)
    extra = np.fmod(x, span)
    flips = np.floor(x / span).astype(np.int64)
    return np.where(flips % 2 == 0, extra + mn, span - extra + mn)


def _source_index(coord, size, pad, T):
    """unnormalise + padding; returns (index, d(index)/d(coord))."""
    # torch's CPU (gcc -ffp-contract) and CUDA (nvcc fmad) kernels both fuse (g+1)*size-1 into
    # one rounding; emulate the fma through float64 (exact for a 24-bit x small-integer product)
    a = (coord + T(1)).astype(T)
    idx = ((a.astype(np.float64) * float(size) - 1.0).astype(T) / T(2)).astype(T)
    mult = np.full(coord.shape, T(size) / T(2), dtype=T)
    if pad == "border":
        clipped = (idx <= 0) | (idx >= T(size - 1))
        idx = np.minimum(T(size - 1), np.maximum(idx, T(0)))
        mult = np.where(clipped, T(0), mult)
    elif pad == "reflection":
        # gradient sign through the reflection is not needed by the hot path
        # (no caller uses reflection); forward only.
        idx = _reflect(idx, -1, 2 * size - 1, T)
        idx = np.minimum(T(size - 1), np.maximum(idx, T(0)))
    elif pad != "zeros":
        raise ValueError(pad)
    return idx.astype(T), mult.astype(T)


def _sample_positions(flow, pad):
    T = flow.dtype.type
    B, _, H, W = flow.shape
    xs = np.arange(W, dtype=T)[None, None, :]
    ys = np.arange(H, dtype=T)[None, :, None]
    vx = (xs + flow[:, 0]).astype(T)
    vy = (ys + flow[:, 1]).astype(T)
    gx = (T(2.0) * vx / T(W - 1) - T(1.0)).astype(T)   # norm_grid :30
    gy = (T(2.0) * vy / T(H - 1) - T(1.0)).astype(T)   # norm_grid :31
    ix, mx = _source_index(gx, W, pad, T)
    iy, my = _source_index(gy, H, pad, T)
    return ix, iy, mx, my


def flow_warp_numpy(image, flow, pad="border", mode="bilinear"):
    image = np.asarray(image)
    flow = np.asarray(flow, dtype=image.dtype)
    T = image.dtype.type
    B, C, H, W = image.shape
    ix, iy, _, _ = _sample_positions(flow, pad)
    bidx = np.arange(B)[:, None, None, None]
    cidx = np.arange(C)[None, :, None, None]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        xs = np.clip(xx, 0, W - 1).astype(np.int64)
        ys = np.clip(yy, 0, H - 1).astype(np.int64)
        vals = image[bidx, cidx, ys[:, None], xs[:, None]]
        return np.where(ok[:, None], vals, T(0))

    if mode == "nearest":
        xn = np.rint(ix)
        yn = np.rint(iy)
        return tap(yn, xn).astype(T)
    if mode != "bilinear":
        raise ValueError(mode)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    nw = ((x1 - ix) * (y1 - iy)).astype(T)
    ne = ((ix - x0) * (y1 - iy)).astype(T)
    sw = ((x1 - ix) * (iy - y0)).astype(T)
    se = ((ix - x0) * (iy - y0)).astype(T)
    out = (tap(y0, x0) * nw[:, None] + tap(y0, x1) * ne[:, None] +
           tap(y1, x0) * sw[:, None] + tap(y1, x1) * se[:, None])
    return out.astype(T)


def flow_warp_numpy_grads(image, flow, gout, pad="border"):
    """Bilinear only.  Returns (grad_image, grad_flow)."""
    image = np.asarray(image)
    T = image.dtype.type
    flow = np.asarray(flow, dtype=T)
    gout = np.asarray(gout, dtype=T)
    B, C, H, W = image.shape
    ix, iy, mx, my = _sample_positions(flow, pad)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    nw = (x1 - ix) * (y1 - iy)
    ne = (ix - x0) * (y1 - iy)
    sw = (x1 - ix) * (iy - y0)
    se = (ix - x0) * (iy - y0)
    gimg = np.zeros_like(image)
    bidx = np.broadcast_to(np.arange(B)[:, None, None, None], gout.shape)
    cidx = np.broadcast_to(np.arange(C)[None, :, None, None], gout.shape)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        xs = np.clip(xx, 0, W - 1).astype(np.int64)
        ys = np.clip(yy, 0, H - 1).astype(np.int64)
        vals = image[bidx, cidx, np.broadcast_to(ys[:, None], gout.shape),
                     np.broadcast_to(xs[:, None], gout.shape)]
        return np.where(ok[:, None], vals, T(0)), ok, ys, xs

    def scatter(yy, xx, wgt):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        xs = np.clip(xx, 0, W - 1).astype(np.int64)
        ys = np.clip(yy, 0, H - 1).astype(np.int64)
        contrib = np.where(ok[:, None], gout * wgt[:, None], T(0))
        np.add.at(gimg, (bidx, cidx, np.broadcast_to(ys[:, None], gout.shape),
                         np.broadcast_to(xs[:, None], gout.shape)), contrib)

    scatter(y0, x0, nw.astype(T))
    scatter(y0, x1, ne.astype(T))
    scatter(y1, x0, sw.astype(T))
    scatter(y1, x1, se.astype(T))

    v_nw, *_ = tap(y0, x0)
    v_ne, *_ = tap(y0, x1)
    v_sw, *_ = tap(y1, x0)
    v_se, *_ = tap(y1, x1)
    gix = (-v_nw * (y1 - iy)[:, None] + v_ne * (y1 - iy)[:, None]
           - v_sw * (iy - y0)[:, None] + v_se * (iy - y0)[:, None]) * gout
    giy = (-v_nw * (x1 - ix)[:, None] - v_ne * (ix - x0)[:, None]
           + v_sw * (x1 - ix)[:, None] + v_se * (ix - x0)[:, None]) * gout
    ggx = mx * gix.sum(axis=1)       # d/d(normalised grid x)
    ggy = my * giy.sum(axis=1)
    gflow = np.stack([ggx / T(W - 1) * T(2.0), ggy / T(H - 1) * T(2.0)], 1)
    return gimg.astype(T), gflow.astype(T)
