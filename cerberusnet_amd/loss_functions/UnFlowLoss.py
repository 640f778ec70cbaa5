"""Drop-in for the flow-warp part of ``nnet_training/loss_functions/UnFlowLoss.py``.

``flow_warp(image, flow12, pad='border', mode='bilinear')`` (reference :83-94)
keeps its name, signature, defaults and semantics -- including quirk Q2 (the
grid is normalised by (W-1),(H-1) but sampled with align_corners=False, so zero
flow is not the identity) -- but runs as ONE fused HIP kernel per direction
(``cerberus::flow_warp``): no CPU-built mesh, no H2D copy, no grid tensor.
Gradients flow to both the image and the flow.

``mesh_grid`` / ``norm_grid`` (:11-32) are kept for importers; they are not on
the hot path any more.
"""
import torch

from .. import ops as _ops

__all__ = ["flow_warp", "mesh_grid", "norm_grid", "area_resize", "area_pyramid", "unFlowLoss"]


def area_resize(image, size):
    """``F.interpolate(image, size, mode='area')`` as one HIP launch: how the photometric loss
    brings the target images to every flow scale (reference :279-280).  Forward only."""
    return torch.ops.cerberus.area_resize(image, int(size[0]), int(size[1]))


def area_pyramid(image, sizes):
    """``[F.interpolate(image, size, mode='area') for size in sizes]`` with ONE pass over the image for all
    integer-ratio scales; a scale of the image's own size is the image itself (reference :279-280, per scale)."""
    flat = [int(v) for size in sizes for v in size]
    return list(torch.ops.cerberus.area_pyramid(image, flat))


def mesh_grid(batch_sz, height, width):
    """Pixel-coordinate grid, (B,2,H,W), channel 0 = x, channel 1 = y."""
    xs = torch.arange(0, width).view(1, 1, width).expand(batch_sz, height, width)
    ys = torch.arange(0, height).view(1, height, 1).expand(batch_sz, height, width)
    return torch.stack([xs, ys], 1)


def norm_grid(v_grid):
    """Scale pixel coordinates to [-1, 1] by (W-1), (H-1); returns (B,H,W,2)."""
    _, _, height, width = v_grid.size()
    v_grid_norm = torch.zeros_like(v_grid)
    v_grid_norm[:, 0, :, :] = 2.0 * v_grid[:, 0, :, :] / (width - 1) - 1.0
    v_grid_norm[:, 1, :, :] = 2.0 * v_grid[:, 1, :, :] / (height - 1) - 1.0
    return v_grid_norm.permute(0, 2, 3, 1)


def flow_warp(image, flow12, pad='border', mode='bilinear'):
    '''
    Warps an image given a flow prediction (fused HIP grid_sample)
    '''
    if pad not in _ops.PAD_MODES:
        raise ValueError("nn.functional.grid_sample(): expected padding_mode to be 'zeros', "
                         "'border', or 'reflection', but got: '%s'" % pad)
    if mode not in _ops.INTERP_MODES:
        raise ValueError("flow_warp: mode must be 'bilinear' or 'nearest', got '%s'" % mode)
    modes = (_ops.PAD_MODES[pad], _ops.INTERP_MODES[mode])
    if torch.is_grad_enabled() and image.requires_grad:
        # training: the forward also saves the sample positions for the backward (what
        # autograd's save_for_backward is to grid_sample in the reference)
        return torch.ops.cerberus.flow_warp_ctx(image, flow12, *modes)[0]
    # inference, and training warps of an image that carries no gradient (the photometric loss's target
    # images, :282-283): the positions are only needed by grad_image's tiles -- grad_flow recomputes them from
    # the flow -- so no context is written (2 floats per pixel at full resolution)
    return torch.ops.cerberus.flow_warp(image, flow12, *modes)


# ---- the photometric flow loss: a CALLER of the warp (host model of the hot path) ---------------
def _torch_flow_warp(image, flow12, pad="border", mode="bilinear"):
    """The reference op sequence (:83-94) in stock torch ops -- CPU wiring tests only."""
    import torch.nn.functional as F
    b, _, h, w = image.size()
    grid = norm_grid(mesh_grid(b, h, w).type_as(image) + flow12)
    return F.grid_sample(image, grid, mode=mode, padding_mode=pad, align_corners=False)


def _ssim_distance(x, y):
    """(1 - SSIM) / 2 on 3x3 windows with reflection padding, clamped to [0, 1]
    (``loss_functions.py:47-77``)."""
    import torch.nn.functional as F
    pool = lambda t: F.avg_pool2d(F.pad(t, (1, 1, 1, 1), mode="reflect"), 3, 1)
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mu_x, mu_y = pool(x), pool(y)
    var_x = pool(x ** 2) - mu_x ** 2
    var_y = pool(y ** 2) - mu_y ** 2
    cov = pool(x * y) - mu_x * mu_y
    num = (2 * mu_x * mu_y + c1) * (2 * cov + c2)
    den = (mu_x ** 2 + mu_y ** 2 + c1) * (var_x + var_y + c2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def _edge_aware_smoothness(flow, image, alpha, degree):
    """First / second order smoothness of the flow, damped across image edges (:162-187)."""
    dx = lambda t: t[:, :, :, 1:] - t[:, :, :, :-1]
    dy = lambda t: t[:, :, 1:] - t[:, :, :-1]
    wx = torch.exp(-dx(image).abs().mean(1, keepdim=True) * alpha)
    wy = torch.exp(-dy(image).abs().mean(1, keepdim=True) * alpha)
    if degree == 1:
        return ((wx * dx(flow).abs() / 2.).mean() + (wy * dy(flow).abs() / 2).mean()) / 2.
    if degree == 2:
        return ((wx[:, :, :, 1:] * dx(dx(flow)).abs()).mean() +
                (wy[:, :, 1:, :] * dy(dy(flow)).abs()).mean()) / 2.
    raise NotImplementedError(degree)


class unFlowLoss(torch.nn.Module):
    """Counterpart of ``unFlowLoss`` (:189-322) for the terms the Cerberus configs use: L1 and SSIM
    photometric terms on the image pair warped by the predicted flow at every pyramid scale,
    edge-aware smoothness, forward / backward consistency.  Same constructor keywords, same
    ``forward(predictions, targets)`` with ``predictions['flow'|'flow_b']`` (lists, full resolution
    first) and ``targets['l_img'|'l_seq']``.  The census ("ternary") term and the occlusion masks
    (dead code in the reference, :285-297: the mask is all ones) are not built.

    ``backend='hip'`` (default): the two warps per scale are ``cerberus::flow_warp`` and the area
    resizes of the targets ``cerberus::area_resize``; ``'torch'``: stock ops, CPU tests only."""

    def __init__(self, weight=1.0, weights=None, consistency=True, back_occ_only=False,
                 backend="hip", **kwargs):
        super().__init__()
        weights = weights or {"l1": 0.15, "ssim": 0.85}
        if "ternary" in weights:
            raise NotImplementedError("the census (ternary) term is outside the hot-path scope")
        self.weight = weight
        self.l1_weight = weights.get("l1")
        self.ssim_weight = weights.get("ssim")
        self.smooth_args = kwargs.get("smooth", {"degree": 2, "alpha": 0.2, "weighting": 75.0})
        self.w_sm_scales = kwargs.get("w_sm_scales", [1.0, 0.0, 0.0, 0.0, 0.0])
        self.w_wrp_scales = kwargs.get("w_wrp_scales", [1.0, 1.0, 1.0, 1.0, 0.0])
        self.consistency = consistency
        self.back_occ_only = back_occ_only
        if backend not in ("hip", "torch"):
            raise ValueError("backend must be 'hip' or 'torch'")
        self.backend = backend

    def _pyramid(self, image, sizes):
        if self.backend == "hip":
            return area_pyramid(image, sizes)
        return [torch.nn.functional.interpolate(image, size, mode="area") for size in sizes]

    def _warp(self, image, flow):
        return flow_warp(image, flow, pad="border") if self.backend == "hip" \
            else _torch_flow_warp(image, flow, pad="border")

    def loss_photometric(self, im_orig, im_recons):
        terms = []
        if self.l1_weight is not None:
            terms.append(self.l1_weight * (im_orig - im_recons).abs())
        if self.ssim_weight is not None:
            terms.append(self.ssim_weight * _ssim_distance(im_recons, im_orig))
        return sum(t.mean() for t in terms)          # (/ mean of the all-ones mask = 1)

    def forward(self, predictions, targets):
        total_warp, total_smooth, s = 0., 0., 1.
        used = [i for i in range(min(len(predictions["flow"]), len(predictions["flow_b"]))) if self.w_wrp_scales[i] != 0]
        sizes = [tuple(predictions["flow"][i].shape[2:]) for i in used]
        # both target images at every scale the loss uses: one pass over each image (reference: one
        # F.interpolate per scale and image, :279-280)
        pyr1 = dict(zip(used, self._pyramid(targets["l_img"], sizes)))
        pyr2 = dict(zip(used, self._pyramid(targets["l_seq"], sizes)))
        for i, (f12, f21) in enumerate(zip(predictions["flow"], predictions["flow_b"])):
            if self.w_wrp_scales[i] == 0:
                continue
            size = tuple(f12.shape[2:])
            im1, im2 = pyr1[i], pyr2[i]
            if i == 0:
                s = min(size)
            warp = self.loss_photometric(im1, self._warp(im2, f12))
            smooth = _edge_aware_smoothness(f12 / s, im1, self.smooth_args["alpha"],
                                            self.smooth_args["degree"])
            if self.consistency:
                warp = (warp + self.loss_photometric(im2, self._warp(im1, f21))) / 2.
                smooth = (smooth + _edge_aware_smoothness(f21 / s, im2, self.smooth_args["alpha"],
                                                          self.smooth_args["degree"])) / 2.
            total_warp = total_warp + warp * self.w_wrp_scales[i]
            total_smooth = total_smooth + smooth * self.w_sm_scales[i]
        return self.weight * (total_warp + self.smooth_args["weighting"] * total_smooth)
