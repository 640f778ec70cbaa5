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

__all__ = ["flow_warp", "mesh_grid", "norm_grid", "area_resize"]


def area_resize(image, size):
    """``F.interpolate(image, size, mode='area')`` as one HIP launch: how the photometric loss
    brings the target images to every flow scale (reference :279-280).  Forward only."""
    return torch.ops.cerberus.area_resize(image, int(size[0]), int(size[1]))


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
    if torch.is_grad_enabled() and (image.requires_grad or flow12.requires_grad):
        # training: the forward also saves the sample positions for the backward (what
        # autograd's save_for_backward is to grid_sample in the reference)
        return torch.ops.cerberus.flow_warp_ctx(image, flow12, *modes)[0]
    return torch.ops.cerberus.flow_warp(image, flow12, *modes)
