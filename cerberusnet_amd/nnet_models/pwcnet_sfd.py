"""The caller of the hot path: a coarse-to-fine PWC flow head on an HRNet feature
pyramid -- counterpart of ``PWCNetHead`` in
``nnet_training/nnet_models/pwcnet_sfd.py`` (reference ctor :121-161, forward :163-203).

Same constructor (``channels_in``, ``upsample``, kwargs ``correlation_args``,
``flow_est_network``, ``context_network``, ``1x1_conv_out``, ``output_level``), same
``state_dict`` keys/shapes, same op sequence per level:

    flow = interpolate(flow*2, x2, bilinear, align_corners=True)     (:176)
    im2_warp = flow_warp(im2, flow)                                  (:178)  <- HIP
    out_corr = leaky_relu(corr(im1, im2_warp), 0.1)                  (:181-182)  <- HIP
    feat, dflow = flow_estimator(cat[out_corr, conv1x1(im1), flow])  (:185-187)
    flow = flow + dflow; flow = flow + context(cat[feat, flow])      (:188-191)

MI355X-specific options (off the reference surface, all default to reference behaviour
except where noted):
  * ``fuse_leaky=True`` (default): LeakyReLU is applied in the correlation kernel's store
    (``cerberus::correlation_leaky``) -- same values, one pass less over
    the 81-channel volume.
  * ``fuse_concat=True`` (default, HIP backend with ``fuse_leaky``): the correlation kernel
    writes its 81 channels straight into the ``torch.cat([out_corr, im1_1by1, flow])`` buffer
    (:186-187) -- ``CostVolumeConcat``; same values, the 81-channel volume is not re-read and
    re-written by a concatenation pass.
  * ``fuse_upsample=True`` (default, HIP backend): ``F.interpolate(flow * k, scale_factor=k,
    mode='bilinear', align_corners=True)`` (:176 with k = 2, :199-201 with k = 4) runs as ONE
    launch (``cerberus::flow_upsample``) with a deterministic gather backward instead of a
    multiply + ATen's upsample and its float-atomic backward.
  * ``correlation_backend``: ``"hip"`` (default; no fallback: CPU tensors raise) or
    ``"torch"`` -- the reference's own pure-PyTorch ``CorrelationTorch`` + ``grid_sample``,
    for CPU-side wiring tests (DDP over gloo) only; never selected automatically.
"""
from typing import List, Sequence

import torch
import torch.nn.functional as F
from torch import nn

from ..correlation_package.correlation import Correlation, CorrelationTorch, cost_volume_concat
from ..loss_functions.UnFlowLoss import flow_warp, mesh_grid, norm_grid
from .pwcnet_modules import ContextNetwork, FlowEstimatorDense, FlowEstimatorLite, conv_block

_ESTIMATORS = {"FlowEstimatorDense": FlowEstimatorDense, "FlowEstimatorLite": FlowEstimatorLite}
_CONTEXTS = {"ContextNetwork": ContextNetwork}


def _torch_flow_warp(image, flow12, pad="border", mode="bilinear"):
    """Reference op sequence (UnFlowLoss.py:83-94) in stock torch ops; CPU wiring tests only."""
    b, _, h, w = image.size()
    grid = norm_grid(mesh_grid(b, h, w).type_as(image) + flow12)
    return F.grid_sample(image, grid, mode=mode, padding_mode=pad, align_corners=False)


class PWCNetHead(nn.Module):
    """Self-contained PWC head for a Cerberus-style multi-task model."""

    def __init__(self, channels_in: Sequence[int], upsample=True, **kwargs):
        super().__init__()
        self.upsample = upsample
        self.output_level = kwargs.get("output_level", 4)
        self.fuse_leaky = bool(kwargs.get("fuse_leaky", True))
        self.fuse_concat = bool(kwargs.get("fuse_concat", True))
        self.fuse_upsample = bool(kwargs.get("fuse_upsample", True))
        # f2: warp + correlation + LeakyReLU as ONE forward kernel that never writes the warped features
        # (torch.ops.cerberus.warp_correlation_leaky; its backward recomputes the warp).  Off by default: measured slower than
        # the two tuned launches (bench.py extra.f2_fused); what it saves is the warped tensor and the warp context per level
        self.fuse_warp = bool(kwargs.get("fuse_warp", False))
        self.correlation_backend = kwargs.get("correlation_backend", "hip")
        if self.correlation_backend not in ("hip", "torch"):
            raise ValueError("correlation_backend must be 'hip' or 'torch'")

        corr_args = kwargs.get("correlation_args") or dict(
            pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
        search_range = corr_args["max_displacement"]
        self.corr = Correlation(**corr_args)

        width_1x1 = kwargs.get("1x1_conv_out", 32)
        # one 1x1 projection per pyramid level, coarse level first (reversed HRNet order)
        self.conv_1x1 = nn.ModuleList(
            conv_block(c, width_1x1, kernel_size=1) for c in reversed(list(channels_in)))

        est_in = width_1x1 + (2 * search_range + 1) ** 2 + 2
        est_type = (kwargs.get("flow_est_network") or {}).get("type", "FlowEstimatorDense")
        if est_type not in _ESTIMATORS:
            raise NotImplementedError(est_type)
        self.flow_estimator = _ESTIMATORS[est_type](est_in)

        ctx_type = (kwargs.get("context_network") or {}).get("type", "ContextNetwork")
        if ctx_type not in _CONTEXTS:
            raise NotImplementedError(ctx_type)
        self.context_networks = _CONTEXTS[ctx_type](self.flow_estimator.feat_dim + 2)

    # ---- the two hot-path ops, by backend --------------------------------------------------
    def _warp(self, feat, flow):
        if self.correlation_backend == "torch":
            return _torch_flow_warp(feat, flow)
        return flow_warp(feat, flow)

    def _upsample(self, flow, factor):
        if self.correlation_backend == "hip" and self.fuse_upsample:
            return torch.ops.cerberus.flow_upsample(flow, factor)
        return F.interpolate(flow * factor, scale_factor=factor, mode="bilinear", align_corners=True)

    def _cost_volume(self, im1, im2_warp):
        c = self.corr
        if self.correlation_backend == "torch":
            if not (c.pad_size == c.max_displacement and c.kernel_size == 1 and
                    c.stride1 == 1 and c.stride2 == 1):
                raise ValueError("the torch backend only covers pad=d, k=1, s1=s2=1")
            return F.leaky_relu(CorrelationTorch(c.max_displacement)(im1, im2_warp), 0.1)
        if self.fuse_leaky:
            return torch.ops.cerberus.correlation_leaky(
                im1, im2_warp, c.pad_size, c.kernel_size, c.max_displacement, c.stride1,
                c.stride2, c.corr_multiply, 0.1)
        out = c(im1, im2_warp)
        return F.leaky_relu(out, 0.1, inplace=True)

    def forward(self, im1_pyr, im2_pyr) -> List[torch.Tensor]:
        feats1, feats2 = im1_pyr[1], im2_pyr[1]  # HRNet returns (concat, [low-res .. high-res])
        coarse = feats1[0]
        flow = coarse.new_zeros((coarse.size(0), 2, coarse.size(2), coarse.size(3)))
        flows = []
        for level, (im1, im2) in enumerate(zip(feats1, feats2)):
            c = self.corr
            fused = (level > 0 and self.fuse_warp and self.correlation_backend == "hip" and c.pad_size == 4 and
                     c.max_displacement == 4 and c.kernel_size == 1 and c.stride1 == 1 and c.stride2 == 1)
            if level == 0:
                im2_warp = im2
            else:
                flow = self._upsample(flow, 2)
                if not fused:
                    im2_warp = self._warp(im2, flow).type(im1.dtype)
            if fused:
                out_corr = torch.ops.cerberus.warp_correlation_leaky(im1, im2, flow, 1, 0.1)      # pad_mode 1 = border, as flow_warp's default
                est_in = torch.cat([out_corr, self.conv_1x1[level](im1), flow], dim=1)
            elif self.correlation_backend == "hip" and self.fuse_leaky and self.fuse_concat:
                c = self.corr
                est_in = cost_volume_concat(
                    im1, im2_warp, [self.conv_1x1[level](im1), flow],
                    (c.pad_size, c.kernel_size, c.max_displacement, c.stride1, c.stride2,
                     c.corr_multiply), 0.1)
            else:
                out_corr = self._cost_volume(im1, im2_warp)
                est_in = torch.cat([out_corr, self.conv_1x1[level](im1), flow], dim=1)
            feat, dflow = self.flow_estimator(est_in)
            flow = flow + dflow
            flow = flow + self.context_networks(torch.cat([feat, flow], dim=1))
            flows.append(flow)
            if level == self.output_level:
                break
        if self.upsample:
            flows = [self._upsample(f, 4) for f in flows]
        return flows[::-1]

    def forward_both(self, im1_pyr, im2_pyr):
        """Both flow directions in ONE pass: what ``cerberus.py:131,135`` computes with two calls of
        the head on swapped inputs, ``(flows_1to2, flows_2to1) == (self(im1_pyr, im2_pyr),
        self(im2_pyr, im1_pyr))``.  The two directions are stacked along the batch axis
        (``[f1; f2]`` against ``[f2; f1]``): every op of the head -- the hot-path kernels and the
        convolutions -- runs once on 2B items instead of twice on B, which halves the launches and
        doubles the workgroups per launch (the coarse pyramid levels are latency-bound: their
        kernels take 0.70-0.85x the time per pair at twice the batch).  Batch
        items never interact in any op of the head, so the values are those of the two separate
        calls (bit for bit where both batch sizes dispatch to the same kernels).  Costs one
        concatenation of each level's features per frame order; an encoder that runs both frames
        as one batch already holds ``[f1; f2]``."""
        feats1, feats2 = im1_pyr[1], im2_pyr[1]
        b = feats1[0].size(0)
        a = [torch.cat([x, y], dim=0) for x, y in zip(feats1, feats2)]
        sw = [torch.cat([y, x], dim=0) for x, y in zip(feats1, feats2)]
        flows = self.forward((None, a), (None, sw))
        return [f[:b] for f in flows], [f[b:] for f in flows]
