"""The host model of the hot path: HRNetV2 backbone + PWC flow head, the part of
``CerberusBase`` (``nnet_training/nnet_models/cerberus.py:88-146``) that reaches the correlation
and warp ops.  SURVEY.md section 8(d) "end-to-end timing", section 7 step 9; VERDICT r3 #6.

Same constructor keywords (``name``, ``backbone_config = {type, cfg[, pretrained]}``,
``flow_config = {type, cfg}``), same ``forward(l_img, consistency=True, l_seq=...)`` returning
``{'flow': [...], 'flow_b': [...]}``, same ``state_dict`` keys for the two sub-modules
(``backbone.*``, ``flow.*``): a reference checkpoint loads with ``strict=False`` (its
``segmentation.*`` / ``depth.*`` entries have no counterpart here).

OUT OF SCOPE: the segmentation and depth heads -- dense
convolutions / matmuls on stock ops with no hot-path op in them.  Passing a
``segmentation_config`` / ``depth_config`` raises instead of silently dropping a head.
"""
from typing import Dict

import torch
from torch import nn

from .hrnetv2 import HighResolutionNet, hrnet_config, W32
from .pwcnet_sfd import PWCNetHead

_BACKBONES = {"HighResolutionNet": HighResolutionNet}
_FLOW_HEADS = {"PWCNetHead": PWCNetHead}

CORRELATION_ARGS = dict(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1,
                        corr_multiply=1)            # configs/HRNetV2_kt.json:77-84


def cerberus_flow_config(widths=W32, estimator="FlowEstimatorLite", **head_options) -> Dict:
    """Keyword arguments of :class:`CerberusBase` for BASELINE configs 3-5: HRNetV2-W32 backbone,
    ``PWCNetHead`` with corr d = 4 (SURVEY.md 8(d) "synthetic inputs -- model level")."""
    flow_cfg = {"correlation_args": dict(CORRELATION_ARGS),
                "flow_est_network": {"type": estimator, "args": {}},
                "context_network": {"type": "ContextNetwork", "args": {}},
                "1x1_conv_out": 32}
    flow_cfg.update(head_options)
    return {"name": "CerberusFlow", "backbone_config": {"type": "HighResolutionNet", "cfg": hrnet_config(widths)},
            "flow_config": {"type": "PWCNetHead", "cfg": flow_cfg}}


class CerberusBase(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        for head in ("segmentation_config", "depth_config"):
            if kwargs.get(head):
                raise NotImplementedError(
                    "%s: the segmentation / depth heads are outside this package's scope (stock dense ops, "
                    "no correlation or warp in them); build them from the reference tree" % head)
        self.modelname = kwargs.get("name", "CerberusFlow")
        bb = kwargs["backbone_config"]
        if bb["type"] not in _BACKBONES:
            raise NotImplementedError("%s backbone does not exist" % bb["type"])
        self.backbone = _BACKBONES[bb["type"]](**bb["cfg"])
        if bb.get("pretrained"):
            self.backbone.init_weights(bb["pretrained"])
        fl = kwargs["flow_config"]
        if fl["type"] not in _FLOW_HEADS:
            raise NotImplementedError("%s flow decoder type does not exist" % fl["type"])
        self.flow = _FLOW_HEADS[fl["type"]](self.backbone.output_ch, **fl["cfg"])
        # both frames through the backbone as ONE batch of 2B (cerberus.py:112,128 run it twice):
        # BatchNorm statistics then span both frames, so it is opt-in (changes training numerics)
        self.stack_frames = bool(kwargs.get("stack_frames", False))

    def forward(self, l_img: torch.Tensor, consistency=True, **kwargs) -> Dict[str, torch.Tensor]:
        out = {}
        seq = consistency if isinstance(consistency, torch.Tensor) else kwargs.get("l_seq")
        if seq is None:
            self.backbone(l_img)                      # (nothing consumes it without a second frame)
            return out
        if self.stack_frames:
            b = l_img.size(0)
            _, both = self.backbone(torch.cat([l_img, seq], 0))
            enc = (None, [f[:b] for f in both])
            enc_bw = (None, [f[b:] for f in both])
        else:
            enc = self.backbone(l_img)
            enc_bw = self.backbone(seq)
        out["flow"] = self.flow(enc, enc_bw)
        if not isinstance(consistency, torch.Tensor) and consistency:
            out["flow_b"] = self.flow(enc_bw, enc)
        return out
