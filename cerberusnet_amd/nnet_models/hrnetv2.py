"""HRNetV2 backbone: the HOST MODEL of the hot path (SURVEY.md section 7 step 9, section 8(d)
"end-to-end timing"), not part of it.  Stock ``torch.nn`` only -- on the MI355X its
convolutions and batch norms are MIOpen's; nothing here is a hand-written kernel.

Counterpart of ``HighResolutionNet`` in ``nnet_training/nnet_models/hrnetv2.py``
(constructor :265-292, forward :368-417): same constructor keywords (``STAGE1`` .. ``STAGE4``
dictionaries with ``NUM_MODULES / NUM_BRANCHES / BLOCK / NUM_BLOCKS / NUM_CHANNELS /
FUSE_METHOD``), same ``state_dict`` keys, shapes and parameter order (reference checkpoints
load; ``tests/test_model_cpu.py`` pins keys, shapes and forward values against goldens captured
from the reference class in the build container), same return value ``(concatenated features
at 1/4 resolution, [x3, x2, x1, x0] low resolution first)`` -- the list is what ``PWCNetHead``
walks coarse to fine (``pwcnet_sfd.py:171``).

Written from the architecture, table-driven: a residual unit is a list of (kernel, width)
rows, an exchange unit is an (i, j) table of what branch j contributes to branch i.
"""
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F
from torch import nn

BN_MOMENTUM = 0.1          # hrnetv2.py:21
W18 = (18, 36, 72, 144)
W32 = (32, 64, 128, 256)
W48 = (48, 96, 192, 384)


def _bn(ch):
    return nn.BatchNorm2d(ch, momentum=BN_MOMENTUM)


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)


def _cbr(cin, cout, k, stride=1, relu=True):
    """conv -> bn [-> relu] as a Sequential whose children are '0', '1' (['2'])."""
    layers = [_conv(cin, cout, k, stride), _bn(cout)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class ResidualUnit(nn.Module):
    """BASIC (3x3, 3x3; expansion 1) or BOTTLENECK (1x1, 3x3, 1x1 x4; expansion 4) unit
    (hrnetv2.py:32-105).  Children are named conv1/bn1 .. convN/bnN and, when the shortcut
    needs a projection, ``downsample`` = Sequential(conv1x1, bn)."""
    ROWS = {"BASIC": ((3, 1), (3, 1)), "BOTTLENECK": ((1, 1), (3, 1), (1, 4))}
    EXPANSION = {"BASIC": 1, "BOTTLENECK": 4}

    def __init__(self, kind: str, cin: int, planes: int, stride: int = 1):
        super().__init__()
        rows = self.ROWS[kind]
        self.depth = len(rows)
        ch = cin
        for n, (k, mult) in enumerate(rows, 1):
            # the stride sits on the first 3x3 of the unit
            s = stride if (k == 3 and not any(r[0] == 3 for r in rows[:n - 1])) else 1
            setattr(self, "conv%d" % n, _conv(ch, planes * mult, k, s))
            setattr(self, "bn%d" % n, _bn(planes * mult))
            ch = planes * mult
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != ch:
            self.downsample = nn.Sequential(nn.Conv2d(cin, ch, 1, stride, bias=False), _bn(ch))
        self.out_channels = ch

    def forward(self, x):
        y = x
        for n in range(1, self.depth + 1):
            y = getattr(self, "bn%d" % n)(getattr(self, "conv%d" % n)(y))
            if n < self.depth:
                y = self.relu(y)
        shortcut = x if self.downsample is None else self.downsample(x)
        return self.relu(y + shortcut)


def _chain(kind, cin, planes, count):
    units, ch = [], cin
    for _ in range(count):
        units.append(ResidualUnit(kind, ch, planes))
        ch = units[-1].out_channels
    return nn.Sequential(*units), ch


class ExchangeUnit(nn.Module):
    """One multi-resolution module (hrnetv2.py:108-245): per-branch residual chains, then every
    output branch i sums all branches j brought to its resolution: j > i by a 1x1 conv + bn and a
    bilinear (align_corners=True) upsample, j < i by (i - j) stride-2 3x3 convs (the last one
    without ReLU and widening to branch i's channels), j == i as it is; ReLU after the sum."""

    def __init__(self, kind: str, blocks: Sequence[int], cin: List[int], planes: Sequence[int],
                 multi_scale_output: bool = True):
        super().__init__()
        if not (len(blocks) == len(cin) == len(planes)):
            raise ValueError("NUM_BRANCHES <> NUM_BLOCKS / NUM_CHANNELS / NUM_INCHANNELS")
        chains, width = [], []
        for b in range(len(cin)):
            chain, ch = _chain(kind, cin[b], planes[b], blocks[b])
            chains.append(chain)
            width.append(ch)
        self.branches = nn.ModuleList(chains)
        self.out_channels = width
        self.fuse_layers = None
        nb = len(width)
        if nb > 1:
            rows = []
            for i in range(nb if multi_scale_output else 1):
                row = []
                for j in range(nb):
                    if j > i:
                        row.append(_cbr(width[j], width[i], 1, relu=False))
                    elif j == i:
                        row.append(None)
                    else:
                        steps = i - j
                        row.append(nn.Sequential(*[
                            _cbr(width[j], width[i] if s == steps - 1 else width[j], 3, 2,
                                 relu=(s != steps - 1)) for s in range(steps)]))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        ys = [chain(x) for chain, x in zip(self.branches, xs)]
        if self.fuse_layers is None:
            return ys
        out = []
        for i, row in enumerate(self.fuse_layers):
            # summation order of the reference (hrnetv2.py:228-243): branch 0's term first
            acc = None
            for j, y in enumerate(ys):
                if j == i:
                    term = y
                elif j > i:
                    term = F.interpolate(row[j](y), size=ys[i].shape[-2:], mode="bilinear",
                                         align_corners=True)
                else:
                    term = row[j](y)
                acc = term if acc is None else acc + term
            out.append(self.relu(acc))
        return out


def _transition(prev: Sequence[int], cur: Sequence[int]) -> nn.ModuleList:
    """hrnetv2.py:294-322: an existing branch gets a 3x3 conv only if its width changes; a new
    branch is made from the LAST previous branch by stride-2 3x3 convs."""
    layers = []
    for i, ch in enumerate(cur):
        if i < len(prev):
            layers.append(_cbr(prev[i], ch, 3) if ch != prev[i] else None)
        else:
            steps = i + 1 - len(prev)
            layers.append(nn.Sequential(*[
                _cbr(prev[-1], ch if s == steps - 1 else prev[-1], 3, 2) for s in range(steps)]))
    return nn.ModuleList(layers)


def hrnet_config(widths: Sequence[int] = W32) -> Dict[str, dict]:
    """The stage layout of ``configs/HRNetV2_kt.json:36-70`` with NUM_CHANNELS [w, 2w, 4w, 8w]."""
    w = list(widths)
    return {
        "STAGE1": dict(NUM_MODULES=1, NUM_BRANCHES=1, BLOCK="BOTTLENECK", NUM_BLOCKS=[4],
                       NUM_CHANNELS=[64], FUSE_METHOD="SUM"),
        "STAGE2": dict(NUM_MODULES=1, NUM_BRANCHES=2, BLOCK="BASIC", NUM_BLOCKS=[4, 4],
                       NUM_CHANNELS=w[:2], FUSE_METHOD="SUM"),
        "STAGE3": dict(NUM_MODULES=4, NUM_BRANCHES=3, BLOCK="BASIC", NUM_BLOCKS=[4, 4, 4],
                       NUM_CHANNELS=w[:3], FUSE_METHOD="SUM"),
        "STAGE4": dict(NUM_MODULES=3, NUM_BRANCHES=4, BLOCK="BASIC", NUM_BLOCKS=[4, 4, 4, 4],
                       NUM_CHANNELS=w[:4], FUSE_METHOD="SUM"),
    }


class HighResolutionNet(nn.Module):
    """``HighResolutionNet(**cfg)``; ``cfg`` as :func:`hrnet_config` returns it.
    ``concat_features=False`` skips the 1/4-resolution concatenation the segmentation / depth
    heads consume (hrnetv2.py:407-413) when only the flow head follows (first return value None)."""

    def __init__(self, concat_features: bool = True, **cfg):
        super().__init__()
        self.concat_features = concat_features
        self.conv1, self.bn1 = _conv(3, 64, 3, 2), _bn(64)
        self.conv2, self.bn2 = _conv(64, 64, 3, 2), _bn(64)
        self.relu = nn.ReLU(inplace=True)
        s1 = cfg["STAGE1"]
        self.layer1, width = _chain(s1["BLOCK"], 64, s1["NUM_CHANNELS"][0], s1["NUM_BLOCKS"][0])
        prev = [width]
        self.branches_per_stage = []
        for n in (2, 3, 4):
            sc = cfg["STAGE%d" % n]
            exp = ResidualUnit.EXPANSION[sc["BLOCK"]]
            cur = [c * exp for c in sc["NUM_CHANNELS"]]
            setattr(self, "transition%d" % (n - 1), _transition(prev, cur))
            units, cin = [], list(cur)
            for _ in range(sc["NUM_MODULES"]):
                units.append(ExchangeUnit(sc["BLOCK"], sc["NUM_BLOCKS"], cin, sc["NUM_CHANNELS"]))
                cin = list(units[-1].out_channels)
            setattr(self, "stage%d" % n, nn.Sequential(*units))
            self.branches_per_stage.append(sc["NUM_BRANCHES"])
            prev = cin
        self.output_ch = list(cfg["STAGE4"]["NUM_CHANNELS"])

    def forward(self, x_in):
        x = self.relu(self.bn1(self.conv1(x_in)))
        x = self.relu(self.bn2(self.conv2(x)))
        ys = [self.layer1(x)]
        for n in (2, 3, 4):
            trans = getattr(self, "transition%d" % (n - 1))
            xs = []
            for i, layer in enumerate(trans):
                src = ys[i] if i < len(ys) else ys[-1]
                xs.append(src if layer is None else layer(src))
            ys = getattr(self, "stage%d" % n)(xs)
        feats = None
        if self.concat_features:
            size = ys[0].shape[-2:]
            feats = torch.cat([ys[0]] + [F.interpolate(y, size=size, mode="bilinear", align_corners=True)
                                         for y in ys[1:]], 1)
        return feats, ys[::-1]

    def init_weights(self, pretrained=None):
        """hrnetv2.py:419-431 without the checkpoint file handling: N(0, 0.001) convolutions,
        unit batch norms."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if pretrained:
            state = torch.load(pretrained, map_location="cpu")
            own = self.state_dict()
            own.update({k.replace("model.", ""): v for k, v in state.items()
                        if k.replace("model.", "") in own})
            self.load_state_dict(own)
